"""Model zoo structure: parameter counts, layer names, TF SAME padding, shapes."""
import pytest
import torch

from distributed_tensorflow_amd import models
from distributed_tensorflow_amd.engine.params import ParamStore
from distributed_tensorflow_amd.engine.torch_executor import TorchExecutor


def test_same_padding():
  assert models.same_padding(56, 3, 2) == (0, 1)     # stride-2 3x3 on an even extent
  assert models.same_padding(112, 3, 2) == (0, 1)    # the ResNet-50 max-pool
  assert models.same_padding(56, 3, 1) == (1, 1)
  assert models.same_padding(56, 1, 2) == (0, 0)
  assert models.same_padding(7, 3, 2) == (1, 1)


def test_resnet50_inventory():
  m = models.resnet50()
  assert m.count_params(True) == 25559081
  assert m.count_params(False) == 53120
  convs = [o for o in m.ops if o.kind == "conv"]
  bns = [o for o in m.ops if o.kind == "bn"]
  assert len(convs) == 53 and len(bns) == 53
  assert len(m.trainable_variables) == 161
  assert m.shapes["conv1:out"] == (112, 112, 64)
  assert m.shapes["max_pooling2d:out"] == (56, 56, 64)
  assert m.shapes["bn5c_branch2c:out"] == (7, 7, 2048)
  names = set(m.layer_names())
  for n in ("conv1", "bn_conv1", "res2a_branch1", "res3a_branch2b", "bn4f_branch2c",
            "res5c_branch2c", "fc1000"):
    assert n in names
  # v1.5: the stride sits on the 3x3 and on the projection, not on the first 1x1
  byname = {o.name: o for o in convs}
  assert byname["res3a_branch2a"].attrs["stride"] == 1
  assert byname["res3a_branch2b"].attrs["stride"] == 2
  assert byname["res3a_branch2b"].attrs["pad"] == (0, 1, 0, 1)
  assert byname["res3a_branch1"].attrs["stride"] == 2
  assert byname["conv1"].attrs["pad"] == (3, 3, 3, 3)
  assert m.var("fc1000/kernel").shape == (2048, 1001)
  assert m.var("fc1000/bias").l2 == 1e-4 and m.var("bn_conv1/gamma").l2 == 0.0
  assert m.macs_per_image() == pytest.approx(4.09e9, rel=0.01)


@pytest.mark.parametrize("fn,blocks,params", [(models.resnet20, 3, None),
                                              (models.resnet56, 9, 856058),
                                              (models.resnet110, 18, None)])
def test_resnet_cifar_inventory(fn, blocks, params):
  m = fn()
  convs = [o for o in m.ops if o.kind == "conv"]
  assert len(convs) == 6 * blocks + 1 + 3
  if params:
    assert m.count_params(True) == params and m.count_params(False) == 4288
  byname = {o.name: o for o in convs}
  assert byname["res3block_0_branch2a"].attrs["stride"] == 2     # stride on the first 3x3
  assert byname["res3block_0_branch2a"].attrs["pad"] == (0, 1, 0, 1)
  assert byname["res2block_0_branch1"].attrs["stride"] == 1      # projection even at stride 1
  assert m.shapes["logits"] == (10,)
  assert m.name == "resnet%d" % (6 * blocks + 2)


def test_trivial_model():
  m = models.trivial_model(1001)
  assert [o.kind for o in m.ops] == ["flatten", "dense", "dense"]
  assert m.count_params() == 224 * 224 * 3 + 1 + 1001 + 1001


def test_param_store_layout_and_buckets():
  m = models.resnet56()
  st = ParamStore(m, "cpu", seed=1)
  assert st.num_elements >= 856058
  # views alias the flat buffer
  v = st.view("conv1/kernel")
  v.fill_(3.0)
  e = st.entries["conv1/kernel"]
  assert torch.all(st.params[e.offset:e.offset + e.size] == 3.0)
  for packs in (1, 2, 4, 0):
    bs = st.make_buckets(packs)
    covered = sorted((b.start, b.end) for b in bs)
    assert covered[0][0] == 0 and covered[-1][1] == st.num_elements
    for (a0, a1), (b0, b1) in zip(covered, covered[1:]):
      assert a1 == b0
    # readiness order: bucket 0 holds the last layers
    assert "fc10/bias" in bs[0].var_names
    if packs:
      assert len(bs) <= packs + 1
  assert len(st.make_buckets(0)) == len(st.trainable)
  assert len(st.make_buckets(bucket_bytes=1 << 20)) >= 3
  # large model, several packs: the bucket that completes last (first layers) is latency-sized
  big = ParamStore(models.resnet50(), "cpu", seed=0)
  bs = big.make_buckets(4)
  covered = sorted((b.start, b.end) for b in bs)
  assert covered[0][0] == 0 and covered[-1][1] == big.num_elements
  assert all(a1 == b0 for (_, a1), (b0, _) in zip(covered, covered[1:]))
  assert len(bs) == 5 and "conv1/kernel" in bs[-1].var_names
  assert (bs[-1].end - bs[-1].start) * 4 <= ParamStore.TAIL_BUCKET_BYTES
  assert len(big.make_buckets(1)) == 1                 # --num_packs 1 stays one collective
  # deterministic init
  st2 = ParamStore(m, "cpu", seed=1)
  st.initialize(1)
  assert torch.equal(st.params, st2.params)
  # he_normal statistics: var = 2 / fan_in
  k = st.view("res4block_3_branch2a/kernel")
  assert k.var().item() == pytest.approx(2.0 / (3 * 3 * 64), rel=0.1)
  assert torch.all(st.view("bn_conv1/moving_variance") == 1)


def test_torch_executor_bn_semantics_and_learning():
  """Moving stats use the unbiased variance and Keras momentum; loss goes down."""
  m = models.resnet20()
  st = ParamStore(m, "cpu", seed=0)
  ex = TorchExecutor(m, st, 8)
  g = torch.Generator().manual_seed(0)
  x = torch.randn(8, 32, 32, 3, generator=g) * 50 + 100
  y = torch.randint(0, 10, (8,), generator=g)
  # expected moving stats of the stem BN after one step
  with torch.no_grad():
    w = st.view("conv1/kernel").permute(3, 2, 0, 1)
    c = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2), w, padding=1)
    mean = c.mean((0, 2, 3))
    var_unbiased = c.var((0, 2, 3), unbiased=True)
  l0, _ = ex.train_step(x, y)
  mm = st.view("bn_conv1/moving_mean")
  mv = st.view("bn_conv1/moving_variance")
  assert torch.allclose(mm, 0.003 * mean, rtol=1e-3, atol=1e-5)
  assert torch.allclose(mv, 0.997 + 0.003 * var_unbiased, rtol=1e-3, atol=1e-4)
  assert st.grads.abs().sum() > 0
  from distributed_tensorflow_amd.train.optimizers import SGD
  opt = SGD(0.05, momentum=0.9)
  for _ in range(15):
    l, _ = ex.train_step(x, y)
    opt.apply_gradients(st, 1.0, ex)
  assert l.item() < l0.item() * 0.5
  p = ex.predict(x)
  assert p.shape == (8, 10) and torch.allclose(p.sum(1), torch.ones(8), atol=1e-5)


@pytest.mark.parametrize("packs", [1, 4, 8])
def test_gradient_buckets_complete_in_backward_order(packs):
  """ResNet-50's fusion buckets are contiguous slices of the flat gradient buffer that fill up in
  index order as the backward pass walks the layers in reverse: bucket k is complete before
  any gradient of bucket k + 1 is written, so its all-reduce can start while the rest of the
  backward pass runs (the overlap the multi-GPU step relies on)."""
  import torch
  from distributed_tensorflow_amd import models
  from distributed_tensorflow_amd.engine.params import ParamStore
  spec = models.resnet50()
  store = ParamStore(spec, torch.device("cpu"), 0)
  buckets = store.make_buckets(num_packs=packs)
  assert len(buckets) >= packs
  owner = {}
  for b in buckets:
    for v in b.var_names:
      owner[v] = b.index
  # contiguous cover of the gradient buffer, from its tail (last layers) to its head
  assert buckets[0].end == store.num_elements and buckets[-1].start == 0
  for a, b in zip(buckets, buckets[1:]):
    assert b.end == a.start
  # replay the backward pass: the variables of an op get their gradients when the op's backward
  # runs, ops run in reverse execution order
  remaining = {b.index: len(b.var_names) for b in buckets}
  completed = []
  started = set()
  for op in reversed(spec.ops):
    for suffix in ("kernel", "bias", "gamma", "beta"):
      name = op.name + "/" + suffix
      if name in owner:
        k = owner[name]
        # a bucket is only written once every earlier bucket is complete
        assert all(remaining[j] == 0 for j in range(k)), (name, k)
        started.add(k)
        remaining[k] -= 1
        if remaining[k] == 0:
          completed.append(k)
  assert completed == list(range(len(buckets)))
  assert set(owner) == {e.spec.name for e in store.trainable}
