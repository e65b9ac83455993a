"""Teacher-forced check of the planned backward pass (GPU).

End-to-end gradient comparisons of a deep BatchNorm network are ill-conditioned: two correct fp32
implementations of ResNet-50 at batch 8 disagree by 2-3 % per variable, and 16-bit executions by
far more, so such a test cannot tell a wiring mistake from rounding.  Here every layer is checked
on its own instead.  After one native training step, each op's backward is recomputed with
PyTorch autograd in fp32 *from the executor's own buffers* - the op's stored input activations and
the stored gradient of its output - and compared with what the plan left behind:

* the gradient of every variable (convolution / dense kernels and biases, BatchNorm gamma / beta),
* the gradient of every activation tensor, i.e. the sum of the contributions of all its consumers
  (residual fan-out, masked residual gradients added in data-gradient epilogues, strided data
  gradients as parity launches, deferred projection gradients, dual BatchNorm sweeps, the fused
  stem BatchNorm + max-pool).

Tensors the plan never materialises (the normalised shortcut of a projection block, the stem's
full-resolution activation, masked residual gradients) are recomputed from their producers /
consumers the same way.  Errors do not accumulate across layers, so the bounds are tight:
5e-3 relative L2 in fp32 for variables and 8e-3 for activation gradients (each ReLU mask bit that flips at a pre-activation within rounding of
zero moves a gradient tensor of a million elements by about 1e-3 of its norm; a wiring mistake
moves it by its whole norm), 1e-2 in bf16 (one rounding of the stored result).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from distributed_tensorflow_amd import models


def _rel(a, b):
  return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-20)).item()


class _Teacher:
  def __init__(self, hx, spec, store, images, labels):
    self.hx, self.spec, self.store = hx, spec, store
    self.images, self.labels = images, labels
    self.producer = {op.output: op for op in spec.ops}
    self.consumers = {}
    for op in spec.ops:
      for t in op.inputs:
        self.consumers.setdefault(t, []).append(op)
    # forward tensors the plan does not write: outputs of lazily normalised shortcut BatchNorms
    # and of a BatchNorm fused with its max-pool
    self.virtual = {b.output for b in spec.ops if b.kind == "bn" and
                    (b.name in hx._bn_lazy or b.name in hx._bn_pool or b.name in hx._bn_in_tile)}
    self._act, self._grad, self._vjp = {}, {}, {}

  # -- activations (NCHW fp32) ---------------------------------------------------------------
  def act(self, name):
    if name not in self._act:
      if name == "input":
        a = self.images.permute(0, 3, 1, 2).float()
      elif name in self.virtual:
        op = self.producer[name]
        a = self.forward(op, [self.act(t) for t in op.inputs], {k: v.detach() for k, v in self.params(op).items()})
      else:
        t = self.hx.acts[name].float()
        c = self.spec.shapes[name][-1]
        a = t[..., :c].permute(0, 3, 1, 2) if t.dim() == 4 else t[..., :c]
      self._act[name] = a.detach().contiguous()
    return self._act[name]

  def params(self, op):
    names = {"conv": ["kernel"], "dense": ["kernel", "bias"], "bn": ["gamma", "beta"]}.get(op.kind, [])
    return {n: self.store.view(op.name + "/" + n).detach().clone().requires_grad_(True) for n in names}

  def forward(self, op, ins, prm):
    a = op.attrs
    if op.kind == "conv":
      pt, pb, pl, pr = a["pad"]
      x = ins[0]
      if self.hx.dt != 2:       # weights are rounded to the storage type before the product
        w = prm["kernel"].detach().to(self.hx.act_dtype).float() + (prm["kernel"] - prm["kernel"].detach())
      else:
        w = prm["kernel"]
      return F.conv2d(F.pad(x, (pl, pr, pt, pb)), w.permute(3, 2, 0, 1), None, a["stride"])
    if op.kind == "bn":
      y = F.batch_norm(ins[0], None, None, prm["gamma"], prm["beta"], True, 0.0, a["eps"])
      if len(ins) > 1:
        y = y + ins[1]
      return torch.relu(y) if a["relu"] else y
    if op.kind == "maxpool":
      pt, pb, pl, pr = a["pad"]
      return F.max_pool2d(F.pad(ins[0], (pl, pr, pt, pb), value=float("-inf")), a["k"], a["stride"])
    if op.kind == "mean":
      return ins[0].mean(dim=(2, 3))
    if op.kind == "dense":
      w = prm["kernel"]
      if self.hx.dt != 2:
        w = w.detach().to(self.hx.act_dtype).float() + (w - w.detach())
      return ins[0] @ w + prm["bias"]
    raise NotImplementedError(op.kind)

  # -- gradients -------------------------------------------------------------------------------
  def stored_grad(self, name):
    g = self.hx.act_grads.get(name)
    if g is None or name in self.virtual:
      return None
    g = g.float()
    c = self.spec.shapes[name][-1]
    return (g[..., :c].permute(0, 3, 1, 2) if g.dim() == 4 else g[..., :c]).contiguous()

  def grad(self, name):
    """Gradient of tensor `name` as the plan holds it, or recomputed from its consumers."""
    if name not in self._grad:
      g = self.stored_grad(name)
      self._grad[name] = g if g is not None else self.expected_grad(name)
    return self._grad[name]

  def vjp(self, op):
    if op.name not in self._vjp:
      ins = [self.act(t).clone().requires_grad_(t != "input") for t in op.inputs]
      prm = self.params(op)
      out = self.forward(op, ins, prm)
      out.backward(self.grad(op.output))
      self._vjp[op.name] = ([i.grad for i in ins], {k: v.grad for k, v in prm.items()})
    return self._vjp[op.name]

  def expected_grad(self, name):
    total = None
    for op in self.consumers.get(name, []):
      gi, _ = self.vjp(op)
      for t, g in zip(op.inputs, gi):
        if t == name and g is not None:
          total = g if total is None else total + g
    return total


@pytest.mark.parametrize("name,batch,dtype,tol", [
    ("resnet20", 16, "fp32", 5e-3), ("resnet50", 8, "fp32", 5e-3),
    ("resnet20", 16, "bf16", 1e-2), ("resnet50", 8, "bf16", 1e-2), ("resnet50", 8, "bf16+frag", 1e-2),
    ("resnet50", 8, "fp16+frag", 1e-2),
])
def test_every_layer_backward_from_the_executors_own_buffers(name, batch, dtype, tol, monkeypatch):
  from distributed_tensorflow_amd.engine.hip_executor import HipExecutor
  from distributed_tensorflow_amd.engine.params import ParamStore
  frag = dtype.endswith("+frag")
  if frag:
    # the BatchNorms in front of pointwise layers too: applied to the fragments of the pipeline kernel
    # and of the weight gradient
    dtype = dtype[:-5]
    monkeypatch.setenv("DTF_BN_IN_FRAG", "2")
    monkeypatch.setenv("DTF_BN_PROLOGUE_BYTES", "0")     # (at this batch size the halo kernel's prologue would take them)
  if name == "resnet50" and dtype == "bf16":
    # also the BatchNorms that are applied inside their consumer's tile (at this batch size their
    # tensors are small and would otherwise keep the apply sweep)
    monkeypatch.setenv("DTF_BN_IN_TILE", "2")
  torch.backends.cudnn.allow_tf32 = False
  torch.backends.cuda.matmul.allow_tf32 = False
  spec = getattr(models, name)()
  dev = torch.device("cuda")
  store = ParamStore(spec, dev, 11)
  hx = HipExecutor(spec, store, batch, dtype, store.make_buckets(4), use_graph=False)
  g = torch.Generator(device="cpu").manual_seed(9)
  H, W, C = spec.input_shape
  x = (torch.randn(batch, H, W, C, generator=g) * 60 + 120).to(dev)
  y = torch.randint(0, spec.num_classes, (batch,), generator=g).to(dev)
  hx.train_step(x, y, 1.0)
  torch.cuda.synchronize()
  xin = x if dtype == "fp32" else x.to(hx.act_dtype).float()     # what the first layer saw
  t = _Teacher(hx, spec, store, xin, y)
  # the loss gradient itself: (softmax - onehot) / N
  logits = t.act(spec.logits_name)
  want = (torch.softmax(logits, 1) - F.one_hot(y, spec.num_classes).float()) / batch
  assert _rel(t.stored_grad(spec.logits_name), want) <= tol
  checked_vars = checked_acts = 0
  worst = ("", 0.0)
  for op in reversed(spec.ops):
    gi, gp = t.vjp(op)
    for k, gref in gp.items():
      got = store.view(op.name + "/" + k, "grad")
      r = _rel(got, gref)
      # dgamma / dbeta are sums of terms of both signs gated by the ReLU mask: an element whose
      # pre-activation is within rounding of zero flips its mask bit between two evaluations of
      # the same formula and moves the sum by one term, ~1/sqrt(P) of the total
      # (a flipped bit also moves the weight gradients of the layers behind it: the same allowance
      # as for the activation gradients below)
      bound = max(tol, 1e-2) if op.kind == "bn" else max(tol, 8e-3)
      assert r <= bound, (op.name, k, r)
      worst = max(worst, (op.name + "/" + k, r), key=lambda p: p[1])
      checked_vars += 1
  for tname in t.consumers:
    got = t.stored_grad(tname)
    if got is None or tname == "input":
      continue
    r = _rel(got, t.expected_grad(tname))
    # the statistics come out of the convolution epilogues through float atomics: their last bits
    # vary from run to run and with them a handful of ReLU mask bits at pre-activations within
    # rounding of zero (~1e-3 of a gradient tensor's norm per flipped bit)
    assert r <= max(tol, 8e-3), (tname, r)
    checked_acts += 1
  if name == "resnet50" and dtype == "bf16":
    # the BatchNorm in front of the 56x56x64 and 28x28x128 3x3 layers (+ the sixteen in front of the
    # last pointwise layer of a block)
    assert len(hx._bn_in_tile) == (22 if frag else 6)
  elif frag:
    assert len(hx._bn_in_tile) >= 16
  assert checked_vars == len(store.trainable)
  assert checked_acts >= len([o for o in spec.ops if o.kind in ("conv", "dense")]) - 1
