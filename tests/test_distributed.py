"""Multi-process tests of the distribution strategies on CPU (`gloo`, world size 2-8).

They use the framework's own launcher to start one process per rank on localhost - the
same trick the reference uses for manual testing (two terminals with `localhost:8080/8081`,
`README.md:120-124`) - and check what the reference never tests automatically: gradient
equality across replicas, equivalence of N-replica training with single-process large-batch
training, Horovod-API semantics, and parameter-server push/pull.
"""
import json
import os
import subprocess
import sys
import textwrap

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_ranks(tmp_path, body: str, nproc: int, num_ps: int = 0, timeout: int = 300):
  """Writes `body` as a script and runs it under the launcher; returns per-rank JSON results."""
  script = tmp_path / "worker.py"
  script.write_text("import sys, os, json\nsys.path.insert(0, %r)\n" % ROOT +
                    "import torch\ntorch.set_num_threads(1)\n" + textwrap.dedent(body))
  sys.path.insert(0, ROOT)
  from distributed_tensorflow_amd import launch
  env = {"DTF_TEST_OUT": str(tmp_path), "OMP_NUM_THREADS": "1"}
  code = launch.launch([str(script)], nproc, num_ps=num_ps, log_dir=str(tmp_path / "logs"),
                       env_extra=env, timeout_s=timeout)
  logs = ""
  for i in range(nproc):
    p = tmp_path / "logs" / ("log%d.log" % i)
    if p.exists():
      logs += "\n--- rank %d ---\n" % i + p.read_text()[-3000:]
  assert code == 0, logs
  out = []
  for i in range(nproc):
    p = tmp_path / ("result%d.json" % i)
    out.append(json.loads(p.read_text()) if p.exists() else None)
  return out


COMMON = """
from distributed_tensorflow_amd import models
from distributed_tensorflow_amd.parallel import strategy as S, cluster
from distributed_tensorflow_amd.train.model import Model
from distributed_tensorflow_amd.train.optimizers import SGD
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
def save(obj):
  with open(os.path.join(os.environ["DTF_TEST_OUT"], "result%d.json" % rank), "w") as f:
    json.dump(obj, f)
g = torch.Generator().manual_seed(123)
X = torch.randn(8, 32, 32, 3, generator=g) * 50 + 100
Y = torch.randint(0, 10, (8, 1), generator=g).float()
"""


@pytest.mark.parametrize("alg,packs", [(None, 1), ("hierarchical_copy", 3), ("ring", 0)])
def test_mirrored_matches_single_process_large_batch(tmp_path, alg, packs):
  body = COMMON + """
strat = S.get_distribution_strategy("mirrored", num_gpus=0, all_reduce_alg=%r, num_packs=%d)
assert strat.num_replicas_in_sync == world
m = Model(models.resnet20(), strategy=strat, dtype="fp32", seed=5 + rank)   # different seeds: broadcast must fix it
m.compile(SGD(0.05, momentum=0.9), "sparse_categorical_crossentropy", ["sparse_categorical_accuracy"])
per = strat.per_replica_batch_size(8)
xs, ys = X[rank * per:(rank + 1) * per], Y[rank * per:(rank + 1) * per]
m._sync_initial_variables()
p0 = m.store.params.clone()
losses = []
for _ in range(3):
  l, c = m.train_on_batch(xs, ys)
  losses.append(l.item())
save({"params_sum": m.store.params.double().sum().item(), "p0_sum": p0.double().sum().item(),
      "fc": m.store.view("fc10/kernel").flatten()[:16].tolist(),
      "conv1": m.store.view("conv1/kernel").flatten()[:16].tolist(),
      "grad_head": m.store.grads[:32].tolist(), "losses": losses,
      "collectives": m._grad_reducer().collectives_launched})
strat.barrier()
""" % (alg, packs)
  res = _run_ranks(tmp_path, body, 2)
  # replicas hold identical variables and identical (reduced) gradients
  assert res[0]["p0_sum"] == res[1]["p0_sum"]
  assert res[0]["fc"] == res[1]["fc"] and res[0]["conv1"] == res[1]["conv1"]
  assert res[0]["grad_head"] == res[1]["grad_head"]
  assert res[0]["collectives"] > 0
  # single process, global batch 8, same initial weights (seed 5 = rank 0's)
  sys.path.insert(0, ROOT)
  from distributed_tensorflow_amd import models
  from distributed_tensorflow_amd.train.model import Model
  from distributed_tensorflow_amd.train.optimizers import SGD
  torch.set_num_threads(1)
  g = torch.Generator().manual_seed(123)
  X = torch.randn(8, 32, 32, 3, generator=g) * 50 + 100
  Y = torch.randint(0, 10, (8, 1), generator=g).float()
  m = Model(models.resnet20(), dtype="fp32", seed=5)
  m.compile(SGD(0.05, momentum=0.9), "sparse_categorical_crossentropy")
  # BatchNorm statistics are per replica in the reference (no SyncBN), so the equivalence
  # with one big batch is exact only for the dense head of the first step; later layers see
  # per-shard statistics.  Compare against a manual two-shard emulation instead.
  from distributed_tensorflow_amd.engine.params import ParamStore
  from distributed_tensorflow_amd.engine.torch_executor import TorchExecutor
  spec = models.resnet20()
  shards = [ParamStore(spec, "cpu", 5), ParamStore(spec, "cpu", 5)]
  exs = [TorchExecutor(spec, s, 4) for s in shards]
  vel = torch.zeros_like(shards[0].params)
  for _ in range(3):
    for r in range(2):
      exs[r].train_step(X[r * 4:(r + 1) * 4], Y[r * 4:(r + 1) * 4].reshape(-1).long())
    gsum = shards[0].grads + shards[1].grads
    for s in shards:
      gg = gsum / 2 + s.l2_vector() * s.params
      v = 0.9 * vel - 0.05 * gg
      s.params.add_(v)
    vel = 0.9 * vel - 0.05 * (gsum / 2 + shards[0].l2_vector() * (shards[0].params - v))
  ref = shards[0].view("fc10/kernel").flatten()[:16]
  got = torch.tensor(res[0]["fc"])
  assert torch.allclose(got, ref, rtol=1e-4, atol=1e-6)
  ref_c = shards[0].view("conv1/kernel").flatten()[:16]
  assert torch.allclose(torch.tensor(res[0]["conv1"]), ref_c, rtol=1e-3, atol=1e-6)


def test_mirrored_eight_ranks_stay_in_sync(tmp_path):
  """The 8-replica job of the scaling benchmark, rehearsed on CPU: eight ranks, four fusion
  buckets, different initial seeds and different data per rank; after the broadcast and two
  steps every replica holds the same variables and the same reduced gradients."""
  body = COMMON + """
torch.set_num_threads(1)
strat = S.get_distribution_strategy("mirrored", num_gpus=0, all_reduce_alg=None, num_packs=4)
assert strat.num_replicas_in_sync == world == 8
m = Model(models.resnet20(), strategy=strat, dtype="fp32", seed=rank)
m.compile(SGD(0.05, momentum=0.9), "sparse_categorical_crossentropy", ["sparse_categorical_accuracy"])
g2 = torch.Generator().manual_seed(1000 + rank)
xs = torch.randn(2, 32, 32, 3, generator=g2) * 50 + 100
ys = torch.randint(0, 10, (2, 1), generator=g2).float()
for _ in range(2):
  l, c = m.train_on_batch(xs, ys)
red = m._grad_reducer()
save({"params": m.store.params.double().sum().item(), "state": m.store.state.double().sum().item(),
      "grads": m.store.grads.double().sum().item(), "head": m.store.params[:8].tolist(),
      "buckets": len(red.buckets), "collectives": red.collectives_launched, "loss": l.item()})
strat.barrier()
"""
  res = _run_ranks(tmp_path, body, 8, timeout=600)
  assert len(res) == 8
  for r in res[1:]:
    assert r["params"] == res[0]["params"] and r["head"] == res[0]["head"]
    assert r["grads"] == res[0]["grads"]
  assert res[0]["buckets"] >= 4 and res[0]["collectives"] == 2 * res[0]["buckets"]
  # moving statistics are per replica (no SyncBN in the reference): they differ with the data
  assert len({r["state"] for r in res}) > 1


def test_checkpoint_under_mirrored_holds_replica_mean_of_moving_statistics(tmp_path):
  """Saving under a synchronous strategy is collective: the chief's file carries the mean over
  the replicas of the BatchNorm moving statistics (Keras ON_READ / MEAN aggregation), every
  replica keeps its own copy, and only the chief writes."""
  body = COMMON + """
from distributed_tensorflow_amd.train.callbacks import ModelCheckpoint
strat = S.get_distribution_strategy("mirrored", num_gpus=0, all_reduce_alg=None, num_packs=2)
m = Model(models.resnet20(), strategy=strat, dtype="fp32", seed=0)
m.compile(SGD(0.05, momentum=0.9), "sparse_categorical_crossentropy", ["sparse_categorical_accuracy"])
g2 = torch.Generator().manual_seed(77 + rank)
xs = torch.randn(4, 32, 32, 3, generator=g2) * (30 + 40 * rank) + 100
ys = torch.randint(0, 10, (4, 1), generator=g2).float()
for _ in range(2):
  m.train_on_batch(xs, ys)
mine = m.store.state.clone()
path = os.path.join(os.environ["DTF_TEST_OUT"], "ck-{epoch}.h5")
cb = ModelCheckpoint(path); cb.set_model(m)
cb.on_epoch_end(0, {})
assert torch.equal(m.store.state, mine)                 # the replica's own copy is untouched
mean = mine.clone(); strat.average_state(mean)
out = {"mine": mine.double().sum().item(), "mean": mean.double().sum().item(),
       "wrote": os.path.exists(path.format(epoch=1))}
if rank == 0:
  m2 = Model(models.resnet20(), dtype="fp32", seed=5)
  m2.compile(SGD(0.05, momentum=0.9), "sparse_categorical_crossentropy")
  m2.load_weights(path.format(epoch=1))
  out["loaded_equals_mean"] = bool(torch.allclose(m2.store.state, mean, rtol=0, atol=1e-6))
  out["loaded_equals_mine"] = bool(torch.allclose(m2.store.state, mine, rtol=0, atol=1e-6))
strat.barrier()
save(out)
"""
  res = _run_ranks(tmp_path, body, 2)
  assert res[0]["mine"] != res[1]["mine"]
  assert res[0]["mean"] == pytest.approx(res[1]["mean"], rel=1e-9)
  assert res[0]["loaded_equals_mean"] and not res[0]["loaded_equals_mine"]
  assert len([f for f in os.listdir(tmp_path) if f.endswith(".h5")]) == 1


def test_deterministic_replica_gradient_check(tmp_path):
  """`deterministic=True` compares gradient checksums across replicas after the all-reduce
  and raises when a replica's buffer was tampered with."""
  body = COMMON + """
strat = S.get_distribution_strategy("mirrored", num_gpus=0)
m = Model(models.resnet20(), strategy=strat, dtype="fp32", seed=1, deterministic=True)
m.compile(SGD(0.05, momentum=0.9), "sparse_categorical_crossentropy")
xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
m.train_on_batch(xs, ys)                   # passes: all-reduced gradients agree
red = m._grad_reducer()
orig = red.finish
def corrupt():
  orig()
  if rank == 1:
    m.store.grads[7] += 1.0
red.finish = corrupt
caught = False
try:
  m.train_on_batch(xs, ys)
except RuntimeError as e:
  caught = "replicas disagree" in str(e)
save({"caught": caught})
strat.barrier()
"""
  res = _run_ranks(tmp_path, body, 2)
  assert res[0]["caught"] and res[1]["caught"]


def test_multi_worker_mirrored_from_tf_config(tmp_path):
  """Rank and world size come from a TF_CONFIG cluster spec (two 'workers' on localhost)."""
  sys.path.insert(0, ROOT)
  from distributed_tensorflow_amd.launch import free_port
  port = free_port()
  script = tmp_path / "w.py"
  script.write_text("import sys, os, json\nsys.path.insert(0, %r)\nimport torch\ntorch.set_num_threads(1)\n" % ROOT + textwrap.dedent("""
    from distributed_tensorflow_amd import models
    from distributed_tensorflow_amd.parallel import strategy as S, cluster
    from distributed_tensorflow_amd.train.model import Model
    from distributed_tensorflow_amd.train.optimizers import SGD
    n = cluster.configure_cluster()
    strat = S.get_distribution_strategy("multi_worker_mirrored", num_gpus=0, num_workers=n)
    m = Model(models.resnet20(), strategy=strat, dtype="fp32", seed=strat.rank)
    m.compile(SGD(0.1, momentum=0.9), "sparse_categorical_crossentropy")
    g = torch.Generator().manual_seed(strat.rank)
    x = torch.randn(2, 32, 32, 3, generator=g); y = torch.randint(0, 10, (2, 1), generator=g).float()
    m.train_on_batch(x, y)
    json.dump({"n": n, "rank": strat.rank, "world": strat.num_replicas_in_sync, "chief": strat.is_chief,
               "psum": m.store.params.double().sum().item()},
              open(os.path.join(os.environ["DTF_TEST_OUT"], "r%d.json" % strat.rank), "w"))
    strat.barrier()
  """))
  procs = []
  for idx in range(2):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
      env.pop(k, None)
    env["TF_CONFIG"] = json.dumps({"cluster": {"worker": ["localhost:%d" % port, "localhost:%d" % (port + 1)]},
                                   "task": {"type": "worker", "index": idx}})
    env["DTF_TEST_OUT"] = str(tmp_path)
    procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stderr=subprocess.PIPE))
  errs = [p.communicate(timeout=300)[1].decode()[-2000:] for p in procs]
  assert all(p.returncode == 0 for p in procs), errs
  r = [json.load(open(tmp_path / ("r%d.json" % i))) for i in range(2)]
  assert [x["rank"] for x in r] == [0, 1] and r[0]["world"] == 2 and r[0]["n"] == 2
  assert r[0]["chief"] and not r[1]["chief"]
  assert r[0]["psum"] == r[1]["psum"]


def test_horovod_api(tmp_path):
  body = COMMON + """
from distributed_tensorflow_amd import hvd
hvd.init(use_gpu=False)
assert hvd.size() == world and hvd.rank() == rank and hvd.local_rank() == rank
assert hvd.local_size() == world                       # from the launcher's LOCAL_WORLD_SIZE
del os.environ["LOCAL_WORLD_SIZE"]; hvd._state.pop("local_size", None)
hvd._publish_host(hvd._c())                            # what init() does without the variable:
assert hvd._state["local_size"] == world               # counted from the ranks' host names
if rank == 0:
  assert hvd.local_size() == world                     # a local query: one rank alone may ask
t = torch.full((4,), float(rank + 1))
avg = hvd.allreduce(t); tot = hvd.allreduce(t, average=False)
b = hvd.broadcast(torch.full((2,), float(rank + 7)), root_rank=1)
m = Model(models.resnet20(), dtype="fp32", seed=rank)
opt = hvd.DistributedOptimizer(SGD(0.1 * hvd.size(), momentum=0.9), fusion_threshold=1 << 20)
m.compile(opt, "sparse_categorical_crossentropy", ["sparse_categorical_accuracy"])
cbs = [hvd.callbacks.BroadcastGlobalVariablesCallback(0), hvd.callbacks.MetricAverageCallback(),
       hvd.callbacks.LearningRateWarmupCallback(warmup_epochs=3, verbose=0)]
class DS:
  def __iter__(self):
    while True:
      yield X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
lrs = []
from distributed_tensorflow_amd.train.callbacks import Callback
class Rec(Callback):
  def on_batch_begin(self, batch, logs=None):
    lrs.append(self.model.optimizer.current_lr())
h = m.fit(DS(), epochs=2, steps_per_epoch=2, callbacks=cbs + [Rec()], verbose=0)
save({"avg": avg.tolist(), "tot": tot.tolist(), "b": b.tolist(), "psum": m.store.params.double().sum().item(),
      "loss": [float(v) for v in h.history["loss"]], "lrs": lrs,
      "buckets": len(m._bucket_list())})
hvd.shutdown()
"""
  res = _run_ranks(tmp_path, body, 2)
  assert res[0]["avg"] == [1.5] * 4 and res[0]["tot"] == [3.0] * 4
  assert res[0]["b"] == [8.0, 8.0] and res[1]["b"] == [8.0, 8.0]
  assert res[0]["psum"] == res[1]["psum"]                   # broadcast + averaged gradients
  assert res[0]["loss"] == res[1]["loss"]                   # MetricAverageCallback
  assert res[0]["buckets"] >= 2                              # 1 MB fusion threshold on 1.1 MB of grads
  # warm-up: lr = lr0/size * (epoch*(size-1)/warmup + 1), epoch fractional (+ one step)
  lr0, size, spe = 0.2, 2, 2
  exp = [lr0 / size * (((e + (b + 1) / spe)) * (size - 1) / 3 + 1) for e in range(2) for b in range(spe)]
  assert res[0]["lrs"] == pytest.approx(exp, rel=1e-6)


def test_parameter_server_async(tmp_path):
  """1 PS + 2 workers: pull -> compute -> push, PS applies SGD on arrival, clean shutdown."""
  body = COMMON + """
strat = S.get_distribution_strategy("parameter_server", num_gpus=0)
m = Model(models.resnet20(), strategy=strat, dtype="fp32", seed=11)
m.compile(SGD(0.05, momentum=0.9), "sparse_categorical_crossentropy", ["sparse_categorical_accuracy"])
class DS:
  def __iter__(self):
    w = max(strat.worker_index, 0)
    while True:
      yield X[w * 4:(w + 1) * 4], Y[w * 4:(w + 1) * 4]
p0 = m.store.params.clone()
h = m.fit(DS(), epochs=1, steps_per_epoch=3, verbose=0)
out = {"ps": strat.is_ps_task, "chief": strat.is_chief}
if strat.is_ps_task:
  out["updates"] = strat.updates_applied
  out["moved"] = (m.store.params - p0).abs().sum().item()
  out["mom"] = m.store.slot("momentum").abs().sum().item()
  out["psum"] = m.store.params.double().sum().item()
else:
  ev = m.evaluate(DS(), steps=1)
  out["loss"] = float(h.history["loss"][-1]); out["eval"] = [float(v) for v in ev]
  out["psum"] = m.store.params.double().sum().item()
  strat.shutdown()
save(out)
"""
  res = _run_ranks(tmp_path, body, 3, num_ps=1)
  assert res[0]["ps"] and not res[1]["ps"] and res[1]["chief"] and not res[2]["chief"]
  assert res[0]["updates"] == 6                       # 2 workers x 3 pushes, none lost
  assert res[0]["moved"] > 0 and res[0]["mom"] > 0
  for r in res[1:]:
    assert r["loss"] > 0 and len(r["eval"]) == 2
  # a worker that pulls after everybody finished sees exactly the PS copy
  assert any(abs(r["psum"] - res[0]["psum"]) < 1e-6 * abs(res[0]["psum"]) for r in res[1:])


@pytest.mark.parametrize("num_ps,workers,fmt,protocol", [
    (1, 7, "fp32", "stream"), (2, 6, "bf16", "stream"),
    # the RCCL data plane's protocol (a bucket is announced when it is ready to leave, a commit
    # closes the push, every worker lands in its own buffer), here over gloo
    (1, 7, "bf16", "pieces"), (2, 6, "fp32", "pieces"), (1, 2, "fp32", "whole")])
def test_parameter_server_many_workers_no_deadlock_no_lost_update(tmp_path, num_ps, workers, fmt,
                                                                  protocol):
  """The reference's shape of job (`ps_server/run.sh`: 1 ps + 15 workers) at 8 ranks: every
  worker steps four times with overlapped pulls and bucket-wise pushes (three gradient buckets
  split over the ps shards, fp32 or bfloat16 on the wire, one push of one worker carrying a
  non-finite gradient).  Nothing deadlocks, every ps rank applies exactly one update per
  finite push and drops the other, and the shards a worker reads back afterwards are the ps
  ranks' copies."""
  body = COMMON + """
os.environ["DTF_PS_PUSH_DTYPE"] = %r
os.environ["DTF_PS_PROTOCOL"] = %r
strat = S.get_distribution_strategy("parameter_server", num_gpus=0)
assert strat.protocol == %r
strat.bucket_bytes = 300000
from distributed_tensorflow_amd.train.optimizers import LossScaleOptimizer
m = Model(models.resnet20(), strategy=strat, dtype="fp32", seed=11)
opt = LossScaleOptimizer(SGD(0.01, momentum=0.9), "dynamic", initial_scale=8.0)
m.compile(opt, "sparse_categorical_crossentropy", ["sparse_categorical_accuracy"])
out = {"ps": strat.is_ps_task}
if strat.is_ps_task:
  strat.serve(m)
  out.update(applied=strat.updates_applied, dropped=strat.updates_dropped,
             shard=list(strat._shards(m.store)[strat.task.task_index]),
             params=m.store.params.double().sum().item())
  a, b = out["shard"]
  out["shard_sum"] = m.store.params[a:b].double().sum().item()
else:
  w = strat.worker_index
  xs, ys = X[w:w + 2], Y[w:w + 2]
  for step in range(4):
    if w == 1 and step == 2:
      xs = xs * float("inf")              # this worker's third gradient is not finite
    l, c = m.train_on_batch(xs, ys)
    if w == 1 and step == 2:
      xs = X[w:w + 2]
  out["scale"] = opt.scale
  out["skipped"] = opt.skipped_steps
  out["buckets"] = len(m._bucket_list())
  # wait until every worker's pushes have left, then read the ps copies back
  st = strat._store
  strat.flush()
  st.add("test/finished", 1)
  import time
  while int(st.add("test/finished", 0)) < strat.num_workers:
    time.sleep(0.05)
  strat.pull(m.store, fresh=True)
  out["sums"] = [m.store.params[a:b].double().sum().item() for a, b in strat._shards(m.store)]
  strat.shutdown()
save(out)
""" % (fmt, protocol, protocol)
  res = _run_ranks(tmp_path, body, num_ps + workers, num_ps=num_ps, timeout=600)
  ps, wk = res[:num_ps], res[num_ps:]
  assert all(r["ps"] for r in ps) and not any(r["ps"] for r in wk)
  for r in ps:
    assert r["applied"] == 4 * workers - 1 and r["dropped"] == 1
  assert all(r["buckets"] >= 3 for r in wk)
  assert wk[1]["skipped"] == 1 and wk[1]["scale"] == 4.0 and wk[0]["scale"] == 8.0
  for r in wk:
    for k, want in enumerate(ps):
      assert abs(r["sums"][k] - want["shard_sum"]) <= 1e-9 * max(1.0, abs(want["shard_sum"]))


def test_launcher_fail_fast(tmp_path):
  body = """
import time
rank = int(os.environ["RANK"])
if rank == 1:
  sys.exit(3)
time.sleep(60)
"""
  script = tmp_path / "f.py"
  script.write_text("import sys, os\n" + textwrap.dedent(body))
  sys.path.insert(0, ROOT)
  from distributed_tensorflow_amd import launch
  import time
  t0 = time.time()
  code = launch.launch([str(script)], 2, log_dir=str(tmp_path / "logs"))
  assert code == 3
  assert time.time() - t0 < 30          # the surviving rank was torn down, not waited for


def _rehearsal_script(tmp_path, device: str, dtype: str, alg):
  script = tmp_path / "rehearse.py"
  script.write_text("import sys, os, json\nsys.path.insert(0, %r)\nimport torch\n" % ROOT + textwrap.dedent("""
    import faulthandler
    faulthandler.dump_traceback_later(240, exit=True)
    os.environ["DTF_REHEARSE_COLLECTIVES"] = "1"
    from distributed_tensorflow_amd import models
    from distributed_tensorflow_amd.parallel import strategy as S
    from distributed_tensorflow_amd.train.model import Model
    from distributed_tensorflow_amd.train.optimizers import SGD
    dev = torch.device(%r)
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(16, 32, 32, 3, generator=g) * 50 + 100).to(dev)
    y = torch.randint(0, 10, (16, 1), generator=g).float().to(dev)
    out = {}
    for name, strat in (("plain", None),
                        ("rehearsed", S.MirroredStrategy(use_gpu=dev.type == "cuda", num_packs=3,
                                                         all_reduce_alg=%r))):
      m = Model(models.resnet20(), strategy=strat, dtype=%r, device=dev, seed=1)
      m.compile(SGD(0.05, momentum=0.9), "sparse_categorical_crossentropy")
      losses = [m.train_on_batch(x, y)[0].item() for _ in range(10)]
      red = m._grad_reducer()
      out[name] = {"losses": losses, "psum": m.store.params.double().sum().item(),
                   "ncoll": red.collectives_launched if red is not None else 0,
                   "backend": strat.comm.backend if strat is not None else None}
    json.dump(out, open(%r, "w"))
  """ % (device, alg, dtype, str(tmp_path / "out.json"))))
  return script


def test_one_rank_collective_rehearsal_issues_real_collectives(tmp_path):
  """`DTF_REHEARSE_COLLECTIVES=1` forms a one-rank group and runs the segmented backward +
  per-bucket all-reduce path; the result equals training without a strategy."""
  script = _rehearsal_script(tmp_path, "cpu", "fp32", "ring")
  env = dict(os.environ)
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TF_CONFIG"):
    env.pop(k, None)
  r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True,
                     timeout=300)
  assert r.returncode == 0, r.stderr[-3000:]
  out = json.loads((tmp_path / "out.json").read_text())
  assert out["rehearsed"]["backend"] == "gloo"
  assert out["rehearsed"]["ncoll"] == 10 * 3 and out["plain"]["ncoll"] == 0
  assert out["rehearsed"]["losses"] == pytest.approx(out["plain"]["losses"], rel=1e-5)
  assert out["rehearsed"]["psum"] == pytest.approx(out["plain"]["psum"], rel=1e-7)


def test_torchrun_launch_with_agent_store(tmp_path):
  """`python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P`
  (how the multi-GPU benchmark is started): ranks come from RANK / LOCAL_RANK / WORLD_SIZE, the
  rendezvous reuses the launcher's store on MASTER_PORT, and the strategy reduces gradients."""
  script = tmp_path / "w.py"
  script.write_text("import sys, os, json\nsys.path.insert(0, %r)\nimport torch\n" % ROOT + textwrap.dedent("""
    torch.set_num_threads(1)
    from distributed_tensorflow_amd import models
    from distributed_tensorflow_amd.parallel import strategy as S, comm
    from distributed_tensorflow_amd.train.model import Model
    from distributed_tensorflow_amd.train.optimizers import SGD
    strat = S.MirroredStrategy(use_gpu=False, num_packs=2)
    assert strat.num_replicas_in_sync == 2 and strat.rank == int(os.environ["RANK"])
    m = Model(models.resnet20(), strategy=strat, dtype="fp32", seed=strat.rank)
    m.compile(SGD(0.05, momentum=0.9), "sparse_categorical_crossentropy")
    g = torch.Generator().manual_seed(strat.rank)
    x = torch.randn(4, 32, 32, 3, generator=g) * 50 + 100
    y = torch.randint(0, 10, (4, 1), generator=g).float()
    for _ in range(2):
      m.train_on_batch(x, y)
    t = torch.tensor([m.store.params.double().sum().item()], dtype=torch.float64)
    lo, hi = t.clone(), t.clone()
    torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
    torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
    assert lo.item() == hi.item(), (lo, hi)
    strat.barrier()
    if strat.rank == 0:
      open(os.path.join(%r, "ok"), "w").write("ok")
    comm.shutdown()
  """ % str(tmp_path)))
  import socket
  s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
  env = dict(os.environ, OMP_NUM_THREADS="1")
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TF_CONFIG", "MASTER_ADDR", "MASTER_PORT"):
    env.pop(k, None)
  r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                      "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                      str(port), str(script)], env=env, capture_output=True, text=True, timeout=300)
  assert r.returncode == 0, r.stderr[-3000:]
  assert (tmp_path / "ok").exists()


def test_horovodrun_command_line(tmp_path):
  """`horovodrun -np 2 -H localhost:2 python script.py` (`README.md:239-250`): host-major ranks,
  the Horovod facade sees size 2, gradients are averaged."""
  from distributed_tensorflow_amd import horovodrun
  assert horovodrun.parse_hosts("a:4,b:4", 6) == [("a", 4), ("b", 4)]
  assert horovodrun.plan([("a", 4), ("b", 4)], 6) == [("a", 0, 4), ("b", 4, 2)]
  with pytest.raises(ValueError):
    horovodrun.parse_hosts("a:1", 2)
  script = tmp_path / "hv.py"
  script.write_text("import sys, os, json\nsys.path.insert(0, %r)\nimport torch\n" % ROOT + textwrap.dedent("""
    torch.set_num_threads(1)
    import distributed_tensorflow_amd.hvd as hvd
    hvd.init(use_gpu=False)
    assert hvd.size() == 2 and hvd.local_rank() == hvd.rank()
    t = hvd.allreduce(torch.tensor([float(hvd.rank() + 1)]), average=True)
    assert abs(t.item() - 1.5) < 1e-6
    open(os.path.join(%r, "ok%%d" %% hvd.rank()), "w").write("ok")
    hvd.shutdown()
  """ % str(tmp_path)))
  env = dict(os.environ, OMP_NUM_THREADS="1")
  for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "TF_CONFIG", "MASTER_ADDR", "MASTER_PORT"):
    env.pop(k, None)
  r = subprocess.run([os.path.join(ROOT, "horovodrun"), "--start-timeout", "60", "-np", "2", "-H",
                      "localhost:2", "--verbose", "python", str(script)],
                     env=env, capture_output=True, text=True, timeout=300)
  assert r.returncode == 0, r.stderr[-3000:]
  assert (tmp_path / "ok0").exists() and (tmp_path / "ok1").exists()


def test_parameter_server_half_precision_pull(tmp_path):
  """A 16-bit worker asks for bfloat16 parameters: it receives exactly the bf16 rounding of the
  fp32 master copy, which stays on the ps task and keeps receiving fp32 updates."""
  body = COMMON + """
from distributed_tensorflow_amd.parallel.parameter_server import ParameterServerStrategy
strat = ParameterServerStrategy(use_gpu=False)
m = Model(models.resnet20(), strategy=strat, dtype="fp32", seed=7)
m.compile(SGD(0.1, momentum=0.9), "sparse_categorical_crossentropy")
if strat.is_ps_task:
  n = strat.serve(m)
  save({"updates": n, "psum": m.store.params.double().sum().item()})
else:
  expect = m.store.params.to(torch.bfloat16).float()      # same seed as the ps task
  m.store.params.zero_()
  strat.pull(m.store, half=True)
  ok_half = bool(torch.equal(m.store.params, expect))
  m.store.params.zero_()
  strat.pull(m.store)
  ok_full = bool(torch.equal(m.store.params.to(torch.bfloat16).float(), expect)) and \\
      not torch.equal(m.store.params, expect)
  m.store.grads.fill_(0.5)
  strat.push(m.store, 0.1, 1.0)
  strat.shutdown()
  save({"ok_half": ok_half, "ok_full": ok_full})
"""
  res = _run_ranks(tmp_path, body, 2, num_ps=1)
  assert res[0]["updates"] == 1
  assert res[1]["ok_half"] and res[1]["ok_full"]


def test_native_communicator_probe_verdicts_are_agreed_through_the_store():
  """Every rank must see the same verdict on the start-up all-reduce of the native
  communicator, or some would train on it while others fall back."""
  from distributed_tensorflow_amd.parallel import rccl

  class Store(dict):
    def set(self, k, v):
      self[k] = v

    def get(self, k):
      return self[k]

  st = Store()
  for r in (1, 2):
    st.set("probe/%d" % r, b"1")
  assert rccl._agree(st, "probe", 0, 3, True)
  assert not rccl._agree(st, "probe", 0, 3, False)       # this rank failed
  st.set("probe/2", b"0")
  assert not rccl._agree(st, "probe", 0, 3, True)        # another rank failed
