"""Headline benchmark: ResNet-50 v1.5 / ImageNet-224 training throughput on N MI355X.

    python bench.py --gpus 1 --steps 30 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29511 bench.py --gpus 8 --steps 30 --warmup 5

Config (the one BASELINE.json names): ResNet-50 v1.5, 1001 classes, 224x224x3, batch 256
per GPU, bf16 compute with fp32 master weights, synthetic data resident on the device,
random-init weights, SGD-momentum + L2 (the full optimizer step is inside the timed region).
N > 1 is synchronous data parallelism, one process per GPU, fused-bucket RCCL all-reduce
overlapped with the backward pass; weak scaling (per-GPU batch fixed).

Rank 0 prints ONE JSON line. `value` is whole-job images/second (global batch * steps /
max-over-ranks wall time).  `vs_baseline` divides by the reference's published Horovod
throughput at the nearest published GPU count >= N (1 GPU: 175.5 img/s, 4: 597.8, 16: 1428.3,
Tesla P40, fp32).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

import torch  # noqa: E402

REFERENCE_IMG_PER_SEC = {1: 175.5, 4: 597.8, 16: 1428.3}   # reference README.md:258-264


def baseline_for(n: int) -> float:
  for k in sorted(REFERENCE_IMG_PER_SEC):
    if n <= k:
      return REFERENCE_IMG_PER_SEC[k]
  return REFERENCE_IMG_PER_SEC[16]


def main() -> int:
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=30)
  ap.add_argument("--warmup", type=int, default=5)
  ap.add_argument("--batch", type=int, default=256, help="per-GPU batch size")
  ap.add_argument("--model", default="resnet50", choices=["resnet50", "resnet56", "resnet20"])
  ap.add_argument("--device", default="cuda", choices=["cuda", "cpu"],
                  help="cpu: rehearsal of the launch / collective / reporting path without a GPU "
                       "(gloo, PyTorch-operator executor, fp32); not a benchmark")
  ap.add_argument("--dtype", default=None, choices=["bf16", "fp16", "fp32"],
                  help="compute precision (default: bf16 on the GPU, fp32 on the CPU)")
  ap.add_argument("--num_packs", type=int, default=4, help="gradient fusion buckets")
  ap.add_argument("--all_reduce_alg", default=None,
                  help="nccl (default) | ring | hierarchical_copy | mesh")
  ap.add_argument("--comm", default="auto", choices=["auto", "native", "torch"],
                  help="data plane of the gradient collectives: the engine's RCCL communicator "
                       "(auto on GPUs) or torch.distributed")
  ap.add_argument("--no_graph", action="store_true")
  ap.add_argument("--rehearse_collectives", action="store_true",
                  help="1 GPU only: run the multi-GPU code path (segmented backward, one RCCL "
                       "all-reduce per bucket on the communication stream) in a one-rank group")
  ap.add_argument("--executor", default="hip", choices=["hip", "torch"])
  a = ap.parse_args()
  # stdout carries the result line and nothing else: libraries write there too (RCCL prints a
  # five-line version banner to stdout when a communicator is created), so file descriptor 1 is
  # pointed at stderr for the duration of the run and the JSON goes to the saved descriptor.
  sys.stdout.flush()
  result_fd = os.dup(1)
  os.dup2(2, 1)
  if os.environ.get("DTF_DEBUG_HANG"):      # dump every thread's stack if still running after N s
    import faulthandler
    faulthandler.dump_traceback_later(float(os.environ["DTF_DEBUG_HANG"]), exit=False)

  from distributed_tensorflow_amd import models
  from distributed_tensorflow_amd.data.synthetic import SyntheticDataset
  from distributed_tensorflow_amd.parallel import strategy as strategy_lib
  from distributed_tensorflow_amd.train.model import Model
  from distributed_tensorflow_amd.train.optimizers import SGD
  from distributed_tensorflow_amd.utils import logging

  logging.set_verbosity(logging.WARNING)
  if a.no_graph:
    os.environ["DTF_HIP_GRAPH"] = "0"
  world = int(os.environ.get("WORLD_SIZE", "1"))
  if world != a.gpus:
    if world == 1 and a.gpus > 1:
      sys.stderr.write("bench.py: --gpus %d needs a launcher (torch.distributed.run "
                       "--nproc-per-node %d)\n" % (a.gpus, a.gpus))
      return 2
    a.gpus = world
  on_gpu = a.device == "cuda"
  if on_gpu and not torch.cuda.is_available():
    sys.stderr.write("bench.py needs a GPU (--device cpu rehearses the code path without one)\n")
    return 2
  dtype = a.dtype or ("bf16" if on_gpu else "fp32")
  if not on_gpu:
    a.executor = "torch"
    torch.set_num_threads(max(1, (os.cpu_count() or 8) // max(world, 1)))

  if a.gpus == 1 and a.rehearse_collectives:
    os.environ["DTF_REHEARSE_COLLECTIVES"] = "1"
  if a.gpus > 1 or a.rehearse_collectives:
    strat = strategy_lib.MirroredStrategy(use_gpu=on_gpu, all_reduce_alg=a.all_reduce_alg,
                                          num_packs=a.num_packs, comm_kind=a.comm)
    device = strat.device
    rank = strat.rank
  else:
    strat = None
    if on_gpu:
      device = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
      torch.cuda.set_device(device)
    else:
      device = torch.device("cpu")
    rank = 0

  if a.model == "resnet50":
    spec, shape, classes = models.resnet50(), (224, 224, 3), 1001
    loss, metric = "sparse_categorical_crossentropy", "sparse_categorical_accuracy"
  else:
    spec, shape, classes = getattr(models, a.model)(), (32, 32, 3), 10
    loss, metric = "sparse_categorical_crossentropy", "sparse_categorical_accuracy"

  model = Model(spec, strategy=strat, dtype=dtype, device=device, executor=a.executor)
  from distributed_tensorflow_amd.train import schedules
  global_batch = a.batch * a.gpus
  if a.model == "resnet50":
    steps_per_epoch = 1281167 / global_batch
    lr_at = lambda i: schedules.imagenet_learning_rate_schedule(0, i, steps_per_epoch, global_batch)
  else:
    lr_at = lambda i: schedules.cifar_learning_rate_schedule(0, i, 50000 / global_batch, global_batch)
  opt = SGD(lr_at(0), momentum=0.9)
  model.compile(opt, loss, [metric])
  # one random image and label per sample, resident on the device (no input pipeline)
  ds = SyntheticDataset(shape[0], shape[1], shape[2], classes, a.batch, device,
                        seed=int(os.environ.get("RANK", "0")), per_sample_random=True)
  images, labels = next(iter(ds))
  model._sync_initial_variables()  # pylint: disable=protected-access

  def sync():
    if strat is not None:
      strat.barrier()
    if on_gpu:
      torch.cuda.synchronize(device)

  last = None
  # the executor times its first 8 steps to choose between hipGraph replay and eager launch;
  # keep that one-off calibration out of both the warm-up and the timed steps
  for _ in range(8):
    opt.learning_rate = lr_at(0)
    model.train_on_batch(images, labels)
  # effective environment, process group, reducer and launch mode: a failing first multi-GPU run
  # must be diagnosable from its log alone
  from distributed_tensorflow_amd.utils.runtime_report import log_runtime_report
  log_runtime_report(strat, model, rank)
  for i in range(a.warmup):
    opt.learning_rate = lr_at(i)          # the reference's per-batch schedule (warm-up phase)
    last = model.train_on_batch(images, labels)
  sync()
  t0 = time.perf_counter()
  for i in range(a.steps):
    opt.learning_rate = lr_at(a.warmup + i)
    last = model.train_on_batch(images, labels)
  sync()
  dt = time.perf_counter() - t0
  if strat is not None:
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    dt = float(t.item())
  loss_val = float(last[0].item())
  ex = next(iter(model._executors.values()))  # pylint: disable=protected-access
  launch_mode = {True: "hipgraph", False: "eager"}.get(getattr(ex, "use_graph", False), "auto")
  if rank == 0:
    value = global_batch * a.steps / dt
    out = {
        "metric": ("images/sec (whole node) ResNet-50 ImageNet-224 bs%d/GPU" % a.batch if a.model == "resnet50"
                   else "images/sec (whole node) %s CIFAR-10 bs%d/GPU" % (
                       {"resnet56": "ResNet-56", "resnet20": "ResNet-20"}[a.model], a.batch)),
        "value": round(value, 2), "unit": "images/sec", "n_gpus": a.gpus, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(1000.0 * dt / a.steps, 3),
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": round(value / baseline_for(a.gpus), 3) if a.model == "resnet50" else None,
        "dtype": dtype,
        "data": "synthetic (device-resident batch), random-init weights" + (
            "" if on_gpu else "; CPU rehearsal, not a benchmark"),
        "config": {"model": "resnet50_v1.5" if a.model == "resnet50" else a.model,
                   "global_batch": global_batch, "per_gpu_batch": a.batch,
                   "image": "%dx%dx%d" % shape, "classes": classes,
                   "optimizer": "sgd_momentum_l2", "parallelism": "dp%d" % a.gpus + (
                       " (one-rank collective rehearsal)" if a.rehearse_collectives and a.gpus == 1
                       else ""),
                   "executor": model.executor_kind, "launch": launch_mode,
                   "collectives": ("none" if strat is None else
                                   "native RCCL communicator, all-reduce inside the launch plan"
                                   if getattr(model._reducer, "_skip", False) else
                                   "native RCCL communicator" if getattr(model._reducer, "_native", None)
                                   else "torch.distributed (%s)" % strat.comm.backend),
                   "final_loss": round(loss_val, 4),
                   "baseline_img_per_sec": baseline_for(a.gpus)}}
    sys.stdout.flush()
    os.write(result_fd, (json.dumps(out) + "\n").encode())
  if strat is not None:
    strat.barrier()
    from distributed_tensorflow_amd.parallel import comm
    try:
      comm.shutdown()
    finally:
      # the result is out and every rank has passed the barrier: leave without running the
      # communicators' destructors (a rank that is slow to tear down RCCL must not turn a
      # finished measurement into a launcher error)
      sys.stdout.flush()
      sys.stderr.flush()
      if on_gpu and a.gpus > 1:
        os._exit(0)
  return 0


if __name__ == "__main__":
  sys.exit(main())
