#!/usr/bin/env python
"""In-step tuning of convolution and weight-gradient tiles.

The tile table (`ops/tuned_gfx950.json`) is measured on kernels running alone, back to back on
hot caches.  Inside a training step a launch meets cold caches and - in the backward pass - the
other lane's kernels, so the tile that is fastest alone is not always the one that gives the
shortest step.  This script times *whole training steps* (the model and batch of bench.py) and
searches, one launch class at a time (largest share of the work first), over the tuner's
candidates for that shape; a candidate is kept when it beats the incumbent by more than the noise
margin in two measurements.  Result: `ops/tuned_gfx950_step.json`, consulted by
`HipExecutor._in_step_choice` before the stand-alone table.

    python scripts/tune_in_step.py [--model resnet50] [--batch 256] [--kinds conv,wgrad]
                                   [--out path] [--margin 0.02] [--budget_s 1000]
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DTF_HIP_GRAPH", "0")          # the large steps run eager anyway
os.environ["DTF_AUTOTUNE_IGNORE_STEP"] = "1"         # the incumbent is the stand-alone table

import torch  # noqa: E402

from distributed_tensorflow_amd import models  # noqa: E402
from distributed_tensorflow_amd.data.synthetic import SyntheticDataset  # noqa: E402
from distributed_tensorflow_amd.ops import autotune  # noqa: E402
from distributed_tensorflow_amd.train.model import Model  # noqa: E402
from distributed_tensorflow_amd.train.optimizers import SGD  # noqa: E402
from distributed_tensorflow_amd.utils import logging  # noqa: E402


def step_ms(args, device, images, labels, warm=6):
  steps = args.steps
  """Builds the executor with the current `autotune.STEP_OVERRIDE` and times `steps` steps."""
  spec = getattr(models, args.model)()
  model = Model(spec, dtype=args.dtype, device=device, executor="hip")
  model.compile(SGD(0.1, momentum=0.9), "sparse_categorical_crossentropy",
                ["sparse_categorical_accuracy"])
  for _ in range(warm):
    model.train_on_batch(images, labels)
  torch.cuda.synchronize(device)
  t0 = time.perf_counter()
  for _ in range(steps):
    model.train_on_batch(images, labels)
  torch.cuda.synchronize(device)
  ms = (time.perf_counter() - t0) * 1000.0 / steps
  ex = next(iter(model._executors.values()))  # pylint: disable=protected-access
  launches = list(getattr(ex, "step_launches", []))
  del model, ex
  gc.collect()
  torch.cuda.empty_cache()
  return ms, launches


def candidates(kind, lane, d):
  if kind == "conv":
    return [dict(cfg=c) for c in autotune._fwd_candidates(d)]  # pylint: disable=protected-access
  cur = int(d.get("splits", 1))
  if d["Cout"] < 128:                       # narrow layers: everything the stand-alone tuner tries
    return [dict(cfg=c, splits=s) for c, s in autotune._wgrad_candidates(d)]  # pylint: disable=protected-access
  tiles = [16, 24, 17, 18, 0]
  out = []
  for t in tiles:
    for mul in (1.0, 0.75, 1.5):
      c = dict(cfg=t, splits=max(1, int(round(cur * mul))))
      if c not in out:
        out.append(c)
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--model", default="resnet50")
  ap.add_argument("--batch", type=int, default=256)
  ap.add_argument("--dtype", default="bf16")
  ap.add_argument("--kinds", default="conv,wgrad")
  ap.add_argument("--out", default=autotune.STEP_PATH)
  ap.add_argument("--margin", type=float, default=0.02, help="ms a candidate must win by")
  ap.add_argument("--steps", type=int, default=30, help="timed steps per measurement")
  ap.add_argument("--budget_s", type=float, default=1000.0, help="stop starting classes after this")
  ap.add_argument("--skip", type=int, default=0, help="classes to skip (continue an earlier run)")
  ap.add_argument("--resume", action="store_true", help="start from the entries already in --out")
  a = ap.parse_args()
  logging.set_verbosity(logging.WARNING)
  t_start = time.time()
  device = torch.device("cuda", 0)
  torch.cuda.set_device(device)
  shape = (224, 224, 3) if a.model == "resnet50" else (32, 32, 3)
  ncls = 1001 if a.model == "resnet50" else 10
  ds = SyntheticDataset(shape[0], shape[1], shape[2], ncls, a.batch, device, per_sample_random=True)
  images, labels = next(iter(ds))

  autotune.STEP_OVERRIDE.clear()
  chosen = {}
  if os.path.exists(a.out) and (a.skip or a.resume):
    chosen = json.load(open(a.out))
    autotune.STEP_OVERRIDE.update(chosen)
  base, launches = step_ms(a, device, images, labels)
  base = min(base, step_ms(a, device, images, labels)[0])
  kinds = a.kinds.split(",")
  classes = {}
  for kind, lane, d in launches:
    if kind not in kinds or (kind == "wgrad" and lane != 1):
      continue
    sig = autotune.step_signature(kind, d, lane)
    if kind == "conv":
      work = d["N"] * d["OH"] * d["OW"] * d["TI"] * d["TJ"] * d["Ctap"] * d["Cout"]
    else:
      work = d["N"] * d["OH"] * d["OW"] * d["TI"] * d["TJ"] * d["Ctap"] * d["Cout"]
    c = classes.setdefault(sig, dict(kind=kind, lane=lane, desc=d, n=0, work=0))
    c["n"] += 1
    c["work"] += work
  order = sorted(classes.items(), key=lambda kv: -kv[1]["work"])
  print("baseline %.3f ms/step, %d launches in %d classes" % (base, len(launches), len(classes)),
        flush=True)
  for idx, (sig, c) in enumerate(order):
    if idx < a.skip:
      continue
    if time.time() - t_start > a.budget_s:
      print("time budget used up after %d classes" % idx, flush=True)
      break
    d = c["desc"]
    best_ms, best = base, None
    for cand in candidates(c["kind"], c["lane"], d):
      if cand.get("cfg") == d.get("cfg") and cand.get("splits", d.get("splits")) == d.get("splits"):
        continue
      autotune.STEP_OVERRIDE[sig] = cand
      try:
        ms, _ = step_ms(a, device, images, labels)
      except (ValueError, RuntimeError):            # the tile cannot serve this shape
        continue
      if ms < best_ms - a.margin:
        ms2, _ = step_ms(a, device, images, labels)           # confirm
        if max(ms, ms2) < best_ms - a.margin:
          best_ms, best = max(ms, ms2), cand
    if best is not None:
      autotune.STEP_OVERRIDE[sig] = best
      chosen[sig] = best
      base = best_ms
      with open(a.out, "w") as f:
        json.dump(chosen, f, indent=0, sort_keys=True)
    else:
      autotune.STEP_OVERRIDE.pop(sig, None)
    print("[%d] %s lane %d K=%d Cout=%d OH=%d x%d (cfg %s): %s -> %.3f ms/step"
          % (idx, c["kind"], c["lane"], d["TI"] * d["TJ"] * d["Ctap"], d["Cout"], d["OH"], c["n"],
             d.get("cfg"), best or "unchanged", base), flush=True)
  with open(a.out, "w") as f:
    json.dump(chosen, f, indent=0, sort_keys=True)
  print("wrote %d entries to %s; step %.3f ms" % (len(chosen), a.out, base))


if __name__ == "__main__":
  main()
