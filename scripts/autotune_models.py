#!/usr/bin/env python
"""Measures tile configurations for every convolution shape of the bundled models on the
current GPU and writes the cache file used by `ops/autotune.py`.

    DTF_AUTOTUNE_OUT=gpurun_out/tuned_gfx950.json python scripts/autotune_models.py \
        [--retune_rows | --retune_wgrad_rows | --retune_addend] [model:batch[:dtype] ...]
                                   (default: resnet50:256 resnet56:128 resnet50:128 resnet20:128
                                    resnet50:256:fp16)

--retune_rows measures again the shapes the whole-rows kernels can serve (--retune_wgrad_rows: their
weight gradients), --retune_addend the
launches whose epilogue adds another tensor; entries of other shapes are kept.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DTF_AUTOTUNE"] = "1"

import torch  # noqa: E402
from distributed_tensorflow_amd import models  # noqa: E402
from distributed_tensorflow_amd.engine.hip_executor import HipExecutor  # noqa: E402
from distributed_tensorflow_amd.engine.params import ParamStore  # noqa: E402
from distributed_tensorflow_amd.ops import autotune  # noqa: E402


def main():
  jobs = [a for a in sys.argv[1:] if ":" in a] or ["resnet50:256", "resnet56:128",
                                                   "resnet50:128", "resnet20:128",
                                                   "resnet50:256:fp16"]
  if "--retune_addend" in sys.argv:     # launches whose epilogue adds another tensor
    autotune.RETUNE = lambda kind, desc: kind == "conv" and desc.get("addend") is not None
  if "--retune_rows" in sys.argv:
    autotune.RETUNE = lambda kind, desc: kind == "conv" and bool(autotune._rows_candidates(desc))
  if "--retune_wgrad_rows" in sys.argv:
    autotune.RETUNE = lambda kind, desc: kind == "wgrad" and any(
        c == 192 for c, _s in autotune._wgrad_candidates(desc))
  dev = torch.device("cuda", 0)
  for job in jobs:
    name, batch, dtype = (job.split(":") + ["bf16"])[:3]
    t0 = time.time()
    spec = getattr(models, name)()
    store = ParamStore(spec, dev, 0)
    HipExecutor(spec, store, int(batch), dtype, use_graph=False)
    torch.cuda.synchronize()
    print("%s batch %s tuned in %.1fs, cache now %d entries" % (
        name, batch, time.time() - t0, len(autotune._load())), flush=True)
    del store
    torch.cuda.empty_cache()
  autotune.save()


if __name__ == "__main__":
  main()
