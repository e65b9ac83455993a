"""Integration-test helpers (`official.utils.testing.integration` surface)."""
from __future__ import annotations

import os
import shutil
import sys
import tempfile

from . import flags as flags_core


def run_synthetic(main, tmp_root, extra_flags=None, synth=True, max_train=1):
  """Runs `main(FLAGS)` with synthetic data for one step, then removes the model dir.

  Builds `argv = [prog, --model_dir <tmp>, --train_epochs 1, --epochs_between_evals 1]
  + extra_flags [+ --use_synthetic_data] [+ --max_train_steps N]`, parses the flags
  and calls `main(FLAGS)` (reference usage: `resnet_cifar_test.py:73-77`).
  """
  extra_flags = [] if extra_flags is None else list(extra_flags)
  model_dir = tempfile.mkdtemp(dir=tmp_root)
  args = [sys.argv[0], "--model_dir", model_dir, "--train_epochs", "1",
          "--epochs_between_evals", "1"] + extra_flags
  if synth:
    args.append("--use_synthetic_data")
  if max_train is not None:
    args.extend(["--max_train_steps", str(max_train)])
  try:
    flags_core.FLAGS(args)
    return main(flags_core.FLAGS)
  finally:
    if os.path.exists(model_dir):
      shutil.rmtree(model_dir)
