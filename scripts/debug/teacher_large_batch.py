#!/usr/bin/env python
"""The per-layer backward check of tests/test_hip_teacher.py (every layer's gradients recomputed in
fp32 from the executor's own buffers) at a batch close to the training step's, where the launches
fill the GPU:  python scripts/debug/teacher_large_batch.py [batch=96] [dtype=bf16]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest
import test_hip_teacher as T

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 96
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
mp = pytest.MonkeyPatch()
try:
  T.test_every_layer_backward_from_the_executors_own_buffers("resnet50", batch, dtype, 1e-2, mp)
  print("every layer of resnet50 at batch %d (%s) within tolerance" % (batch, dtype))
finally:
  mp.undo()
