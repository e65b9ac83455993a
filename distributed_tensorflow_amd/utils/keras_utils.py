"""Session-level configuration (`official.utils.misc.keras_utils` surface).

`set_session_config` is the place where the reference switches eager/graph mode
and XLA (`resnet_imagenet_main.py:86-88`).  The engine here has one execution
mode (a pre-planned stream of HIP kernels), so the function configures what is
configurable: host thread pools and deterministic library behaviour.
"""
from __future__ import annotations

import torch

from ..train.callbacks import (BatchTimestamp, ProfilerCallback, TimeHistory,  # noqa: F401
                               get_profiler_callback)
from . import logging


def set_session_config(enable_eager=False, enable_xla=False, inter_op=0, intra_op=0):
  del enable_eager
  if enable_xla:
    logging.info("--enable_xla: the native engine already runs a fused, pre-planned step; "
                 "only drop_remainder batching is forced.")
  if intra_op and intra_op > 0:
    torch.set_num_threads(int(intra_op))
  if inter_op and inter_op > 0:
    try:
      torch.set_num_interop_threads(int(inter_op))
    except RuntimeError:   # can only be set once per process
      pass


def get_config_proto_v1(enable_xla=False):
  """Kept for call-site parity with the reference tests (`resnet_cifar_test.py:63`)."""
  return {"enable_xla": bool(enable_xla)}
