"""Loader and thin Python API of the in-tree HIP extension `_dtf_native`.

The extension is built ahead of time by `ops/build.py` (hipcc, gfx950) and lives next to
this file.  Importing this module on a machine with a GPU and no built extension raises: the
device path never falls back to library kernels silently.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Optional, Sequence

import torch

try:
  from . import _dtf_native as _ext          # noqa: F401
except ImportError as e:                      # pragma: no cover - exercised only without a build
  _ext = None
  _import_error = e

__all__ = ["available", "ext", "run_op", "Plan", "stream_handle", "sgd_momentum_l2", "loss_scale_step",
           "pack_desc_bytes", "pack_blocks", "PACK_PER_BLOCK"]

PACK_PER_BLOCK = 256 * 8


def available() -> bool:
  return _ext is not None


def ext():
  if _ext is None:
    raise RuntimeError(
        "native extension _dtf_native is not built; run "
        "`python -m distributed_tensorflow_amd.ops.build` (%s)" % (_import_error,))
  return _ext


def stream_handle(device: Optional[torch.device] = None) -> int:
  """Raw hipStream_t of torch's current stream (kernels are ordered with torch ops)."""
  return int(torch.cuda.current_stream(device).cuda_stream)


def run_op(kind: str, args: Dict, stream: Optional[int] = None):
  ext().run_op(kind, args, stream_handle() if stream is None else stream)


def Plan():
  return ext().Plan()


def sgd_momentum_l2(params: torch.Tensor, grads: torch.Tensor, velocity: torch.Tensor,
                    l2x2: torch.Tensor, lr: float, momentum: float, grad_scale: float = 1.0,
                    skip: Optional[torch.Tensor] = None):
  """`skip`: two device floats [skip flag, extra gradient factor] of a device-side loss-scale
  state (`loss_scale_step`); flag != 0 leaves the update out."""
  for t in (params, grads, velocity, l2x2):
    if t.dtype != torch.float32 or not t.is_contiguous() or t.device != params.device:
      raise ValueError("sgd_momentum_l2 wants contiguous fp32 tensors on one device")
  n = params.numel()
  if grads.numel() != n or l2x2.numel() != n or (momentum and velocity.numel() != n):
    raise ValueError("sgd_momentum_l2: buffer sizes differ")
  run_op("sgd", dict(p=params.data_ptr(), g=grads.data_ptr(), v=velocity.data_ptr(),
                     l2x2=l2x2.data_ptr(), n=n, lr=float(lr), momentum=float(momentum),
                     grad_scale=float(grad_scale),
                     skip=skip.data_ptr() if skip is not None else None))


def loss_scale_step(grads: torch.Tensor, state: torch.Tensor, flag: torch.Tensor,
                    loss_scale_dev: Optional[torch.Tensor], growth: int, multiplier: float):
  """Dynamic loss scaling without a host round trip: one pass over the flat gradients sets
  `flag` when any element is inf / nan, then a one-thread kernel advances
  state = [scale, clean steps, skipped steps, skip, 1 / scale used] and writes the next
  step's scale to `loss_scale_dev` (the executor's loss-gradient factor)."""
  if state.dtype != torch.float32 or state.numel() < 5 or flag.dtype != torch.int32:
    raise ValueError("loss_scale_step: state is 5 floats, flag one int32")
  run_op("nonfinite_flag", dict(g=grads.data_ptr(), n=grads.numel(), flag=flag.data_ptr()))
  run_op("loss_scale_update", dict(
      state=state.data_ptr(), flag=flag.data_ptr(),
      loss_scale_dev=loss_scale_dev.data_ptr() if loss_scale_dev is not None else None,
      growth=float(growth), multiplier=float(multiplier)))


def pack_blocks(d: Dict) -> int:
  """Thread blocks the weight-pack kernel spends on descriptor `d`: tiles of 32 input channels
  x 64 output channels per tap for the forward layout (mode 0, a transpose), 2048 elements
  per block for the data-gradient layout (mode 1, a strided copy)."""
  if d["mode"] == 0:
    return d["TI"] * d["TJpad"] * (-(-d["Cpad"] // 32)) * (-(-d["Cout"] // 64))
  return -(-d["count"] // PACK_PER_BLOCK)


def pack_desc_bytes(descs: Sequence[Dict]) -> bytes:
  """Serialises weight-pack descriptors to the C `PackDesc` layout (64 bytes each)."""
  out = bytearray()
  for d in descs:
    out += struct.pack("<qq8iqq", d["src_off"], d["dst_off"], d["TI"], d["TJ"], d["Cin"],
                       d["Cout"], d["TJpad"], d["Cpad"], d["CoutPad"], d["mode"], d["count"],
                       d["first_block"])
  if _ext is not None and descs and len(out) // len(descs) != _ext.pack_desc_size():
    raise RuntimeError("PackDesc layout mismatch between Python and C++")
  return bytes(out)
