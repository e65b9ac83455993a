"""Device-dispatching entry points of the fused element-wise ops.

On a GPU these call the hand-written HIP kernels of the in-tree extension
(`ops/native.py`); on the CPU (tests, gloo runs) the same arithmetic is a few
torch ops.  A GPU tensor with the extension missing raises - there is no silent
library fallback on the device path.
"""
from __future__ import annotations

import torch

__all__ = ["sgd_momentum_l2"]


def sgd_momentum_l2(params: torch.Tensor, grads: torch.Tensor, velocity: torch.Tensor,
                    l2x2: torch.Tensor, lr: float, momentum: float, grad_scale: float = 1.0,
                    skip=None):
  """Keras-form SGD step on flat fp32 buffers, in place.

      g  = grads * grad_scale + l2x2 * params      (l2x2 = 2*lambda per element)
      v <- momentum * v - lr * g
      w <- w + v
  """
  if params.device.type == "cuda":
    from . import native  # pylint: disable=import-outside-toplevel
    native.sgd_momentum_l2(params, grads, velocity, l2x2, lr, momentum, grad_scale, skip)
    return
  if skip is not None:
    raise ValueError("a device-side skip flag needs the native optimizer kernel")
  with torch.no_grad():
    g = grads * grad_scale if grad_scale != 1.0 else grads
    g = torch.addcmul(g, l2x2, params)
    if momentum:
      velocity.mul_(momentum).add_(g, alpha=-lr)
      params.add_(velocity)
    else:
      params.add_(g, alpha=-lr)
