"""Optimizers over the flat variable buffers.

`SGD` reproduces the Keras momentum rule the reference uses
(`common.py:169-172`, `gradient_descent_v2.SGD(lr, momentum=0.9)`):

    v <- momentum * v - lr * g          (learning rate *inside* the velocity)
    w <- w + v

with the gradient `g = grad / grad_divisor + 2*lambda*w` where the second term
is the derivative of the L2 regulariser `lambda * sum(w^2)` Keras attaches to
conv kernels, the dense kernel and the dense bias (`resnet_model.py:42-43,377-379`)
but not to BatchNorm scale/offset.  On a GPU with the native extension the
whole update is one fused multi-tensor HIP kernel; otherwise a handful of torch
ops over the flat buffers.

`LossScaleOptimizer` adds static or dynamic loss scaling
(`resnet_imagenet_main.py:182-187`): gradients arrive multiplied by the scale,
are unscaled inside the update, and with dynamic scaling a non-finite gradient
skips the step and halves the scale, while 2000 clean steps double it.
"""
from __future__ import annotations

import math
from typing import Callable, Optional, Union

import torch

from ..engine.params import ParamStore

__all__ = ["SGD", "LossScaleOptimizer", "Optimizer"]

Schedule = Union[float, Callable[[int], float]]


class Optimizer:
  """Base class: holds `learning_rate` (float or schedule) and `iterations`."""

  def __init__(self, learning_rate: Schedule = 0.01):
    self._lr = learning_rate
    self.iterations = 0

  @property
  def learning_rate(self):
    return self._lr

  @learning_rate.setter
  def learning_rate(self, value):
    self._lr = value

  lr = learning_rate

  def current_lr(self) -> float:
    lr = self._lr
    if callable(lr):
      lr = lr(self.iterations)
    return float(lr)

  def apply_gradients(self, store: ParamStore, grad_divisor: float = 1.0,
                      executor=None) -> bool:  # pragma: no cover - interface
    raise NotImplementedError

  # Slots the optimizer keeps in the store (for checkpoints / broadcast).
  slot_names = ()

  def get_config(self):
    return {"name": type(self).__name__, "learning_rate": self.current_lr()}


class SGD(Optimizer):
  """Keras-form SGD with (non-Nesterov) momentum and fused L2."""

  slot_names = ("momentum",)

  def __init__(self, learning_rate: Schedule = 0.01, momentum: float = 0.0,
               nesterov: bool = False):
    super().__init__(learning_rate)
    if nesterov:
      raise NotImplementedError("the reference uses plain momentum")
    self.momentum = float(momentum)

  def apply_gradients(self, store: ParamStore, grad_divisor: float = 1.0,
                      executor=None, skip=None) -> bool:
    """`skip`: device floats [skip flag, extra gradient factor] of a device-side loss-scale
    state; the kernel leaves the update out when the flag is set."""
    lr = self.current_lr()
    from ..ops import fused  # pylint: disable=import-outside-toplevel
    v = store.slot("momentum") if self.momentum else store.params  # unused when momentum == 0
    fused.sgd_momentum_l2(store.params, store.grads, v, store.l2_vector(), lr,
                          self.momentum, 1.0 / grad_divisor, skip)
    store.version += 1
    if executor is not None:
      executor.params_updated()
    self.iterations += 1
    return True

  def get_config(self):
    c = super().get_config()
    c.update(momentum=self.momentum, nesterov=False)
    return c


class LossScaleOptimizer(Optimizer):
  """Wraps an optimizer with a static or dynamic loss scale."""

  def __init__(self, inner: Optimizer, loss_scale: Union[float, str] = "dynamic",
               initial_scale: float = 2.0 ** 15, growth_steps: int = 2000,
               multiplier: float = 2.0):
    self.inner = inner
    self.dynamic = loss_scale == "dynamic"
    self._dev_state = None
    self.scale = float(initial_scale if self.dynamic else loss_scale)
    self.growth_steps = growth_steps
    self.multiplier = multiplier
    self._good_steps = 0
    self._skipped = 0
    # device-side state of dynamic scaling on the native engine:
    # [scale, clean steps, skipped steps, skip this step, 1 / scale of this step's gradients]
    self._dev_state: Optional[torch.Tensor] = None
    self._dev_flag: Optional[torch.Tensor] = None
    self._bound = None       # the executor whose device-side scale the state kernel maintains

  # learning-rate plumbing goes to the wrapped optimizer
  @property
  def learning_rate(self):
    return self.inner.learning_rate

  @learning_rate.setter
  def learning_rate(self, v):
    self.inner.learning_rate = v

  lr = learning_rate

  @property
  def iterations(self):
    return self.inner.iterations

  @iterations.setter
  def iterations(self, v):
    self.inner.iterations = v

  @property
  def slot_names(self):
    return self.inner.slot_names

  @property
  def momentum(self):
    return getattr(self.inner, "momentum", 0.0)

  def current_lr(self):
    return self.inner.current_lr()

  @property
  def scale(self) -> float:
    """Current loss scale (reads the device state back when scaling runs on the device)."""
    if self._dev_state is not None:
      return float(self._dev_state[0].item())
    return self._scale

  @scale.setter
  def scale(self, v: float):
    self._scale = float(v)
    if self._dev_state is not None:
      self._dev_state[0] = float(v)

  @property
  def skipped_steps(self) -> int:
    if self._dev_state is not None:
      return int(self._dev_state[2].item())
    return self._skipped

  @property
  def loss_scale(self) -> float:
    return self.scale

  def loss_scale_for(self, executor) -> Optional[float]:
    """The factor for the next step's loss gradient on `executor`; None: it already sits in
    that executor's device-side scale (`loss_scale_dev`), written by the scaling kernel of the
    previous step.  Another executor (a different batch size) gets the value read back."""
    if self._dev_state is not None and executor is self._bound:
      return None
    return self.scale

  def _apply_on_device(self, store: ParamStore, grad_divisor: float, executor) -> bool:
    """Native engine: finiteness pass, scale state machine and the skip decision all run on
    the GPU - two small launches in front of the fused optimizer, no `.item()` per step."""
    from ..ops import native  # pylint: disable=import-outside-toplevel
    if self._dev_state is None:
      dev = store.grads.device
      self._dev_state = torch.tensor([self._scale, float(self._good_steps), float(self._skipped),
                                      0.0, 1.0], dtype=torch.float32, device=dev)
      self._dev_flag = torch.zeros(1, dtype=torch.int32, device=dev)
    native.loss_scale_step(store.grads, self._dev_state, self._dev_flag,
                           executor.loss_scale_dev, self.growth_steps, self.multiplier)
    self._bound = executor
    executor._loss_scale = None     # its host-side copy of the scale is stale from here on
    return self.inner.apply_gradients(store, grad_divisor, executor, skip=self._dev_state[3:5])

  def device_verdict(self, store: ParamStore, executor) -> torch.Tensor:
    """Parameter-server worker on the native engine: runs the finiteness pass and the scale
    state machine on the GPU (ordered behind whatever the current stream has waited for) and
    returns two device floats `[skip, 1 / scale of these gradients]` - the trailer of the push,
    which the PS hands to its fused optimizer unread.  A copy: the state itself is overwritten
    by the next step while the trailer may still be waiting to be sent."""
    from ..ops import native  # pylint: disable=import-outside-toplevel
    if self._dev_state is None:
      dev = store.grads.device
      self._dev_state = torch.tensor([self._scale, float(self._good_steps), float(self._skipped),
                                      0.0, 1.0], dtype=torch.float32, device=dev)
      self._dev_flag = torch.zeros(1, dtype=torch.int32, device=dev)
    native.loss_scale_step(store.grads, self._dev_state, self._dev_flag,
                           executor.loss_scale_dev, self.growth_steps, self.multiplier)
    self._bound = executor
    executor._loss_scale = None     # its host-side copy of the scale is stale from here on
    return self._dev_state[3:5].clone()

  def update_scale(self, finite: bool):
    """Advances the host-side state machine by one step whose gradients were / were not finite
    (used directly by the parameter-server worker, whose update is applied elsewhere)."""
    if not self.dynamic:
      return
    if not finite:
      self.scale = max(self.scale / self.multiplier, 1.0)
      self._good_steps = 0
      self._skipped += 1
      return
    self._good_steps += 1
    if self._good_steps >= self.growth_steps:
      self.scale *= self.multiplier
      self._good_steps = 0

  def apply_gradients(self, store: ParamStore, grad_divisor: float = 1.0,
                      executor=None) -> bool:
    if (self.dynamic and store.grads.device.type == "cuda"
        and getattr(executor, "loss_scale_dev", None) is not None):
      return self._apply_on_device(store, grad_divisor, executor)
    if self.dynamic and not bool(torch.isfinite(store.grads).all().item()):
      self.update_scale(False)
      return False
    ok = self.inner.apply_gradients(store, grad_divisor * self.scale, executor)
    self.update_scale(True)
    return ok

  def get_config(self):
    c = self.inner.get_config()
    c.update(loss_scale="dynamic" if self.dynamic else self.scale)
    return c
