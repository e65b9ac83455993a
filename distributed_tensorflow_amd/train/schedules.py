"""Learning-rate schedules of the two reference drivers.

* CIFAR (`resnet_cifar_main.py:34-65`): `lr0 = 0.1 * batch / 128`, multiplied by
  0.1 / 0.01 / 0.001 from epochs 91 / 136 / 182, epoch-granular.
* ImageNet (`resnet_imagenet_main.py:37-71`): `lr0 = 0.1 * batch / 256`, linear
  warm-up over the first 5 (fractional) epochs, then x1, x0.1 @30, x0.01 @60,
  x0.001 @80.
* `PiecewiseConstantDecayWithWarmup` (`common.py:76-140`): the step-indexed
  variant selected by `--use_tensor_lr`; a plain callable `schedule(step)`.
"""
from __future__ import annotations

from typing import List, Sequence

BASE_LEARNING_RATE = 0.1   # reference common.py:32

CIFAR_LR_SCHEDULE = [(0.1, 91), (0.01, 136), (0.001, 182)]           # (multiplier, start epoch)
IMAGENET_LR_SCHEDULE = [(1.0, 5), (0.1, 30), (0.01, 60), (0.001, 80)]


def cifar_learning_rate_schedule(current_epoch, current_batch, batches_per_epoch, batch_size):
  """Piecewise-constant CIFAR schedule with linear batch-size scaling."""
  del current_batch, batches_per_epoch
  initial = BASE_LEARNING_RATE * batch_size / 128
  lr = initial
  for mult, start_epoch in CIFAR_LR_SCHEDULE:
    if current_epoch >= start_epoch:
      lr = initial * mult
    else:
      break
  return lr


def imagenet_learning_rate_schedule(current_epoch, current_batch, batches_per_epoch,
                                    batch_size):
  """Linear scaling rule, gradual warm-up and step decay."""
  initial = BASE_LEARNING_RATE * batch_size / 256
  epoch = current_epoch + float(current_batch) / batches_per_epoch
  warm_mult, warm_end = IMAGENET_LR_SCHEDULE[0]
  if epoch < warm_end:
    return initial * warm_mult * epoch / warm_end
  lr = initial
  for mult, start_epoch in IMAGENET_LR_SCHEDULE:
    if epoch >= start_epoch:
      lr = initial * mult
    else:
      break
  return lr


class PiecewiseConstantDecayWithWarmup:
  """Step-indexed schedule: linear warm-up, then piecewise-constant decay."""

  def __init__(self, batch_size: int, epoch_size: int, warmup_epochs: int,
               boundaries: Sequence[int], multipliers: Sequence[float],
               compute_lr_on_cpu: bool = True, name: str = None):
    if len(boundaries) != len(multipliers) - 1:
      raise ValueError("The length of boundaries must be 1 less than the "
                       "length of multipliers")
    base_lr_batch_size = 256
    steps_per_epoch = epoch_size // batch_size
    self.rescaled_lr = BASE_LEARNING_RATE * batch_size / base_lr_batch_size
    self.step_boundaries: List[float] = [float(steps_per_epoch) * b for b in boundaries]
    self.lr_values: List[float] = [self.rescaled_lr * m for m in multipliers]
    self.warmup_steps = warmup_epochs * steps_per_epoch
    self.compute_lr_on_cpu = compute_lr_on_cpu
    self.name = name

  def __call__(self, step: int) -> float:
    if step < self.warmup_steps:
      return self.rescaled_lr * (float(step) / float(self.warmup_steps))
    # tf piecewise_constant: values[i] for boundaries[i-1] < step <= boundaries[i]
    for i, b in enumerate(self.step_boundaries):
      if step <= b:
        return self.lr_values[i]
    return self.lr_values[-1]

  def get_config(self):
    return {"rescaled_lr": self.rescaled_lr, "step_boundaries": self.step_boundaries,
            "lr_values": self.lr_values, "warmup_steps": self.warmup_steps,
            "compute_lr_on_cpu": self.compute_lr_on_cpu, "name": self.name}
