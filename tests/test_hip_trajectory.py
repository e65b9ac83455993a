"""Loss-trajectory parity of the native engine with PyTorch fp32 (GPU).

Fifty optimisation steps of ResNet-20 from the same initial weights on the same batches (a fixed
set of eight learnable synthetic batches, cycled), SGD with momentum and the L2 terms of the model
(`resnet_cifar_model.py:32-34`): the native bf16 / fp16 / fp32 steps must follow the loss curve of
the PyTorch fp32 executor.  Single steps of two executions of a BatchNorm network drift apart
chaotically (and the bf16 engine's atomically accumulated statistics make two of its own runs
differ), so the curves are compared as a whole - the mean loss over the fifty steps within 2 % -
and as means over windows of five steps: within 5 % of the window's loss over the first twenty
steps, within 10 % afterwards (with a floor of 0.5 in the denominator: a relative bound on a loss
that approaches zero measures nothing).  Observed: up to 2.6 % in bf16 on the steep part of the
curve; fp32 agrees to four digits over the first twenty-five steps and then drifts like any other
pair of runs - 0.9 % in most, 5.4 % in one (split-K atomics here, library kernels there).  The
first step is compared exactly (same weights, same batch: only rounding differs).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

from distributed_tensorflow_amd import models
from distributed_tensorflow_amd.train.optimizers import SGD, LossScaleOptimizer


def _batches(dev, n_batches=8, bs=64):
  g = torch.Generator().manual_seed(123)
  # class-dependent mean images + noise: learnable within a few dozen steps
  protos = torch.randn(10, 32, 32, 3, generator=g)
  out = []
  for _ in range(n_batches):
    y = torch.randint(0, 10, (bs,), generator=g)
    x = protos[y] + 1.2 * torch.randn(bs, 32, 32, 3, generator=g)
    out.append((x.to(dev), y.reshape(-1, 1).float().to(dev)))
  return out


def _run(kind, dtype, steps=50):
  from distributed_tensorflow_amd.train.model import Model
  torch.backends.cudnn.allow_tf32 = False
  torch.backends.cuda.matmul.allow_tf32 = False
  dev = torch.device("cuda")
  m = Model(models.resnet20(), dtype=dtype, device=dev, executor=kind, seed=7)
  opt = SGD(0.05, momentum=0.9)
  if dtype == "fp16":
    opt = LossScaleOptimizer(opt, "dynamic", initial_scale=2.0 ** 12)
  m.compile(opt, "sparse_categorical_crossentropy", ["sparse_categorical_accuracy"])
  data = _batches(dev)
  losses = []
  for i in range(steps):
    x, y = data[i % len(data)]
    losses.append(m.train_on_batch(x, y)[0])
  return torch.stack(losses).float().cpu()


@pytest.fixture(scope="module")
def reference_curve():
  return _run("torch", "fp32")


@pytest.mark.parametrize("dtype", ["bf16", "fp16", "fp32"])
def test_fifty_step_loss_curve_follows_pytorch_fp32(reference_curve, dtype):
  ref = reference_curve
  got = _run("hip", dtype)
  print("losses", dtype, ref.reshape(10, 5).mean(1).tolist(), got.reshape(10, 5).mean(1).tolist())
  assert torch.isfinite(got).all()
  assert ref[-5:].mean() < 0.75 * ref[:5].mean()         # the task is being learnt
  tol0 = 1e-4 if dtype == "fp32" else 1e-2
  assert abs(got[0] - ref[0]) <= tol0 * ref[0], (got[0].item(), ref[0].item())
  w_ref = ref.reshape(10, 5).mean(1)
  w_got = got.reshape(10, 5).mean(1)
  rels = (w_got - w_ref).abs() / w_ref.clamp_min(0.5)
  print("windowed loss curves", dtype, w_ref.tolist(), w_got.tolist(), rels.tolist())
  assert rels[:4].max().item() <= 0.05, (dtype, rels.tolist(), w_ref.tolist(), w_got.tolist())
  assert rels.max().item() <= 0.10, (dtype, rels.tolist(), w_ref.tolist(), w_got.tolist())
  assert abs(got.mean() - ref.mean()) <= 0.02 * ref.mean(), (got.mean().item(), ref.mean().item())
