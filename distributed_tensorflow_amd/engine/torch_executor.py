"""Reference executor: interprets a `ModelSpec` with PyTorch library ops.

This is the portable back end (CPU, or any torch device) used for plumbing
tests, for the multi-process `gloo` tests of the distribution strategies, and as
the numerical oracle the hand-written HIP engine is checked against.  It follows
the semantics the reference gets from Keras/TensorFlow:

* TensorFlow `SAME` padding (extra pixel after), explicit zero padding before
  `VALID` stems (`resnet_model.py:253-262`, `resnet_cifar_model.py:221-227`);
* fused-BatchNorm training rule: biased variance to normalise, *unbiased*
  variance into the moving average, `moving = moving*m + batch*(1-m)`;
* reduced precision keeps fp32 variables and casts the logits to fp32 before the
  softmax (`resnet_model.py:385-386`).

Variables are leaf views of the flat `ParamStore` buffers and their `.grad`
fields are views of the flat gradient buffer, so autograd accumulates straight
into the fusion buckets.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from ..models.graph import ModelSpec, Op
from .params import Bucket, ParamStore

__all__ = ["TorchExecutor", "TORCH_DTYPES"]

TORCH_DTYPES = {"fp32": torch.float32, "float32": torch.float32,
                "bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
                "fp16": torch.float16, "float16": torch.float16}


class TorchExecutor:
  """Runs forward / backward of one replica with torch ops and autograd."""

  native = False

  def __init__(self, spec: ModelSpec, store: ParamStore, batch_size: int,
               dtype: str = "fp32", buckets: Optional[List[Bucket]] = None,
               data_format: str = "channels_last"):
    """`data_format` is the memory layout of the activations (`resnet_model.py:243-251`
    transposes to NCHW after the NHWC input layer for `channels_first`): logical tensors are
    NCHW for the library ops either way, stored channel-major (`channels_first`) or
    pixel-major (`channels_last`, torch's channels_last memory format)."""
    if data_format not in ("channels_first", "channels_last"):
      raise ValueError("data_format must be channels_first or channels_last")
    self.data_format = data_format
    self.spec = spec
    self.store = store
    self.batch_size = batch_size
    self.dtype = TORCH_DTYPES[dtype]
    self.device = store.device
    self.buckets = buckets if buckets is not None else store.make_buckets(1)
    self.debug_keep_activations = False     # tests: keep (and retain the grads of) activations
    self.last_activations: Dict[str, torch.Tensor] = {}
    self._leaves: Dict[str, torch.Tensor] = {}
    for e in store.trainable:
      p = store.view(e.spec.name)           # view of the flat buffer, not requiring grad
      p.requires_grad_(True)                # -> a leaf that shares the flat storage
      p.grad = store.view(e.spec.name, "grad")
      self._leaves[e.spec.name] = p

  # -- helpers -------------------------------------------------------------------
  def _w(self, name: str) -> torch.Tensor:
    return self._leaves[name]

  def _run(self, images: torch.Tensor, training: bool) -> torch.Tensor:
    """images: [N,H,W,C] float. Returns fp32 logits [N,classes]."""
    st = self.store
    dt = self.dtype
    t: Dict[str, torch.Tensor] = {}
    x = images.to(self.device)
    if x.dim() != 4:
      raise ValueError("expected NHWC images, got shape %s" % (tuple(x.shape),))
    x = x.permute(0, 3, 1, 2).to(dt)
    t["input"] = x.contiguous() if self.data_format == "channels_first" else \
        x.contiguous(memory_format=torch.channels_last)
    for op in self.spec.ops:
      a = op.attrs
      src = t[op.inputs[0]]
      if op.kind == "conv":
        w = self._w(op.name + "/kernel").permute(3, 2, 0, 1).to(dt)   # HWIO -> OIHW
        pt, pb, pl, pr = a["pad"]
        if pt or pb or pl or pr:
          src = F.pad(src, (pl, pr, pt, pb))
        out = F.conv2d(src, w, None, a["stride"])
      elif op.kind == "bn":
        y = src.float()
        gamma, beta = self._w(op.name + "/gamma"), self._w(op.name + "/beta")
        rm, rv = st.view(op.name + "/moving_mean"), st.view(op.name + "/moving_variance")
        y = F.batch_norm(y, rm, rv, gamma, beta, training, 1.0 - a["momentum"], a["eps"])
        if len(op.inputs) > 1:
          y = y + t[op.inputs[1]].float()
        if a["relu"]:
          y = torch.relu(y)
        out = y.to(dt)
      elif op.kind == "maxpool":
        pt, pb, pl, pr = a["pad"]
        if pt or pb or pl or pr:
          src = F.pad(src, (pl, pr, pt, pb), value=float("-inf"))
        out = F.max_pool2d(src, a["k"], a["stride"])
      elif op.kind == "mean":
        out = src.float().mean(dim=(2, 3)).to(dt)
      elif op.kind == "flatten":
        # Keras flattens NHWC order.
        out = src.permute(0, 2, 3, 1).reshape(src.shape[0], -1) if src.dim() == 4 else src
      elif op.kind == "dense":
        w = self._w(op.name + "/kernel").to(dt)
        b = self._w(op.name + "/bias").to(dt)
        out = torch.addmm(b, src, w)
      else:
        raise NotImplementedError(op.kind)
      if self.debug_keep_activations and training and out.requires_grad:
        out.retain_grad()
      t[op.output] = out
    if self.debug_keep_activations:
      self.last_activations = t
    return t[self.spec.logits_name].float()

  # -- public API ------------------------------------------------------------------
  def train_step(self, images: torch.Tensor, labels: torch.Tensor, loss_scale: float = 1.0,
                 on_bucket_ready: Optional[Callable[[Bucket], None]] = None
                 ) -> Tuple[torch.Tensor, torch.Tensor]:
    """One forward+backward.  Gradients (of the local-batch *mean* loss, times
    `loss_scale`) are left in `store.grads`.  Returns (mean loss, #correct)."""
    self.store.grads.zero_()
    labels = labels.to(self.device)
    logits = self._run(images, training=True)
    loss = F.cross_entropy(logits, labels)
    (loss * loss_scale).backward()
    with torch.no_grad():
      correct = (logits.argmax(dim=1) == labels).sum()
    if on_bucket_ready is not None:
      for b in self.buckets:
        on_bucket_ready(b)
    return loss.detach(), correct

  @torch.no_grad()
  def eval_step(self, images: torch.Tensor, labels: torch.Tensor
                ) -> Tuple[torch.Tensor, torch.Tensor]:
    labels = labels.to(self.device)
    logits = self._run(images, training=False)
    loss = F.cross_entropy(logits, labels)
    correct = (logits.argmax(dim=1) == labels).sum()
    return loss, correct

  @torch.no_grad()
  def predict(self, images: torch.Tensor) -> torch.Tensor:
    """Softmax probabilities in fp32 (the model output of the reference)."""
    return torch.softmax(self._run(images, training=False), dim=1)

  def params_updated(self):
    """Called after the optimizer changed `store.params` (nothing cached here)."""

  def synchronize(self):
    if self.device.type == "cuda":
      torch.cuda.synchronize(self.device)
