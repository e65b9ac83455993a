"""Flat variable storage: one fp32 buffer for every trainable variable of a model.

All trainable variables live back to back (forward-execution order, each start
aligned to 16 bytes) in `params`; `grads` and every optimizer slot mirror that
layout; BatchNorm moving statistics live in a second flat buffer `state`.
Consequences the rest of the framework relies on:

* a gradient *fusion bucket* is a contiguous slice of `grads`, so one RCCL
  all-reduce per bucket needs no packing copy (the tensor-fusion buffer that
  Horovod builds at run time, cf. `hvd.DistributedOptimizer` used at reference
  `resnet_imagenet_main_horovod.py:190`, and the `--num_packs` concatenation of
  `resnet_imagenet_main.py:116-117`, exist here by construction);
* the fused SGD kernel walks the flat buffers with a small per-variable table;
* broadcast / parameter-server pull / push move one buffer, not 161 tensors.

Variables keep Keras layouts (conv HWIO, dense `[in,out]`) so a checkpoint is a
plain per-name copy of these views.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from ..models.graph import ModelSpec, VarSpec

__all__ = ["ParamStore", "Bucket", "VarEntry", "init_variable"]

_ALIGN = 4  # elements (16 bytes of fp32)


@dataclasses.dataclass
class VarEntry:
  spec: VarSpec
  offset: int          # element offset into params (trainable) or state
  size: int            # number of real elements
  padded: int          # size rounded up to the alignment
  index: int           # position among trainable (or non-trainable) variables


@dataclasses.dataclass
class Bucket:
  index: int
  start: int           # element range [start, end) of the flat gradient buffer
  end: int
  var_names: List[str]


def init_variable(spec: VarSpec, gen: torch.Generator) -> torch.Tensor:
  """Initial value of one variable on the CPU (fp32), Keras initialiser semantics."""
  shape = tuple(spec.shape)
  if spec.init == "zeros":
    return torch.zeros(shape)
  if spec.init == "ones":
    return torch.ones(shape)
  if spec.init == "he_normal":
    # Keras VarianceScaling(scale=2, mode=fan_in, truncated_normal): the stddev is
    # divided by the std of a unit normal truncated to [-2, 2] so that the
    # resulting variance is exactly 2 / fan_in.
    std = math.sqrt(2.0 / max(spec.fan_in, 1)) / 0.87962566103423978
    t = torch.empty(shape)
    torch.nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2 * std, b=2 * std, generator=gen)
    return t
  if spec.init == "normal0.01":
    return torch.randn(shape, generator=gen) * 0.01
  if spec.init == "glorot_uniform":
    fan_in, fan_out = shape[0], shape[-1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=gen) * 2 - 1) * lim
  raise ValueError("unknown initialiser %r" % spec.init)


class ParamStore:
  """Owns the flat buffers of one model replica."""

  def __init__(self, spec: ModelSpec, device: torch.device | str = "cpu", seed: int = 0):
    self.spec = spec
    self.device = torch.device(device)
    self.entries: Dict[str, VarEntry] = {}
    off = 0
    soff = 0
    ti = si = 0
    for v in spec.variables:
      padded = -(-v.size // _ALIGN) * _ALIGN
      if v.trainable:
        self.entries[v.name] = VarEntry(v, off, v.size, padded, ti)
        off += padded
        ti += 1
      else:
        self.entries[v.name] = VarEntry(v, soff, v.size, padded, si)
        soff += padded
        si += 1
    self.num_elements = off
    self.num_state_elements = soff
    self.params = torch.zeros(off, dtype=torch.float32, device=self.device)
    self.grads = torch.zeros(off, dtype=torch.float32, device=self.device)
    self.state = torch.zeros(max(soff, 1), dtype=torch.float32, device=self.device)
    self.slots: Dict[str, torch.Tensor] = {}
    self.version = 0  # bumped whenever params change (lets executors refresh bf16 copies)
    self.initialize(seed)

  # -- views -------------------------------------------------------------------
  def _flat(self, e: VarEntry, which: str) -> torch.Tensor:
    if not e.spec.trainable:
      if which != "value":
        raise KeyError("%s has no %s buffer" % (e.spec.name, which))
      buf = self.state
    elif which == "value":
      buf = self.params
    elif which == "grad":
      buf = self.grads
    else:
      buf = self.slot(which)
    return buf[e.offset:e.offset + e.size]

  def view(self, name: str, which: str = "value") -> torch.Tensor:
    """Shaped view (no copy) of a variable, its gradient, or an optimizer slot."""
    e = self.entries[name]
    return self._flat(e, which).view(e.spec.shape)

  def slot(self, name: str) -> torch.Tensor:
    if name not in self.slots:
      self.slots[name] = torch.zeros_like(self.params)
    return self.slots[name]

  @property
  def trainable(self) -> List[VarEntry]:
    return [e for e in self.entries.values() if e.spec.trainable]

  @property
  def non_trainable(self) -> List[VarEntry]:
    return [e for e in self.entries.values() if not e.spec.trainable]

  # -- init / copy -------------------------------------------------------------
  def initialize(self, seed: int = 0):
    gen = torch.Generator(device="cpu")
    gen.manual_seed(int(seed))
    p = torch.zeros(self.num_elements)
    s = torch.zeros(max(self.num_state_elements, 1))
    for e in self.entries.values():
      val = init_variable(e.spec, gen).reshape(-1)
      (p if e.spec.trainable else s)[e.offset:e.offset + e.size] = val
    self.params.copy_(p)
    self.state.copy_(s)
    self.grads.zero_()
    for t in self.slots.values():
      t.zero_()
    self.version += 1

  def l2_table(self) -> List[Tuple[int, int, float]]:
    """(offset, size, λ) for every trainable variable, in layout order."""
    return [(e.offset, e.size, e.spec.l2) for e in self.trainable]

  def l2_vector(self) -> torch.Tensor:
    """Per-element 2λ (the gradient coefficient of λ·Σw²); built once, cached."""
    if getattr(self, "_l2vec", None) is None:
      v = torch.zeros(self.num_elements)
      for off, size, lam in self.l2_table():
        if lam:
          v[off:off + size] = 2.0 * lam
      self._l2vec = v.to(self.device)
    return self._l2vec

  def regularization_loss(self) -> torch.Tensor:
    """Σ λ·‖w‖² over regularised variables (what Keras adds to the reported loss)."""
    return 0.5 * torch.dot(self.l2_vector(), self.params * self.params)

  # -- buckets -----------------------------------------------------------------
  TAIL_BUCKET_BYTES = 2 << 20

  def make_buckets(self, num_packs: int = 1, bucket_bytes: Optional[int] = None) -> List[Bucket]:
    """Splits the gradient buffer into contiguous fusion buckets.

    Buckets are returned in *readiness order*: backward produces the gradients of
    the last layers first, so bucket 0 is the tail of the buffer.  `num_packs`
    equal-sized packs (reference flag `--num_packs`), or a byte budget per bucket
    (Horovod's fusion threshold); `num_packs=0` gives one bucket per variable.

    With several buckets on a large model the last one is additionally cut so that what
    completes at the very end of the backward pass (the first layers of the network) is a small
    bucket of its own: its collective is the only one that cannot overlap any compute, so it
    should be latency-sized (`TAIL_BUCKET_BYTES`), not a full pack.
    """
    tv = self.trainable
    if not tv:
      return []
    total = self.num_elements
    if bucket_bytes:
      target = max(int(bucket_bytes) // 4, 1)
    elif num_packs and num_packs > 0:
      target = -(-total // num_packs)
    else:
      target = 1
    tail = self.TAIL_BUCKET_BYTES // 4
    split_tail = (bucket_bytes or (num_packs and num_packs > 1)) and total > 8 * tail \
        and target > 2 * tail
    buckets: List[Bucket] = []
    cur: List[VarEntry] = []
    cur_n = 0
    left = total
    for e in reversed(tv):
      if split_tail and cur and left <= tail:
        buckets.append(self._mk_bucket(len(buckets), cur))   # what follows is the tail bucket
        cur, cur_n = [], 0
        split_tail = False
      cur.append(e)
      cur_n += e.padded
      left -= e.padded
      if cur_n >= target:
        buckets.append(self._mk_bucket(len(buckets), cur))
        cur, cur_n = [], 0
    if cur:
      buckets.append(self._mk_bucket(len(buckets), cur))
    return buckets

  @staticmethod
  def _mk_bucket(idx: int, entries: List[VarEntry]) -> Bucket:
    start = min(e.offset for e in entries)
    end = max(e.offset + e.padded for e in entries)
    return Bucket(idx, start, end, [e.spec.name for e in entries])

  # -- named weights -----------------------------------------------------------
  def get_weights(self) -> Dict[str, torch.Tensor]:
    return {n: self.view(n).detach().cpu().clone() for n in self.entries}

  def set_weights(self, weights: Dict[str, "torch.Tensor"], strict: bool = True):
    missing = [n for n in self.entries if n not in weights]
    if strict and missing:
      raise KeyError("missing variables: %s" % missing[:5])
    for n, w in weights.items():
      if n not in self.entries:
        if strict:
          raise KeyError("unexpected variable %r" % n)
        continue
      w = torch.as_tensor(w, dtype=torch.float32)
      if tuple(w.shape) != tuple(self.entries[n].spec.shape):
        raise ValueError("shape mismatch for %s: %s vs %s"
                         % (n, tuple(w.shape), self.entries[n].spec.shape))
      self.view(n).copy_(w)
    self.version += 1
