"""Static layer-graph IR shared by every executor.

A model is a topologically ordered list of `Op`s over named NHWC tensors plus a
list of `VarSpec`s that carry the Keras variable names and layouts (conv kernels
HWIO `[kh,kw,cin,cout]`, dense kernels `[in,out]`, BatchNorm
`gamma, beta, moving_mean, moving_variance`), so that checkpoints round-trip with
the Keras definitions in the reference (`image_classification/resnet_model.py:224-389`,
`resnet_cifar_model.py:192-256`).

Activation and residual add are attributes of the `bn` op (`relu`, second input)
because that is the unit the native engine fuses: `dst = relu(bn(src) + residual)`.
The graph is what both back ends consume: the PyTorch reference executor
interprets it with library ops, the HIP engine compiles it into a launch plan.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Any, Dict, List, Optional, Sequence, Tuple

__all__ = ["Op", "VarSpec", "ModelSpec", "GraphBuilder", "same_padding"]


def same_padding(in_size: int, k: int, stride: int) -> Tuple[int, int]:
  """TensorFlow `SAME` padding: (before, after). The extra pixel goes after.

  A 3x3/2 window on an even extent therefore pads (0, 1), unlike the symmetric
  padding=1 of most other frameworks; Keras weights only round-trip with this rule.
  """
  out = -(-in_size // stride)
  total = max((out - 1) * stride + k - in_size, 0)
  return total // 2, total - total // 2


@dataclasses.dataclass
class VarSpec:
  name: str                 # "<layer>/<var>", e.g. "res2a_branch2a/kernel"
  shape: Tuple[int, ...]
  init: str                 # he_normal | normal0.01 | zeros | ones
  trainable: bool = True
  l2: float = 0.0           # coefficient λ of the λ·Σw² regulariser (0 = none)
  fan_in: int = 0

  @property
  def layer(self) -> str:
    return self.name.split("/")[0]

  @property
  def var(self) -> str:
    return self.name.split("/")[1]

  @property
  def size(self) -> int:
    return int(math.prod(self.shape))


@dataclasses.dataclass
class Op:
  kind: str                 # conv | bn | maxpool | mean | flatten | dense | transpose
  name: str
  inputs: List[str]
  output: str
  attrs: Dict[str, Any] = dataclasses.field(default_factory=dict)


@dataclasses.dataclass
class ModelSpec:
  name: str
  input_shape: Tuple[int, int, int]        # H, W, C (NHWC, batch excluded)
  num_classes: int
  ops: List[Op]
  variables: List[VarSpec]
  shapes: Dict[str, Tuple[int, ...]]       # tensor name -> shape without batch
  input_name: str = "input"
  logits_name: str = "logits"

  def var(self, name: str) -> VarSpec:
    for v in self.variables:
      if v.name == name:
        return v
    raise KeyError(name)

  @property
  def trainable_variables(self) -> List[VarSpec]:
    return [v for v in self.variables if v.trainable]

  @property
  def non_trainable_variables(self) -> List[VarSpec]:
    return [v for v in self.variables if not v.trainable]

  def count_params(self, trainable: Optional[bool] = None) -> int:
    return sum(v.size for v in self.variables
               if trainable is None or v.trainable == trainable)

  def layer_names(self) -> List[str]:
    seen, out = set(), []
    for v in self.variables:
      if v.layer not in seen:
        seen.add(v.layer)
        out.append(v.layer)
    return out

  def macs_per_image(self) -> int:
    """Forward multiply-accumulates per image (conv + dense)."""
    total = 0
    for op in self.ops:
      if op.kind == "conv":
        a = op.attrs
        ho, wo, _ = self.shapes[op.output]
        total += ho * wo * a["cout"] * a["kh"] * a["kw"] * a["cin"]
      elif op.kind == "dense":
        total += op.attrs["in_features"] * op.attrs["out_features"]
    return total


class GraphBuilder:
  """Small helper that tracks tensor shapes while a model function emits ops."""

  def __init__(self, name: str, input_shape: Sequence[int], num_classes: int):
    self.name = name
    self.input_shape = tuple(int(x) for x in input_shape)
    self.num_classes = num_classes
    self.ops: List[Op] = []
    self.vars: List[VarSpec] = []
    self.shapes: Dict[str, Tuple[int, ...]] = {"input": self.input_shape}
    self._n = 0

  def _tmp(self, base: str) -> str:
    self._n += 1
    return "%s:%d" % (base, self._n)

  # -- layers ---------------------------------------------------------------
  def conv(self, x: str, name: str, filters: int, kernel: int, stride: int = 1,
           padding: str = "same", explicit_pad: int = 0, l2: float = 0.0) -> str:
    """`ZeroPadding2D(explicit_pad)` (optional) followed by `Conv2D(use_bias=False)`."""
    h, w, c = self.shapes[x]
    if padding == "same":
      pt, pb = same_padding(h, kernel, stride)
      pl, pr = same_padding(w, kernel, stride)
    elif padding == "valid":
      pt = pb = pl = pr = explicit_pad
    else:
      raise ValueError(padding)
    ho = (h + pt + pb - kernel) // stride + 1
    wo = (w + pl + pr - kernel) // stride + 1
    out = name + ":out"
    self.vars.append(VarSpec(name + "/kernel", (kernel, kernel, c, filters), "he_normal",
                             True, l2, fan_in=kernel * kernel * c))
    self.ops.append(Op("conv", name, [x], out, dict(
        kh=kernel, kw=kernel, cin=c, cout=filters, stride=stride,
        pad=(pt, pb, pl, pr), padding=padding, explicit_pad=explicit_pad, l2=l2)))
    self.shapes[out] = (ho, wo, filters)
    return out

  def bn(self, x: str, name: str, momentum: float, eps: float, relu: bool = False,
         residual: Optional[str] = None) -> str:
    h, w, c = self.shapes[x]
    out = name + ":out"
    self.vars.append(VarSpec(name + "/gamma", (c,), "ones"))
    self.vars.append(VarSpec(name + "/beta", (c,), "zeros"))
    self.vars.append(VarSpec(name + "/moving_mean", (c,), "zeros", trainable=False))
    self.vars.append(VarSpec(name + "/moving_variance", (c,), "ones", trainable=False))
    inputs = [x] + ([residual] if residual is not None else [])
    if residual is not None and self.shapes[residual] != (h, w, c):
      raise ValueError("residual shape %s != %s" % (self.shapes[residual], (h, w, c)))
    self.ops.append(Op("bn", name, inputs, out, dict(
        channels=c, momentum=momentum, eps=eps, relu=relu)))
    self.shapes[out] = (h, w, c)
    return out

  def maxpool(self, x: str, name: str, k: int = 3, stride: int = 2) -> str:
    h, w, c = self.shapes[x]
    pt, pb = same_padding(h, k, stride)
    pl, pr = same_padding(w, k, stride)
    ho = (h + pt + pb - k) // stride + 1
    wo = (w + pl + pr - k) // stride + 1
    out = name + ":out"
    self.ops.append(Op("maxpool", name, [x], out, dict(k=k, stride=stride,
                                                      pad=(pt, pb, pl, pr))))
    self.shapes[out] = (ho, wo, c)
    return out

  def global_mean(self, x: str, name: str = "reduce_mean") -> str:
    h, w, c = self.shapes[x]
    out = name + ":out"
    self.ops.append(Op("mean", name, [x], out, dict(h=h, w=w, channels=c)))
    self.shapes[out] = (c,)
    return out

  def flatten(self, x: str, name: str = "flatten") -> str:
    n = int(math.prod(self.shapes[x]))
    out = name + ":out"
    self.ops.append(Op("flatten", name, [x], out, {}))
    self.shapes[out] = (n,)
    return out

  def dense(self, x: str, name: str, units: int, l2: float = 0.0,
            init: str = "normal0.01", out: Optional[str] = None) -> str:
    (n,) = self.shapes[x]
    out = out or (name + ":out")
    self.vars.append(VarSpec(name + "/kernel", (n, units), init, True, l2, fan_in=n))
    self.vars.append(VarSpec(name + "/bias", (units,), "zeros", True, l2))
    self.ops.append(Op("dense", name, [x], out, dict(in_features=n, out_features=units, l2=l2,
                                                     init=init)))
    self.shapes[out] = (units,)
    return out

  def build(self, logits: str) -> ModelSpec:
    return ModelSpec(self.name, self.input_shape, self.num_classes, self.ops, self.vars,
                     self.shapes, "input", logits)
