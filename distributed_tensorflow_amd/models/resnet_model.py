"""ResNet-50 v1.5 for ImageNet as a layer graph.

Architecture parity with the reference Keras builder
(`image_classification/resnet_model.py:224-389`): explicit 3-pixel zero pad +
7x7/2 VALID stem, 3x3/2 SAME max-pool, bottleneck stages [3, 4, 6, 3] with the
stride on the 3x3 convolution (`:172-176`, the "v1.5" variant) and on the
projection shortcut (`:204-211`), global mean pool, `fc1000` with L2 on kernel
and bias (`:375-381`).  Layer names follow the reference grammar
`res{stage}{block}_branch{2a|2b|2c|1}` / `bn{stage}{block}_branch...` (`:70-71`).
"""
from __future__ import annotations

from .graph import GraphBuilder, ModelSpec

L2_WEIGHT_DECAY = 1e-4      # reference resnet_model.py:37
BATCH_NORM_DECAY = 0.9      # :38
BATCH_NORM_EPSILON = 1e-5   # :39

_STAGES = ((2, (64, 64, 256), "abc", 1),
           (3, (128, 128, 512), "abcd", 2),
           (4, (256, 256, 1024), "abcdef", 2),
           (5, (512, 512, 2048), "abc", 2))


def _bn(g, x, name, relu=False, residual=None):
  return g.bn(x, name, BATCH_NORM_DECAY, BATCH_NORM_EPSILON, relu=relu, residual=residual)


def _bottleneck(g: GraphBuilder, x: str, filters, stage: int, block: str, stride: int,
                project: bool, l2: float) -> str:
  f1, f2, f3 = filters
  cn = "res%d%s_branch" % (stage, block)
  bnn = "bn%d%s_branch" % (stage, block)
  y = g.conv(x, cn + "2a", f1, 1, l2=l2)
  y = _bn(g, y, bnn + "2a", relu=True)
  y = g.conv(y, cn + "2b", f2, 3, stride=stride, padding="same", l2=l2)
  y = _bn(g, y, bnn + "2b", relu=True)
  y = g.conv(y, cn + "2c", f3, 1, l2=l2)
  if project:
    s = g.conv(x, cn + "1", f3, 1, stride=stride, l2=l2)
    s = _bn(g, s, bnn + "1")
  else:
    s = x
  # bn2c + add + relu is one fused unit in the engine.
  return _bn(g, y, bnn + "2c", relu=True, residual=s)


def resnet50(num_classes: int = 1001, use_l2_regularizer: bool = True) -> ModelSpec:
  l2 = L2_WEIGHT_DECAY if use_l2_regularizer else 0.0
  g = GraphBuilder("resnet50", (224, 224, 3), num_classes)
  x = g.conv("input", "conv1", 64, 7, stride=2, padding="valid", explicit_pad=3, l2=l2)
  x = _bn(g, x, "bn_conv1", relu=True)
  x = g.maxpool(x, "max_pooling2d", 3, 2)
  for stage, filters, blocks, stride in _STAGES:
    for i, b in enumerate(blocks):
      x = _bottleneck(g, x, filters, stage, b, stride if i == 0 else 1, i == 0, l2)
  x = g.global_mean(x)
  x = g.dense(x, "fc1000", num_classes, l2=l2, out="logits")
  return g.build(x)
