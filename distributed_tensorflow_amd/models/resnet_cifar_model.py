"""ResNet-(6n+2) for CIFAR-10 (ResNet-20/32/56/110) as a layer graph.

Parity with the reference builder (`image_classification/resnet_cifar_model.py:192-256`):
1-pixel zero pad + 3x3 VALID stem with 16 filters, three stages of `num_blocks`
basic blocks with 16/32/64 filters, stride 2 on the *first* 3x3 of stages 3 and 4
(`:127`) and a 1x1 projection shortcut in the first block of every stage
(`:145-151`, also in stage 2 where the stride is 1), global mean pool and `fc10`.
Block names follow `res{stage}block_{i}_branch{2a|2b|1}` (`:65-66,184-188`).

The reference also exports `resnet10 = partial(resnet, num_blocks=110)` (`:262`),
a typo for ResNet-110 (18 blocks); `resnet110` here is the 18-block network.
"""
from __future__ import annotations

import functools

from .graph import GraphBuilder, ModelSpec

BATCH_NORM_DECAY = 0.997    # reference resnet_cifar_model.py:34
BATCH_NORM_EPSILON = 1e-5   # :35
L2_WEIGHT_DECAY = 2e-4      # :36


def _bn(g, x, name, relu=False, residual=None):
  return g.bn(x, name, BATCH_NORM_DECAY, BATCH_NORM_EPSILON, relu=relu, residual=residual)


def _basic_block(g: GraphBuilder, x: str, filters: int, stage: int, block: str,
                 stride: int, project: bool) -> str:
  cn = "res%d%s_branch" % (stage, block)
  bnn = "bn%d%s_branch" % (stage, block)
  y = g.conv(x, cn + "2a", filters, 3, stride=stride, padding="same", l2=L2_WEIGHT_DECAY)
  y = _bn(g, y, bnn + "2a", relu=True)
  y = g.conv(y, cn + "2b", filters, 3, padding="same", l2=L2_WEIGHT_DECAY)
  if project:
    s = g.conv(x, cn + "1", filters, 1, stride=stride, l2=L2_WEIGHT_DECAY)
    s = _bn(g, s, bnn + "1")
  else:
    s = x
  return _bn(g, y, bnn + "2b", relu=True, residual=s)


def resnet(num_blocks: int, classes: int = 10) -> ModelSpec:
  g = GraphBuilder("resnet%d" % (6 * num_blocks + 2), (32, 32, 3), classes)
  x = g.conv("input", "conv1", 16, 3, stride=1, padding="valid", explicit_pad=1,
             l2=L2_WEIGHT_DECAY)
  x = _bn(g, x, "bn_conv1", relu=True)
  for stage, filters, stride in ((2, 16, 1), (3, 32, 2), (4, 64, 2)):
    for i in range(num_blocks):
      x = _basic_block(g, x, filters, stage, "block_%d" % i, stride if i == 0 else 1, i == 0)
  x = g.global_mean(x)
  x = g.dense(x, "fc10", classes, l2=L2_WEIGHT_DECAY, out="logits")
  return g.build(x)


resnet20 = functools.partial(resnet, num_blocks=3)
resnet32 = functools.partial(resnet, num_blocks=5)
resnet56 = functools.partial(resnet, num_blocks=9)
resnet110 = functools.partial(resnet, num_blocks=18)
