"""Two-layer "trivial" model used to benchmark the input pipeline and step overhead.

Parity with the reference (`image_classification/trivial_model.py:26-41`):
Flatten -> Dense(1) -> Dense(num_classes) -> fp32 softmax, Keras default
initialisers (glorot uniform), no regulariser.
"""
from __future__ import annotations

from .graph import GraphBuilder, ModelSpec


def trivial_model(num_classes: int = 1001, input_shape=(224, 224, 3)) -> ModelSpec:
  g = GraphBuilder("trivial", input_shape, num_classes)
  x = g.flatten("input")
  x = g.dense(x, "dense", 1, init="glorot_uniform")
  x = g.dense(x, "dense_1", num_classes, init="glorot_uniform", out="logits")
  return g.build(x)
