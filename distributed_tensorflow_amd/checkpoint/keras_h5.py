"""Keras-layout `.h5` checkpoints (full model: weights + optimizer state).

The reference's Horovod drivers save `tf.keras.callbacks.ModelCheckpoint('./checkpoint-{epoch}.h5')`
on rank 0 (`resnet_cifar_main_horovod.py:235-237`), i.e. Keras full-model HDF5 files:

    /                       attrs: keras_version, backend, model_config, training_config
    /model_weights          attrs: layer_names, backend, keras_version
        /<layer>            attrs: weight_names = [b'<layer>/kernel:0', ...]
            /<layer>/kernel:0, /<layer>/gamma:0, ...     (datasets, Keras layouts)
    /optimizer_weights      attrs: weight_names = [b'SGD/iter:0', b'SGD/<layer>/<var>/momentum:0', ...]

Variables keep the Keras layouts (conv HWIO, dense [in, out], BatchNorm order gamma, beta,
moving_mean, moving_variance), so `model.load_weights(path, by_name=True)` of the stock Keras
`resnet50` / `resnet56` definitions reads files written here, and files written by Keras
load here; `model_config` is a functional-model description in the `Model.get_config()` schema
(see `_model_config`).  Checked against the HDF5 specification, the Keras source and this
module's own reader; parity with a real Keras / h5py is unpinned - neither is available in the
build environment and the reference ships no `.h5` file.
"""
from __future__ import annotations

import json
from typing import Dict, List

import numpy as np
import torch

from . import hdf5

__all__ = ["save_model", "load_model", "read_weights", "KERAS_VERSION"]

KERAS_VERSION = b"2.2.4-tf"
_BN_ORDER = ("gamma", "beta", "moving_mean", "moving_variance")
_DEFAULT_ORDER = ("kernel", "bias") + _BN_ORDER


def _layer_vars(spec) -> Dict[str, List[str]]:
  out: Dict[str, List[str]] = {}
  for v in spec.variables:
    out.setdefault(v.layer, []).append(v.var)
  for layer, vs in out.items():
    vs.sort(key=lambda n: _DEFAULT_ORDER.index(n) if n in _DEFAULT_ORDER else 99)
  return out


def _model_config(spec, initializers: bool = True) -> str:
  """`model_config` of a Keras 2.2.4-tf functional model (`Model.get_config()` schema: every
  layer with `name`, `class_name`, `config` and `inbound_nodes` = [[[source layer, node index,
  tensor index, {}], ...]], plus `input_layers` / `output_layers`), equivalent to the reference's
  builders (`resnet_model.py:224-389`, `resnet_cifar_model.py:192-256`, `trivial_model.py:26-41`).

  The graph IR fuses what Keras keeps as separate layers, so a `bn` op expands into
  `BatchNormalization` (+ `Add` with its residual) (+ `Activation('relu')`), a convolution with
  an explicit border into `ZeroPadding2D` + `Conv2D(padding='valid')`, and the model ends with
  `Activation('softmax')` as the reference's do.  Layer configs list the fields that differ from
  the Keras constructors' defaults (`from_config` fills in the rest), which keeps the attribute
  under the 64 KB a classic HDF5 object-header message can hold.  Two deliberate differences:
  the global mean is a `GlobalAveragePooling2D` named `reduce_mean` (the reference's `Lambda`
  serialises Python byte code) and the fp32 cast in front of the softmax is the softmax layer's
  `dtype`.  `initializers=False` leaves the kernel initialisers out (they do not matter to a
  model whose weights are loaded; the 110-layer CIFAR ResNet needs the room).  No Keras / h5py is available where this is built and the reference ships no `.h5`
  file: the schema follows the Keras source, loading by a stock Keras is unpinned."""
  layers = [{"name": "input_1", "class_name": "InputLayer",
             "config": {"batch_input_shape": [None] + list(spec.input_shape), "dtype": "float32",
                        "sparse": False, "name": "input_1"}, "inbound_nodes": []}]
  src = {spec.input_name: "input_1"}          # tensor name -> Keras layer that produces it
  counts = {"add": 0, "activation": 0}

  def emit(name, cls, cfg, inputs):
    cfg = dict(cfg, name=name)
    layers.append({"name": name, "class_name": cls, "config": cfg,
                   "inbound_nodes": [[[src_name, 0, 0, {}] for src_name in inputs]]})
    return name

  def auto(kind):
    counts[kind] += 1
    return kind if counts[kind] == 1 else "%s_%d" % (kind, counts[kind] - 1)

  def l2(coef):
    return {"class_name": "L1L2", "config": {"l1": 0.0, "l2": float(coef)}} if coef else None

  for op in spec.ops:
    at = op.attrs
    x = src[op.inputs[0]]
    if op.kind == "conv":
      padding = at.get("padding", "same")
      if padding == "valid" and at.get("explicit_pad", 0):
        e = int(at["explicit_pad"])
        x = emit(op.name + "_pad", "ZeroPadding2D", {"padding": [[e, e], [e, e]]}, [x])
      cfg = {"filters": at["cout"], "kernel_size": [at["kh"], at["kw"]],
             "strides": [at["stride"], at["stride"]], "padding": padding, "use_bias": False}
      if initializers:
        cfg["kernel_initializer"] = {"class_name": "VarianceScaling", "config": {
            "scale": 2.0, "mode": "fan_in", "distribution": "truncated_normal", "seed": None}}
      if at.get("l2"):
        cfg["kernel_regularizer"] = l2(at["l2"])
      src[op.output] = emit(op.name, "Conv2D", cfg, [x])
    elif op.kind == "bn":
      y = emit(op.name, "BatchNormalization", {"axis": [3], "momentum": at["momentum"],
                                               "epsilon": at["eps"]}, [x])
      if len(op.inputs) > 1:
        y = emit(auto("add"), "Add", {}, [y, src[op.inputs[1]]])
      if at.get("relu"):
        y = emit(auto("activation"), "Activation", {"activation": "relu"}, [y])
      src[op.output] = y
    elif op.kind == "maxpool":
      src[op.output] = emit(op.name, "MaxPooling2D", {
          "pool_size": [at["k"], at["k"]], "strides": [at["stride"], at["stride"]],
          "padding": "same"}, [x])
    elif op.kind == "mean":
      src[op.output] = emit(op.name, "GlobalAveragePooling2D", {}, [x])
    elif op.kind == "flatten":
      src[op.output] = emit(op.name, "Flatten", {}, [x])
    elif op.kind == "dense":
      cfg = {"units": at["out_features"]}
      if initializers and at.get("init", "normal0.01") == "normal0.01":
        cfg["kernel_initializer"] = {"class_name": "RandomNormal", "config": {
            "mean": 0.0, "stddev": 0.01, "seed": None}}
      if at.get("l2"):
        cfg["kernel_regularizer"] = l2(at["l2"])
        cfg["bias_regularizer"] = l2(at["l2"])
      src[op.output] = emit(op.name, "Dense", cfg, [x])
    else:
      raise ValueError("no Keras layer for op kind %r" % op.kind)
  out = emit("softmax" if all(l["name"] != "softmax" for l in layers) else "softmax_out",
             "Activation", {"activation": "softmax", "dtype": "float32"}, [src[spec.logits_name]])
  cfg = {"class_name": "Model", "config": {
      "name": spec.name, "layers": layers, "input_layers": [["input_1", 0, 0]],
      "output_layers": [[out, 0, 0]]}}
  return json.dumps(cfg, separators=(",", ":"))


def save_model(path: str, model, include_optimizer: bool = True):
  """Writes a Keras-style full-model (or weights-only) HDF5 file."""
  spec, store = model.spec, model.store
  w = hdf5.Writer(path)
  root = w.root
  root.attrs["keras_version"] = KERAS_VERSION
  root.attrs["backend"] = b"tensorflow"
  # an attribute is one object-header message: 64 KB at most in the classic file format.  The
  # 110-layer CIFAR ResNet (390 Keras layers, 81 KB) does not fit either way: its files carry
  # weights and optimizer state only (what `load_weights` reads)
  cfg = _model_config(spec)
  if len(cfg) >= 64000:
    cfg = _model_config(spec, initializers=False)
  if len(cfg) < 64000:
    root.attrs["model_config"] = cfg.encode("utf8")
  mw = root.create_group("model_weights")
  layers = _layer_vars(spec)
  mw.attrs["layer_names"] = [n.encode("utf8") for n in layers]
  mw.attrs["backend"] = b"tensorflow"
  mw.attrs["keras_version"] = KERAS_VERSION
  for layer, vs in layers.items():
    g = mw.create_group(layer)
    g.attrs["weight_names"] = [("%s/%s:0" % (layer, v)).encode("utf8") for v in vs]
    for v in vs:
      arr = store.view("%s/%s" % (layer, v)).detach().cpu().numpy().astype(np.float32)
      g.create_dataset("%s/%s:0" % (layer, v), arr)
  opt = getattr(model, "optimizer", None)
  if include_optimizer and opt is not None:
    root.attrs["training_config"] = json.dumps({
        "optimizer_config": {"class_name": "SGD", "config": opt.get_config()},
        "loss": model.loss, "metrics": model.metrics_names[1:]}).encode("utf8")
    ow = root.create_group("optimizer_weights")
    names = [b"SGD/iter:0"]
    ow.create_dataset("SGD/iter:0", np.array(int(opt.iterations), dtype=np.int64))
    if "momentum" in store.slots:
      for e in store.trainable:
        n = "SGD/%s/momentum:0" % e.spec.name
        names.append(n.encode("utf8"))
        ow.create_dataset(n, store.view(e.spec.name, "momentum").detach().cpu().numpy())
    ow.attrs["weight_names"] = names
  w.close()


def _decode(x) -> str:
  return x.decode("utf8") if isinstance(x, (bytes, np.bytes_)) else str(x)


def read_weights(path: str) -> Dict[str, np.ndarray]:
  """Returns {'<layer>/<var>': ndarray} from a Keras HDF5 file (full model or weights only)."""
  f = hdf5.File(path)
  g = f["model_weights"] if "model_weights" in f else f
  out: Dict[str, np.ndarray] = {}
  layer_names = g.attrs.get("layer_names")
  layers = [_decode(n) for n in np.asarray(layer_names).reshape(-1)] if layer_names is not None \
      else g.keys()
  for layer in layers:
    if layer not in g:
      continue
    lg = g[layer]
    wn = lg.attrs.get("weight_names")
    names = [_decode(n) for n in np.asarray(wn).reshape(-1)] if wn is not None else \
        [p for p, _ in lg.visit_datasets()]
    for n in names:
      ds = lg[n]
      key = n[:-2] if n.endswith(":0") else n
      parts = key.split("/")
      # Keras may prefix nested scopes; the variable is the last part, the layer the group
      out["%s/%s" % (layer, parts[-1])] = np.asarray(ds.read())
  return out


def load_model(path: str, model, load_optimizer: bool = True, strict: bool = True):
  """Restores variables (and SGD slots / iteration count) into `model`."""
  weights = read_weights(path)
  store = model.store
  missing = [n for n in store.entries if n not in weights]
  if strict and missing:
    raise KeyError("checkpoint %s lacks variables: %s" % (path, missing[:5]))
  for n, arr in weights.items():
    if n not in store.entries:
      if strict:
        raise KeyError("checkpoint variable %s does not exist in model %s" % (n, model.name))
      continue
    t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32))
    if tuple(t.shape) != tuple(store.entries[n].spec.shape):
      raise ValueError("shape mismatch for %s: checkpoint %s vs model %s"
                       % (n, tuple(t.shape), store.entries[n].spec.shape))
    store.view(n).copy_(t)
  store.version += 1
  opt = getattr(model, "optimizer", None)
  if load_optimizer and opt is not None:
    f = hdf5.File(path)
    if "optimizer_weights" in f:
      ow = f["optimizer_weights"]
      for p, ds in ow.visit_datasets():
        if p.endswith("iter:0"):
          opt.iterations = int(np.asarray(ds.read()).reshape(-1)[0])
        elif p.endswith("/momentum:0"):
          var = p[:-len("/momentum:0")].split("/", 1)[1]
          if var in store.entries:
            store.view(var, "momentum").copy_(torch.from_numpy(
                np.ascontiguousarray(ds.read(), dtype=np.float32)))
