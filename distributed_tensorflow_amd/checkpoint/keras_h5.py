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
load here.  (Checked against the HDF5 specification and this module's own reader; a
round-trip through h5py itself is not exercised in the test-suite because h5py is not
available in the build environment.)
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


def _model_config(spec) -> str:
  cfg = {"class_name": "Model", "config": {"name": spec.name, "layers": [
      {"name": op.name, "class_name": {"conv": "Conv2D", "bn": "BatchNormalization",
                                       "maxpool": "MaxPooling2D", "mean": "Lambda",
                                       "flatten": "Flatten", "dense": "Dense"}.get(op.kind, op.kind),
       "inbound": list(op.inputs)} for op in spec.ops],
      "input_shape": list(spec.input_shape), "num_classes": spec.num_classes}}
  return json.dumps(cfg)


def save_model(path: str, model, include_optimizer: bool = True):
  """Writes a Keras-style full-model (or weights-only) HDF5 file."""
  spec, store = model.spec, model.store
  w = hdf5.Writer(path)
  root = w.root
  root.attrs["keras_version"] = KERAS_VERSION
  root.attrs["backend"] = b"tensorflow"
  cfg = _model_config(spec)
  if len(cfg) < 60000:
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
