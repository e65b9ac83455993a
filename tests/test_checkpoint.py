"""Pure-Python HDF5 writer/reader and the Keras-layout checkpoint round trip."""
import struct

import numpy as np
import pytest
import torch

from distributed_tensorflow_amd import models
from distributed_tensorflow_amd.checkpoint import hdf5, keras_h5
from distributed_tensorflow_amd.train.model import Model
from distributed_tensorflow_amd.train.optimizers import SGD


def test_hdf5_roundtrip(tmp_path):
  p = str(tmp_path / "t.h5")
  w = hdf5.Writer(p)
  w.root.attrs["title"] = b"hello"
  w.root.attrs["version"] = np.int64(7)
  g = w.root.create_group("a/b")
  g.attrs["names"] = [b"x:0", b"longer_name:0"]
  a = np.arange(24, dtype=np.float32).reshape(2, 3, 4)
  g.create_dataset("x:0", a)
  g.create_dataset("nested/y", np.array([1, 2, 3], dtype=np.int64))
  w.root.create_dataset("scalar", np.array(3.5, dtype=np.float64))
  many = w.root.create_group("many")
  for i in range(300):                      # more links than one symbol-table node holds
    many.create_dataset("d%03d" % i, np.full((2,), i, dtype=np.float32))
  w.close()
  raw = open(p, "rb").read()
  assert raw[:8] == b"\x89HDF\r\n\x1a\n"
  assert struct.unpack_from("<Q", raw, 40)[0] == len(raw)       # end-of-file address
  f = hdf5.File(p)
  assert f.attrs["title"] == b"hello" and int(f.attrs["version"]) == 7
  assert sorted(f.keys()) == ["a", "many", "scalar"]
  b = f["a/b"]
  assert [bytes(n) for n in b.attrs["names"]] == [b"x:0", b"longer_name:0"]
  assert np.array_equal(b["x:0"].read(), a) and b["x:0"].shape == (2, 3, 4)
  assert np.array_equal(f["a/b/nested/y"][...], [1, 2, 3])
  assert float(f["scalar"].read()) == 3.5
  assert len(f["many"].keys()) == 300
  assert f["many/d123"].read()[0] == 123
  with pytest.raises(KeyError):
    f["a/missing"]


def test_keras_checkpoint_roundtrip(tmp_path):
  spec = models.resnet20()
  m = Model(spec, dtype="fp32", device=torch.device("cpu"), seed=3)
  opt = SGD(0.1, momentum=0.9)
  m.compile(opt, "categorical_crossentropy", ["categorical_accuracy"])
  # take two steps so that momentum slots and moving statistics are non-trivial
  g = torch.Generator().manual_seed(0)
  x = torch.randn(4, 32, 32, 3, generator=g)
  y = torch.nn.functional.one_hot(torch.randint(0, 10, (4,), generator=g), 10).float()
  m.train_on_batch(x, y)
  m.train_on_batch(x, y)
  path = str(tmp_path / "checkpoint-1.h5")
  m.save(path)
  f = hdf5.File(path)
  assert f.attrs["backend"] == b"tensorflow"
  mw = f["model_weights"]
  names = [bytes(n).decode() for n in mw.attrs["layer_names"]]
  assert names[0] == "conv1" and "fc10" in names and "bn2block_0_branch2a" in names
  assert [bytes(n) for n in mw["bn_conv1"].attrs["weight_names"]] == [
      b"bn_conv1/gamma:0", b"bn_conv1/beta:0", b"bn_conv1/moving_mean:0",
      b"bn_conv1/moving_variance:0"]
  k = mw["conv1/conv1/kernel:0"].read()
  assert k.shape == (3, 3, 3, 16)                       # HWIO
  assert mw["fc10/fc10/kernel:0"].shape == (64, 10)      # [in, out]
  assert np.array_equal(k, m.store.view("conv1/kernel").numpy())
  ow = f["optimizer_weights"]
  assert int(ow["SGD/iter:0"].read()) == 2
  # restore into a differently initialised model
  m2 = Model(spec, dtype="fp32", device=torch.device("cpu"), seed=99)
  m2.compile(SGD(0.1, momentum=0.9), "categorical_crossentropy", ["categorical_accuracy"])
  assert not torch.equal(m2.store.params, m.store.params)
  m2.load_weights(path)
  assert torch.equal(m2.store.params, m.store.params)
  assert torch.equal(m2.store.state, m.store.state)
  assert torch.equal(m2.store.slot("momentum"), m.store.slot("momentum"))
  assert m2.optimizer.iterations == 2
  # and the restored model continues identically
  l1, _ = m.train_on_batch(x, y)
  l2, _ = m2.train_on_batch(x, y)
  assert l1.item() == pytest.approx(l2.item(), rel=1e-6)
  # weights-only file
  wpath = str(tmp_path / "w.h5")
  m.save_weights(wpath)
  assert "optimizer_weights" not in hdf5.File(wpath)
  assert set(keras_h5.read_weights(wpath)) == set(m.store.entries)


def test_model_config_is_a_keras_functional_model_description(tmp_path):
  """`model_config` follows the `Model.get_config()` schema of Keras 2.2.4-tf: every layer has
  name / class_name / config / inbound_nodes, sources precede their consumers, the fused IR ops
  are expanded into the reference's Keras layers (`resnet_model.py:224-389`: 53 Conv2D, 53
  BatchNormalization, 16 Add, ReLU activations, `conv1_pad`, `fc1000`, a final softmax), and
  the attribute survives the HDF5 round trip.  (Loading it with a real Keras is unpinned: none
  is installed and the reference ships no `.h5` file.)"""
  import collections
  import json
  from distributed_tensorflow_amd import models
  from distributed_tensorflow_amd.checkpoint import hdf5, keras_h5
  from distributed_tensorflow_amd.train.model import Model
  from distributed_tensorflow_amd.train.optimizers import SGD
  for spec, convs, adds in ((models.resnet50(), 53, 16), (models.resnet56(), 58, 27),
                            (models.resnet110(), 112, 54)):
    cfg = json.loads(keras_h5._model_config(spec))
    assert cfg["class_name"] == "Model"
    layers = cfg["config"]["layers"]
    kinds = collections.Counter(l["class_name"] for l in layers)
    assert kinds["Conv2D"] == convs and kinds["BatchNormalization"] == convs and kinds["Add"] == adds
    assert kinds["InputLayer"] == 1 and kinds["ZeroPadding2D"] == 1 and kinds["Dense"] == 1
    seen = set()
    for l in layers:
      assert set(l) == {"name", "class_name", "config", "inbound_nodes"} and l["config"]["name"] == l["name"]
      for node in l["inbound_nodes"]:
        for source, node_index, tensor_index, kwargs in node:
          assert source in seen and node_index == 0 and tensor_index == 0 and kwargs == {}
      assert l["name"] not in seen
      seen.add(l["name"])
    assert cfg["config"]["input_layers"] == [["input_1", 0, 0]]
    assert cfg["config"]["output_layers"] == [[layers[-1]["name"], 0, 0]]
    assert layers[-1]["config"]["activation"] == "softmax"
    by_name = {l["name"]: l for l in layers}
    assert by_name["conv1"]["config"]["padding"] == "valid"
    assert by_name["conv1"]["inbound_nodes"][0][0][0] == "conv1_pad"
    assert set(spec.layer_names()) <= set(by_name)          # every layer that owns variables
  r50 = {l["name"]: l for l in json.loads(keras_h5._model_config(models.resnet50()))["config"]["layers"]}
  assert r50["conv1_pad"]["config"]["padding"] == [[3, 3], [3, 3]]
  assert r50["res3a_branch2b"]["config"]["strides"] == [2, 2]      # v1.5: the stride sits on the 3x3
  assert r50["res3a_branch2b"]["config"]["kernel_regularizer"]["config"]["l2"] == 1e-4
  assert r50["fc1000"]["config"]["units"] == 1001
  assert r50["fc1000"]["config"]["bias_regularizer"]["config"]["l2"] == 1e-4
  assert r50["bn_conv1"]["config"]["momentum"] == 0.9 and r50["bn_conv1"]["config"]["epsilon"] == 1e-5
  # the attribute is written and read back; the 110-layer model's description (81 KB even without
  # initialisers) exceeds what one HDF5 object-header message holds: its file carries weights and
  # optimizer state only, which `load_weights` restores all the same
  for spec, has_config in ((models.resnet20(), True), (models.resnet110(), False)):
    m = Model(spec, dtype="fp32")
    m.compile(SGD(0.1, momentum=0.9))
    path = str(tmp_path / (spec.name + ".h5"))
    m.save(path)
    attrs = hdf5.File(path).attrs
    assert ("model_config" in attrs) == has_config
    if has_config:
      raw = attrs["model_config"]
      back = json.loads(raw.decode("utf8") if isinstance(raw, bytes) else str(raw))
      assert back["config"]["name"] == spec.name and len(back["config"]["layers"]) > 50
    m2 = Model(spec, dtype="fp32", seed=5)
    m2.compile(SGD(0.1, momentum=0.9))
    m2.load_weights(path)
    assert torch.equal(m2.store.params, m.store.params)
