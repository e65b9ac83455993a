"""Model.fit protocol, callbacks, optimizers (CPU)."""
import os

import numpy as np
import pytest
import torch

from distributed_tensorflow_amd import models
from distributed_tensorflow_amd.data.synthetic import SyntheticDataset
from distributed_tensorflow_amd.engine.params import ParamStore
from distributed_tensorflow_amd.train import callbacks as cb
from distributed_tensorflow_amd.train.model import Model
from distributed_tensorflow_amd.train.optimizers import SGD, LossScaleOptimizer


def _model(loss="sparse_categorical_crossentropy", metric="sparse_categorical_accuracy", opt=None):
  m = Model(models.resnet20(), dtype="fp32", seed=1)
  m.compile(opt or SGD(0.05, momentum=0.9), loss, [metric])
  return m


def test_fit_history_validation_and_callbacks(tmp_path):
  m = _model()
  ds = SyntheticDataset(32, 32, 3, 10, 4)
  events = []

  class Rec(cb.Callback):
    def on_train_begin(self, logs=None): events.append("train_begin")
    def on_epoch_begin(self, epoch, logs=None): events.append("epoch_begin%d" % epoch)
    def on_batch_begin(self, batch, logs=None): events.append("b%d" % batch)
    def on_epoch_end(self, epoch, logs=None): events.append(("epoch_end", sorted(logs)))
    def on_train_end(self, logs=None): events.append("train_end")

  th = cb.TimeHistory(4, 2)
  sched = cb.LearningRateBatchScheduler(lambda e, b, bpe, bs: 0.1 / (1 + e), batch_size=4,
                                        num_images=12)
  ck = cb.ModelCheckpoint(str(tmp_path / "checkpoint-{epoch}.h5"))
  h = m.fit(ds, epochs=2, steps_per_epoch=3, callbacks=[Rec(), th, sched, ck],
            validation_data=ds, validation_steps=1, validation_freq=2, verbose=0)
  assert events[0] == "train_begin" and events[-1] == "train_end"
  assert events.count("b0") == 2 and "epoch_begin1" in events
  assert h.history["loss"][0] > h.history["loss"][1]
  assert set(h.history) == {"loss", "sparse_categorical_accuracy", "val_loss",
                            "val_sparse_categorical_accuracy"}
  assert len(h.history["val_loss"]) == 1                      # validation_freq=2
  assert isinstance(h.history["loss"][-1].item(), float)
  assert [t.batch_index for t in th.timestamp_log] == [1, 2, 4, 6]
  assert m.optimizer.current_lr() == pytest.approx(0.05)      # schedule of epoch 1
  assert os.path.exists(tmp_path / "checkpoint-1.h5") and os.path.exists(tmp_path / "checkpoint-2.h5")
  out = m.evaluate(ds, steps=2)
  assert len(out) == 2 and 0 <= out[1] <= 1
  # reported loss includes the L2 regularisation term, like Keras
  reg = m.store.regularization_loss().item()
  assert reg > 0 and out[0] >= reg


def test_categorical_loss_accepts_one_hot_and_sparse_column():
  m = _model("categorical_crossentropy", "categorical_accuracy")
  x = torch.randn(4, 32, 32, 3)
  onehot = torch.nn.functional.one_hot(torch.tensor([1, 2, 3, 4]), 10).float()
  l1, _ = m.test_on_batch(x, onehot)
  l2, _ = m.test_on_batch(x, torch.tensor([[1.], [2.], [3.], [4.]]))
  assert l1.item() == pytest.approx(l2.item())
  with pytest.raises(ValueError):
    m.compile(SGD(0.1), "mse")


def test_sgd_is_keras_momentum_with_l2():
  spec = models.resnet20()
  st = ParamStore(spec, "cpu", 0)
  st.grads.normal_()
  p0 = st.params.clone()
  opt = SGD(0.1, momentum=0.9)
  opt.apply_gradients(st, grad_divisor=2.0)
  g = st.grads / 2 + st.l2_vector() * p0
  assert torch.allclose(st.slot("momentum"), -0.1 * g, atol=1e-7)
  assert torch.allclose(st.params, p0 - 0.1 * g, atol=1e-6)
  p1, v1 = st.params.clone(), st.slot("momentum").clone()
  opt.apply_gradients(st, grad_divisor=2.0)
  g2 = st.grads / 2 + st.l2_vector() * p1
  assert torch.allclose(st.slot("momentum"), 0.9 * v1 - 0.1 * g2, atol=1e-7)
  assert opt.iterations == 2
  # L2 applies to conv/dense kernels and the dense bias, not to BatchNorm scale/offset
  l2 = st.l2_vector()
  e = st.entries
  assert l2[e["conv1/kernel"].offset] == pytest.approx(4e-4)
  assert l2[e["fc10/bias"].offset] == pytest.approx(4e-4)
  assert l2[e["bn_conv1/gamma"].offset] == 0


def test_loss_scale_optimizer_static_and_dynamic():
  spec = models.resnet20()
  st = ParamStore(spec, "cpu", 0)
  ref = ParamStore(spec, "cpu", 0)
  ref.grads.normal_()
  st.grads.copy_(ref.grads * 128)
  SGD(0.1, momentum=0.9).apply_gradients(ref)
  LossScaleOptimizer(SGD(0.1, momentum=0.9), 128).apply_gradients(st)
  assert torch.allclose(st.params, ref.params, atol=1e-6)
  dyn = LossScaleOptimizer(SGD(0.1), "dynamic", initial_scale=1024.0, growth_steps=2)
  st.grads[5] = float("inf")
  before = st.params.clone()
  assert dyn.apply_gradients(st) is False and dyn.loss_scale == 512.0 and dyn.skipped_steps == 1
  assert torch.equal(st.params, before)                      # the step was skipped
  st.grads.normal_()
  dyn.apply_gradients(st); dyn.apply_gradients(st)
  assert dyn.loss_scale == 1024.0                            # grew after 2 clean steps


def test_profiler_callback_writes_trace(tmp_path):
  m = _model()
  ds = SyntheticDataset(32, 32, 3, 10, 2)
  p = cb.get_profiler_callback(str(tmp_path), "2,3")
  m.fit(ds, epochs=1, steps_per_epoch=4, callbacks=[p], verbose=0)
  assert os.path.exists(tmp_path / "profile_steps_2_3.trace.json")


def test_stop_at_threshold_callback():
  from distributed_tensorflow_amd.train.callbacks import StopAtThreshold

  class _M:
    stop_training = False
  cb = StopAtThreshold(0.5, "categorical_accuracy")
  cb.model = _M()
  cb.on_epoch_end(0, {"val_categorical_accuracy": 0.4})
  assert not cb.model.stop_training
  cb.on_epoch_end(1, {"loss": 1.0})                       # no validation this epoch
  assert not cb.model.stop_training
  cb.on_epoch_end(2, {"val_categorical_accuracy": 0.6})
  assert cb.model.stop_training


def test_data_format_selects_the_activation_layout_not_the_result():
  """`--data_format` is a layout choice (`resnet_model.py:243-251`): the PyTorch-operator
  executor really stores activations channel-major or pixel-major, and the losses agree."""
  import torch
  from distributed_tensorflow_amd import models
  from distributed_tensorflow_amd.train.model import Model
  from distributed_tensorflow_amd.train.optimizers import SGD
  runs = {}
  for df in ("channels_first", "channels_last"):
    m = Model(models.resnet20(), dtype="fp32", seed=1, data_format=df)
    m.compile(SGD(0.1, momentum=0.9))
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4, 32, 32, 3, generator=g)
    y = torch.randint(0, 10, (4, 1), generator=g).float()
    losses = [m.train_on_batch(x, y)[0].item() for _ in range(3)]
    ex = m._executor(4)
    ex.debug_keep_activations = True
    m.train_on_batch(x, y)
    a = ex.last_activations["input"]
    runs[df] = (losses, a.is_contiguous(), a.is_contiguous(memory_format=torch.channels_last))
  assert runs["channels_first"][1] and not runs["channels_first"][2]
  assert runs["channels_last"][2] and not runs["channels_last"][1]
  for a, b in zip(runs["channels_first"][0], runs["channels_last"][0]):
    assert abs(a - b) < 1e-3 * max(1.0, abs(a))
