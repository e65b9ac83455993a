"""Keras-like `Model` (compile / fit / evaluate / predict / save) over the engine.

The reference drives training through `tf.keras.Model.compile/fit/evaluate`
(`resnet_imagenet_main.py:199-259`).  This class keeps that calling convention
and the callback protocol while the work underneath is done by an executor
(native HIP engine on MI355X, PyTorch reference elsewhere), a distribution
strategy (bucketed RCCL all-reduce or parameter-server push/pull) and a fused
optimizer over flat buffers.

Per training step:  callbacks.on_batch_begin -> next batch -> forward + backward
(gradient buckets are all-reduced while backward is still running) -> fused
SGD-momentum(+L2, +loss-scale, +1/replicas) -> metric accumulation on the device
(no host sync) -> callbacks.on_batch_end.
"""
from __future__ import annotations

import math
import os
import sys
import time
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from ..engine.params import ParamStore
from ..engine.torch_executor import TorchExecutor
from ..models.graph import ModelSpec
from ..utils import logging
from .callbacks import Callback, CallbackList, History
from .optimizers import LossScaleOptimizer, Optimizer

__all__ = ["Model", "SUPPORTED_LOSSES"]


def _next_batch(it, what: str, step: int, steps: int):
  """`next(it)` with an error that says what ran out (a bare StopIteration from inside `fit`
  reads like a bug in the loop)."""
  try:
    return next(it)
  except StopIteration:
    raise RuntimeError("the %s input ran out of data at step %d of %d: the dataset holds fewer "
                       "batches than steps were requested (repeat it, or lower the step count)"
                       % (what, step, steps)) from None

SUPPORTED_LOSSES = ("sparse_categorical_crossentropy", "categorical_crossentropy")


# executor kind of the most recently constructed Model ("hip" / "torch"): lets a driver-level test
# or a log reader tell which engine a run used
LAST_EXECUTOR_KIND: Optional[str] = None


def _pick_executor_kind(requested: str, device: torch.device, dtype: str,
                        spec: Optional[ModelSpec] = None) -> str:
  """auto: the native engine on a GPU, PyTorch ops on the CPU.

  The native engine stores activations as bfloat16 (the type it is tuned for), IEEE half
  (`--dtype fp16`, the reference's mixed-precision switch `resnet_imagenet_main.py:97-100`, with
  its loss scale) or fp32 (the reference's default precision): fp32 accumulation and fp32 master
  weights in all three.  `--executor torch` runs any of them through PyTorch's operators."""
  if requested != "auto":
    return requested
  if device.type == "cuda" and dtype in ("bf16", "bfloat16", "fp16", "fp32"):
    if spec is not None:
      from ..engine import hip_executor  # pylint: disable=import-outside-toplevel
      if not hip_executor.supports(spec):
        logging.warning("model %s has ops without a native kernel: running it with PyTorch ops",
                        spec.name)
        return "torch"
    return "hip"      # a missing extension is an error on a GPU machine, not a silent fallback
  return "torch"


class Model:
  """A model replica bound to a device, an executor and (optionally) a strategy."""

  def __init__(self, spec: ModelSpec, strategy=None, dtype: str = "fp32",
               device: Optional[torch.device] = None, seed: int = 0,
               executor: str = "auto", data_format: str = "channels_last",
               use_hip_graph: Optional[bool] = None, deterministic: bool = False):
    """`use_hip_graph`: replay the native training step as a captured hipGraph (None: the
    executor's default).  `deterministic`: every sum of the step in a fixed order (BatchNorm
    statistics, weight gradients, bias gradients: two runs agree bit for bit), and after every
    gradient all-reduce the replicas compare a checksum of the reduced gradients."""
    self.spec = spec
    self.name = spec.name
    self.strategy = strategy
    self.dtype = {"float32": "fp32", "bfloat16": "bf16", "float16": "fp16"}.get(dtype, dtype)
    if device is None:
      if strategy is not None:
        device = strategy.device
      else:
        device = torch.device("cpu")
    self.device = torch.device(device)
    self.data_format = data_format
    self.store = ParamStore(spec, self.device, seed)
    self.executor_kind = _pick_executor_kind(executor, self.device, self.dtype, spec)
    global LAST_EXECUTOR_KIND  # pylint: disable=global-statement
    LAST_EXECUTOR_KIND = self.executor_kind
    self._executors: Dict[int, object] = {}
    self.optimizer: Optional[Optimizer] = None
    self.loss: Optional[str] = None
    self.metrics_names: List[str] = ["loss"]
    self._acc_name: Optional[str] = None
    self.stop_training = False
    self.use_hip_graph = use_hip_graph
    self.deterministic = bool(deterministic)
    self._reducer = None
    self._buckets = None
    self._synced_initial = False
    self.history: Optional[History] = None

  # -- configuration -----------------------------------------------------------------
  def compile(self, optimizer: Optimizer, loss: str = "sparse_categorical_crossentropy",
              metrics: Optional[Sequence[str]] = None, run_eagerly=None,
              experimental_run_tf_function=None):
    del run_eagerly, experimental_run_tf_function   # accepted for call-site parity
    if loss not in SUPPORTED_LOSSES:
      raise ValueError("loss must be one of %s" % (SUPPORTED_LOSSES,))
    self.optimizer = optimizer
    self.loss = loss
    self._acc_name = None
    for m in metrics or []:
      if m in ("sparse_categorical_accuracy", "categorical_accuracy", "accuracy", "acc"):
        self._acc_name = m if m.endswith("accuracy") and m != "accuracy" else (
            "sparse_categorical_accuracy" if loss.startswith("sparse") else
            "categorical_accuracy")
      else:
        raise ValueError("unsupported metric %r" % m)
    self.metrics_names = ["loss"] + ([self._acc_name] if self._acc_name else [])
    for s in optimizer.slot_names:
      self.store.slot(s)

  # -- executors ---------------------------------------------------------------------
  def _bucket_list(self):
    if self._buckets is None:
      st = self.strategy
      hv = getattr(self.optimizer, "fusion_bytes", None)
      if hv is not None:
        self._buckets = self.store.make_buckets(bucket_bytes=hv)
      elif st is not None and getattr(st, "bucket_bytes", None):
        self._buckets = self.store.make_buckets(bucket_bytes=st.bucket_bytes)
      else:
        self._buckets = self.store.make_buckets(
            num_packs=getattr(st, "num_packs", 1) if st is not None else 1)
    return self._buckets

  def _executor(self, batch: int):
    ex = self._executors.get(batch)
    if ex is None:
      if self.executor_kind == "hip":
        from ..engine.hip_executor import HipExecutor  # pylint: disable=import-outside-toplevel
        if self.data_format == "channels_first":
          logging.warning("data_format channels_first has no effect on the native engine: its "
                          "kernels are written for pixel-major (NHWC) activations; results do "
                          "not depend on the layout")
        use_graph = self.use_hip_graph
        if use_graph is None and self.strategy is not None and not self.strategy.synchronous \
            and "DTF_HIP_GRAPH" not in os.environ:
          # asynchronous parameter server: the step is interleaved with host-driven transfers
          # at every bucket boundary, and the first steps (no parameters prefetched yet) are no
          # basis for the eager / replay calibration: eager launch measured 21.6 ms of GPU time
          # per ResNet-50 step against 28.4 ms for eight replayed graph segments
          use_graph = False
        ex = HipExecutor(self.spec, self.store, batch, self.dtype, self._bucket_list(),
                         use_graph=use_graph,
                         fused_bn_stats=False if self.deterministic else None,
                         deterministic=self.deterministic)
      else:
        ex = TorchExecutor(self.spec, self.store, batch, self.dtype, self._bucket_list(),
                           data_format=self.data_format)
      self._executors[batch] = ex
    return ex

  def _grad_reducer(self):
    if self._reducer is None:
      mk = getattr(self.optimizer, "make_reducer", None)       # hvd.DistributedOptimizer
      if mk is not None:
        self._reducer = mk(self.store, self._bucket_list()) or False
      elif self.strategy is not None:
        self._reducer = self.strategy.make_reducer(self.store, self._bucket_list()) or False
      else:
        self._reducer = False
    return self._reducer or None

  def synchronize(self):
    if self.device.type == "cuda":
      torch.cuda.synchronize(self.device)

  def _sync_initial_variables(self):
    if not self._synced_initial and self.strategy is not None:
      self.strategy.broadcast_variables(self.store, 0)
      for ex in self._executors.values():
        ex.params_updated()
    self._synced_initial = True

  # -- label handling -----------------------------------------------------------------
  def _class_indices(self, labels: torch.Tensor) -> torch.Tensor:
    if self.loss == "categorical_crossentropy" and not (labels.dim() == 2 and labels.shape[1] == 1):
      # one-hot targets; a [N, 1] tensor is the synthetic-data label format the reference
      # feeds to both drivers (common.py:344-350) and is taken as class indices
      if labels.dim() != 2:
        raise ValueError("categorical_crossentropy expects one-hot labels [N, classes]")
      return labels.argmax(dim=1).long()
    return labels.reshape(-1).long()

  # -- one step ---------------------------------------------------------------------
  def _loss_scale(self, executor=None) -> Optional[float]:
    """Loss-gradient factor of the next step; None when the native engine's device-side scale
    is already current (dynamic scaling decided on the GPU)."""
    opt = self._loss_scale_optimizer()
    if opt is None:
      return 1.0
    return opt.loss_scale_for(executor) if executor is not None else opt.loss_scale

  def _loss_scale_optimizer(self) -> Optional[LossScaleOptimizer]:
    opt = self.optimizer
    while opt is not None and not isinstance(opt, LossScaleOptimizer):
      opt = getattr(opt, "inner", None)      # e.g. hvd.DistributedOptimizer(LossScaleOptimizer(..))
    return opt

  def train_on_batch(self, images: torch.Tensor, labels: torch.Tensor):
    """Runs one optimisation step; returns (mean loss tensor, correct-count tensor)."""
    st = self.strategy
    if not self._synced_initial:
      self._sync_initial_variables()
    ex = self._executor(int(images.shape[0]))
    idx = self._class_indices(labels.to(self.device))
    if st is not None and not st.synchronous:
      # asynchronous parameter server: pull -> compute -> push
      # (the next step's parameters are requested before this step computes, and every
      # gradient bucket leaves while the backward pass is still running)
      half = self.dtype in ("bf16", "fp16")
      st.pull(self.store, half=half)
      ex.params_updated()
      ls = self._loss_scale_optimizer()
      # dynamic loss scaling on the native engine: finiteness pass, scale state machine and the
      # verdict stay on the GPU; the verdict travels to the PS as the push's trailer and the
      # host never synchronises with the step
      on_device = (ls is not None and ls.dynamic and getattr(ex, "native", False)
                   and self.store.grads.device.type == "cuda")
      scale = self._loss_scale(ex) if on_device else self._loss_scale()
      st.begin_push(self.store, self._bucket_list(), self.optimizer.current_lr(),
                    1.0 if on_device else scale, half)
      soft = getattr(ex, "native", False) and st.orders_side_lane(ex)
      loss, correct = ex.train_step(images, idx, scale, st.on_bucket_ready, soft_joins=soft) \
          if soft else ex.train_step(images, idx, scale, st.on_bucket_ready)
      finite, trailer = True, None
      if ls is not None and ls.dynamic:
        if soft and self.store.grads.device.type == "cuda":
          # the executor skipped its joins: weight gradients may still be in flight on the side
          # lane, and a check that ran ahead of them would miss their inf / nan
          side = ex.side_stream()
          if side is not None:
            torch.cuda.current_stream(self.device).wait_stream(side)
        if on_device:
          trailer = ls.device_verdict(self.store, ex)
        else:
          # the PS applies a push only when its trailer says so: a non-finite gradient is
          # dropped there and the scale adapts here, as in the synchronous path
          finite = bool(torch.isfinite(self.store.grads).all().item())
          ls.update_scale(finite)
      st.end_push(finite, trailer)
      self.optimizer.iterations += 1
      return loss, correct
    red = self._grad_reducer()
    if red is not None and getattr(ex, "native", False) and hasattr(red, "attach_to_plan"):
      red._skip = red.attach_to_plan(ex)          # single bucket + native communicator
    if red is not None and getattr(ex, "native", False) and \
        getattr(red, "orders_side_lane", None) is not None and red.orders_side_lane(ex):
      loss, correct = ex.train_step(images, idx, self._loss_scale(ex), red.on_bucket_ready,
                                    soft_joins=True)
    else:
      loss, correct = ex.train_step(images, idx, self._loss_scale(ex),
                                    red.on_bucket_ready if red is not None else None)
    if red is not None:
      red.finish()
      if self.deterministic:
        self._check_replica_gradients()
    self.optimizer.apply_gradients(self.store, red.divisor if red is not None else 1.0, ex)
    return loss, correct

  def _check_replica_gradients(self):
    """Every replica must hold bit-identical gradients after the all-reduce: compares a
    (sum, sum of squares) checksum of the flat gradient buffer across ranks."""
    comm = getattr(self._reducer, "comm", None)
    if comm is None or comm.world_size <= 1:
      return
    g = self.store.grads.double()
    mine = torch.stack([g.sum(), (g * g).sum()])
    lo, hi = mine.clone(), mine.clone()
    import torch.distributed as dist  # pylint: disable=import-outside-toplevel
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    if not torch.equal(lo, hi) or not bool(torch.isfinite(mine).all()):
      raise RuntimeError("replicas disagree on the reduced gradients at iteration %d: "
                         "checksum min %s max %s" % (self.optimizer.iterations, lo.tolist(),
                                                     hi.tolist()))

  def test_on_batch(self, images, labels):
    ex = self._executor(int(images.shape[0]))
    return ex.eval_step(images, self._class_indices(labels.to(self.device)))

  # -- fit ---------------------------------------------------------------------------
  def fit(self, x: Iterable, epochs: int = 1, steps_per_epoch: Optional[int] = None,
          callbacks: Optional[List[Callback]] = None, validation_data: Optional[Iterable] = None,
          validation_steps: Optional[int] = None, validation_freq: int = 1,
          verbose: int = 1, initial_epoch: int = 0) -> History:
    if self.optimizer is None:
      raise RuntimeError("compile() the model before fit()")
    st = self.strategy
    history = History()
    self.history = history
    if st is not None and st.is_ps_task:
      # A parameter-server task does not train: it serves until every worker is done.
      st.serve(self)
      return history
    if steps_per_epoch is None:
      steps_per_epoch = len(x)  # type: ignore[arg-type]
    # History goes last (as in Keras) so it records what earlier callbacks put into `logs`
    cbs = CallbackList(list(callbacks or []) + [history], self,
                       dict(epochs=epochs, steps=steps_per_epoch, verbose=verbose))
    self._sync_initial_variables()
    self.stop_training = False
    it = iter(x)
    cbs.call("on_train_begin", None)
    acc = torch.zeros(3, dtype=torch.float64, device=self.device)  # loss_sum, correct, count
    for epoch in range(initial_epoch, epochs):
      cbs.call("on_epoch_begin", epoch, None)
      acc.zero_()
      t_epoch = time.time()
      if verbose:
        print("Epoch %d/%d" % (epoch + 1, epochs), flush=True)
      # per-step results are kept as the tensors the step returned and folded into `acc` only
      # when somebody looks (batch logs, progress line, end of epoch): three tiny device
      # launches per step are a measurable share of a 2.5 ms step
      pending: list = []

      def fold():
        by_n: Dict[int, list] = {}
        for l, c, k in pending:
          by_n.setdefault(k, []).append((l, c))
        for k, items in by_n.items():
          acc[0] += torch.stack([l for l, _ in items]).double().sum() * k
          acc[1] += torch.stack([c for _, c in items]).double().sum()
          acc[2] += k * len(items)
        pending.clear()

      for step in range(steps_per_epoch):
        cbs.call("on_batch_begin", step, None)
        images, labels = _next_batch(it, "training", epoch * steps_per_epoch + step,
                                     epochs * steps_per_epoch)
        n = int(images.shape[0])
        loss, correct = self.train_on_batch(images, labels)
        pending.append((loss, correct, n))
        logs = {"batch": step, "size": n}
        if cbs.wants_batch_logs or (verbose == 1 and self._progress_due(step, steps_per_epoch)):
          fold()
          vals = acc.tolist()
          logs["loss"] = vals[0] / max(vals[2], 1)
          if self._acc_name:
            logs[self._acc_name] = vals[1] / max(vals[2], 1)
          if verbose == 1:
            self._print_progress(step, steps_per_epoch, logs, t_epoch)
        cbs.call("on_batch_end", step, logs)
        if len(pending) >= 256:
          fold()
        if self.stop_training:
          break
      fold()
      logs = self._epoch_logs(acc)
      if validation_data is not None and (epoch + 1) % max(validation_freq, 1) == 0:
        val = self.evaluate(validation_data, steps=validation_steps, verbose=0)
        logs["val_loss"] = val[0]
        if self._acc_name:
          logs["val_" + self._acc_name] = val[1]
      if verbose:
        dt = time.time() - t_epoch
        print("%d/%d - %.0fs - %s" % (steps_per_epoch, steps_per_epoch, dt, " - ".join(
            "%s: %.4f" % (k, v) for k, v in logs.items())), flush=True)
      cbs.call("on_epoch_end", epoch, logs)
      if self.stop_training:
        break
    cbs.call("on_train_end", None)
    return history

  @staticmethod
  def _progress_due(step, total):
    every = max(total // 20, 1)
    return (step + 1) % every == 0 or step + 1 == total

  def _print_progress(self, step, total, logs, t0):
    dt = time.time() - t0
    msg = " - ".join("%s: %.4f" % (k, v) for k, v in logs.items() if k not in ("batch", "size"))
    print("%5d/%d - %.0fs - %s" % (step + 1, total, dt, msg), flush=True)

  def _epoch_logs(self, acc: torch.Tensor) -> Dict[str, np.floating]:
    st = self.strategy
    if st is not None and st.synchronous and st.num_replicas_in_sync > 1:
      acc = st.reduce_scalars(acc.clone())
    vals = acc.tolist()
    count = max(vals[2], 1.0)
    reg = float(self.store.regularization_loss().item())
    logs = {"loss": np.float64(vals[0] / count + reg)}
    if self._acc_name:
      logs[self._acc_name] = np.float64(vals[1] / count)
    return logs

  # -- evaluate / predict ----------------------------------------------------------------
  def evaluate(self, x: Iterable, steps: Optional[int] = None, verbose: int = 0):
    """Returns `[loss, accuracy]` (numpy scalars), like Keras with one metric."""
    st = self.strategy
    if st is not None and st.is_ps_task:
      return None
    if st is not None and not st.synchronous:
      st.pull(self.store, fresh=True)
      for ex in self._executors.values():
        ex.params_updated()
    self._sync_initial_variables()
    if steps is None:
      steps = len(x)  # type: ignore[arg-type]
    acc = torch.zeros(3, dtype=torch.float64, device=self.device)
    it = iter(x)
    for i in range(steps):
      images, labels = _next_batch(it, "evaluation", i, steps)
      n = int(images.shape[0])
      loss, correct = self.test_on_batch(images, labels)
      acc[0] += loss.double() * n
      acc[1] += correct.double()
      acc[2] += n
    if st is not None and st.synchronous and st.num_replicas_in_sync > 1:
      acc = st.reduce_scalars(acc)
    vals = acc.tolist()
    count = max(vals[2], 1.0)
    reg = float(self.store.regularization_loss().item())
    out = [np.float64(vals[0] / count + reg), np.float64(vals[1] / count)]
    if verbose:
      print("%d/%d - loss: %.4f - %s: %.4f" % (steps, steps, out[0],
                                               self._acc_name or "accuracy", out[1]), flush=True)
    return out if self._acc_name else [out[0]]

  def predict(self, images: torch.Tensor) -> torch.Tensor:
    ex = self._executor(int(images.shape[0]))
    return ex.predict(images)

  __call__ = predict

  # -- weights ---------------------------------------------------------------------------
  def count_params(self) -> int:
    return self.spec.count_params()

  def get_weights(self) -> Dict[str, torch.Tensor]:
    return self.store.get_weights()

  def set_weights(self, weights, strict: bool = True):
    self.store.set_weights(weights, strict)
    for ex in self._executors.values():
      ex.params_updated()

  def save_weights(self, path: str):
    from ..checkpoint import keras_h5  # pylint: disable=import-outside-toplevel
    keras_h5.save_model(path, self, include_optimizer=False)

  def save(self, path: str, include_optimizer: bool = True):
    from ..checkpoint import keras_h5  # pylint: disable=import-outside-toplevel
    keras_h5.save_model(path, self, include_optimizer=include_optimizer)

  def save_replicated(self, path: str, include_optimizer: bool = True,
                      weights_only: bool = False) -> bool:
    """Checkpoint of a model trained under a synchronous strategy.  COLLECTIVE: every replica
    calls it.  The BatchNorm moving statistics written are the mean over the replicas (they are
    per-replica quantities, aggregated MEAN on read in Keras' `MirroredStrategy`); the replicas'
    own copies are left as they were.  Only the chief writes the file; returns True there."""
    st = self.strategy
    sync = st is not None and st.synchronous and st.num_replicas_in_sync > 1
    chief = st is None or st.is_chief
    keep = None
    if sync:
      keep = self.store.state.clone()
      st.average_state(self.store.state)
    try:
      if chief:
        if weights_only:
          self.save_weights(path)
        else:
          self.save(path, include_optimizer)
    finally:
      if keep is not None:
        self.store.state.copy_(keep)
    return chief

  def load_weights(self, path: str, load_optimizer: bool = True):
    from ..checkpoint import keras_h5  # pylint: disable=import-outside-toplevel
    keras_h5.load_model(path, self, load_optimizer=load_optimizer)
    for ex in self._executors.values():
      ex.params_updated()
    self._synced_initial = False   # a restored rank 0 must be re-broadcast

  def summary(self, print_fn=print):
    print_fn('Model: "%s"' % self.name)
    for op in self.spec.ops:
      print_fn("  %-28s %-8s -> %s" % (op.name, op.kind, self.spec.shapes[op.output]))
    print_fn("Total params: %d (trainable %d)" % (
        self.spec.count_params(), self.spec.count_params(True)))
