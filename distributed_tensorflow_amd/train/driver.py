"""The training/eval driver shared by the CIFAR and ImageNet entry points.

The reference carries 27 near-identical copies of one `run(flags_obj)` function
(baseline, `_dist`, `_dist_ps_*`, `_horovod`, `ps_server/*_ps_{0..15}.py`) that
differ only in a hard-coded `TF_CONFIG`, the eager/graph switch and a few
Horovod lines.  Here there is one `run_classifier`, parameterised by a
`DatasetConfig`; role and cluster come from the environment
(`parallel/cluster.py`).  Flow parity with `resnet_imagenet_main.py:74-265`:

  perf knobs -> dtype / loss scale -> cluster + strategy -> input_fn (synthetic or
  real) -> optimizer (+ loss-scale wrapper, + Horovod wrapper) -> model ->
  compile -> callbacks -> step accounting -> fit -> evaluate -> build_stats
"""
from __future__ import annotations

import dataclasses
import os
from typing import Callable, Dict, Optional

import torch

from .. import common
from ..models.graph import ModelSpec
from ..parallel import cluster as cluster_lib
from ..parallel import strategy as strategy_lib
from ..utils import flags as flags_core
from ..utils import keras_utils
from ..utils import logging
from .callbacks import ModelCheckpoint, StopAtThreshold
from .model import Model
from .optimizers import LossScaleOptimizer
from .schedules import PiecewiseConstantDecayWithWarmup

__all__ = ["DatasetConfig", "run_classifier", "maybe_launch"]


@dataclasses.dataclass
class DatasetConfig:
  name: str
  height: int
  width: int
  channels: int
  num_classes: int
  num_images: Dict[str, int]                    # {'train': ..., 'validation': ...}
  model_fn: Callable[[], ModelSpec]
  loss: str
  accuracy_metric: str
  learning_rate_schedule: Callable              # (epoch, batch, batches_per_epoch, batch_size)
  lr_schedule_table: Optional[list] = None      # [(multiplier, start_epoch)], for tensor LR
  real_input_fn: Optional[Callable] = None      # input_fn for on-disk data
  parse_record_fn: Optional[Callable] = None
  allow_fp16: bool = True
  scale_lr_by_size_in_horovod: bool = False     # CIFAR-Horovod: lr = 0.1 * size
  divide_steps_by_size_in_horovod: bool = True  # ImageNet-Horovod: steps // size
  trivial_model_fn: Optional[Callable[[], ModelSpec]] = None


def maybe_launch(flags_obj, argv) -> Optional[int]:
  """Fans a directly started driver out into one process per GPU when needed.

  Returns the job's exit code if this process acted as the launcher, else None.
  """
  from .. import launch as launch_lib  # pylint: disable=import-outside-toplevel
  ds = (flags_obj.distribution_strategy or "default").lower()
  if "RANK" in os.environ:
    return None
  n = strategy_lib.strategy_needs_launch(ds, flags_obj.num_gpus)
  _configure_cluster(flags_obj)
  tfc = cluster_lib.parse_tf_config()
  if tfc is not None and ds == "multi_worker_mirrored" and flags_obj.num_gpus > 1:
    # One TF "worker" task with several GPUs: fan out into one rank per GPU; the task's
    # position in the cluster gives the node rank, the first task's address the rendezvous.
    task = cluster_lib.resolve_task(gpus_per_task=flags_obj.num_gpus)
    tasks = [t for t in task.cluster.ordered_tasks() if t[0] != "ps"]
    node = tasks.index((task.job, task.task_index))
    logging.info("multi_worker_mirrored: task %s/%d launches %d local ranks (node %d of %d)",
                 task.job, task.task_index, flags_obj.num_gpus, node, len(tasks))
    return launch_lib.launch(argv, flags_obj.num_gpus, master_addr=task.master_addr,
                             master_port=task.master_port, nnodes=len(tasks), node_rank=node)
  if ds == "parameter_server" and tfc is None:
    # Single command: 1 PS + (num_gpus - 1) workers on this node, one rank per GPU.
    # N GPUs give 1 PS rank + (N - 1) worker ranks, one rank per GPU over RCCL.
    world = max(flags_obj.num_gpus, 2)
    env_extra = {}
    import torch  # pylint: disable=import-outside-toplevel
    have = torch.cuda.device_count() if flags_obj.num_gpus > 0 else 0
    if have < world and "DTF_BACKEND" not in os.environ:
      # RCCL refuses two ranks on one GPU: with fewer GPUs than ranks (the minimum job is
      # 1 PS + 1 worker) the ranks share devices and talk over gloo instead
      logging.warning("parameter_server: %d ranks on %d GPU(s): ranks share devices, using the "
                      "gloo backend (one rank per GPU is needed for RCCL)", world, have)
      env_extra["DTF_BACKEND"] = "gloo"
    logging.info("parameter_server: launching 1 PS rank + %d worker ranks", world - 1)
    return launch_lib.launch(argv, world, num_ps=1, env_extra=env_extra or None)
  if n and tfc is None:
    logging.info("%s: launching %d ranks (one per GPU)", ds, n)
    return launch_lib.launch(argv, n)
  return None


def _configure_cluster(flags_obj) -> int:
  return cluster_lib.configure_cluster(
      flags_obj.worker_hosts, flags_obj.task_index,
      ps_hosts=getattr(flags_obj, "ps_hosts", None),
      task_type=getattr(flags_obj, "task_type", None))


def _make_strategy(flags_obj, use_horovod: bool):
  num_workers = _configure_cluster(flags_obj)
  if use_horovod:
    return None, num_workers
  bucket_bytes = None
  if flags_obj.fusion_threshold_mb:
    bucket_bytes = int(flags_obj.fusion_threshold_mb) * 1024 * 1024
  elif flags_obj.num_packs == 1:
    # The reference's default (one pack) would leave a single collective that can only start
    # when the whole backward pass is done.  Unless packs are asked for explicitly, gradients
    # are reduced in 25 MB buckets that overlap the backward pass (small models still end up
    # with one bucket).
    bucket_bytes = 25 * 1024 * 1024
  strategy = strategy_lib.get_distribution_strategy(
      distribution_strategy=flags_obj.distribution_strategy,
      num_gpus=flags_obj.num_gpus, num_workers=num_workers,
      all_reduce_alg=flags_obj.all_reduce_alg, num_packs=flags_obj.num_packs,
      bucket_bytes=bucket_bytes, comm_kind=getattr(flags_obj, "comm", None))
  return strategy, num_workers


def run_classifier(flags_obj, cfg: DatasetConfig, use_horovod: bool = False):
  """Trains and evaluates; returns the stats dictionary of `build_stats`."""
  keras_utils.set_session_config(
      enable_eager=flags_obj.enable_eager, enable_xla=flags_obj.enable_xla,
      inter_op=flags_obj.inter_op_parallelism_threads,
      intra_op=flags_obj.intra_op_parallelism_threads)

  # Flag override logic for better performance.
  if flags_obj.tf_gpu_thread_mode:
    common.set_gpu_thread_mode_and_count(flags_obj)
  if flags_obj.data_delay_prefetch:
    common.data_delay_prefetch()
  common.set_cudnn_batchnorm_mode()

  dtype = flags_obj.dtype
  if dtype == "fp16" and not cfg.allow_fp16:
    raise ValueError("dtype fp16 is not supported for %s. Use fp32 or bf16." % cfg.name)

  data_format = flags_obj.data_format or "channels_last"

  hvd = None
  if use_horovod:
    from .. import hvd as hvd_mod  # pylint: disable=import-outside-toplevel
    hvd = hvd_mod
    hvd.init(use_gpu=flags_obj.num_gpus > 0)

  strategy, _ = _make_strategy(flags_obj, use_horovod)

  if hvd is not None:
    device = hvd.device()
  elif strategy is not None:
    device = strategy.device
  elif flags_obj.num_gpus > 0 and torch.cuda.is_available():
    device = torch.device("cuda", 0)
  else:
    device = torch.device("cpu")

  # ---- input ------------------------------------------------------------------------
  global_batch = flags_obj.batch_size
  per_replica_batch = strategy.per_replica_batch_size(global_batch) if strategy else global_batch
  rank = strategy.rank if strategy else (hvd.rank() if hvd else 0)
  num_shards = strategy.num_replicas_in_sync if strategy else 1
  if strategy is not None and not strategy.synchronous:
    num_shards = max(strategy.num_workers, 1)
    rank = max(strategy.worker_index, 0)
  if hvd is not None:
    num_shards = 1          # the Horovod drivers do not shard the input (reference parity)

  if flags_obj.use_synthetic_data:
    strategy_lib.set_up_synthetic_data()
    input_fn = common.get_synth_input_fn(
        height=cfg.height, width=cfg.width, num_channels=cfg.channels,
        num_classes=cfg.num_classes, dtype=torch.float32, drop_remainder=True, device=device)
  else:
    if cfg.real_input_fn is None:
      raise ValueError("no on-disk input pipeline for %s" % cfg.name)
    strategy_lib.undo_set_up_synthetic_data()
    input_fn = cfg.real_input_fn

  drop_remainder = flags_obj.enable_xla or not flags_obj.enable_get_next_as_optional
  is_ps = strategy is not None and strategy.is_ps_task

  train_input_dataset = None if is_ps else input_fn(
      is_training=True, data_dir=flags_obj.data_dir, batch_size=per_replica_batch,
      num_epochs=flags_obj.train_epochs, parse_record_fn=cfg.parse_record_fn,
      datasets_num_private_threads=flags_obj.datasets_num_private_threads,
      dtype=dtype, drop_remainder=drop_remainder, device=device,
      shard_index=rank, num_shards=num_shards,
      tf_data_experimental_slack=flags_obj.tf_data_experimental_slack)

  eval_input_dataset = None
  if not flags_obj.skip_eval and not is_ps:
    eval_input_dataset = input_fn(
        is_training=False, data_dir=flags_obj.data_dir, batch_size=per_replica_batch,
        num_epochs=flags_obj.train_epochs, parse_record_fn=cfg.parse_record_fn,
        dtype=dtype, drop_remainder=drop_remainder, device=device,
        shard_index=rank, num_shards=num_shards)

  # ---- optimizer / model ---------------------------------------------------------------
  lr_schedule = 0.1
  if flags_obj.use_tensor_lr and cfg.lr_schedule_table:
    table = cfg.lr_schedule_table
    lr_schedule = PiecewiseConstantDecayWithWarmup(
        batch_size=global_batch, epoch_size=cfg.num_images["train"],
        warmup_epochs=table[0][1], boundaries=[p[1] for p in table[1:]],
        multipliers=[p[0] for p in table], compute_lr_on_cpu=True)
  if hvd is not None and cfg.scale_lr_by_size_in_horovod:
    lr_schedule = 0.1 * hvd.size()

  with strategy_lib.get_strategy_scope(strategy):
    optimizer = common.get_optimizer(lr_schedule)
    if hvd is not None:
      optimizer = hvd.DistributedOptimizer(optimizer)
    loss_scale = flags_core.get_loss_scale(flags_obj, default_for_fp16=128)
    if dtype == "fp16" or loss_scale != 1:
      optimizer = _wrap_loss_scale(optimizer, loss_scale)

    if flags_obj.use_trivial_model:
      if cfg.trivial_model_fn is None:
        raise ValueError("no trivial model for %s" % cfg.name)
      spec = cfg.trivial_model_fn()
    else:
      spec = cfg.model_fn()
    model = Model(spec, strategy=strategy, dtype=dtype, device=device, seed=flags_obj.seed,
                  executor=flags_obj.executor, data_format=data_format,
                  use_hip_graph=None if flags_obj.use_hip_graph else False,
                  deterministic=flags_obj.deterministic)
    model.compile(loss=cfg.loss, optimizer=optimizer,
                  metrics=[cfg.accuracy_metric] if flags_obj.report_accuracy_metrics else None,
                  run_eagerly=flags_obj.run_eagerly,
                  experimental_run_tf_function=flags_obj.force_v2_in_keras_compile)
    if flags_obj.resume:
      model.load_weights(flags_obj.resume)
      logging.info("restored variables from %s", flags_obj.resume)

  logging.info("model %s: %d parameters, executor=%s, dtype=%s, device=%s, strategy=%s",
               spec.name, spec.count_params(), model.executor_kind, dtype, device,
               strategy.name if strategy else ("horovod" if hvd else "off"))

  # ---- callbacks and step accounting ------------------------------------------------------
  callbacks = common.get_callbacks(cfg.learning_rate_schedule, cfg.num_images["train"])

  if flags_obj.stop_threshold is not None and not flags_obj.skip_eval:
    callbacks.append(StopAtThreshold(flags_obj.stop_threshold, cfg.accuracy_metric))

  train_steps = cfg.num_images["train"] // global_batch
  train_epochs = flags_obj.train_epochs
  if flags_obj.train_steps:
    train_steps = min(flags_obj.train_steps, train_steps)
    train_epochs = 1
  if flags_obj.max_train_steps:
    train_steps = min(flags_obj.max_train_steps, train_steps)
    train_epochs = 1

  num_eval_steps = cfg.num_images["validation"] // global_batch
  validation_data = eval_input_dataset
  if flags_obj.skip_eval:
    num_eval_steps = None
    validation_data = None

  verbose = 2
  steps_per_epoch = train_steps
  if hvd is not None:
    # The Horovod drivers replace the common callbacks (no TimeHistory, no LR
    # scheduler): resnet_imagenet_main_horovod.py:255-259.
    callbacks = [hvd.callbacks.BroadcastGlobalVariablesCallback(0),
                 hvd.callbacks.MetricAverageCallback(),
                 hvd.callbacks.LearningRateWarmupCallback(warmup_epochs=3, verbose=1)]
    if hvd.rank() == 0:
      callbacks.append(ModelCheckpoint("./checkpoint-{epoch}.h5"))
    verbose = 1 if hvd.rank() == 0 else 0
    if cfg.divide_steps_by_size_in_horovod:
      steps_per_epoch = max(train_steps // hvd.size(), 1)
  elif strategy is not None and not strategy.synchronous:
    # ps_server/resnet_imagenet_main_dist_ps_0.py:263  (train_steps // 15)
    steps_per_epoch = max(train_steps // max(strategy.num_workers, 1), 1)
    verbose = 1 if strategy.is_chief else 2
  elif strategy is not None and not strategy.is_chief:
    verbose = 0

  # effective environment, process group, reducer and launch mode, once, from the first rank
  from ..utils.runtime_report import log_runtime_report  # pylint: disable=import-outside-toplevel
  log_runtime_report(strategy, None if is_ps else model,
                     rank=(hvd.rank() if hvd is not None else None))

  history = model.fit(train_input_dataset, epochs=train_epochs,
                      steps_per_epoch=steps_per_epoch, callbacks=callbacks,
                      validation_steps=num_eval_steps, validation_data=validation_data,
                      validation_freq=flags_obj.epochs_between_evals, verbose=verbose)

  eval_output = None
  if not flags_obj.skip_eval and not is_ps:
    eval_output = model.evaluate(eval_input_dataset, steps=num_eval_steps,
                                 verbose=2 if verbose else 0)

  if flags_obj.export_dir and not is_ps and (hvd is None or hvd.rank() == 0):
    # collective under a synchronous strategy (replica mean of the BatchNorm moving statistics);
    # the chief writes
    if strategy is None or strategy.is_chief:
      os.makedirs(flags_obj.export_dir, exist_ok=True)
    model.save_replicated(os.path.join(flags_obj.export_dir, spec.name + ".h5"))

  stats = common.build_stats(history, eval_output, callbacks)
  if strategy is not None:
    strategy.shutdown()
  return stats


def _wrap_loss_scale(optimizer, loss_scale):
  """LossScaleOptimizer around the (possibly Horovod-wrapped) optimizer.

  The reference wraps in this order (`resnet_imagenet_main_horovod.py:189-196`), so
  the reducer hook of the inner Horovod optimizer must stay reachable."""
  wrapped = LossScaleOptimizer(optimizer, loss_scale)
  for attr in ("make_reducer", "fusion_bytes"):
    if hasattr(optimizer, attr):
      setattr(wrapped, attr, getattr(optimizer, attr))
  return wrapped
