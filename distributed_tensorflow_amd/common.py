"""Shared driver utilities: flags, callbacks, optimizer factory, stats, perf knobs.

Counterpart of the reference's `image_classification/common.py`; each function
cites the lines it stands in for.
"""
from __future__ import annotations

import multiprocessing
import os

from .data.synthetic import get_synth_input_fn  # noqa: F401  (re-export, common.py:311-359)
from .train import callbacks as callbacks_lib
from .train.callbacks import LearningRateBatchScheduler, TimeHistory  # noqa: F401
from .train.optimizers import SGD
from .train.schedules import BASE_LEARNING_RATE, PiecewiseConstantDecayWithWarmup  # noqa: F401
from .train.stats import TRAIN_TOP_1, build_stats  # noqa: F401
from .utils import flags as flags_core
from .utils import logging

FLAGS = flags_core.FLAGS


def set_gpu_thread_mode_and_count(flags_obj):
  """Host-thread budgeting (`common.py:143-166`).

  The reference dedicates `per_gpu_thread_count` private TF threads per GPU to
  kernel launches and shrinks the input-pipeline pool accordingly.  Here one
  process drives one GPU and the step is a pre-recorded launch plan (optionally
  a hipGraph), so the launch side needs a single thread; what remains meaningful
  is the budget for the input workers, computed with the same formula.
  """
  cpu_count = multiprocessing.cpu_count()
  logging.info("Logical CPU cores: %s", cpu_count)
  per_gpu_thread_count = flags_obj.per_gpu_thread_count or 2
  os.environ["DTF_GPU_THREAD_MODE"] = str(flags_obj.tf_gpu_thread_mode)
  os.environ["DTF_GPU_THREAD_COUNT"] = str(per_gpu_thread_count)
  logging.info("DTF_GPU_THREAD_COUNT: %s", os.environ["DTF_GPU_THREAD_COUNT"])
  logging.info("DTF_GPU_THREAD_MODE: %s", os.environ["DTF_GPU_THREAD_MODE"])
  total_gpu_thread_count = per_gpu_thread_count * flags_obj.num_gpus
  num_runtime_threads = flags_obj.num_gpus
  if not flags_obj.datasets_num_private_threads:
    flags_obj.datasets_num_private_threads = max(1, min(
        cpu_count - total_gpu_thread_count - num_runtime_threads, flags_obj.num_gpus * 8))
    logging.info("Set datasets_num_private_threads to %s",
                 flags_obj.datasets_num_private_threads)


def get_optimizer(learning_rate=0.1):
  """SGD with momentum 0.9 (`common.py:169-172`); the schedule overwrites the rate."""
  return SGD(learning_rate=learning_rate, momentum=0.9)


def get_callbacks(learning_rate_schedule_fn, num_images):
  """TimeHistory + LR scheduler (+ TensorBoard, profiler) (`common.py:175-199`)."""
  time_callback = TimeHistory(FLAGS.batch_size, FLAGS.log_steps)
  callbacks = [time_callback]
  if not FLAGS.use_tensor_lr:
    callbacks.append(LearningRateBatchScheduler(
        learning_rate_schedule_fn, batch_size=FLAGS.batch_size, num_images=num_images))
  if FLAGS.enable_tensorboard:
    callbacks.append(callbacks_lib.TensorBoard(log_dir=FLAGS.model_dir))
  if FLAGS.profile_steps:
    callbacks.append(callbacks_lib.get_profiler_callback(
        FLAGS.model_dir, FLAGS.profile_steps, FLAGS.enable_tensorboard))
  return callbacks


def define_keras_flags(dynamic_loss_scale=True):
  """All flags of the Keras drivers (`common.py:248-309`)."""
  flags_core.define_base(run_eagerly=True)
  flags_core.define_performance(num_parallel_calls=False, tf_gpu_thread_mode=True,
                                datasets_num_private_threads=True,
                                dynamic_loss_scale=dynamic_loss_scale, loss_scale=True,
                                tf_data_experimental_slack=True, enable_xla=True,
                                force_v2_in_keras_compile=True)
  flags_core.define_image()
  flags_core.define_benchmark()
  flags_core.define_distribution()

  f = flags_core
  f.DEFINE_boolean("enable_eager", False, "Accepted for compatibility (the engine is "
                   "always stream-ordered eager launch or hipGraph replay).")
  f.DEFINE_boolean("skip_eval", False, "Skip evaluation?")
  f.DEFINE_boolean("set_learning_phase_to_train", True,
                   "If skip eval, build only the training plan.")
  f.DEFINE_boolean("explicit_gpu_placement", False,
                   "Without a distribution strategy, pin the run to GPU 0 explicitly.")
  f.DEFINE_boolean("use_trivial_model", False, "Whether to use a trivial Keras model.")
  f.DEFINE_boolean("report_accuracy_metrics", True,
                   "Report metrics during training and evaluation.")
  f.DEFINE_boolean("use_tensor_lr", False,
                   "Use the step-indexed learning-rate schedule object instead of the "
                   "per-batch callback.")
  f.DEFINE_boolean("enable_tensorboard", False, "Whether to enable Tensorboard callback.")
  f.DEFINE_integer("train_steps", None,
                   "The number of steps to run for training. If it is larger than # batches "
                   "per epoch, then use # batches per epoch. When this flag is set, only one "
                   "epoch is going to run for training.")
  f.DEFINE_string("profile_steps", None,
                  "Save profiling data to model dir at given range of steps: a comma "
                  "separated pair of positive integers (first,last step).")
  f.DEFINE_boolean("data_delay_prefetch", False,
                   "Delay the input prefetch thread slightly so its host->device copy does "
                   "not compete with the gradient traffic of the current step.")
  f.DEFINE_boolean("batchnorm_spatial_persistent", True,
                   "Use the single-pass BatchNorm statistics path (statistics produced by "
                   "the convolution epilogue instead of a separate reduction).")
  f.DEFINE_boolean("enable_get_next_as_optional", False,
                   "Allow a final partial batch in the input pipeline.")
  # Extensions of this framework (not in the reference).
  f.DEFINE_string("ps_hosts", None, "Comma-separated host:port list of parameter-server tasks "
                  "(with --worker_hosts, --task_type, --task_index: a TF_CONFIG on the "
                  "command line).")
  f.DEFINE_string("task_type", None, "ps | worker | chief: job of this process in the cluster.")
  f.DEFINE_string("executor", "auto", "auto | hip | torch: execution back end.")
  f.DEFINE_string("comm", "auto", "auto | native | torch: data plane of the gradient collectives "
                  "and parameter-server transfers on GPUs - the engine's own RCCL communicator "
                  "(issued from C++, capturable into the step's hipGraph, fail-fast watchdog) or "
                  "torch.distributed.  auto: native whenever the job runs on RCCL.")
  f.DEFINE_integer("seed", 0, "Seed of the variable initialisers.")
  f.DEFINE_string("resume", None, "Keras-layout HDF5 checkpoint to restore before training.")
  f.DEFINE_integer("fusion_threshold_mb", None,
                   "Gradient fusion bucket size in MB (overrides --num_packs; with neither "
                   "flag given the buckets are 25 MB).")
  f.DEFINE_boolean("use_hip_graph", True, "Replay the training step as a captured hipGraph "
                   "when the strategy allows it.")
  f.DEFINE_boolean("deterministic", False, "Every floating-point sum of a step in a fixed order "
                   "(BatchNorm statistics, weight and bias gradients: two runs agree bit for "
                   "bit; slower), and a check that replicas hold identical gradients after the "
                   "all-reduce (debug aid).")


def data_delay_prefetch():
  """`--data_delay_prefetch` (`common.py:362-365`): 11 ms prefetch delay for real data."""
  if not FLAGS.use_synthetic_data:
    os.environ["DTF_DATA_PREFETCH_DELAY_MS"] = "11"


def set_cudnn_batchnorm_mode():
  """`--batchnorm_spatial_persistent` (`common.py:368-377`).

  The cuDNN "spatial persistent" mode is a faster single-kernel BatchNorm.  The
  counterpart in the HIP engine is taking the per-channel sum / sum-of-squares
  from the convolution epilogue, which removes the statistics pass.
  """
  if FLAGS.batchnorm_spatial_persistent:
    os.environ["DTF_BN_FUSED_STATS"] = "1"
  else:
    os.environ["DTF_BN_FUSED_STATS"] = "0"
