"""Command-line flag registry with absl-compatible syntax and the Model-Garden flag groups.

The reference drives every entry point through `absl.flags` plus the un-vendored
`official.utils.flags.core` helpers (call sites: reference
`image_classification/common.py:248-309`, `resnet_imagenet_main.py:268-271`,
`resnet_cifar_main.py:223-230`).  absl is not a dependency here, so this module
provides the same surface:

* `DEFINE_string / integer / float / boolean / enum / list`, short aliases,
  `FLAGS.name` attribute access, `FLAGS(argv)` parsing, `set_defaults`.
* absl syntax: `--name=value`, `--name value`, single dash `-name`, `--name` /
  `--noname` for booleans (a boolean never consumes the next token, so
  `-use_synthetic_data true` leaves `true` as an ignored positional, exactly as
  the reference tests rely on, `resnet_cifar_test.py:33-44`).
* `define_base / define_performance / define_image / define_benchmark /
  define_distribution`, `get_dtype` (`get_tf_dtype`), `get_loss_scale`,
  `get_num_gpus`.
"""
from __future__ import annotations

import sys
from typing import Any, Callable, Dict, Iterable, List, Optional, Sequence

__all__ = [
    "FLAGS", "FlagValues", "DEFINE_string", "DEFINE_integer", "DEFINE_float",
    "DEFINE_boolean", "DEFINE_bool", "DEFINE_enum", "DEFINE_list", "DEFINE_alias",
    "set_defaults", "define_base", "define_performance", "define_image",
    "define_benchmark", "define_distribution", "get_dtype", "get_tf_dtype",
    "get_loss_scale", "get_num_gpus", "UnrecognizedFlagError", "IllegalFlagValueError",
    "DTYPE_MAP",
]


class UnrecognizedFlagError(ValueError):
  pass


class IllegalFlagValueError(ValueError):
  pass


class _Flag:
  __slots__ = ("name", "default", "help", "parser", "kind", "value", "present",
               "enum_values")

  def __init__(self, name, default, help_, parser, kind, enum_values=None):
    self.name = name
    self.default = default
    self.help = help_
    self.parser = parser
    self.kind = kind
    self.value = default
    self.present = False
    self.enum_values = enum_values

  def parse(self, text):
    try:
      self.value = self.parser(text)
    except (ValueError, TypeError) as e:
      raise IllegalFlagValueError(
          "flag --%s=%s: %s" % (self.name, text, e)) from None
    self.present = True


def _parse_bool(text):
  if isinstance(text, bool):
    return text
  t = str(text).strip().lower()
  if t in ("true", "t", "1", "yes", "y"):
    return True
  if t in ("false", "f", "0", "no", "n"):
    return False
  raise ValueError("not a boolean: %r" % (text,))


def _parse_list(text):
  if isinstance(text, (list, tuple)):
    return list(text)
  if text is None or text == "":
    return []
  return [s.strip() for s in str(text).split(",")]


class FlagValues:
  """Holds flag definitions and their parsed values (absl `FlagValues` analogue)."""

  def __init__(self):
    object.__setattr__(self, "_flags", {})      # name -> _Flag
    object.__setattr__(self, "_aliases", {})    # alias -> name
    object.__setattr__(self, "_parsed", False)

  # -- definition ---------------------------------------------------------
  def _define(self, flag: _Flag, short_name: Optional[str] = None):
    # Re-definition with identical kind is tolerated (test classes call the
    # define_* functions repeatedly); the newest default wins.
    self._flags[flag.name] = flag
    if short_name:
      self._aliases[short_name] = flag.name

  def _resolve(self, name: str) -> Optional[_Flag]:
    name = self._aliases.get(name, name)
    return self._flags.get(name)

  # -- access --------------------------------------------------------------
  def __getattr__(self, name):
    f = self._resolve(name)
    if f is None:
      raise AttributeError("unknown flag %r" % name)
    return f.value

  def __setattr__(self, name, value):
    f = self._resolve(name)
    if f is None:
      raise AttributeError("unknown flag %r" % name)
    f.value = value
    f.present = True

  def __contains__(self, name):
    return self._resolve(name) is not None

  def __getitem__(self, name):
    f = self._resolve(name)
    if f is None:
      raise KeyError(name)
    return f

  def flag_values_dict(self) -> Dict[str, Any]:
    return {n: f.value for n, f in self._flags.items()}

  def is_parsed(self):
    return self._parsed

  def unparse_flags(self):
    for f in self._flags.values():
      f.value = f.default
      f.present = False
    object.__setattr__(self, "_parsed", False)

  def set_default(self, name, value):
    f = self._resolve(name)
    if f is None:
      raise UnrecognizedFlagError("cannot set default of unknown flag %r" % name)
    f.default = value
    if not f.present:
      f.value = value

  # -- parsing -------------------------------------------------------------
  def __call__(self, argv: Sequence[str], known_only: bool = False) -> List[str]:
    """Parses `argv` (argv[0] is the program name). Returns leftover positionals."""
    args = list(argv)
    rest = [args[0]] if args else []
    i = 1
    while i < len(args):
      a = args[i]
      i += 1
      if a == "--":
        rest.extend(args[i:])
        break
      if not a.startswith("-") or a == "-":
        rest.append(a)
        continue
      body = a[2:] if a.startswith("--") else a[1:]
      if "=" in body:
        name, val = body.split("=", 1)
        has_val = True
      else:
        name, val, has_val = body, None, False
      f = self._resolve(name)
      if f is None and name.startswith("no"):
        g = self._resolve(name[2:])
        if g is not None and g.kind == "bool" and not has_val:
          g.parse(False)
          continue
      if f is None:
        if known_only:
          rest.append(a)
          continue
        raise UnrecognizedFlagError("Unknown command line flag %r" % name)
      if f.kind == "bool":
        f.parse(val if has_val else True)
        continue
      if not has_val:
        if i >= len(args):
          raise IllegalFlagValueError("Missing value for flag --%s" % name)
        val = args[i]
        i += 1
      f.parse(val)
    object.__setattr__(self, "_parsed", True)
    return rest

  def get_help(self) -> str:
    lines = []
    inv = {}
    for a, n in self._aliases.items():
      inv.setdefault(n, []).append(a)
    for n in sorted(self._flags):
      f = self._flags[n]
      al = ",".join("-" + a for a in inv.get(n, []))
      lines.append("  --%s%s: %s (default: %r)" % (n, (" [" + al + "]") if al else "",
                                                f.help, f.default))
    return "\n".join(lines)


FLAGS = FlagValues()


def DEFINE_string(name, default, help, short_name=None, flag_values=FLAGS):  # pylint: disable=redefined-builtin
  flag_values._define(_Flag(name, default, help, str, "string"), short_name)


def DEFINE_integer(name, default, help, short_name=None, lower_bound=None,  # pylint: disable=redefined-builtin
                   upper_bound=None, flag_values=FLAGS):
  def parser(t):
    v = int(t)
    if lower_bound is not None and v < lower_bound:
      raise ValueError("%d below lower bound %d" % (v, lower_bound))
    if upper_bound is not None and v > upper_bound:
      raise ValueError("%d above upper bound %d" % (v, upper_bound))
    return v
  flag_values._define(_Flag(name, default, help, parser, "int"), short_name)


def DEFINE_float(name, default, help, short_name=None, flag_values=FLAGS):  # pylint: disable=redefined-builtin
  flag_values._define(_Flag(name, default, help, float, "float"), short_name)


def DEFINE_boolean(name, default, help, short_name=None, flag_values=FLAGS):  # pylint: disable=redefined-builtin
  flag_values._define(_Flag(name, default, help, _parse_bool, "bool"), short_name)


DEFINE_bool = DEFINE_boolean


def DEFINE_enum(name, default, enum_values, help, short_name=None,  # pylint: disable=redefined-builtin
                case_sensitive=True, flag_values=FLAGS):
  vals = list(enum_values)

  def parser(t):
    for v in vals:
      if t == v or (not case_sensitive and str(t).lower() == str(v).lower()):
        return v
    raise ValueError("value should be one of <%s>" % "|".join(map(str, vals)))
  flag_values._define(_Flag(name, default, help, parser, "enum", vals), short_name)


def DEFINE_list(name, default, help, short_name=None, flag_values=FLAGS):  # pylint: disable=redefined-builtin
  flag_values._define(_Flag(name, _parse_list(default) if default is not None else None,
                            help, _parse_list, "list"), short_name)


def DEFINE_alias(alias, original, flag_values=FLAGS):
  flag_values._aliases[alias] = original


def set_defaults(flag_values=FLAGS, **kwargs):
  """Overrides per-driver defaults (reference: `flags_core.set_defaults`)."""
  for k, v in kwargs.items():
    flag_values.set_default(k, v)


# ---------------------------------------------------------------------------
# Model-Garden flag groups.
# ---------------------------------------------------------------------------

def _has_gpu() -> bool:
  try:
    import torch  # pylint: disable=import-outside-toplevel
    return torch.cuda.device_count() > 0
  except Exception:  # pragma: no cover
    return False


def define_base(data_dir=True, model_dir=True, clean=True, train_epochs=True,
                epochs_between_evals=True, stop_threshold=True, batch_size=True,
                num_gpu=True, hooks=True, export_dir=True,
                distribution_strategy=True, run_eagerly=False):
  """Base flags (reference call: `common.py:250`)."""
  if data_dir:
    DEFINE_string("data_dir", "/tmp", "The location of the input data.", "dd")
  if model_dir:
    DEFINE_string("model_dir", "/tmp", "The location of the model checkpoint files.", "md")
  if clean:
    DEFINE_boolean("clean", False, "If set, model_dir will be removed if it exists.")
  if train_epochs:
    DEFINE_integer("train_epochs", 1, "The number of epochs used to train.", "te")
  if epochs_between_evals:
    DEFINE_integer("epochs_between_evals", 1,
                   "The number of training epochs to run between evaluations.", "ebe")
  if stop_threshold:
    DEFINE_float("stop_threshold", None,
                 "If passed, training will stop at the earlier of train_epochs and when "
                 "the evaluation metric is greater than or equal to stop_threshold.", "st")
  if batch_size:
    DEFINE_integer("batch_size", 32,
                   "Global batch size for training and evaluation. With a synchronous "
                   "strategy it is split evenly over the replicas.", "bs")
  if num_gpu:
    DEFINE_integer("num_gpus", 1 if _has_gpu() else 0,
                   "How many GPUs to use at each worker with the distribution strategy; "
                   "0 runs on the CPU.", "ng")
  if run_eagerly:
    DEFINE_boolean("run_eagerly", False,
                   "Kept for command-line compatibility; the engine always runs "
                   "stream-ordered kernels.")
  if hooks:
    DEFINE_list("hooks", "LoggingTensorHook", "Kept for command-line compatibility.", "hk")
  if export_dir:
    DEFINE_string("export_dir", None,
                  "If set, the final weights are exported to this directory as "
                  "a Keras-layout HDF5 file.", "ed")
  if distribution_strategy:
    DEFINE_string("distribution_strategy", "default",
                  "off | default | one_device | mirrored | multi_worker_mirrored | "
                  "parameter_server (case-insensitive).", "ds")


def define_performance(num_parallel_calls=False, inter_op=True, intra_op=True,
                       synthetic_data=True, max_train_steps=True, dtype=True,
                       all_reduce_alg=True, num_packs=True,
                       tf_gpu_thread_mode=False, datasets_num_private_threads=False,
                       datasets_num_parallel_batches=False, dynamic_loss_scale=False,
                       loss_scale=False, fp16_implementation=False,
                       tf_data_experimental_slack=False, enable_xla=False,
                       force_v2_in_keras_compile=False):
  """Performance flags (reference call: `common.py:251-258`)."""
  if num_parallel_calls:
    DEFINE_integer("num_parallel_calls", 8,
                   "Number of records processed in parallel by the input workers.", "npc")
  if inter_op:
    DEFINE_integer("inter_op_parallelism_threads", 0,
                   "Host threads for independent ops (0 = library default).", "inter")
  if intra_op:
    DEFINE_integer("intra_op_parallelism_threads", 0,
                   "Host threads inside one CPU op (0 = library default).", "intra")
  if synthetic_data:
    DEFINE_boolean("use_synthetic_data", False,
                   "Use one device-resident synthetic batch instead of the input pipeline.",
                   "synth")
  if max_train_steps:
    DEFINE_integer("max_train_steps", None,
                   "Stop training after this many global steps.", "mts")
  if dtype:
    DEFINE_enum("dtype", "fp32", ["fp16", "bf16", "fp32"],
                "Storage type of activations and packed weights; variables, accumulation and "
                "statistics are always fp32.  fp32 (the reference's default): exact fp32 matrix "
                "instructions.  fp16 (the reference's mixed precision): IEEE half with loss "
                "scaling.  bf16: the type the MI355X engine is tuned for (fused statistics, "
                "multi-stage pipelines), no loss scaling needed.  All three run on the native "
                "kernels on a GPU.", "dt")
    if loss_scale:
      DEFINE_string("loss_scale", None,
                    "Loss scale for reduced-precision training: a positive number, or "
                    "'dynamic'%s. Default 1 for fp32/bf16 and 128 for fp16."
                    % ("" if dynamic_loss_scale else " (not accepted here)"), "ls")
  if all_reduce_alg:
    DEFINE_string("all_reduce_alg", None,
                  "Gradient all-reduce algorithm: nccl (RCCL all-reduce per fusion bucket), "
                  "ring (reduce-scatter + all-gather), hierarchical_copy (reduce to rank 0 "
                  "then broadcast), mesh (all-link shard exchange over the point-to-point "
                  "xGMI links). Default: nccl.", "ara")
  if num_packs:
    DEFINE_integer("num_packs", 1,
                   "Number of fusion buckets the gradients are packed into for the "
                   "all-reduce (0 = one collective per tensor).")
  if tf_gpu_thread_mode:
    DEFINE_string("tf_gpu_thread_mode", None,
                  "gpu_private | gpu_shared: dedicate host launch threads per GPU.", "gt_mode")
    DEFINE_integer("per_gpu_thread_count", 0,
                   "Host threads dedicated to each GPU when tf_gpu_thread_mode is set.",
                   "pgtc")
  if datasets_num_private_threads:
    DEFINE_integer("datasets_num_private_threads", None,
                   "Number of private worker processes of the input pipeline.")
  if datasets_num_parallel_batches:
    DEFINE_integer("datasets_num_parallel_batches", None,
                   "Number of batches assembled in parallel by the input pipeline.")
  if tf_data_experimental_slack:
    DEFINE_boolean("tf_data_experimental_slack", False,
                   "Lets the last prefetch stage of the input pipeline run ahead.")
  if enable_xla:
    DEFINE_boolean("enable_xla", False,
                   "Accepted for compatibility: the native engine always runs a fused, "
                   "graph-captured step; the flag forces drop_remainder batching.")
  if force_v2_in_keras_compile:
    DEFINE_boolean("force_v2_in_keras_compile", None,
                   "Accepted for compatibility with the reference command lines.")
  if fp16_implementation:
    DEFINE_enum("fp16_implementation", "keras", ["keras", "graph_rewrite"],
                "Accepted for compatibility.")


def define_image(data_format=True):
  if data_format:
    DEFINE_enum("data_format", None, ["channels_first", "channels_last"],
                "Memory layout of the activations; inputs are NHWC either way, as in the "
                "reference, and results do not depend on it.  The PyTorch-operator executor "
                "stores activations channel-major (channels_first) or pixel-major "
                "(channels_last).  The native MI355X engine is written for pixel-major "
                "tensors only: it accepts channels_first with a warning and computes in NHWC.  "
                "Default: channels_last.", "df")


def define_benchmark(benchmark_log_dir=True, bigquery_uploader=True):
  DEFINE_enum("benchmark_logger_type", "BaseBenchmarkLogger",
              ["BaseBenchmarkLogger", "BenchmarkFileLogger", "BenchmarkBigQueryLogger"],
              "Where benchmark metrics go: the log, or JSON files under "
              "benchmark_log_dir.", "blt")
  DEFINE_string("benchmark_test_id", None, "Identifier recorded with the benchmark run.",
                "bti")
  DEFINE_integer("log_steps", 100, "Interval (in steps) of the throughput log lines.")
  if benchmark_log_dir:
    DEFINE_string("benchmark_log_dir", None, "Directory for BenchmarkFileLogger output.",
                  "bld")
  if bigquery_uploader:
    DEFINE_string("gcp_project", None, "Accepted for compatibility (no uploader).", "gp")
    DEFINE_string("bigquery_data_set", "test_benchmark", "Accepted for compatibility.", "bds")
    DEFINE_string("bigquery_run_table", "benchmark_run", "Accepted for compatibility.", "brt")
    DEFINE_string("bigquery_run_status_table", "benchmark_run_status",
                  "Accepted for compatibility.", "brst")
    DEFINE_string("bigquery_metric_table", "benchmark_metric",
                  "Accepted for compatibility.", "bmt")


def define_distribution(worker_hosts=True, task_index=True):
  if worker_hosts:
    DEFINE_string("worker_hosts", None,
                  "Comma-separated list of worker ip:port pairs for multi-worker training.")
  if task_index:
    DEFINE_integer("task_index", -1,
                   "Index of this worker in worker_hosts (for multi-worker training).")


DTYPE_MAP = {
    # name -> (torch dtype name, default loss scale)
    "fp16": ("float16", 128),
    "bf16": ("bfloat16", 1),
    "fp32": ("float32", 1),
}


def get_dtype(flags_obj) -> str:
  """Returns the torch dtype *name* for `--dtype` (reference `get_tf_dtype`)."""
  return DTYPE_MAP[flags_obj.dtype][0]


get_tf_dtype = get_dtype


def get_loss_scale(flags_obj, default_for_fp16=128):
  """`"dynamic"`, or a float (reference: `flags_core.get_loss_scale`)."""
  ls = getattr(flags_obj, "loss_scale", None) if "loss_scale" in flags_obj else None
  if ls == "dynamic":
    return "dynamic"
  if ls is not None:
    v = float(ls)
    if v <= 0:
      raise IllegalFlagValueError("loss_scale should be a positive number or 'dynamic'")
    return v
  if flags_obj.dtype == "fp16":
    return default_for_fp16
  return 1


def get_num_gpus(flags_obj) -> int:
  """Resolves `--num_gpus` (-1 means "all visible")."""
  n = flags_obj.num_gpus
  if n != -1:
    return n
  import torch  # pylint: disable=import-outside-toplevel
  return torch.cuda.device_count()


def run(main: Callable, argv: Optional[Sequence[str]] = None, flag_values=FLAGS):
  """`absl.app.run` analogue: parse flags, call `main(remaining_argv)`."""
  argv = list(sys.argv if argv is None else argv)
  if any(a in ("--help", "-help", "--helpfull", "-h") for a in argv[1:]):
    print(flag_values.get_help())
    sys.exit(0)
  rest = flag_values(argv)
  sys.exit(main(rest) or 0)
