"""Native MI355X executor: compiles a `ModelSpec` into a plan of hand-written HIP kernels.

Planning happens once per (model, batch size): every activation, gradient, statistic and
packed-weight buffer is allocated up front (288 GB of HBM3E per GPU means nothing is
recomputed and nothing is freed during training), tile configurations are chosen per layer
shape, and every kernel launch of the forward and backward pass is recorded into a native
`Plan` with its final device addresses.  A training step is then

    copy batch into the staging buffers -> plan.run(forward) -> plan.run(backward segment 0)
    -> bucket 0 ready (all-reduce starts on the communication stream) -> segment 1 -> ...

i.e. a handful of C++ calls that enqueue a few hundred kernels on the HIP stream; with no
gradient reducer the whole step can be captured once and replayed as a single hipGraph.

Kernel inventory per layer kind:

* conv / dense forward and data gradient: `conv_igemm.hip` (implicit GEMM on MFMA, also the
  four-parity decomposition of strided data gradients);
* conv / dense weight gradient: `conv_wgrad.hip` (split-K, transposing LDS reads, fp32 atomics
  straight into the flat gradient bucket, Keras HWIO layout);
* BatchNorm(+add)(+ReLU): `bn.hip`; max / mean pooling: `pool.hip`; loss: `loss.hip`;
  optimizer, weight repacking, input preprocessing: `optim.hip`.

Mixed precision: bf16 activations and packed weights, fp32 master weights, fp32 accumulation,
statistics, loss and gradient buffers (`resnet_imagenet_main.py:97-100` keeps fp32 variables
the same way for fp16).
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional, Tuple

import torch

from ..models.graph import ModelSpec, Op
from ..ops import autotune, native
from ..ops.conv_plan import ConvGeom, round_up
from ..utils import logging
from .params import Bucket, ParamStore

__all__ = ["HipExecutor", "supports"]


def supports(spec: ModelSpec) -> bool:
  """True when every op of the model has a native kernel path."""
  first = spec.ops[0] if spec.ops else None
  if _is_mlp(spec):
    return True           # flatten + dense chain: streamed by the kernels of dense.hip
  if first is None or first.kind != "conv" or first.inputs[0] != "input":
    return False          # the planner stages the input batch for a leading convolution
  for op in spec.ops:
    if op.kind not in ("conv", "bn", "maxpool", "mean", "dense"):
      return False
    if op.kind == "dense" and op.attrs["in_features"] % 8:
      return False
    if op.kind == "bn":
      c = op.attrs["channels"]
      if c < 8 or c > 2048 or c & (c - 1):
        return False
  return True


def _is_mlp(spec: ModelSpec) -> bool:
  """Flatten followed by a chain of dense layers (the reference's trivial model,
  `image_classification/trivial_model.py:26-41`), at most 4096 logits."""
  ops = spec.ops
  if len(ops) < 2 or ops[0].kind != "flatten" or ops[0].inputs[0] != "input":
    return False
  prev = ops[0].output
  for op in ops[1:]:
    if op.kind != "dense" or list(op.inputs) != [prev]:
      return False
    prev = op.output
  return prev == spec.logits_name and spec.num_classes <= 4096


class _TypedPlan:
  """A native `Plan` whose every op description carries the executor's element type."""

  def __init__(self, plan, dt: int):
    self._plan = plan
    self._dt = dt

  def add(self, kind, desc, name="", lane=0):
    if self._dt and "dt" not in desc:
      desc = dict(desc, dt=self._dt)
    return self._plan.add(kind, desc, name, lane)

  def __getattr__(self, item):
    return getattr(self._plan, item)


# element types of the native engine: name -> (code passed to the kernels, torch dtype)
ELEMENT_TYPES = {"bf16": (0, torch.bfloat16), "bfloat16": (0, torch.bfloat16),
                 "fp16": (1, torch.float16), "float16": (1, torch.float16),
                 "fp32": (2, torch.float32), "float32": (2, torch.float32)}


class HipExecutor:
  native = True
  # rows of atomically accumulated BatchNorm sums per layer (ResNet-50 step: 8 rows 20.30 ms, 16: 20.13, 32: 20.12, 64: 20.18)
  STATS_REPLICAS = int(os.environ.get("DTF_BN_STATS_REPLICAS", "32"))
  _PIPE_F16_CFGS = (17, 18, 20, 21, 23, 24, 30, 31)   # pipeline tiles instantiated for IEEE half
  # A data gradient also accumulates the backward statistics of the BatchNorm it feeds when the
  # gradient tensor is at most this many bytes.  The epilogue re-reads the BatchNorm input at
  # the size of its output: for small tensors that replaces a reduction sweep which is pure
  # launch latency (ResNet-56 step 3.19 -> 2.97 ms).  For large tensors the two are equal: the
  # sweep reads dz and y, the epilogue y only, but dz was written by the launch before and the
  # sweep - walking it from the end that launch touched last - finds it in the 256 MB
  # memory-side cache, so both forms fetch the same bytes from memory (ResNet-50 bs256, serial
  # kernel time with every layer fused: reduction sweeps -1.48 ms, convolutions +1.53 ms; step
  # 20.15 ms at 8 MB, 20.07 - 20.10 at 16 - 128 MB, 20.19 at 256 MB, 20.35 with every layer;
  # `results/bstats_threshold_sweep.txt`).
  BWD_STATS_SMALL_BYTES = 64 << 20
  # A training BatchNorm (+ ReLU) whose only consumer is a unit-stride convolution is folded
  # into that convolution (it normalises its operand on the way into LDS and writes the
  # normalised tensor on the side) when the tensor is at most this many bytes: for the
  # latency-bound layers of the CIFAR ResNets that removes a launch per layer; large layers
  # keep the sweep (their convolutions fill LDS by DMA, which has no place for arithmetic).
  BN_PROLOGUE_BYTES = 8 << 20

  def __init__(self, spec: ModelSpec, store: ParamStore, batch_size: int, dtype: str = "bf16",
               buckets: Optional[List[Bucket]] = None, use_graph: Optional[bool] = None,
               fused_bn_stats: Optional[bool] = None, deterministic: bool = False):
    if store.device.type != "cuda":
      raise RuntimeError("HipExecutor needs a GPU device")
    if dtype not in ELEMENT_TYPES:
      raise ValueError("the native engine stores activations as bf16, fp16 or fp32 (got %s)" % dtype)
    # bf16 is the type the engine is tuned for; fp16 (the reference's mixed precision, with loss
    # scaling) runs the same kernels and fusions instantiated for IEEE half; fp32 (the
    # reference's default precision, exact fp32 matrix instructions) runs the same plan without
    # the fused statistics, the folded inference BatchNorm and the multi-stage pipelines
    self.dt, self.act_dtype = ELEMENT_TYPES[dtype]
    self.esize = torch.empty((), dtype=self.act_dtype).element_size()
    if not supports(spec):
      raise ValueError("model %s contains ops without a native kernel" % spec.name)
    native.ext()   # raises if the extension is missing: no silent fallback on the GPU
    self.spec = spec
    self.store = store
    self.N = int(batch_size)
    self.device = store.device
    self.buckets = buckets if buckets is not None else store.make_buckets(1)
    # True: replay the step as a hipGraph; False: enqueue the plan from C++ every step; None
    # ("auto", or DTF_HIP_GRAPH unset): time a few steps each way and keep the faster - graph
    # replay removes the launch overhead that dominates small models, eager launch lets the
    # side lane really overlap the main lane, which wins on large ones
    if use_graph is None and "DTF_HIP_GRAPH" in os.environ:
      use_graph = os.environ["DTF_HIP_GRAPH"] != "0"
    self.use_graph = use_graph
    self._pre_index: Optional[int] = None
    self._direct_in = None
    self._staged = [None, None]          # identity keys of the images / labels staged last
    self._staged_keep = [None, None]     # ... and the tensors themselves (see _stage)
    self._calib: Dict[str, list] = {"eager": [], "graph": []}
    self._calib_step = 0
    # BatchNorm statistics accumulated in the convolution epilogue with float atomics (one
    # sweep over the activations less, summation order not fixed) or by the separate
    # deterministic sweep
    # deterministic: every floating-point sum of the step has a fixed order - BatchNorm
    # statistics by the partial-row sweeps, weight gradients without a split of the pixel range
    # (one block per output tile: its single atomic add lands on a zeroed element), bias
    # gradients by one block per column group.  Slower (a 1x1 layer with 64 input channels has
    # two weight-gradient tiles for 800 000 pixels); two runs then agree bit for bit.
    self.deterministic = bool(deterministic)
    if self.deterministic:
      fused_bn_stats = False
    if fused_bn_stats is None:
      fused_bn_stats = os.environ.get("DTF_BN_FUSED_STATS", "1") != "0"
    self.fused_bn_stats = bool(fused_bn_stats)
    self.overlap_wgrad = os.environ.get("DTF_WGRAD_STREAM", "1") != "0"
    self.wgrad_group = int(os.environ.get("DTF_WGRAD_GROUP", "0"))   # 0: decided while planning
    # several layers' weight gradients in one launch: opt-in, fewer launches did not pay
    # (ResNet-56: 2.40 against 2.35 ms per step - the merged launch waits for its last layer)
    self.wgrad_batch = os.environ.get("DTF_WGRAD_BATCH", "0") == "1"
    self._wgrad_queue: List[Tuple[Dict, str]] = []
    self._keep: List[torch.Tensor] = []      # owns every planned buffer
    self._soft_joins = False
    self._graph_stream: Optional[torch.cuda.Stream] = None
    self._train_graph = None
    self._pool_idx: Dict[str, torch.Tensor] = {}
    self._seen_version = -1
    with torch.cuda.device(self.device):
      if _is_mlp(spec):
        self._plan_mlp()
      else:
        self._plan_model()

  # ----------------------------------------------------------------------------------------
  # buffers
  # ----------------------------------------------------------------------------------------
  def _alloc(self, shape, dtype=None, zero=True) -> torch.Tensor:
    """A planned buffer; `dtype` None: an activation tensor in the executor's element type."""
    t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype or self.act_dtype,
                                               device=self.device)
    self._keep.append(t)
    return t

  def _var_ptr(self, name: str, which: str = "value") -> int:
    return self.store.view(name, which).data_ptr()

  # ----------------------------------------------------------------------------------------
  # planning
  # ----------------------------------------------------------------------------------------
  def _plan_model(self):
    spec, N = self.spec, self.N
    H, W, C = spec.input_shape
    self.images = self._alloc((N, H, W, C), torch.float32)
    self.labels = self._alloc((N,), torch.int32)
    self.accum = self._alloc((2,), torch.float32)            # [loss sum, correct count]
    self.loss_scale_dev = self._alloc((1,), torch.float32)
    self.loss_scale_dev.fill_(1.0)
    self._loss_scale = 1.0

    # ---- first pass: geometry of every layer, activation tensors -------------------------
    acts: Dict[str, torch.Tensor] = {}
    self.acts = acts
    geoms: Dict[str, ConvGeom] = {}
    self.geoms = geoms
    first_conv = next((o for o in spec.ops if o.kind == "conv"), None)
    consumers: Dict[str, int] = {}
    for op in spec.ops:
      for t in op.inputs:
        consumers[t] = consumers.get(t, 0) + 1

    # input staging for the first convolution
    if first_conv is None or first_conv.inputs[0] != "input":
      raise ValueError("native engine expects the model to start with a convolution")
    a = first_conv.attrs
    merged = a["cin"] <= 4 and a["kw"] <= 8 and a["stride"] % 2 == 0
    g0 = ConvGeom(N, H, W, a["cin"], a["kh"], a["kw"], a["cout"], a["stride"], tuple(a["pad"]),
                  mode="merged" if merged else "normal", dt=self.dt)
    geoms[first_conv.name] = g0
    acts["input"] = self._alloc(g0.input_shape)
    self._pre = dict(N=N, H=H, W=W, Cin=C, pt=a["pad"][0] if merged else 0,
                     pl=a["pad"][2] if merged else 0, OH=g0.Hs, OW=g0.Ws, Cpad=g0.Cs)
    self._pre["in"] = self.images.data_ptr()
    self._pre["out"] = acts["input"].data_ptr()

    shapes = spec.shapes
    for op in spec.ops:
      at = op.attrs
      if op.kind == "conv":
        if op.name not in geoms:
          h, w, c = shapes[op.inputs[0]]
          geoms[op.name] = ConvGeom(N, h, w, c, at["kh"], at["kw"], at["cout"], at["stride"],
                                    tuple(at["pad"]), dt=self.dt)
        acts[op.output] = self._alloc(geoms[op.name].output_shape)
      elif op.kind == "dense":
        geoms[op.name] = ConvGeom(N, 1, 1, at["in_features"], 1, 1, at["out_features"], dt=self.dt)
        acts[op.output] = self._alloc((N, geoms[op.name].ldc))
      elif op.kind in ("bn", "maxpool"):
        ho, wo, c = shapes[op.output]
        acts[op.output] = self._alloc((N, ho, wo, c))
      elif op.kind == "mean":
        acts[op.output] = self._alloc((N, shapes[op.output][0]))

    # ---- packed weights ------------------------------------------------------------------
    self._plan_weight_packing()

    # ---- BatchNorm + ReLU whose only consumer is a max-pool (the ResNet-50 stem): one fused
    # sweep each way, the full-resolution activation and its gradient are never materialised
    self._bn_pool: Dict[str, Op] = {}       # BatchNorm name -> max-pool op
    self._pool_bn: Dict[str, Op] = {}       # max-pool name -> BatchNorm op
    if os.environ.get("DTF_FUSE_BN_POOL", "1") != "0":
      by_input: Dict[str, List[Op]] = {}
      for op in spec.ops:
        for t in op.inputs:
          by_input.setdefault(t, []).append(op)
      for b in spec.ops:
        c = b.attrs.get("channels", 0)
        users = by_input.get(b.output, [])
        if (b.kind == "bn" and b.attrs["relu"] and len(b.inputs) == 1 and len(users) == 1
            and users[0].kind == "maxpool" and c >= 8 and c & (c - 1) == 0 and c <= 2048
            and users[0].attrs["stride"] <= users[0].attrs["k"] <= 2 * users[0].attrs["stride"]
            and users[0].attrs["k"] ** 2 < 255 and b.output != spec.logits_name):
          self._bn_pool[b.name] = users[0]
          self._pool_bn[users[0].name] = b

    # ---- BatchNorm of a projection shortcut (no ReLU, no residual) whose only consumer is
    # another BatchNorm that takes it as its residual: its normalised output is never
    # materialised in training, the consumer applies scale / shift to the shortcut
    # convolution's output while adding it
    self._bn_lazy: Dict[str, Op] = {}        # BatchNorm name -> consumer
    self._bn_lazy_out: Dict[str, Op] = {}    # its output tensor -> the BatchNorm
    if os.environ.get("DTF_BN_LAZY_SHORTCUT", "1") != "0":
      users: Dict[str, List[Op]] = {}
      for op in spec.ops:
        for t in op.inputs:
          users.setdefault(t, []).append(op)
      for b in spec.ops:
        u = users.get(b.output, [])
        if (b.kind == "bn" and not b.attrs["relu"] and len(b.inputs) == 1 and len(u) == 1
            and u[0].kind == "bn" and len(u[0].inputs) > 1 and u[0].inputs[1] == b.output
            and u[0].inputs[0] != b.output and u[0].name not in self._bn_pool
            and b.output != spec.logits_name
            and tuple(acts[b.inputs[0]].shape) == tuple(acts[u[0].inputs[0]].shape)
            and b.attrs["channels"] == u[0].attrs["channels"]):
          self._bn_lazy[b.name] = u[0]
          self._bn_lazy_out[b.output] = b

    # ---- BatchNorm side buffers -------------------------------------------------------------
    bn_ops = [o for o in spec.ops if o.kind == "bn"]
    tot = sum(o.attrs["channels"] for o in bn_ops)
    # per layer: [mean | invstd | scale | shift]; one workspace of per-block partial sums
    self.bn_saved = self._alloc((4 * max(tot, 1),), torch.float32)
    self.bn_ws = self._alloc((native.ext().bn_workspace_floats(),), torch.float32)
    # Replica rows + ticket for backward reductions that finalise in their last block instead
    # of a separate 5 us launch (zeroed once; every launch leaves them zeroed).  Opt-in: the
    # float atomics that replace the per-block partial rows (up to two million per launch onto a
    # few thousand addresses) cost far more than the launches they save - ResNet-50 bs256
    # 24.9 against 21.3 ms per step.
    self.bn_tail = self._alloc((native.ext().bn_tail_floats(),), torch.float32) \
        if self.fused_bn_stats and os.environ.get("DTF_BN_TAIL_FINALIZE", "0") == "1" else None
    bn_off: Dict[str, int] = {}
    o = 0
    for b in bn_ops:
      bn_off[b.name] = o
      o += 4 * b.attrs["channels"]
    self._bn_off = bn_off
    # fused statistics: replicas of [2][C] sums per BatchNorm whose input comes straight from
    # a convolution; ReLU bit masks ([P][C/8] bytes) for the backward sweeps
    producer = {o.output: o for o in spec.ops}
    self._bn_stats_off: Dict[str, int] = {}
    self._bn_mask: Dict[str, torch.Tensor] = {}
    so = 0
    for b in bn_ops:
      src = producer.get(b.inputs[0])
      c = b.attrs["channels"]
      if (self.fused_bn_stats and src is not None and src.kind == "conv"
          and consumers.get(b.inputs[0], 0) == 1 and c % 8 == 0
          and acts[b.inputs[0]].shape[-1] == c):
        self._bn_stats_off[b.name] = so
        so += self.STATS_REPLICAS * 2 * c
      if b.attrs["relu"] and b.name not in self._bn_pool:
        # written as 32-bit words: rounded up to a multiple of 4 bytes
        self._bn_mask[b.name] = self._alloc((round_up(acts[b.output].numel() // 8, 4),),
                                            torch.uint8, zero=False)
    # replica rows for BatchNorm *backward* statistics accumulated by the data-gradient launch
    # that produces the layer's incoming gradient (zeroed by the same memset as the forward ones)
    self.fused_bn_bwd_stats = self.fused_bn_stats and self.dt in (0, 1) and \
        os.environ.get("DTF_BN_FUSED_BWD_STATS", "1") != "0"      # the 16-bit element types
    self._bn_bwd_stats_off: Dict[str, int] = {}
    if self.fused_bn_bwd_stats:
      for b in bn_ops:
        c = b.attrs["channels"]
        if c % 8 == 0:
          self._bn_bwd_stats_off[b.name] = so
          so += self.STATS_REPLICAS * 2 * c
    self._bn_pre_stats: Dict[str, int] = {}      # BatchNorm name -> address, decided while planning
    self.bn_stats = self._alloc((max(so, 1),), torch.float32)
    self._stats_of_conv = {b.inputs[0]: self.bn_stats.data_ptr() + 4 * self._bn_stats_off[b.name]
                           for b in bn_ops if b.name in self._bn_stats_off}

    # ---- BatchNorm folded into its consumer convolution (see BN_PROLOGUE_BYTES) -------------
    self._bn_prologue: Dict[str, Op] = {}       # BatchNorm name -> consumer convolution
    self._prologue_of: Dict[str, Op] = {}       # convolution name -> BatchNorm
    limit = int(os.environ.get("DTF_BN_PROLOGUE_BYTES", self.BN_PROLOGUE_BYTES))
    if self.fused_bn_stats and self.dt in (0, 1) and limit > 0:
      users: Dict[str, List[Op]] = {}
      for op in spec.ops:
        for t in op.inputs:
          users.setdefault(t, []).append(op)
      for b in bn_ops:
        # users are listed in execution order: the first one must be the convolution, any later
        # one (the residual add of the next block) reads the tensor the convolution wrote
        u = users.get(b.output, [])
        c = b.attrs["channels"]
        if (b.name not in self._bn_stats_off or b.name in self._bn_pool
            or b.name in self._bn_lazy or not u or u[0].kind != "conv"
            or u[0].inputs[0] != b.output
            or b.output == spec.logits_name or not 8 <= c <= 128 or c & (c - 1)
            or acts[b.output].numel() * self.esize > limit):
          continue
        if len(b.inputs) > 2 or (len(b.inputs) == 2 and (
            b.inputs[1] in self._bn_lazy_out
            or tuple(acts[b.inputs[1]].shape) != tuple(acts[b.output].shape))):
          continue                     # a residual that is normalised on the fly keeps the sweep
        g = geoms[u[0].name]
        pt, _pb, pl, _pr = g.pad
        if (g.mode == "normal" and g.stride == 1 and g.KH % 2 == 1 and g.KW % 2 == 1
            and pt == g.KH // 2 and pl == g.KW // 2 and g.OH == g.H and g.OW == g.W
            and g.Cin == c and g.Cs == c and g.Cout <= 64
            and g.W <= 64 and 64 % g.W == 0 and (g.H * g.W) % 64 == 0
            and (64 // g.W + g.KH - 1) * (g.W + g.KW - 1) * c * 2 <= 16384
            and -(-g.KH * g.KW * c // 32) * min(32, g.Cout) * 64 <= 40960):
          self._bn_prologue[b.name] = u[0]
          self._prologue_of[u[0].name] = b

    # ---- BatchNorm + ReLU applied inside the consumer's tile (whole-rows kernels) --------------
    # Where the only consumer of a BatchNorm is a 3x3 layer that the whole-rows kernels serve in the
    # forward pass AND in the weight gradient, both stage their operand in LDS once per element: the
    # scale / shift / ReLU runs there (`in_scale` of the convolution, `x_scale` of the weight
    # gradient) and the normalised tensor is never written or read - the apply sweep (one read, one
    # write of the tensor) disappears, the backward sweeps take the ReLU mask from the sign of
    # the recomputed value (and its backward statistics are not accumulated by a data gradient,
    # which would need the mask bits of the apply sweep).  DTF_BN_IN_TILE=0: never, 2: the
    # whole-rows tiles wherever they serve the layer, whatever the table says.
    # The same for a pointwise consumer (the BatchNorm in front of the last layer of every
    # bottleneck block): the pipeline kernel and the general weight-gradient kernel transform
    # their operand fragments in registers, the layers are bound by their memory traffic and the
    # vector unit is idle (ResNet-50 bs256: 19.13 -> 19.01 ms per step in an alternating A/B on one machine).
    self._bn_in_tile: Dict[str, Op] = {}        # BatchNorm name -> consumer convolution
    self._in_tile_of: Dict[str, Op] = {}        # convolution name -> BatchNorm
    self._in_tile_cfg: Dict[str, Tuple[int, int]] = {}   # convolution name -> (forward, weight-gradient) tile
    in_tile = os.environ.get("DTF_BN_IN_TILE", "1")
    in_frag = os.environ.get("DTF_BN_IN_FRAG", "1")
    # (ResNet-50 bs256, per step: off 19.13 ms, from 60 MB 19.07, from 40 MB 19.02, from 20 MB 19.01,
    # every layer 19.07 - a small layer loses more with its own backward reduction sweep than it saves)
    in_frag_bytes = int(os.environ.get("DTF_BN_IN_FRAG_BYTES", str(20 << 20))) if in_frag != "2" else 0
    in_frag_force = int(os.environ.get("DTF_BN_IN_FRAG_FORCE_BYTES", str(64 << 20)))
    if self.fused_bn_stats and self.dt in (0, 1) and in_tile != "0":
      users = {}
      for op in spec.ops:
        for t in op.inputs:
          users.setdefault(t, []).append(op)
      probe = torch.zeros(16, device=self.device)
      pp = probe.data_ptr()
      for b in bn_ops:
        u = users.get(b.output, [])
        if (b.name not in self._bn_stats_off or b.name in self._bn_pool or b.name in self._bn_lazy
            or b.name in self._bn_prologue or not b.attrs["relu"] or len(b.inputs) != 1
            or len(u) != 1 or u[0].kind != "conv" or b.output == spec.logits_name):
          continue
        g = geoms[u[0].name]
        # (placeholder pointers: the table is read, nothing is launched)
        fd = autotune.apply("conv", g.fwd(pp, pp, pp, stats=pp if u[0].output in self._stats_of_conv else None,
                                          stats_replicas=self.STATS_REPLICAS), measure=False)
        wd = autotune.apply("wgrad", g.wgrad(pp, pp, pp), measure=False)
        if in_tile == "2":               # tests: the whole-rows tiles wherever they serve the layer
          fd = next((dict(fd, cfg=c) for c in (52, 49) if native.ext().conv_rows_serves(dict(fd, cfg=c))), fd)
          wd = dict(wd, cfg=192)
        if in_frag != "0" and g.KH == 1 and g.KW == 1 and not 16 <= fd.get("cfg", 0) < 32:
          # a layer the table gives to the two-stage kernel: the pipeline tile that is closest
          # (2 in tests: every pointwise consumer, whatever its size)
          if in_frag == "2":
            fd = dict(fd, cfg=20)
          elif g.N * g.H * g.W * g.Cin * self.esize >= in_frag_force:
            fd = dict(fd, cfg=int(os.environ.get("DTF_BN_IN_FRAG_CFG", "17")))
        if (48 <= fd.get("cfg", 0) < 54 and native.ext().conv_rows_serves(fd)
            and wd.get("cfg") == 192 and native.ext().wgrad_rows_serves(wd)):
          self._bn_in_tile[b.name] = u[0]
          self._in_tile_of[u[0].name] = b
          self._in_tile_cfg[u[0].name] = (fd["cfg"], wd["cfg"])
        elif (in_frag != "0" and g.KH == 1 and g.KW == 1 and g.stride == 1 and g.mode == "normal"
              and g.Cs == g.Cin and g.Cin <= 512 and g.Cin % 32 == 0
              and g.N * g.H * g.W * g.Cin * self.esize >= in_frag_bytes
              and fd.get("cfg", 0) in (self._PIPE_F16_CFGS if self.dt == 1 else range(16, 32))):
          # pointwise consumer: the pipeline kernel and the general weight-gradient kernel apply
          # scale / shift / ReLU to their fragments between the LDS read and the matrix
          # instructions (DTF_BN_IN_FRAG=0: never, 2: with a pipeline tile whatever the table says;
          # DTF_BN_IN_FRAG_BYTES: smallest input tensor)
          self._bn_in_tile[b.name] = u[0]
          self._in_tile_of[u[0].name] = b
          self._in_tile_cfg[u[0].name] = (fd["cfg"], wd["cfg"])
      self._keep.append(probe)

    ext = native.ext()
    self.fwd_plan = _TypedPlan(ext.Plan(), self.dt)
    self.eval_plan = _TypedPlan(ext.Plan(), self.dt)
    self.bwd_plan = _TypedPlan(ext.Plan(), self.dt)
    self._build_forward(self.fwd_plan, training=True)
    hot_after_forward = (dict(self._sweep_hot), self._sweep_clock)
    self._build_forward(self.eval_plan, training=False)
    # the backward plan runs behind the training forward plan: it continues that plan's record
    # of which end of each tensor is still cached
    self._sweep_hot, self._sweep_clock = hot_after_forward
    self._build_backward(consumers)
    autotune.save()
    self.probs = None
    nbytes = sum(t.numel() * t.element_size() for t in self._keep)
    logging.info("HipExecutor planned %s for batch %d: %d fwd + %d bwd launches, %.2f GB of "
                 "buffers, %d gradient buckets", spec.name, N, self.fwd_plan.size(),
                 self.bwd_plan.size(), nbytes / 2 ** 30, len(self.buckets))

  def _plan_mlp(self):
    """Plan of a flatten + dense chain: every layer streams its fp32 master weights
    (`dense.hip`), nothing is packed.  Forward: cast the staged images to the element type as
    [N, H*W*C], one launch per layer, loss.  Backward, last layer first: weight gradient,
    bias gradient (column sums), data gradient - plain stores into the zeroed flat buffer."""
    spec, N = self.spec, self.N
    H, W, C = spec.input_shape
    self.images = self._alloc((N, H, W, C), torch.float32)
    self.labels = self._alloc((N,), torch.int32)
    self.accum = self._alloc((2,), torch.float32)
    self.loss_scale_dev = self._alloc((1,), torch.float32)
    self.loss_scale_dev.fill_(1.0)
    self._loss_scale = 1.0
    self.geoms, self._bn_lazy, self._bn_pool = {}, {}, {}
    self._pack_args = None
    acts: Dict[str, torch.Tensor] = {}
    self.acts = acts
    flat = spec.ops[0]
    layers = spec.ops[1:]
    K0 = H * W * C
    acts["input"] = self.images
    acts[flat.output] = self._alloc((N, K0))
    for op in layers:
      acts[op.output] = self._alloc((N, op.attrs["out_features"]))
    ext = native.ext()
    self.fwd_plan = _TypedPlan(ext.Plan(), self.dt)
    self.eval_plan = _TypedPlan(ext.Plan(), self.dt)
    self.bwd_plan = _TypedPlan(ext.Plan(), self.dt)
    logits = acts[spec.logits_name]
    self.dlogits = self._alloc(tuple(logits.shape))
    for plan, training in ((self.fwd_plan, True), (self.eval_plan, False)):
      plan.add("memset", dict(ptr=self.accum.data_ptr(), nbytes=8), "zero_loss")
      plan.add("cast", {"in": self.images.data_ptr(), "out": acts[flat.output].data_ptr(),
                        "n": N * K0}, flat.name)
      for op in layers:
        at = op.attrs
        plan.add("dense_fwd", dict(x=acts[op.inputs[0]].data_ptr(), w=self._var_ptr(op.name + "/kernel"),
                                   bias=self._var_ptr(op.name + "/bias"), out=acts[op.output].data_ptr(),
                                   N=N, K=at["in_features"], Cout=at["out_features"]), op.name)
      plan.add("softmax_xent", dict(
          logits=logits.data_ptr(), labels=self.labels.data_ptr(),
          dlogits=self.dlogits.data_ptr() if training else None, accum=self.accum.data_ptr(),
          N=N, C=spec.num_classes, ld=logits.shape[1], grad_scale=1.0 / N, loss_weight=1.0 / N,
          grad_scale_dev=self.loss_scale_dev.data_ptr()), "softmax_xent")
    # ---- backward ------------------------------------------------------------------------------
    plan = self.bwd_plan
    grads: Dict[str, torch.Tensor] = {spec.logits_name: self.dlogits}
    self.act_grads = grads
    var_bucket, remaining = {}, {}
    for b in self.buckets:
      remaining[b.index] = len(b.var_names)
      for v in b.var_names:
        var_bucket[v] = b.index
    by_index = {b.index: b for b in self.buckets}
    self._segments = []
    seg_begin = 0
    plan.add("memset", dict(ptr=self.store.grads.data_ptr(), nbytes=self.store.grads.numel() * 4),
             "zero_grads")
    for op in reversed(layers):
      at = op.attrs
      K, Co = at["in_features"], at["out_features"]
      dy = grads[op.output]
      x = acts[op.inputs[0]]
      plan.add("dense_wgrad", dict(x=x.data_ptr(), dy=dy.data_ptr(),
                                   dw=self._var_ptr(op.name + "/kernel", "grad"), N=N, K=K, Cout=Co),
               op.name + "/wgrad")
      plan.add("colsum", dict(x=dy.data_ptr(), out=self._var_ptr(op.name + "/bias", "grad"), N=N,
                              C=Co, ld=Co), op.name + "/dbias")
      if op is not layers[0]:
        grads[op.inputs[0]] = self._alloc((N, K), zero=False)
        plan.add("dense_dgrad", dict(dy=dy.data_ptr(), w=self._var_ptr(op.name + "/kernel"),
                                     dx=grads[op.inputs[0]].data_ptr(), N=N, K=K, Cout=Co),
                 op.name + "/dgrad")
      ready = []
      for v in (op.name + "/kernel", op.name + "/bias"):
        bi = var_bucket[v]
        remaining[bi] -= 1
        if remaining[bi] == 0:
          ready.append(by_index[bi])
      if ready:
        self._segments.append((seg_begin, plan.size(), ready))
        seg_begin = plan.size()
    if seg_begin != plan.size() or not self._segments:
      self._segments.append((seg_begin, plan.size(), []))
    self.probs = None
    logging.info("HipExecutor planned %s for batch %d: %d fwd + %d bwd launches (dense streaming "
                 "kernels), %d gradient buckets", spec.name, N, self.fwd_plan.size(),
                 self.bwd_plan.size(), len(self.buckets))

  def _plan_weight_packing(self):
    """bf16 copies of every conv / dense kernel: forward layout and data-gradient layout."""
    descs = []
    off = 0
    self._wf: Dict[str, int] = {}     # op name -> element offset of the forward copy
    self._wd: Dict[str, int] = {}     # op name -> element offset of the dgrad copy
    first = True
    for op in self.spec.ops:
      if op.kind not in ("conv", "dense"):
        continue
      g = self.geoms[op.name]
      e = self.store.entries[op.name + "/kernel"]
      d = g.pack_fwd()
      d.update(src_off=e.offset, dst_off=off)
      self._wf[op.name] = off
      off += round_up(d["count"], 8)
      descs.append(d)
      needs_dgrad = not (first and op.kind == "conv")
      first = False
      if needs_dgrad:
        d = g.pack_dgrad()
        d.update(src_off=e.offset, dst_off=off)
        self._wd[op.name] = off
        off += round_up(d["count"], 8)
        descs.append(d)
    blk = 0
    for d in descs:
      d["first_block"] = blk
      blk += native.pack_blocks(d)
    self.packed = self._alloc((off,))
    raw = native.pack_desc_bytes(descs)
    self.pack_descs = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
    self._keep.append(self.pack_descs)
    self._pack_args = dict(params=self.store.params.data_ptr(), packed=self.packed.data_ptr(),
                           descs=self.pack_descs.data_ptr(), ndesc=len(descs), nblocks=blk,
                           dt=self.dt)

  # ---- sweep directions --------------------------------------------------------------------
  # The GPU's 256 MB memory-side cache keeps what the last launches touched.  A launch that walks
  # a large tensor front to back leaves the tensor's *tail* there; the next launch over that
  # tensor finds it only if it starts at the tail.  Every planned launch therefore picks its
  # direction from where the tensors it touches were left hot, and leaves its own mark.
  SWEEP_CACHE_BYTES = 256 << 20
  SWEEP_MIN_BYTES = 24 << 20          # smaller tensors stay cached whole: direction irrelevant

  def _sweep_dir(self, tensors: List[Tuple[Optional[int], int]], mark: bool = True) -> int:
    """1 when the launch touching `tensors` ((address, bytes) pairs) should run back to front.

    Votes by bytes: a tensor whose hot end is its tail (last walked front to back, less than a
    cache's worth of traffic ago) votes for a reversed walk, one left hot at the head for a
    forward walk.  With `mark` the launch's own walk is recorded on every tensor it touches
    (side-lane launches only vote: they run next to the main lane's launches, not before them)."""
    mode = os.environ.get("DTF_SWEEP_ORDER", "auto")
    if mode == "0":
      return 0
    hot = self.__dict__.setdefault("_sweep_hot", {})
    clock = self.__dict__.get("_sweep_clock", 0)
    vote = 0
    for ptr, nbytes in tensors:
      if ptr is None or nbytes < self.SWEEP_MIN_BYTES:
        continue
      h = hot.get(ptr)
      if h is not None and clock - h[1] < self.SWEEP_CACHE_BYTES:
        vote += nbytes if h[0] else -nbytes       # h[0]: 1 = tail is hot
    rev = 1 if vote > 0 else 0
    if not mark:
      return rev
    total = sum(n for p, n in tensors if p is not None)
    clock += total
    self._sweep_clock = clock
    for ptr, nbytes in tensors:
      if ptr is not None:
        hot[ptr] = (0 if rev else 1, clock)
    return rev

  def _conv_tensors(self, desc: Dict) -> List[Tuple[Optional[int], int]]:
    e = self.esize
    nin = desc["N"] * desc["H"] * desc["W"] * desc["Cpix"] * e
    nout = desc["N"] * desc.get("OHfull", desc["OH"]) * desc.get("OWfull", desc["OW"]) \
        * desc.get("ldc", desc["Cout"]) * e
    return [(desc.get("in"), nin), (desc.get("out"), nout), (desc.get("addend"), nout),
            (desc.get("stats_y"), nout)]

  def _in_step_choice(self, plan, kind: str, desc: Dict, lane: int) -> Dict:
    """Tile (and split count) of a training-step launch: the entry of the in-step table
    (`ops/tuned_gfx950_step.json`, choices that minimised the time of whole training steps,
    `scripts/tune_in_step.py`) when there is one, else the stand-alone table's choice - for
    weight gradients on the side lane after the neighbour rule of `_lane_tile`."""
    if plan is getattr(self, "eval_plan", None):
      return desc
    self.__dict__.setdefault("step_launches", []).append((kind, lane, dict(desc)))
    hit = autotune.step_lookup(kind, desc, lane)
    if hit is not None:
      out = dict(desc, cfg=hit["cfg"])
      if kind == "wgrad" and "splits" in hit:
        out["splits"] = hit["splits"]
      return out
    return self._lane_tile(desc) if kind == "wgrad" and lane == 1 else desc

  def _lane_tile(self, desc: Dict) -> Dict:
    """Tile of a weight gradient that runs on the side lane of a large step.

    The tile table is measured on kernels running alone, where small tiles (128x64, 64x128,
    64x64) or four wave groups often win by a few percent.  Next to the main lane's memory-bound
    kernels the 128x128 tile with two wave groups is the better neighbour - half the blocks,
    half the operand traffic through L2 per multiply-add: ResNet-50 bs256 20.81 -> 20.44 ms per
    step with the split counts unchanged (doubling them with the tile: 20.74; the 256x128
    tile: 20.52; `scripts/debug/wgrad_force_cfg.sh`).  Launches found in the in-step table
    (`_in_step_choice`) bypass the rule.  `DTF_WGRAD_LANE_TILE=0` keeps the table."""
    if (os.environ.get("DTF_WGRAD_LANE_TILE", "1") == "0" or self.dt not in (0, 1)
        or not getattr(self, "_large_step", False) or desc.get("batch") is not None
        or desc["TI"] * desc["TJ"] * desc["Ctap"] < 256 or desc["Cout"] < 128):
      return desc
    cfg = int(desc.get("cfg", 0))
    tile, groups = cfg % 16, 1 << ((cfg % 64) // 16)
    if tile in (1, 2, 3) or (tile == 0 and groups == 4):
      desc = dict(desc, cfg=16)          # 128x128, two wave groups, two-stage ring
    mul = os.environ.get("DTF_WGRAD_LANE_SPLITS_MUL")            # A/B runs: split counts scaled
    if mul:
      desc = dict(desc, splits=max(1, int(round(desc.get("splits", 1) * float(mul)))))
    remap = os.environ.get("DTF_WGRAD_LANE_MAP")     # A/B runs: "16:24,0:24" = tile -> tile
    if remap:
      table = dict(e.split(":") for e in remap.split(","))
      if str(desc.get("cfg", 0)) in table:
        desc = dict(desc, cfg=int(table[str(desc.get("cfg", 0))]))
    return desc

  def _add(self, plan, kind: str, desc: Dict, name: str, lane: int = 0):
    """plan.add with the measured tile configuration of this shape, when one is known."""
    # a launch that normalises its input inside the tile keeps the whole-rows tile it was planned with
    # (a pointwise weight gradient that normalises X on its fragments runs any general tile)
    pinned = desc.get("cfg") if desc.get("in_scale") is not None or \
        (desc.get("x_scale") is not None and desc.get("cfg") == 192) else None
    if kind in ("conv", "wgrad"):
      desc = autotune.apply(kind, desc)
    if kind in ("conv", "wgrad") and desc.get("batch") is None:
      desc = self._in_step_choice(plan, kind, desc, lane)
    if pinned is not None:
      desc = dict(desc, cfg=pinned)
      if kind == "wgrad":
        desc.setdefault("splits", 16)      # (one split asks for the fixed summation order)
    if kind == "wgrad" and self.deterministic:
      desc = dict(desc, splits=1)
    if kind == "conv" and plan is getattr(self, "bwd_plan", None) and os.environ.get("DTF_DGRAD_LANE_MAP"):
      table = dict(e.split(":") for e in os.environ["DTF_DGRAD_LANE_MAP"].split(","))   # A/B runs
      if str(desc.get("cfg", 0)) in table:
        desc = dict(desc, cfg=int(table[str(desc.get("cfg", 0))]))
    if os.environ.get("DTF_PRINT_CFGS") and kind in ("conv", "wgrad"):
      logging.warning("cfg %s %s lane %d cfg %s splits %s", kind, name, lane, desc.get("cfg"),
                      desc.get("splits"))
    if kind == "conv" and "rev" not in desc and desc.get("pro_sums") is None \
        and desc.get("bpro_sums") is None:
      desc = dict(desc, rev=self._sweep_dir(self._conv_tensors(desc)))
    elif kind == "wgrad" and "rev" not in desc and desc.get("batch") is None:
      e = self.esize
      tensors = [(desc.get("x"), desc["N"] * desc["H"] * desc["W"] * desc["Cpix"] * e),
                 (desc.get("dy"), desc["N"] * desc["OH"] * desc["OW"] * desc.get("ldy", desc["Cout"]) * e)]
      desc = dict(desc, rev=self._sweep_dir(tensors, mark=lane == 0))
    plan.add(kind, desc, name, lane)

  def _add_wgrad(self, plan, desc: Dict, name: str):
    """Weight gradients run on the plan's side lane: they depend only on tensors that are
    complete at this point and nothing but the bucket's all-reduce / the optimizer consumes
    them, so they overlap the BatchNorm / data-gradient chain."""
    skip = os.environ.get("DTF_DEBUG_SKIP_WGRAD")      # timing experiments only: "pw56,pw28,all"
    if skip:
      pw = desc["TI"] == 1 and desc["TJ"] == 1
      if "all" in skip or (pw and ("pw%d" % desc["OH"]) in skip.split(",")):
        return
    if self.overlap_wgrad:
      # A fork is an event record + a cross-stream wait (about 18 us of host time, and a
      # barrier packet in both queues): the weight gradients of `wgrad_group` consecutive
      # layers share one.  Every buffer they read stays valid for the whole step, so a
      # weight gradient may start any number of layers late.
      self._wgrad_queue.append((desc, name))
      if len(self._wgrad_queue) >= self.wgrad_group:
        self._flush_wgrads(plan)
    else:
      self._add(plan, "wgrad", desc, name)

  def _flush_wgrads(self, plan):
    if not self._wgrad_queue:
      return
    plan.add("fork", {}, self._wgrad_queue[0][1] + "/fork")
    # Layers of identical geometry (the eighteen convolutions of a CIFAR ResNet stage) differ
    # only in their three tensors: a run of them becomes one launch with a grid z-slice per
    # layer, which takes a device table of (x, dy, dw) pointers.
    runs: List[List[Tuple[Dict, str]]] = []
    for desc, name in self._wgrad_queue:
      key = {k: v for k, v in desc.items() if k not in ("x", "dy", "dw")}
      if self.wgrad_batch and runs and runs[-1][0][2] == key:
        runs[-1].append((desc, name, key))
      else:
        runs.append([(desc, name, key)])
    for run in runs:
      if len(run) == 1:
        self._add(plan, "wgrad", run[0][0], run[0][1], lane=1)
        continue
      table = torch.tensor([[d["x"], d["dy"], d["dw"]] for d, _n, _k in run], dtype=torch.int64,
                           device=self.device)
      self._keep.append(table)
      desc = dict(run[0][2], batch=table.data_ptr(), nbatch=len(run), x=run[0][0]["x"],
                  dy=run[0][0]["dy"], dw=run[0][0]["dw"])
      # tile configuration as measured for one layer of this geometry
      tuned = autotune.apply("wgrad", run[0][0])
      desc.update(cfg=tuned.get("cfg", 0), splits=tuned.get("splits", desc.get("splits", 1)))
      plan.add("wgrad", desc, "+".join(n for _d, n, _k in run), 1)
    self._wgrad_queue = []

  def _wf_ptr(self, name):
    return self.packed.data_ptr() + self.esize * self._wf[name]

  def _wd_ptr(self, name):
    return self.packed.data_ptr() + self.esize * self._wd[name]

  # ---- forward ---------------------------------------------------------------------------
  def _build_forward(self, plan, training: bool):
    spec, N = self.spec, self.N
    acts = self.acts
    self._sweep_hot, self._sweep_clock = {}, 0
    plan.add("memset", dict(ptr=self.accum.data_ptr(), nbytes=8), "zero_loss")
    if training:
      self._pre_index = plan.size()      # position of the input staging launch in the plan
    plan.add("preprocess_pad", self._pre, "preprocess")
    if training and (self._bn_stats_off or self._bn_bwd_stats_off):
      plan.add("memset", dict(ptr=self.bn_stats.data_ptr(), nbytes=self.bn_stats.numel() * 4),
               "zero_bn_stats")
    folded = {} if training else self._plan_bn_fold(plan)
    deferred = {b.inputs[0] for b in folded.values()}
    for op in spec.ops:
      at = op.attrs
      x = acts[op.inputs[0]]
      if op.kind == "conv" and op.output in deferred:
        continue                       # launched at its BatchNorm's position, see below
      if op.kind == "bn" and op.name in folded:
        # inference: conv + BatchNorm (+ residual) (+ ReLU) is one launch, the intermediate
        # tensor is never written
        conv = self._producer[op.inputs[0]]
        g = self.geoms[conv.name]
        d = g.fwd(acts[conv.inputs[0]].data_ptr(), self._wf_ptr(conv.name),
                  acts[op.output].data_ptr(), bias=self._fold_ptr[op.name][1])
        d["scale"] = self._fold_ptr[op.name][0]
        d["relu"] = int(at["relu"])
        if len(op.inputs) > 1:
          d["addend"] = acts[op.inputs[1]].data_ptr()
        self._add(plan, "conv", d, conv.name + "+" + op.name)
        continue
      if op.kind == "conv" and training and op.name in self._prologue_of:
        # the BatchNorm in front of this layer runs in its operand path: it reads the raw output
        # of the layer before and leaves the normalised tensor (this layer's input, which the
        # weight gradient reads) and the ReLU mask on the side
        b = self._prologue_of[op.name]
        bt = b.attrs
        C = bt["channels"]
        g = self.geoms[op.name]
        mean_p = self.bn_saved.data_ptr() + 4 * self._bn_off[b.name]
        d = g.fwd(acts[b.inputs[0]].data_ptr(), self._wf_ptr(op.name), acts[op.output].data_ptr(),
                  stats=self._stats_of_conv.get(op.output), stats_replicas=self.STATS_REPLICAS)
        d.update(pro_sums=self.bn_stats.data_ptr() + 4 * self._bn_stats_off[b.name],
                 pro_replicas=self.STATS_REPLICAS, pro_gamma=self._var_ptr(b.name + "/gamma"),
                 pro_beta=self._var_ptr(b.name + "/beta"),
                 pro_mm=self._var_ptr(b.name + "/moving_mean"),
                 pro_mv=self._var_ptr(b.name + "/moving_variance"),
                 pro_save_mean=mean_p, pro_save_invstd=mean_p + 4 * C, pro_scale=mean_p + 8 * C,
                 pro_shift=mean_p + 12 * C, pro_z=x.data_ptr(),
                 pro_mask=self._bn_mask[b.name].data_ptr() if b.name in self._bn_mask else None,
                 pro_P=x.numel() // C, pro_eps=float(bt["eps"]), pro_momentum=float(bt["momentum"]),
                 pro_relu=int(bt["relu"]),
                 pro_res=acts[b.inputs[1]].data_ptr() if len(b.inputs) > 1 else None)
        d["cfg"] = 42 if g.Cout <= 16 else 45       # 64-pixel tiles always hold whole image rows
        self._add(plan, "conv", d, b.name + "+" + op.name)
      elif op.kind == "conv" and training and op.name in self._in_tile_of:
        # the BatchNorm in front is applied inside this layer's halo tile: it reads the raw output
        # of the layer before and the constants the BatchNorm's finalize launch stored
        b = self._in_tile_of[op.name]
        g = self.geoms[op.name]
        sp = self.bn_saved.data_ptr() + 4 * self._bn_off[b.name]
        cb = b.attrs["channels"]
        d = g.fwd(acts[b.inputs[0]].data_ptr(), self._wf_ptr(op.name), acts[op.output].data_ptr(),
                  stats=self._stats_of_conv.get(op.output), stats_replicas=self.STATS_REPLICAS)
        d.update(in_scale=sp + 8 * cb, in_shift=sp + 12 * cb, cfg=self._in_tile_cfg[op.name][0])
        self._add(plan, "conv", d, b.name + "+" + op.name)
      elif op.kind == "conv":
        g = self.geoms[op.name]
        st = self._stats_of_conv.get(op.output) if training else None
        self._add(plan, "conv", g.fwd(x.data_ptr(), self._wf_ptr(op.name), acts[op.output].data_ptr(),
                               stats=st, stats_replicas=self.STATS_REPLICAS), op.name)
      elif op.kind == "dense":
        g = self.geoms[op.name]
        self._add(plan, "conv", g.fwd(x.data_ptr(), self._wf_ptr(op.name), acts[op.output].data_ptr(),
                               bias=self._var_ptr(op.name + "/bias")), op.name)
      elif op.kind == "bn" and training and op.name in self._bn_prologue:
        continue                       # normalised by its consumer convolution, see above
      elif op.kind == "bn":
        C = at["channels"]
        P = x.numel() // C
        so = self._bn_off[op.name]
        mean_p = self.bn_saved.data_ptr() + 4 * so
        res = acts[op.inputs[1]].data_ptr() if len(op.inputs) > 1 else None
        common = dict(y=x.data_ptr(), res=res, z=acts[op.output].data_ptr(),
                      gamma=self._var_ptr(op.name + "/gamma"),
                      beta=self._var_ptr(op.name + "/beta"),
                      mm=self._var_ptr(op.name + "/moving_mean"),
                      mv=self._var_ptr(op.name + "/moving_variance"),
                      P=P, C=C, eps=float(at["eps"]), relu=int(at["relu"]))
        if training and op.name in self._bn_pool:
          pool = self._bn_pool[op.name]
          fused = op.name in self._bn_stats_off
          ws = (self.bn_stats.data_ptr() + 4 * self._bn_stats_off[op.name]) if fused \
              else self.bn_ws.data_ptr()
          d = self._bn_pool_geometry(op, pool)
          d.update(y=x.data_ptr(), zp=acts[pool.output].data_ptr(), gamma=common["gamma"],
                   beta=common["beta"], mm=common["mm"], mv=common["mv"], save_mean=mean_p,
                   save_invstd=mean_p + 4 * C, scale=mean_p + 8 * C, shift=mean_p + 12 * C, ws=ws,
                   fused_replicas=self.STATS_REPLICAS if fused else 0, eps=float(at["eps"]),
                   momentum=float(at["momentum"]))
          plan.add("bn_pool_fwd_train", d, op.name + "+" + pool.name)
        elif training:
          fused = op.name in self._bn_stats_off
          ws = (self.bn_stats.data_ptr() + 4 * self._bn_stats_off[op.name]) if fused \
              else self.bn_ws.data_ptr()
          mask = self._bn_mask[op.name].data_ptr() if op.name in self._bn_mask else None
          extra = {}
          if op.name in self._bn_in_tile:
            extra["apply"] = 0           # statistics only: applied inside the consumer's tile
            mask = None
          elif op.name in self._bn_lazy:
            # projection shortcut: statistics only; the block's last BatchNorm normalises this
            # layer's input on the fly when it adds the residual
            extra["apply"] = 0
          elif len(op.inputs) > 1 and op.inputs[1] in self._bn_lazy_out:
            src = self._bn_lazy_out[op.inputs[1]]
            sp = self.bn_saved.data_ptr() + 4 * self._bn_off[src.name]
            cs = src.attrs["channels"]
            common["res"] = acts[src.inputs[0]].data_ptr()
            extra.update(rscale=sp + 8 * cs, rshift=sp + 12 * cs)
          if extra.get("apply", 1):
            nb = x.numel() * self.esize
            extra["rev"] = self._sweep_dir([(common["y"], nb), (common["res"], nb),
                                            (common["z"], nb)])
          plan.add("bn_fwd_train", dict(
              save_mean=mean_p, save_invstd=mean_p + 4 * C, scale=mean_p + 8 * C,
              shift=mean_p + 12 * C, ws=ws, fused_replicas=self.STATS_REPLICAS if fused else 0,
              mask=mask, momentum=float(at["momentum"]), **common, **extra), op.name)
        else:
          plan.add("bn_infer", common, op.name)
      elif op.kind == "maxpool":
        if training and op.name in self._pool_bn:
          continue                     # produced by the fused BatchNorm sweep above
        h, w, c = spec.shapes[op.inputs[0]]
        ho, wo, _ = spec.shapes[op.output]
        if op.name not in self._pool_idx:
          self._pool_idx[op.name] = self._alloc((N, ho, wo, c), torch.uint8)
        plan.add("maxpool_fwd", dict(x=x.data_ptr(), out=acts[op.output].data_ptr(),
                                     idx=self._pool_idx[op.name].data_ptr(), N=N, H=h, W=w, C=c,
                                     OH=ho, OW=wo, k=at["k"], stride=at["stride"],
                                     pt=at["pad"][0], pl=at["pad"][2]), op.name)
      elif op.kind == "mean":
        plan.add("mean_fwd", dict(x=x.data_ptr(), out=acts[op.output].data_ptr(), N=N,
                                  HW=at["h"] * at["w"], C=at["channels"]), op.name)
    logits = acts[spec.logits_name]
    ld = logits.shape[1]
    if training:
      self.dlogits = self._alloc((N, ld))
    plan.add("softmax_xent", dict(
        logits=logits.data_ptr(), labels=self.labels.data_ptr(),
        dlogits=self.dlogits.data_ptr() if training else None, accum=self.accum.data_ptr(),
        N=N, C=spec.num_classes, ld=ld, grad_scale=1.0 / N, loss_weight=1.0 / N,
        grad_scale_dev=self.loss_scale_dev.data_ptr()), "softmax_xent")

  def _bn_pool_geometry(self, bn: Op, pool: Op) -> Dict:
    """Shared descriptor fields of the fused BatchNorm + ReLU + max-pool launches."""
    h, w, c = self.spec.shapes[pool.inputs[0]]
    ho, wo, _ = self.spec.shapes[pool.output]
    if pool.name not in self._pool_idx:
      self._pool_idx[pool.name] = self._alloc((self.N, ho, wo, c), torch.uint8)
    pa = pool.attrs
    return dict(N=self.N, H=h, W=w, C=c, OH=ho, OW=wo, k=pa["k"], stride=pa["stride"],
                pt=pa["pad"][0], pl=pa["pad"][2], idx=self._pool_idx[pool.name].data_ptr())

  def _plan_bn_fold(self, plan) -> Dict:
    """Inference: BatchNorm layers whose input comes straight from a convolution are folded into
    that convolution's epilogue (scale / shift, + residual add, + ReLU;
    reference evaluation pass `image_classification/resnet_imagenet_main.py:256-259`).  Adds the
    launch that derives scale / shift from the current parameters and moving statistics, and
    returns {BatchNorm name: op} of the folded layers."""
    import struct
    spec, acts = self.spec, self.acts
    if os.environ.get("DTF_EVAL_FOLD_BN", "1") == "0" or self.dt == 2:
      return {}       # the folded epilogue exists for the 16-bit element types
    consumers: Dict[str, int] = {}
    for op in spec.ops:
      for t in op.inputs:
        consumers[t] = consumers.get(t, 0) + 1
    self._producer = {o.output: o for o in spec.ops}
    folded = {}
    for b in spec.ops:
      if b.kind != "bn":
        continue
      src = self._producer.get(b.inputs[0])
      c = b.attrs["channels"]
      if (src is not None and src.kind == "conv" and consumers.get(b.inputs[0], 0) == 1
          and acts[b.inputs[0]].shape == acts[b.output].shape and acts[b.output].shape[-1] == c
          and b.inputs[0] != spec.logits_name):
        folded[b.name] = b
    if not folded:
      return {}
    tot = sum(b.attrs["channels"] for b in folded.values())
    self.bn_folded = self._alloc((2 * tot,), torch.float32)
    ent = self.store.entries
    raw = bytearray()
    self._fold_ptr: Dict[str, Tuple[int, int]] = {}
    o = 0
    for b in folded.values():
      c = b.attrs["channels"]
      raw += struct.pack("<5qif", ent[b.name + "/gamma"].offset, ent[b.name + "/beta"].offset,
                         ent[b.name + "/moving_mean"].offset,
                         ent[b.name + "/moving_variance"].offset, o, c, float(b.attrs["eps"]))
      base = self.bn_folded.data_ptr() + 4 * o
      self._fold_ptr[b.name] = (base, base + 4 * c)
      o += 2 * c
    assert len(raw) == native.ext().bn_fold_desc_size() * len(folded)
    self._fold_descs = torch.frombuffer(raw, dtype=torch.uint8).to(self.device)
    self._keep.append(self._fold_descs)
    plan.add("bn_fold", dict(params=self.store.params.data_ptr(),
                             state=self.store.state.data_ptr(), out=self.bn_folded.data_ptr(),
                             descs=self._fold_descs.data_ptr(), nlayers=len(folded)), "bn_fold")
    return folded

  # ---- backward ----------------------------------------------------------------------------
  def _build_backward(self, consumers: Dict[str, int]):
    spec, N = self.spec, self.N
    acts = self.acts
    plan = self.bwd_plan
    grads: Dict[str, torch.Tensor] = {spec.logits_name: self.dlogits}
    written = {spec.logits_name}
    self.act_grads = grads
    # tensor name -> (dz tensor, mask address): a gradient contribution dz * mask that has not
    # been written anywhere yet
    pending: Dict[str, Tuple[torch.Tensor, Optional[int]]] = {}

    def grad_buf(name: str) -> torch.Tensor:
      if name not in grads:
        grads[name] = self._alloc(tuple(acts[name].shape), zero=False)
      return grads[name]

    # which bucket a variable belongs to, and how many variables each bucket still waits for
    var_bucket: Dict[str, int] = {}
    remaining: Dict[int, int] = {}
    for b in self.buckets:
      remaining[b.index] = len(b.var_names)
      for v in b.var_names:
        var_bucket[v] = b.index
    self._segments: List[Tuple[int, int, List[Bucket]]] = []
    seg_begin = 0
    by_index = {b.index: b for b in self.buckets}

    def done(vars_: List[str]):
      nonlocal seg_begin
      ready = []
      for v in vars_:
        bi = var_bucket[v]
        remaining[bi] -= 1
        if remaining[bi] == 0:
          ready.append(by_index[bi])
      if ready:
        self._flush_wgrads(plan)
        if self.overlap_wgrad:
          # The bucket's gradients are complete once the side lane has caught up.  A soft join:
          # a reducer that waits for the side lane on its communication stream lets the main
          # lane run on (train_step(soft_joins=True)); otherwise the main lane waits here.
          plan.add("soft_join", {}, "join")
        self._segments.append((seg_begin, plan.size(), ready))
        seg_begin = plan.size()

    if self.wgrad_group <= 0:
      # Small steps (ResNet-56 bs128: 3 ms for ~450 launches) are bound by the host's launch
      # rate, and a fork per layer costs more host time than the launches themselves: 4 layers
      # share a fork there (2.97 -> 2.84 ms).  Large steps keep one fork per layer: grouping
      # delays the weight gradients and was slightly slower (ResNet-50 bs256: 22.75 -> 22.9 ms).
      act_bytes = sum(t.numel() * t.element_size() for t in acts.values())
      self.wgrad_group = 4 if act_bytes < (2 << 30) else 1
    self._large_step = sum(t.numel() * t.element_size() for t in acts.values()) >= (2 << 30)
    plan.add("memset", dict(ptr=self.store.grads.data_ptr(),
                            nbytes=self.store.grads.numel() * 4), "zero_grads")
    first_conv = next(o for o in spec.ops if o.kind == "conv")
    producer = {o.output: o for o in spec.ops}
    uses_left = dict(consumers)      # tensor -> consumer ops whose backward has not been planned

    small = int(os.environ.get("DTF_BWD_STATS_SMALL_BYTES", self.BWD_STATS_SMALL_BYTES))

    def bwd_stats_for(xname: str, cin: int) -> Dict:
      """Extra descriptor fields for the data gradient that writes the *final* gradient of
      `xname`: when that tensor is the output of a BatchNorm, the launch also accumulates the
      layer's backward reductions (sum g, sum g*y) so that its reduction sweep disappears."""
      src = producer.get(xname)
      if (uses_left.get(xname, 0) != 1 or src is None or src.kind != "bn"
          or src.name in self._bn_in_tile        # no mask bits: its apply sweep never ran
          or src.name not in self._bn_bwd_stats_off or cin % 8
          or cin != src.attrs["channels"]):
        return {}
      if acts[xname].numel() * self.esize > small:
        return {}
      addr = self.bn_stats.data_ptr() + 4 * self._bn_bwd_stats_off[src.name]
      self._bn_pre_stats[src.name] = addr
      return dict(stats=addr, stats_replicas=self.STATS_REPLICAS,
                  stats_y=acts[src.inputs[0]].data_ptr(),
                  stats_mask=self._bn_mask[src.name].data_ptr() if src.attrs["relu"] else None)

    pooled_grad: Dict[str, torch.Tensor] = {}    # fused max-pool name -> gradient of its output
    dual_done: set = set()                       # BatchNorms whose backward ran in a dual launch
    self.bn_ws2 = None
    bwd_pro: Dict[str, Dict] = {}                # conv output -> BatchNorm backward folded into
    prologue_limit = int(os.environ.get("DTF_BN_PROLOGUE_BYTES", self.BN_PROLOGUE_BYTES)) \
        if os.environ.get("DTF_BN_BWD_PROLOGUE", "1") != "0" else 0      # that conv's data gradient
    deferred: Dict[str, List] = {}               # tensor -> strided data gradients waiting for
    defer_ok = os.environ.get("DTF_DEFER_STRIDED_DGRAD", "1") != "0"    # its first writer
    for op in reversed(spec.ops):
      at = op.attrs
      xname = op.inputs[0]
      dout = grads.get(op.output) if op.output in written else None
      if op.kind == "bn" and op.name in dual_done:
        for t in op.inputs:              # its backward ran together with the block's last layer
          uses_left[t] = uses_left.get(t, 0) - 1
        continue
      if op.kind == "bn" and op.name in self._bn_pool:
        # the pooled gradient is scattered back inside the BatchNorm backward sweeps
        pool = self._bn_pool[op.name]
        C = at["channels"]
        mean_p = self.bn_saved.data_ptr() + 4 * self._bn_off[op.name]
        if xname in written:
          raise RuntimeError("BatchNorm input with several consumers is not planned")
        d = self._bn_pool_geometry(op, pool)
        d.update(dzp=pooled_grad.pop(pool.name).data_ptr(), y=acts[xname].data_ptr(),
                 save_mean=mean_p, save_invstd=mean_p + 4 * C,
                 gamma=self._var_ptr(op.name + "/gamma"),
                 dgamma=self._var_ptr(op.name + "/gamma", "grad"),
                 dbeta=self._var_ptr(op.name + "/beta", "grad"), dy=grad_buf(xname).data_ptr(),
                 ws=self.bn_ws.data_ptr())
        plan.add("bn_pool_bwd", d, op.name + "+" + pool.name + "/bwd")
        written.add(xname)
        done([op.name + "/gamma", op.name + "/beta"])
        for t in op.inputs:
          uses_left[t] = uses_left.get(t, 0) - 1
        continue
      if dout is None and op.output in pending and op.kind != "bn":
        dout = self._materialize_pending(plan, op.output, pending, grad_buf, written)
      if dout is None and op.output not in pending:
        raise RuntimeError("no gradient reaches %s" % op.output)
      if op.kind == "dense":
        g = self.geoms[op.name]
        plan.add("colsum", dict(x=dout.data_ptr(), out=self._var_ptr(op.name + "/bias", "grad"),
                                N=N, C=g.Cout, ld=g.ldc, slices=1 if self.deterministic else 0),
                 op.name + "/dbias")
        self._add_wgrad(plan, g.wgrad(acts[xname].data_ptr(), dout.data_ptr(),
                                      self._var_ptr(op.name + "/kernel", "grad")),
                        op.name + "/wgrad")
        dx = grad_buf(xname)
        for d in g.dgrad(dout.data_ptr(), self._wd_ptr(op.name), dx.data_ptr(),
                         accumulate=xname in written):
          self._add(plan, "conv", d, op.name + "/dgrad")
        written.add(xname)
        done([op.name + "/kernel", op.name + "/bias"])
      elif op.kind == "mean":
        dx = grad_buf(xname)
        if xname in written:
          raise RuntimeError("mean-pool input with several consumers is not planned")
        plan.add("mean_bwd", dict(dy=dout.data_ptr(), dx=dx.data_ptr(), N=N,
                                  HW=at["h"] * at["w"], C=at["channels"]), op.name + "/bwd")
        written.add(xname)
      elif op.kind == "maxpool" and op.name in self._pool_bn:
        pooled_grad[op.name] = dout
      elif op.kind == "maxpool":
        h, w, c = spec.shapes[xname]
        ho, wo, _ = spec.shapes[op.output]
        dx = grad_buf(xname)
        if xname in written:
          raise RuntimeError("max-pool input with several consumers is not planned")
        plan.add("maxpool_bwd", dict(dy=dout.data_ptr(), idx=self._pool_idx[op.name].data_ptr(),
                                     dx=dx.data_ptr(), N=N, H=h, W=w, C=c, OH=ho, OW=wo,
                                     k=at["k"], stride=at["stride"], pt=at["pad"][0],
                                     pl=at["pad"][2]), op.name + "/bwd")
        written.add(xname)
      elif op.kind == "bn":
        C = at["channels"]
        x = acts[xname]
        P = x.numel() // C
        so = self._bn_off[op.name]
        mean_p = self.bn_saved.data_ptr() + 4 * so
        inv_p = mean_p + 4 * C
        if xname in written:
          raise RuntimeError("BatchNorm input with several consumers is not planned")
        dy = grad_buf(xname)
        dz_ptr, mask_ptr = self._bn_incoming_grad(plan, op, dout, pending)
        partner = self._dual_bwd_partner(op, dout, mask_ptr, producer, consumers, written, pending)
        if partner is not None:
          # this block's projection-shortcut BatchNorm receives the same masked gradient: one
          # pair of sweeps reads dz and the mask once for both layers
          C2 = partner.attrs["channels"]
          x2 = partner.inputs[0]
          m2 = self.bn_saved.data_ptr() + 4 * self._bn_off[partner.name]
          if self.bn_ws2 is None:
            self.bn_ws2 = self._alloc((native.ext().bn_workspace_floats(),), torch.float32)
          nb = x.numel() * self.esize
          rr = self._sweep_dir([(dz_ptr, nb), (x.data_ptr(), nb), (acts[x2].data_ptr(), nb)])
          ra = self._sweep_dir([(dz_ptr, nb), (x.data_ptr(), nb), (acts[x2].data_ptr(), nb),
                                (dy.data_ptr(), nb), (grad_buf(x2).data_ptr(), nb)])
          plan.add("bn_bwd_dual", dict(
              rev_reduce=rr, rev_apply=ra,
              dz=dz_ptr, mask=mask_ptr, P=P, C=C, ws=self.bn_ws.data_ptr(),
              ws2=self.bn_ws2.data_ptr(),
              y=x.data_ptr(), mean=mean_p, invstd=inv_p, gamma=self._var_ptr(op.name + "/gamma"),
              dgamma=self._var_ptr(op.name + "/gamma", "grad"),
              dbeta=self._var_ptr(op.name + "/beta", "grad"), dy=dy.data_ptr(),
              y2=acts[x2].data_ptr(), mean2=m2, invstd2=m2 + 4 * C2,
              gamma2=self._var_ptr(partner.name + "/gamma"),
              dgamma2=self._var_ptr(partner.name + "/gamma", "grad"),
              dbeta2=self._var_ptr(partner.name + "/beta", "grad"),
              dy2=grad_buf(x2).data_ptr(),
              tail=self.bn_tail.data_ptr() if self.bn_tail is not None else None),
              op.name + "+" + partner.name + "/bwd")
          written.add(xname)
          written.add(x2)
          dual_done.add(partner.name)
          done([op.name + "/gamma", op.name + "/beta", partner.name + "/gamma",
                partner.name + "/beta"])
          for t in op.inputs:
            uses_left[t] = uses_left.get(t, 0) - 1
          continue
        if len(op.inputs) > 1:
          # The gradient of the residual branch is dz gated by this layer's ReLU mask.  It is
          # not written out: whoever consumes it (the data gradient of the block's first
          # convolution, or the BatchNorm of a projection shortcut) reads dz and the mask.
          rname = op.inputs[1]
          if rname in written or rname in pending:
            raise RuntimeError("residual gradient must be the first writer of %s" % rname)
          if dout is None:
            raise RuntimeError("residual BatchNorm %s fed by a deferred gradient" % op.name)
          pending[rname] = (dout, mask_ptr)
        if op.name not in self._bn_in_tile and \
            self._bwd_prologue_ok(op, producer.get(xname), first_conv, prologue_limit):
          # small layer: the data gradient of the convolution in front of this BatchNorm applies
          # the backward formula to dz on its way into LDS and writes dy on the side
          bwd_pro[xname] = dict(
              op=op, dz=dz_ptr,
              fields=dict(bpro_sums=self._bn_pre_stats[op.name], bpro_replicas=self.STATS_REPLICAS,
                          bpro_y=x.data_ptr(), bpro_mask=mask_ptr, bpro_mean=mean_p,
                          bpro_invstd=inv_p, bpro_gamma=self._var_ptr(op.name + "/gamma"),
                          bpro_dgamma=self._var_ptr(op.name + "/gamma", "grad"),
                          bpro_dbeta=self._var_ptr(op.name + "/beta", "grad"),
                          bpro_dy=dy.data_ptr(), bpro_P=P))
          written.add(xname)
          for t in op.inputs:
            uses_left[t] = uses_left.get(t, 0) - 1
          continue
        nb = x.numel() * self.esize
        rr = 0
        if op.name not in self._bn_pre_stats:
          rr = self._sweep_dir([(dz_ptr, nb), (x.data_ptr(), nb)])
        ra = self._sweep_dir([(dz_ptr, nb), (x.data_ptr(), nb), (dy.data_ptr(), nb)])
        relu_from_y = {}
        if op.name in self._bn_in_tile:
          mask_ptr = None
          relu_from_y = dict(relu_scale=mean_p + 8 * C, relu_shift=mean_p + 12 * C)
        plan.add("bn_bwd", dict(
            rev_reduce=rr, rev_apply=ra, **relu_from_y,
            dz=dz_ptr, y=x.data_ptr(), z=None, mask=mask_ptr,
            mean=mean_p, invstd=inv_p,
            gamma=self._var_ptr(op.name + "/gamma"),
            dgamma=self._var_ptr(op.name + "/gamma", "grad"),
            dbeta=self._var_ptr(op.name + "/beta", "grad"), dy=dy.data_ptr(), dres=None,
            ws=self._bn_pre_stats.get(op.name, self.bn_ws.data_ptr()),
            pre_replicas=self.STATS_REPLICAS if op.name in self._bn_pre_stats else 0,
            tail=self.bn_tail.data_ptr() if self.bn_tail is not None else None,
            P=P, C=C), op.name + "/bwd")
        written.add(xname)
        done([op.name + "/gamma", op.name + "/beta"])
      elif op.kind == "conv" and op.output in bwd_pro:
        # data gradient with the BatchNorm backward of the layer behind it in its operand path;
        # the weight gradient reads the dy that launch writes, so it is forked behind it
        g = self.geoms[op.name]
        bp = bwd_pro.pop(op.output)
        dx = grad_buf(xname)
        add, add_mask = None, None
        if xname in pending and xname not in written:
          pt, pm = pending.pop(xname)
          add, add_mask = pt.data_ptr(), pm
        extra = bwd_stats_for(xname, g.Cin)
        descs = g.dgrad(bp["dz"], self._wd_ptr(op.name), dx.data_ptr(),
                        accumulate=xname in written, addend=add, addend_mask=add_mask)
        assert len(descs) == 1
        d = descs[0]
        d.update(extra)
        d.update(bp["fields"])
        d["cfg"] = 42 if g.Cin <= 16 else 45        # 64-pixel tiles always hold whole image rows
        self._add(plan, "conv", d, bp["op"].name + "/bwd+" + op.name + "/dgrad")
        written.add(xname)
        for op2, g2, dout2 in deferred.pop(xname, []):
          for d2 in g2.dgrad(dout2.data_ptr(), self._wd_ptr(op2.name), dx.data_ptr(),
                             accumulate=True, skip_empty=True):
            self._add(plan, "conv", d2, op2.name + "/dgrad")
        self._add_wgrad(plan, g.wgrad(acts[xname].data_ptr(), dout.data_ptr(),
                                      self._var_ptr(op.name + "/kernel", "grad")),
                        op.name + "/wgrad")
        done([bp["op"].name + "/gamma", bp["op"].name + "/beta", op.name + "/kernel"])
      elif op.kind == "conv":
        g = self.geoms[op.name]
        # The weight gradient is forked before the data gradient is enqueued.  Forking it behind
        # the data gradient (so that it starts next to the memory-bound BatchNorm sweeps of the
        # layer below instead of next to a second matrix kernel) measured 21.9 against 21.5 ms:
        # the sweeps are shorter than a weight gradient, the lane falls behind and ends alone.
        wdesc = g.wgrad(acts[xname].data_ptr(), dout.data_ptr(),
                        self._var_ptr(op.name + "/kernel", "grad"))
        if op.name in self._in_tile_of:
          # its input was normalised inside the forward launch's tile and never written: the
          # weight gradient reads the raw tensor and applies the same constants in its tile
          b = self._in_tile_of[op.name]
          sp = self.bn_saved.data_ptr() + 4 * self._bn_off[b.name]
          cb = b.attrs["channels"]
          wdesc.update(x=acts[b.inputs[0]].data_ptr(), x_scale=sp + 8 * cb, x_shift=sp + 12 * cb)
          if self._in_tile_cfg[op.name][1] == 192:
            wdesc.update(cfg=192)
        self._add_wgrad(plan, wdesc, op.name + "/wgrad")
        if (op is not first_conv and g.has_empty_parities() and defer_ok
            and uses_left.get(xname, 0) > 1 and xname not in written and xname not in pending
            and acts[xname].numel() * self.esize > small):
          # A strided projection shortcut reaches one output parity in four.  As the first
          # writer of the block input's gradient it would fill the other three with zeros,
          # which the main branch's data gradient then reads back to accumulate; deferred
          # until that gradient has been written, it only accumulates into its own parity.
          # (Small tensors keep the order: their final writer also accumulates the
          # BatchNorm backward statistics and must cover every pixel.)
          deferred.setdefault(xname, []).append((op, g, dout))
        elif op is not first_conv:
          dx = grad_buf(xname)
          add, add_mask = None, None
          if xname in pending and xname not in written and g.Cin % 8 == 0:
            pt, pm = pending.pop(xname)
            add, add_mask = pt.data_ptr(), pm
          elif xname in pending:
            self._materialize_pending(plan, xname, pending, grad_buf, written)
          extra = bwd_stats_for(xname, g.Cin)
          for d in g.dgrad(dout.data_ptr(), self._wd_ptr(op.name), dx.data_ptr(),
                           accumulate=xname in written, addend=add, addend_mask=add_mask):
            d.update(extra)
            self._add(plan, "conv", d, op.name + "/dgrad")
          written.add(xname)
          for op2, g2, dout2 in deferred.pop(xname, []):
            for d in g2.dgrad(dout2.data_ptr(), self._wd_ptr(op2.name), dx.data_ptr(),
                              accumulate=True, skip_empty=True):
              self._add(plan, "conv", d, op2.name + "/dgrad")
        done([op.name + "/kernel"])
      for t in op.inputs:
        uses_left[t] = uses_left.get(t, 0) - 1
    if deferred:
      raise RuntimeError("deferred data gradients never issued: %s" % sorted(deferred))
    if bwd_pro:
      raise RuntimeError("BatchNorm backward passes never issued: %s" % sorted(bwd_pro))
    self._flush_wgrads(plan)
    if self.overlap_wgrad:
      plan.add("join", {}, "join")               # the step ends with the side lane collected
    if seg_begin < plan.size() or not self._segments:
      self._segments.append((seg_begin, plan.size(), []))
    left = [i for i, r in remaining.items() if r]
    if left:
      raise RuntimeError("buckets never completed: %s" % left)

  def _bwd_prologue_ok(self, bn: Op, conv: Optional[Op], first_conv: Op, limit: int) -> bool:
    """True when the backward pass of `bn` can run inside the data gradient of `conv`, the
    layer that produces its input: the reductions were accumulated by the launch that wrote
    dz, the convolution is a unit-stride one of a small layer (`conv_halo.hip` serves it with
    a 64-pixel tile) and has a data gradient at all."""
    if (limit <= 0 or self.dt not in (0, 1) or bn.name not in self._bn_pre_stats
        or conv is None or conv.kind != "conv" or conv is first_conv):
      return False
    g = self.geoms[conv.name]
    c = bn.attrs["channels"]
    pt, _pb, pl, _pr = g.pad
    k = -(-g.KH * g.KW * g.cout_pad // 32)
    return (g.mode == "normal" and g.stride == 1 and g.KH % 2 == 1 and g.KW % 2 == 1
            and pt == g.KH // 2 and pl == g.KW // 2 and g.OH == g.H and g.OW == g.W
            and g.Cout == c and g.ldc == c and g.Cin % 8 == 0 and g.Cin <= 64
            and 8 <= c <= 128 and c & (c - 1) == 0
            and self.acts[bn.inputs[0]].numel() * self.esize <= limit
            and g.W <= 64 and 64 % g.W == 0 and (g.H * g.W) % 64 == 0
            and (64 // g.W + g.KH - 1) * (g.W + g.KW - 1) * c * 2 <= 16384
            and k * min(32, round_up(g.Cin, 16)) * 64 <= 40960)

  def _dual_bwd_partner(self, op: Op, dout, mask_ptr, producer, consumers, written,
                        pending) -> Optional[Op]:
    """The BatchNorm of the projection shortcut that feeds `op` as its residual, when the two
    backward passes can share their sweeps: same shape, the partner has no ReLU and no
    residual of its own, its output is consumed by `op` only, and neither layer's reductions
    were already accumulated by a data-gradient epilogue."""
    if os.environ.get("DTF_BN_DUAL_BWD", "1") == "0":
      return None
    if len(op.inputs) < 2 or dout is None or mask_ptr is None or op.name in self._bn_pre_stats:
      return None
    rname = op.inputs[1]
    p = producer.get(rname)
    if (p is None or p.kind != "bn" or p.attrs["relu"] or len(p.inputs) != 1
        or consumers.get(rname, 0) != 1 or rname in written or rname in pending
        or p.attrs["channels"] != op.attrs["channels"] or p.name in self._bn_pool
        or p.inputs[0] in written
        or tuple(self.acts[p.inputs[0]].shape) != tuple(self.acts[op.inputs[0]].shape)):
      return None
    return p

  def _materialize_pending(self, plan, name: str, pending, grad_buf, written) -> torch.Tensor:
    """Writes a deferred `dz * mask` gradient into the gradient buffer of `name`."""
    if name in written:
      raise RuntimeError("deferred residual gradient of %s would overwrite a written one" % name)
    src, mask = pending.pop(name)
    dst = grad_buf(name)
    if mask is None:
      raise RuntimeError("unmasked deferred gradient for %s" % name)
    plan.add("mask_apply", dict(x=src.data_ptr(), mask=mask, out=dst.data_ptr(), n=dst.numel()),
             name + "/residual_grad")
    written.add(name)
    return dst

  def _bn_incoming_grad(self, plan, op: Op, dout, pending) -> Tuple[int, Optional[int]]:
    """(dz address, ReLU-mask address) for the backward of BatchNorm `op`.

    When the gradient of the output is a deferred `dz * mask` (this BatchNorm normalises a
    projection shortcut) the incoming mask is used in place of an own one."""
    own = self._bn_mask[op.name].data_ptr() if op.attrs["relu"] else None
    if dout is not None:
      return dout.data_ptr(), own
    src, mask = pending.pop(op.output)
    if own is not None:
      raise RuntimeError("BatchNorm %s has its own ReLU and a deferred masked gradient" % op.name)
    return src.data_ptr(), mask

  # ----------------------------------------------------------------------------------------
  # execution
  # ----------------------------------------------------------------------------------------
  def params_updated(self):
    """Refresh the packed bf16 weights from the fp32 master copy (one multi-tensor launch)."""
    if self._pack_args is not None:      # dense-only plans read the master weights directly
      native.run_op("pack_weights", self._pack_args)
    self._seen_version = self.store.version

  def _ensure_packed(self):
    if self._seen_version != self.store.version:
      self.params_updated()

  def _same_device(self, t: torch.Tensor) -> bool:
    """`t` lives on the executor's GPU (an index-less `cuda` device means the current one)."""
    if not t.is_cuda or self.device.type != "cuda":
      return False
    mine = self.device.index if self.device.index is not None else torch.cuda.current_device()
    return t.device.index == mine

  def _stage(self, images: torch.Tensor, labels: Optional[torch.Tensor],
             allow_direct: bool = False):
    """Makes the batch visible to the plans.  `allow_direct` (training steps only: the eager
    training path is the one that launches the preprocessing kernel on a foreign pointer)
    lets that launch read the caller's device tensor in place; evaluation and prediction run
    `eval_plan`, whose preprocessing launch reads the planned input buffer, so they copy."""
    if tuple(images.shape) != tuple(self.images.shape):
      raise ValueError("HipExecutor planned for input %s, got %s"
                       % (tuple(self.images.shape), tuple(images.shape)))
    self._direct_in = None
    # A batch that was staged last and has not been written since (the device-resident synthetic
    # batch the reference's `set_up_synthetic_data` hands out every step) is already in the
    # planned buffers: no copy, no label conversion.  The staged tensors are kept referenced, so
    # their memory cannot be handed to a different tensor in the meantime: same address, same
    # version counter and same layout then means same content.
    def key(t):
      return (t.data_ptr(), t._version, tuple(t.shape), t.dtype, tuple(t.stride()), t.device)
    kx = key(images)
    if images.data_ptr() != self.images.data_ptr():
      if (allow_direct and self.use_graph is False and self._pre_index is not None
          and self._same_device(images) and images.dtype == self.images.dtype
          and images.is_contiguous() and os.environ.get("DTF_DIRECT_INPUT", "1") != "0"):
        # eager launch: the preprocessing launch of this step reads the caller's batch in place
        # (a prefetcher hands over a fresh device tensor every step; copying it into the
        # planned input buffer first costs a 154 MB round trip per ResNet-50 batch)
        self._direct_in = images
      elif self._staged[0] != kx:
        self.images.copy_(images, non_blocking=True)
        self._staged[0], self._staged_keep[0] = kx, images
    if labels is not None:
      ky = key(labels)
      if self._staged[1] != ky:
        self.labels.copy_(labels.reshape(-1), non_blocking=True)
        self._staged[1], self._staged_keep[1] = ky, labels

  def _set_loss_scale(self, s: Optional[float]):
    """None: the device-side scale is maintained by the loss-scale kernel of the optimizer."""
    if s is not None and s != self._loss_scale:
      self.loss_scale_dev.fill_(float(s))
      self._loss_scale = float(s)

  def train_step(self, images: torch.Tensor, labels: torch.Tensor,
                 loss_scale: Optional[float] = 1.0,
                 on_bucket_ready: Optional[Callable[[Bucket], None]] = None,
                 soft_joins: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """`soft_joins`: `on_bucket_ready` orders its work behind the side lane itself
    (`side_stream()`), so the main lane need not wait for the weight gradients at bucket
    boundaries (eager launch only; captured graphs keep the joins)."""
    self._soft_joins = bool(soft_joins and on_bucket_ready is not None)
    with torch.cuda.device(self.device):
      self._ensure_packed()
      self._stage(images, labels, allow_direct=True)
      self._set_loss_scale(loss_scale)
      if self.use_graph is None:
        self._calibrate_step(on_bucket_ready)
      elif self.use_graph:
        self._run_train_graph(on_bucket_ready)
      else:
        self._run_eager(on_bucket_ready)
      out = self.accum.clone()         # [mean loss, correct count]: the plan zeroes accum next step
    return out[0], out[1]

  def _run_eager(self, on_bucket_ready=None):
    s = native.stream_handle(self.device)
    direct = getattr(self, "_direct_in", None)
    if direct is not None:
      i = self._pre_index
      self.fwd_plan.run(0, i, s)
      desc = dict(self._pre, **{"in": direct.data_ptr()})
      if self.dt:
        desc["dt"] = self.dt
      native.run_op("preprocess_pad", desc, s)
      self.fwd_plan.run(i + 1, self.fwd_plan.size(), s)
      self._direct_in = None
    else:
      self.fwd_plan.run(0, self.fwd_plan.size(), s)
    if on_bucket_ready is None:
      self.bwd_plan.run(0, self.bwd_plan.size(), s)
      return
    for begin, end, ready in self._segments:
      self.bwd_plan.run(begin, end, s, self._soft_joins)
      for b in ready:
        on_bucket_ready(b)

  def attach_collective(self, desc: Dict) -> bool:
    """Appends a collective of the native communicator (`allreduce` plan entry over the whole
    gradient buffer) to the end of the backward plan: a model whose gradients are one bucket
    then reduces them from inside the plan - and inside the captured hipGraph - with no host
    call between backward and optimizer.  Only before the first graph capture."""
    if self._train_graph is not None or getattr(self, "collective_in_plan", False):
      return getattr(self, "collective_in_plan", False)
    self.bwd_plan.add("allreduce", desc, "gradients/allreduce", 0)
    begin, _end, ready = self._segments[-1]
    self._segments[-1] = (begin, self.bwd_plan.size(), ready)
    self.collective_in_plan = True
    return True

  def side_stream(self) -> Optional["torch.cuda.Stream"]:
    """The plans' side lane (weight gradients), once it exists."""
    h = native.ext().Plan.side_stream()
    return torch.cuda.ExternalStream(h, device=self.device) if h else None

  def _calibrate_step(self, on_bucket_ready=None):
    """Auto mode: steps 0-3 eager (2-3 timed), step 4 captures, steps 5-7 replay (6-7 timed).

    With a gradient reducer the timed region contains the launch of the bucket collectives;
    every rank runs the same sequence of modes, and ranks that end up with different choices
    still issue identical collectives (the mode only changes how kernels are enqueued)."""
    i = self._calib_step
    self._calib_step += 1
    mode = "eager" if i < 4 else "graph"
    timed = i in (2, 3, 6, 7)
    if timed:
      t0 = torch.cuda.Event(enable_timing=True)
      t1 = torch.cuda.Event(enable_timing=True)
      t0.record()
    if mode == "eager":
      self._run_eager(on_bucket_ready)
    else:
      self._run_train_graph(on_bucket_ready)
    if timed:
      t1.record()
      self._calib[mode].append((t0, t1))
    if i == 7:
      torch.cuda.synchronize(self.device)
      ms = {m: min(a.elapsed_time(b) for a, b in ev) for m, ev in self._calib.items()}
      self.use_graph = ms["graph"] <= ms["eager"]
      logging.info("HipExecutor %s: eager %.3f ms, hipGraph replay %.3f ms per step -> %s",
                   self.spec.name, ms["eager"], ms["graph"],
                   "graph" if self.use_graph else "eager")
      self._calib = {}

  def _run_train_graph(self, on_bucket_ready=None):
    """Forward and backward as hipGraphs (captured at first use on a side stream): one graph for
    the forward plan and one per backward segment, so that a gradient reducer can start a
    bucket's collective between two replays exactly as in the eager path."""
    cur = torch.cuda.current_stream(self.device)
    if self._train_graph is None:
      if self._graph_stream is None:
        self._graph_stream = torch.cuda.Stream(self.device)
      gs = self._graph_stream
      gs.wait_stream(cur)
      with torch.cuda.stream(gs):
        # one eager run first: lazy module loading must not happen under capture
        s = native.stream_handle(self.device)
        self.fwd_plan.run(0, self.fwd_plan.size(), s)
        self.bwd_plan.run(0, self.bwd_plan.size(), s)
        gs.synchronize()
        h1 = self.fwd_plan.capture(0, self.fwd_plan.size(), s)
        hs = [self.bwd_plan.capture(b, e, s) for b, e, _ in self._segments]
        self._train_graph = (h1, hs)
      cur.wait_stream(gs)
      # The eager run above updated the BatchNorm moving averages once and left complete
      # gradients; it is this training step's forward + backward, so only the collectives
      # remain to be issued.  Replay starts with the next call.
      if on_bucket_ready is not None:
        for _b, _e, ready in self._segments:
          for b in ready:
            on_bucket_ready(b)
      return
    s = int(cur.cuda_stream)
    self.fwd_plan.replay(self._train_graph[0], s)
    for h, (_b, _e, ready) in zip(self._train_graph[1], self._segments):
      self.bwd_plan.replay(h, s)
      if on_bucket_ready is not None:
        for b in ready:
          on_bucket_ready(b)

  @torch.no_grad()
  def eval_step(self, images: torch.Tensor, labels: torch.Tensor):
    with torch.cuda.device(self.device):
      self._ensure_packed()
      self._stage(images, labels)
      self.eval_plan.run(0, self.eval_plan.size(), native.stream_handle(self.device))
      out = self.accum.clone()
    return out[0], out[1]

  @torch.no_grad()
  def predict(self, images: torch.Tensor) -> torch.Tensor:
    with torch.cuda.device(self.device):
      self._ensure_packed()
      self._stage(images, None)
      self.eval_plan.run(0, self.eval_plan.size(), native.stream_handle(self.device))
      logits = self.acts[self.spec.logits_name]
      probs = torch.empty((self.N, self.spec.num_classes), dtype=torch.float32,
                          device=self.device)
      native.run_op("softmax_probs", dict(logits=logits.data_ptr(), probs=probs.data_ptr(),
                                          N=self.N, C=self.spec.num_classes,
                                          ld=logits.shape[1], dt=self.dt))
    return probs

  def synchronize(self):
    torch.cuda.synchronize(self.device)
