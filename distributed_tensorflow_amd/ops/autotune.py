"""Per-shape tile selection for the convolution kernels, measured on the device.

The reference leaves algorithm choice to cuDNN's autotuner (TensorFlow runs every convolution
algorithm once per new shape and keeps the fastest).  The counterpart here: for every distinct
convolution descriptor of a planned model the candidate tile configurations (and, for the
weight gradient, split-K factors) are timed with HIP events on the real buffers, and the
winner is recorded under a shape signature.  Results are cached in `tuned_gfx950.json` next to
this file - the file in the tree holds the shapes of the bundled models measured on an MI355X,
so a normal run only does dictionary look-ups; shapes that are not in the cache fall back to the
heuristics of `conv_plan.pick_fwd_cfg` / `pick_wgrad_cfg` unless tuning is switched on with
`DTF_AUTOTUNE=1` (new entries are then written to `DTF_AUTOTUNE_OUT` or the cache file;
`DTF_AUTOTUNE_CACHE` reads another cache file, `DTF_AUTOTUNE_IGNORE_CACHE=1` starts empty,
`DTF_AUTOTUNE_RETUNE=conv` / `wgrad` / `conv,wgrad` measures the entries of that kind again and
keeps the rest - the thing to do after changing one of the two kernels).
"""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Tuple

from . import conv_plan

__all__ = ["signature", "lookup", "apply", "tune", "save", "enabled", "CACHE_PATH"]

CACHE_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gfx950.json")
_POINTERS = ("in", "wgt", "out", "addend", "addend_mask", "bias", "stats", "stats_y",
             "stats_mask", "x", "dy", "dw", "scale", "batch", "in_scale", "in_shift", "x_scale",
             "x_shift", "ws")
_IGNORED = _POINTERS + ("cfg", "splits", "stats_replicas", "rev")
_cache: Optional[Dict[str, Dict[str, int]]] = None
_dirty = False


def enabled() -> bool:
  return os.environ.get("DTF_AUTOTUNE", "0") == "1"


def _load() -> Dict[str, Dict[str, int]]:
  global _cache
  if _cache is None:
    _cache = {}
    path = os.environ.get("DTF_AUTOTUNE_CACHE") or CACHE_PATH
    if os.environ.get("DTF_AUTOTUNE_IGNORE_CACHE", "0") != "1" and os.path.exists(path):
      with open(path) as f:
        _cache = json.load(f)
  return _cache


def signature(kind: str, desc: Dict) -> str:
  """Shape key of a descriptor: every integer field, plus which optional tensors are present."""
  parts = [kind]
  for k in sorted(desc):
    if k in _IGNORED or (k.startswith("pro_") and k != "pro_relu") or k.startswith("bpro_"):
      continue
    parts.append("%s=%s" % (k, desc[k]))
  for k in ("addend", "addend_mask", "bias", "stats", "scale"):
    if desc.get(k) is not None:
      if k == "stats" and desc.get("stats_y") is not None:
        # backward statistics in a data-gradient epilogue (re-reads the BatchNorm input):
        # a different kernel from the plain launch, and measured as such
        parts.append("+bstats")
        continue
      parts.append("+" + k)
  if desc.get("pro_sums") is not None:
    parts.append("+pro")           # BatchNorm of the input folded into the launch
  if desc.get("bpro_sums") is not None:
    parts.append("+bpro")          # BatchNorm backward of the output's layer folded into a data gradient
  return "|".join(parts)


_retuned = set()


def lookup(kind: str, desc: Dict) -> Optional[Dict[str, int]]:
  sig = signature(kind, desc)
  if enabled() and kind in os.environ.get("DTF_AUTOTUNE_RETUNE", "").split(",") \
      and sig not in _retuned:
    return None
  return _load().get(sig)


def _rows_candidates(desc: Dict) -> List[int]:
  """48..54: the whole-rows kernels of conv_rows.hip for the large 3x3 unit-stride layers (every
  input pixel fetched once instead of once per tap): 56x56x64 and 28x28x128."""
  if (desc.get("TI"), desc.get("TJ"), desc.get("Ctap"), desc.get("Cpix"), desc.get("Cout"),
      desc.get("OW")) == (7, 1, 32, 4, 64, 112):
    return [54]                  # the merged-tap ImageNet stem over pairs of output rows
  if desc.get("TI") != 3 or desc.get("TJ") != 3 or desc.get("ISH", 1) != 1 or desc.get("OSH", 1) != 1:
    return []
  key = (desc.get("Cpix"), desc.get("W"), desc.get("Cout"))
  return {(64, 56, 64): [48, 50, 52, 53], (128, 28, 128): [49, 51]}.get(key, [])


def _fwd_candidates(desc: Dict) -> List[int]:
  cout = desc["Cout"]
  # 40..47: the one-round-trip kernel for small layers (conv_halo.hip); a descriptor it cannot
  # serve falls back to the two-buffer tile of the same footprint, or is refused (ValueError,
  # skipped by the tuner) when it carries a BatchNorm prologue
  halo = [40, 41, 42] if cout <= 16 else [43, 44, 45, 46] if cout <= 32 else [45, 46, 47]
  if desc.get("pro_sums") is not None or desc.get("bpro_sums") is not None:
    return halo
  small = desc.get("N", 0) * desc.get("OH", 0) * desc.get("OW", 0) * desc.get("Cpix", 0) <= (8 << 20)
  if not small:
    halo = []
  if cout <= 16:
    return [3, 2, 8, 12, 7, 33] + halo
  if cout <= 32:
    return [2, 1, 9, 13, 15, 7, 32, 34] + halo
  # 16..: the multi-stage pipeline of conv_pipe.hip (descriptors it cannot serve fall back to
  # the two-stage tile of the same footprint, so listing them is always safe)
  if cout <= 64:
    return [1, 5, 0, 21, 22, 23, 20, 10, 11, 14, 15, 32, 34] + halo + _rows_candidates(desc)
  wide = [29, 30, 31] if cout >= 256 else []       # 256 output channels per block
  return [0, 6, 4, 5, 1, 16, 17, 18, 19, 20, 21, 24, 25, 26, 27, 28] + wide + _rows_candidates(desc)


def _wgrad_candidates(desc: Dict) -> List[Tuple[int, int]]:
  cout = desc["Cout"]
  ktotal = desc["TI"] * desc["TJ"] * desc["Ctap"]
  P = desc["N"] * desc["OH"] * desc["OW"]
  if cout <= 16:
    cfgs = [5, 7]
  elif cout <= 32:
    cfgs = [4, 6]
  elif cout <= 64:
    cfgs = [1, 3]
  else:
    cfgs = [0, 2, 1] + ([8] if ktotal >= 256 else [])
  if ktotal <= 64:
    cfgs = [c for c in cfgs if conv_plan.WGRAD_CFGS[c][0] <= 64] or cfgs
  out = []
  max_splits = max(P // 256, 1)

  def with_depths(c, s):
    """(c, s) plus its deep-pipeline variants (cfg + 64 * (stages - 2)): tiles whose waves all
    issue the same loads per step (both sides at least 64 wide), while the ring fits the LDS."""
    res = [(c, s)]
    bkk, bco = conv_plan.WGRAD_CFGS[c % 16]
    g = 1 << (c // 16)
    if bkk >= 64 and bco >= 64:
      for stages in (3, 4):
        if g * stages * 32 * (bkk + bco) * 2 <= 160 * 1024:
          res.append((c + 64 * (stages - 2), s))
    return res

  for c in cfgs:
    bkk, bco = conv_plan.WGRAD_CFGS[c]
    tiles = -(-ktotal // bkk) * -(-cout // bco)
    seen = set()
    for target in (192, 256, 384, 512, 768, 1024, 1536):
      s = max(1, min(-(-target // tiles), max_splits))
      if s in seen:
        continue
      seen.add(s)
      out.extend(with_depths(c, s))
      # 2 / 4 wave groups per block share one atomic epilogue: same resident waves with
      # half / a quarter of the blocks (cfg + 16 * log2(groups)); the 8-wave tile allows 2
      for lg, g in ((1, 2), (2, 4)):
        if s >= g and not (c == 8 and g == 4) and (c + 16 * lg, -(-s // g)) not in out:
          out.extend(with_depths(c + 16 * lg, -(-s // g)))
  # 192: the whole-rows kernels of conv_wgrad_rows.hip (56x56x64 and 28x28x128 3x3, unit stride; the
  # merged-tap ImageNet stem); the split count is kept for the fallback tile
  if (desc.get("TI"), desc.get("TJ"), desc.get("Ctap"), desc.get("Cpix"), cout, desc.get("OW"),
      desc.get("TJreal")) == (7, 1, 32, 4, 64, 112, 7) and desc.get("batch") is None and out:
    out.append((192, out[0][1]))
  if (desc.get("TI"), desc.get("TJ"), desc.get("Cpix"), desc.get("W"), cout) in ((3, 3, 64, 56, 64),
                                                                                 (3, 3, 128, 28, 128)) \
      and desc.get("ISH", 1) == 1 and desc.get("batch") is None and out:
    out.append((192, out[0][1]))
  return out


def _time_op(kind: str, desc: Dict, iters: int = 10, repeats: int = 3) -> float:
  """Best of `repeats` timings of `iters` back-to-back launches (a single timing is easily off
  by 10 %: clock ramps, a neighbouring process on the GPU).  The launches are recorded in a
  native plan and enqueued from C++: through the Python entry point a launch costs ~8 us of
  host time, which hid every difference between the tile configurations of the small CIFAR
  layers (all of them measured 8 - 9 us)."""
  import torch  # pylint: disable=import-outside-toplevel
  from . import native  # pylint: disable=import-outside-toplevel
  if desc.get("pro_mm") is not None:
    # a prologue launch updates the moving averages of its BatchNorm: not while timing
    desc = dict(desc, pro_mm=None, pro_mv=None)
  plan = native.Plan()
  for _ in range(iters):
    plan.add(kind, desc, "", 0)
  stream = native.stream_handle()
  plan.run(0, 2, stream)
  best = float("inf")
  for _ in range(repeats):
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    start.record()
    plan.run(0, iters, stream)
    stop.record()
    stop.synchronize()
    best = min(best, start.elapsed_time(stop) / iters)
  return best


def tune(kind: str, desc: Dict) -> Dict[str, int]:
  """Times the candidates of one descriptor on the current device; records and returns the best."""
  global _dirty
  best, best_t = None, float("inf")
  trial = dict(desc)
  if kind == "conv":
    for c in _fwd_candidates(desc):
      trial["cfg"] = c
      try:
        t = _time_op(kind, trial)
      except ValueError:          # a configuration this descriptor (element type) cannot use
        continue
      if t < best_t:
        best, best_t = {"cfg": c}, t
  elif kind == "wgrad":
    for c, s in _wgrad_candidates(desc):
      trial["cfg"], trial["splits"] = c, s
      try:
        t = _time_op(kind, trial)
      except ValueError:
        continue
      if t < best_t:
        best, best_t = {"cfg": c, "splits": s}, t
  else:
    raise ValueError("nothing to tune for op kind %r" % kind)
  if best is None:
    raise ValueError("no tile configuration accepts %s" % signature(kind, desc))
  best["us"] = round(best_t * 1000.0, 1)
  _retuned.add(signature(kind, desc))
  _load()[signature(kind, desc)] = best
  _dirty = True
  return best


# predicate (kind, desc) -> bool: entries to measure again although the table has them (a new
# kernel family joined the candidates of some shapes); set by scripts/autotune_models.py
RETUNE = None


def apply(kind: str, desc: Dict, measure: bool = True) -> Dict:
  """Returns `desc` with the tuned configuration filled in (tuning it first when enabled;
  `measure=False`: from the table only - for a descriptor whose pointers are placeholders)."""
  hit = lookup(kind, desc)
  if measure and enabled() and (hit is None or (RETUNE is not None and RETUNE(kind, desc)
                                                and signature(kind, desc) not in _retuned)):
    hit = tune(kind, desc)
  if hit is not None:
    desc = dict(desc)
    desc["cfg"] = hit["cfg"]
    if "splits" in hit:
      desc["splits"] = hit["splits"]
  return desc


def save(path: Optional[str] = None):
  """Writes the cache if tuning added entries."""
  global _dirty
  if not _dirty:
    return
  path = path or os.environ.get("DTF_AUTOTUNE_OUT") or CACHE_PATH
  os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
  with open(path, "w") as f:
    json.dump(_load(), f, indent=0, sort_keys=True)
  _dirty = False


# ---- choices measured in whole training steps -------------------------------------------------
# The table above is measured on kernels running alone, back to back on hot caches.  Inside a step
# a launch meets cold caches and, in the backward pass, the other lane's kernels: a weight
# gradient on the side lane competes with the main lane for L2 and memory bandwidth, and the
# tile that is fastest alone is not always the one that gives the shortest step.
# `tuned_gfx950_step.json` holds, per launch signature, the choice that minimised the *step* time
# (`scripts/tune_in_step.py`: coordinate search over shape classes, whole training steps timed);
# `STEP_OVERRIDE` lets that search try candidates without touching the file.
STEP_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned_gfx950_step.json")
STEP_OVERRIDE: Dict[str, Dict[str, int]] = {}
_step_cache: Optional[Dict[str, Dict[str, int]]] = None


def step_signature(kind: str, desc: Dict, lane: int = 0) -> str:
  """Signature of a launch in the in-step table: the shape key plus the lane it runs on."""
  return signature(kind, desc) + ("|lane=1" if lane else "")


def step_lookup(kind: str, desc: Dict, lane: int = 0) -> Optional[Dict[str, int]]:
  """In-step choice {cfg[, splits]} for a launch, or None."""
  global _step_cache
  sig = step_signature(kind, desc, lane)
  if sig in STEP_OVERRIDE:
    return STEP_OVERRIDE[sig]
  if _step_cache is None:
    _step_cache = {}
    path = os.environ.get("DTF_AUTOTUNE_STEP_CACHE") or STEP_PATH
    if (os.environ.get("DTF_AUTOTUNE_IGNORE_CACHE", "0") != "1"
        and os.environ.get("DTF_AUTOTUNE_IGNORE_STEP", "0") != "1" and os.path.exists(path)):
      with open(path) as f:
        _step_cache = json.load(f)
  return _step_cache.get(sig)
