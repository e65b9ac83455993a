"""Launch-planning policies of the native executor that are plain Python (no GPU needed): the
cache-aware walk direction of consecutive launches, the side lane's weight-gradient tile rule and
the in-step table that overrides both tile tables."""
import types

import pytest

from distributed_tensorflow_amd.engine.hip_executor import HipExecutor
from distributed_tensorflow_amd.ops import autotune

MB = 1 << 20


def _planner():
  ex = types.SimpleNamespace(esize=2, SWEEP_CACHE_BYTES=HipExecutor.SWEEP_CACHE_BYTES,
                             SWEEP_MIN_BYTES=HipExecutor.SWEEP_MIN_BYTES)
  return ex, lambda tensors, mark=True: HipExecutor._sweep_dir(ex, tensors, mark)


def test_consecutive_launches_over_a_large_tensor_alternate_direction(monkeypatch):
  monkeypatch.delenv("DTF_SWEEP_ORDER", raising=False)
  _ex, walk = _planner()
  x, y, r, z, y2 = 1, 2, 3, 4, 5
  assert walk([(x, 100 * MB), (y, 400 * MB)]) == 0           # convolution writes y front to back
  assert walk([(y, 400 * MB), (r, 400 * MB), (z, 400 * MB)]) == 1    # its tail is hot: walk back
  assert walk([(z, 400 * MB), (y2, 100 * MB)]) == 0          # z was finished at its head
  # a side-lane launch votes but leaves no mark: the main-lane launch after it decides the same
  assert walk([(y2, 100 * MB)], mark=False) == 1
  assert walk([(y2, 100 * MB)]) == 1


def test_small_or_stale_tensors_do_not_vote(monkeypatch):
  monkeypatch.delenv("DTF_SWEEP_ORDER", raising=False)
  _ex, walk = _planner()
  assert walk([(1, 8 * MB)]) == 0
  assert walk([(1, 8 * MB)]) == 0                            # fits the caches whole: no preference
  assert walk([(2, 300 * MB)]) == 0
  assert walk([(3, 400 * MB), (4, 400 * MB)]) == 0           # 800 MB of other traffic since
  assert walk([(2, 300 * MB)]) == 0                          # nothing of tensor 2 is cached any more
  monkeypatch.setenv("DTF_SWEEP_ORDER", "0")
  assert walk([(3, 400 * MB)]) == 0                          # switched off: always front to back


def _wgrad_desc(**kw):
  d = dict(N=256, H=14, W=14, Cpix=256, OH=14, OW=14, Cout=1024, TI=1, TJ=1, Ctap=256, cfg=17,
           splits=24)
  d.update(kw)
  return d


def test_side_lane_prefers_the_128x128_two_group_tile_on_large_steps(monkeypatch):
  for k in ("DTF_WGRAD_LANE_TILE", "DTF_WGRAD_LANE_MAP", "DTF_WGRAD_LANE_SPLITS_MUL"):
    monkeypatch.delenv(k, raising=False)
  ex = types.SimpleNamespace(dt=0, _large_step=True)
  rule = lambda d: HipExecutor._lane_tile(ex, d)
  assert rule(_wgrad_desc(cfg=17))["cfg"] == 16 and rule(_wgrad_desc(cfg=17))["splits"] == 24
  assert rule(_wgrad_desc(cfg=18 + 64))["cfg"] == 16         # 64x128, three-stage ring
  assert rule(_wgrad_desc(cfg=32))["cfg"] == 16              # 128x128 with four wave groups
  assert rule(_wgrad_desc(cfg=24))["cfg"] == 24              # 256x128 stays
  assert rule(_wgrad_desc(cfg=17, Cout=64))["cfg"] == 17     # narrow layers keep their tile
  assert rule(_wgrad_desc(cfg=17, Ctap=64))["cfg"] == 17     # short reductions too
  ex_small = types.SimpleNamespace(dt=0, _large_step=False)
  assert HipExecutor._lane_tile(ex_small, _wgrad_desc(cfg=17))["cfg"] == 17
  ex_fp32 = types.SimpleNamespace(dt=2, _large_step=True)
  assert HipExecutor._lane_tile(ex_fp32, _wgrad_desc(cfg=17))["cfg"] == 17
  monkeypatch.setenv("DTF_WGRAD_LANE_TILE", "0")
  assert rule(_wgrad_desc(cfg=17))["cfg"] == 17


def test_in_step_table_overrides_the_tables(monkeypatch):
  for k in ("DTF_WGRAD_LANE_TILE", "DTF_WGRAD_LANE_MAP", "DTF_WGRAD_LANE_SPLITS_MUL"):
    monkeypatch.delenv(k, raising=False)
  ex = types.SimpleNamespace(dt=0, _large_step=True, eval_plan=object())
  ex._lane_tile = lambda d: HipExecutor._lane_tile(ex, d)
  d = _wgrad_desc(cfg=17)
  plan = object()
  assert HipExecutor._in_step_choice(ex, plan, "wgrad", d, 1)["cfg"] == 16        # the rule
  assert HipExecutor._in_step_choice(ex, plan, "wgrad", d, 0)["cfg"] == 17        # main lane: table
  sig = autotune.step_signature("wgrad", d, 1)
  assert sig.endswith("|lane=1") and sig != autotune.step_signature("wgrad", d, 0)
  monkeypatch.setitem(autotune.STEP_OVERRIDE, sig, dict(cfg=24, splits=12))
  out = HipExecutor._in_step_choice(ex, plan, "wgrad", d, 1)
  assert (out["cfg"], out["splits"]) == (24, 12)
  assert HipExecutor._in_step_choice(ex, ex.eval_plan, "wgrad", d, 1)["cfg"] == 17  # never in evaluation
  assert [k for k, _lane, _d in ex.step_launches] == ["wgrad"] * 3
