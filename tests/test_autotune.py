"""Tile-selection cache: signatures, look-ups and candidate lists (CPU only)."""
import json

from distributed_tensorflow_amd.ops import autotune, conv_plan
from distributed_tensorflow_amd.ops.conv_plan import ConvGeom


def _geom():
  return ConvGeom(256, 14, 14, 256, 3, 3, 256, 1, (1, 1, 1, 1))


def test_signature_ignores_addresses_and_choices():
  g = _geom()
  a = g.fwd(0x1000, 0x2000, 0x3000)
  b = g.fwd(0x9000, 0x8000, 0x7000, cfg=4)
  assert autotune.signature("conv", a) == autotune.signature("conv", b)
  c = g.fwd(0x1000, 0x2000, 0x3000, stats=0x4000, stats_replicas=32)
  assert autotune.signature("conv", a) != autotune.signature("conv", c)      # +stats epilogue
  w1 = g.wgrad(1, 2, 3)
  w2 = g.wgrad(4, 5, 6, cfg=16, splits=3)
  assert autotune.signature("wgrad", w1) == autotune.signature("wgrad", w2)
  d1 = g.dgrad(1, 2, 3)[0]
  d2 = g.dgrad(1, 2, 3, addend=7, addend_mask=8)[0]
  assert autotune.signature("conv", d1) != autotune.signature("conv", d2)


def test_placeholder_descriptors_are_never_measured(monkeypatch):
  """The planner asks which tile the table holds for a layer with descriptors whose pointers are
  placeholders (`measure=False`): even with tuning on and the shape missing or marked for another
  measurement nothing may be launched; the pointers of an input normalised inside the tile are not
  part of the signature."""
  g = ConvGeom(7, 56, 56, 64, 3, 3, 64, 1, (1, 1, 1, 1))
  monkeypatch.setenv("DTF_AUTOTUNE", "1")
  monkeypatch.setattr(autotune, "RETUNE", lambda kind, desc: True)

  def boom(kind, desc):
    raise AssertionError("a placeholder descriptor was launched")
  monkeypatch.setattr(autotune, "tune", boom)
  d = g.wgrad(16, 16, 16)
  assert autotune.apply("wgrad", d, measure=False) == d           # batch 7 is in no table
  f = g.fwd(16, 16, 16)
  assert autotune.apply("conv", f, measure=False) == f
  assert autotune.signature("conv", f) == autotune.signature("conv", dict(f, in_scale=123, in_shift=456))
  assert autotune.signature("wgrad", d) == autotune.signature("wgrad", dict(d, x_scale=123, x_shift=456, ws=789))


def test_apply_uses_cache_and_falls_back(tmp_path, monkeypatch):
  g = _geom()
  d = g.wgrad(1, 2, 3)
  cache = {autotune.signature("wgrad", d): {"cfg": 34, "splits": 5, "us": 1.0}}
  p = tmp_path / "tuned.json"
  p.write_text(json.dumps(cache))
  monkeypatch.setenv("DTF_AUTOTUNE_CACHE", str(p))
  monkeypatch.setattr(autotune, "_cache", None)
  out = autotune.apply("wgrad", d)
  assert out["cfg"] == 34 and out["splits"] == 5 and d["cfg"] != 34      # input left untouched
  other = ConvGeom(8, 7, 7, 64, 1, 1, 64).wgrad(1, 2, 3)
  assert autotune.apply("wgrad", other) == other                          # heuristic choice kept
  monkeypatch.setattr(autotune, "_cache", None)


def test_candidates_are_valid_configurations():
  for cout in (10, 16, 32, 64, 128, 1001):
    g = ConvGeom(4, 8, 8, 64, 3, 3, cout, 1, (1, 1, 1, 1))
    for c in autotune._fwd_candidates(g.fwd(1, 2, 3)):
      assert c in conv_plan.FWD_CFGS
    for c, s in autotune._wgrad_candidates(g.wgrad(1, 2, 3)):
      assert c % 16 in conv_plan.WGRAD_CFGS and (c % 64) // 16 in (0, 1, 2) and c // 64 in (0, 1, 2) and s >= 1
      assert not (c % 16 == 8 and (c % 64) // 16 == 2)        # the 8-wave tile has at most 2 groups


def test_bundled_cache_is_well_formed():
  with open(autotune.CACHE_PATH) as f:
    cache = json.load(f)
  assert len(cache) > 100
  for k, v in cache.items():
    kind = k.split("|", 1)[0]
    assert kind in ("conv", "wgrad")
    if kind == "conv":
      assert v["cfg"] in conv_plan.FWD_CFGS
    else:
      assert v["cfg"] % 16 in conv_plan.WGRAD_CFGS and v["splits"] >= 1


def test_whole_rows_kernels_are_candidates_of_their_shapes_only():
  """The whole-rows kernels (conv_rows.hip 48 - 54, conv_wgrad_rows.hip 192) join the candidate lists
  of exactly the layers they serve; the bundled table uses them for ResNet-50 at batch 256."""
  r56 = ConvGeom(256, 56, 56, 64, 3, 3, 64, 1, (1, 1, 1, 1))
  r28 = ConvGeom(256, 28, 28, 128, 3, 3, 128, 1, (1, 1, 1, 1))
  stem = ConvGeom(256, 224, 224, 3, 7, 7, 64, 2, (3, 3, 3, 3), mode="merged")
  other = ConvGeom(256, 14, 14, 256, 3, 3, 256, 1, (1, 1, 1, 1))
  strided = ConvGeom(256, 56, 56, 128, 3, 3, 128, 2, (0, 1, 0, 1))
  assert {48, 50, 52, 53} <= set(autotune._fwd_candidates(r56.fwd(1, 2, 3)))
  assert {49, 51} <= set(autotune._fwd_candidates(r28.fwd(1, 2, 3)))
  assert 54 in autotune._fwd_candidates(stem.fwd(1, 2, 3))
  for g in (other, strided):
    assert not [c for c in autotune._fwd_candidates(g.fwd(1, 2, 3)) if 48 <= c < 56]
  assert {48, 50, 52, 53} <= set(autotune._fwd_candidates(r56.dgrad(1, 2, 3)[0]))     # unit-stride data gradient
  for g in (r56, r28):
    assert (192, autotune._wgrad_candidates(g.wgrad(1, 2, 3))[0][1]) in autotune._wgrad_candidates(g.wgrad(1, 2, 3))
  for g in (other, strided):
    assert not [c for c, _s in autotune._wgrad_candidates(g.wgrad(1, 2, 3)) if c == 192]
  assert [c for c, _s in autotune._wgrad_candidates(stem.wgrad(1, 2, 3)) if c == 192]
  with open(autotune.CACHE_PATH) as f:
    cache = json.load(f)
  assert cache[autotune.signature("conv", r56.fwd(1, 2, 3, stats=4, stats_replicas=32))]["cfg"] in (48, 50, 52, 53)
  assert cache[autotune.signature("conv", stem.fwd(1, 2, 3, stats=4, stats_replicas=32))]["cfg"] == 54
  assert cache[autotune.signature("wgrad", r56.wgrad(1, 2, 3))]["cfg"] == 192
