"""Descriptor builders for the implicit-GEMM convolution kernels.

A `ConvGeom` captures one convolution (or dense layer, as a 1x1 convolution over a 1x1
image) and produces the argument dictionaries of the native ops:

* `fwd()`      forward gather-GEMM,
* `dgrad()`    data gradient: one launch for stride 1, stride*stride output-parity launches
               for strided convolutions (each is a stride-1 gather over dY with the taps of
               matching parity; nothing is multiplied by inserted zeros),
* `wgrad()`    weight gradient with split-K over pixels,
* `pack_fwd()` / `pack_dgrad()`  how the fp32 HWIO master weight is repacked to bf16.

Two input storage modes exist.  `normal`: the input is NHWC with the channel count rounded up
to a multiple of 8 and padding handled by bounds checks.  `merged` (the ImageNet stem,
7x7/2 on 3 channels): the image is stored with its explicit zero border and 4 channels, and
the 7 horizontal taps x 4 channels (padded to 8 taps) form one 32-element "tap" that is a
single contiguous 64-byte read - K becomes 7*32 = 224 instead of 147 padded per tap to 8
channels (392), and no bounds checks are needed.
"""
from __future__ import annotations

import dataclasses
from typing import Dict, List, Optional, Tuple

__all__ = ["ConvGeom", "round_up", "pick_fwd_cfg", "pick_wgrad_cfg"]

# tile configurations compiled into the extension: cfg id -> (BM pixels, BN channels)
FWD_CFGS = {0: (128, 128), 1: (256, 64), 2: (256, 32), 3: (256, 16), 4: (64, 128), 5: (128, 64),
            6: (256, 128), 7: (128, 16), 14: (64, 64), 15: (128, 32),
            # ring of 4 / 6 LDS stages on any gather (short, tap-straddling reductions)
            8: (256, 16), 9: (256, 32), 10: (128, 64), 11: (256, 64), 12: (128, 16), 13: (128, 32),
            # multi-stage pipeline (conv_pipe.hip): (BM, BN); K-step and ring depth vary
            16: (256, 128), 17: (256, 128), 18: (128, 128), 19: (128, 128), 20: (128, 128),
            21: (128, 64), 22: (256, 64), 23: (256, 64), 24: (64, 128), 25: (128, 256),
            26: (256, 128), 27: (128, 128), 28: (256, 128), 29: (256, 256), 30: (256, 256),
            31: (128, 256),
            32: (64, 32), 33: (64, 16), 34: (64, 32),
            # one-round-trip kernel for small layers (conv_halo.hip): halo tile + resident weights
            40: (256, 16), 41: (128, 16), 42: (64, 16), 43: (256, 32), 44: (128, 32), 45: (64, 32),
            46: (64, 32), 47: (64, 64),
            # whole-rows kernel for large 3x3 unit-stride layers (conv_rows.hip): four output rows
            # (224 / 112 pixels) x all output channels; ring depth, K-steps per stage and
            # occupancy vary
            48: (224, 64), 49: (112, 128), 50: (224, 64), 51: (112, 128), 52: (224, 64),
            53: (224, 64),
            # the 7x7 stride-2 stem over pairs of output rows, filter bank resident in LDS
            54: (224, 64)}
# cfg id % 16 -> (BKK rows of dW, BCO output channels); (cfg id % 64) // 16 = log2(wave groups per
# block); cfg id // 64 = LDS ring depth - 2 (conv_wgrad.hip)
WGRAD_CFGS = {0: (128, 128), 1: (128, 64), 2: (64, 128), 3: (64, 64), 4: (128, 32),
              5: (128, 16), 6: (32, 32), 7: (64, 16), 8: (256, 128)}


def round_up(x: int, m: int) -> int:
  return -(-x // m) * m


def pick_fwd_cfg(M: int, cout: int) -> int:
  """Tile choice for the forward / data-gradient GEMM (M pixels x cout channels) when the
  shape has no measured entry; the rules summarise what the measurements of the ResNet
  shapes prefer."""
  if cout <= 16:
    return 3
  if cout <= 32:
    return 2
  if cout <= 64:
    return 5          # 128x64 (beat 256x64 on every measured shape)
  nt = -(-cout // 128)
  if -(-M // 256) * nt >= 256:
    return 6          # 8 waves, 256x128: most FLOP per L2 byte, if it still fills the chip
  if -(-M // 128) * nt >= 256:
    return 0
  return 4            # 64x128: twice the blocks for the small late-stage layers


def pick_wgrad_cfg(ktotal: int, cout: int, P: int, target_blocks: int = 0) -> Tuple[int, int]:
  """(cfg, splits) for the weight gradient."""
  import os  # pylint: disable=import-outside-toplevel
  target_blocks = target_blocks or int(os.environ.get("DTF_WGRAD_TARGET_BLOCKS", "768"))
  if cout <= 16:
    bco = 16
  elif cout <= 32:
    bco = 32
  elif cout <= 64:
    bco = 64
  else:
    bco = 128
  if bco == 16:
    cfg = 5 if ktotal > 64 else 7
  elif bco == 32:
    cfg = 4 if ktotal > 32 else 6
  elif bco == 64:
    cfg = 1 if ktotal > 64 else 3
  else:
    cfg = 0 if ktotal > 64 else 2
  bkk, bco = WGRAD_CFGS[cfg]
  tiles = -(-ktotal // bkk) * -(-cout // bco)
  splits = max(1, min(-(-target_blocks // tiles), max(P // 256, 1)))
  if splits >= 4:
    # two wave groups per block share one atomic epilogue (half the atomic traffic at the
    # same number of resident waves): preferred by most measured shapes
    return cfg + 16, -(-splits // 2)
  return cfg, splits


@dataclasses.dataclass
class ConvGeom:
  N: int
  H: int
  W: int
  Cin: int
  KH: int
  KW: int
  Cout: int
  stride: int = 1
  pad: Tuple[int, int, int, int] = (0, 0, 0, 0)     # top, bottom, left, right
  mode: str = "normal"                              # normal | merged
  out_pitch: Optional[int] = None                   # channel pitch of the output tensor
  dt: int = 0                                       # element type: 0 bf16, 1 fp16, 2 fp32

  def __post_init__(self):
    pt, pb, pl, pr = self.pad
    self.OH = (self.H + pt + pb - self.KH) // self.stride + 1
    self.OW = (self.W + pl + pr - self.KW) // self.stride + 1
    if self.mode == "merged":
      if self.KW > 8 or self.Cin > 4 or (self.stride * 4) % 8:
        raise ValueError("merged-tap mode needs KW <= 8, Cin <= 4 and an even stride")
      self.Cs = 4
      self.Hs = self.H + pt + pb
      # every 8-pixel read must stay inside the stored row
      self.Ws = max(self.W + pl + pr, (self.OW - 1) * self.stride + 8)
      self.Ws = round_up(self.Ws, 2)
    else:
      self.Cs = round_up(self.Cin, 8)
      self.Hs, self.Ws = self.H, self.W
    self.ldc = self.out_pitch or round_up(self.Cout, 8)
    self.M = self.N * self.OH * self.OW

  # -- storage ---------------------------------------------------------------------
  @property
  def input_shape(self) -> Tuple[int, int, int, int]:
    return (self.N, self.Hs, self.Ws, self.Cs)

  @property
  def output_shape(self) -> Tuple[int, int, int, int]:
    return (self.N, self.OH, self.OW, self.ldc)

  # -- packed weights --------------------------------------------------------------------
  @property
  def k_fwd(self) -> int:
    return self.KH * 32 if self.mode == "merged" else self.KH * self.KW * self.Cs

  def pack_fwd(self) -> Dict:
    if self.mode == "merged":
      tjpad, cpad = 8, 4
    else:
      tjpad, cpad = self.KW, self.Cs
    return dict(TI=self.KH, TJ=self.KW, Cin=self.Cin, Cout=self.Cout, TJpad=tjpad, Cpad=cpad,
                CoutPad=self.Cout, mode=0, count=self.Cout * self.KH * tjpad * cpad)

  @property
  def cout_pad(self) -> int:
    return round_up(self.Cout, 8)

  def pack_dgrad(self) -> Dict:
    return dict(TI=self.KH, TJ=self.KW, Cin=self.Cin, Cout=self.Cout, TJpad=self.KW,
                Cpad=self.Cin, CoutPad=self.cout_pad, mode=1,
                count=self.Cin * self.KH * self.KW * self.cout_pad)

  # -- forward ----------------------------------------------------------------------------
  def fwd(self, x: int, w: int, out: int, bias: Optional[int] = None,
          cfg: Optional[int] = None, stats: Optional[int] = None,
          stats_replicas: int = 0) -> Dict:
    """`stats`: fp32 [stats_replicas][2][Cout] buffer (zeroed by the caller) that receives the
    per-channel sum / sum of squares of the output for a following BatchNorm."""
    pt, _, pl, _ = self.pad
    d = dict(N=self.N, H=self.Hs, W=self.Ws, Cpix=self.Cs, OH=self.OH, OW=self.OW,
             Cout=self.Cout, ldc=self.ldc, ISH=self.stride, ISW=self.stride, dhs=1, dws=1,
             ldw=self.k_fwd, cfg=pick_fwd_cfg(self.M, self.Cout) if cfg is None else cfg)
    d["in"] = x
    d["wgt"] = w
    d["out"] = out
    d["bias"] = bias
    if stats is not None:
      d["stats"] = stats
      d["stats_replicas"] = stats_replicas
    if self.dt:
      d["dt"] = self.dt
    if self.mode == "merged":
      d.update(TI=self.KH, TJ=1, Ctap=32, dh0=0, dw0=0, wsi=32, wsj=32)
    else:
      d.update(TI=self.KH, TJ=self.KW, Ctap=self.Cs, dh0=-pt, dw0=-pl)
    return d

  # -- data gradient ------------------------------------------------------------------------
  def dgrad(self, dy: int, wd: int, dx: int, accumulate: bool = False,
            cfg: Optional[int] = None, addend: Optional[int] = None,
            addend_mask: Optional[int] = None, skip_empty: bool = False) -> List[Dict]:
    """dX[N,H,W,Cin] from dY[N,OH,OW,ldc]; `accumulate` adds to the existing dX, `addend`
    (a tensor shaped like dX, optionally gated by the 1-bit `addend_mask`) is added instead.

    A strided convolution is one launch per output parity; parities no tap reaches (three of
    the four of a 1x1 / stride 2 projection) only write zeros or the addend.  With
    `accumulate` they would rewrite dX unchanged: `skip_empty` leaves them out."""
    if accumulate and addend is not None:
      raise ValueError("either accumulate into dX or add another tensor")
    if self.mode != "normal" or self.Cin % 8:
      raise ValueError("data gradient needs a normal-mode convolution with Cin % 8 == 0")
    if self.ldc != self.cout_pad:
      raise ValueError("data gradient reads dY with pitch round_up(Cout, 8)")
    pt, _, pl, _ = self.pad
    s = self.stride
    ct = self.cout_pad                       # channels per tap of the gather over dY
    ldw = self.KH * self.KW * ct
    out: List[Dict] = []
    for ph in range(s):
      kh0 = (ph + pt) % s
      ti = len(range(kh0, self.KH, s))
      ohg = -(-(self.H - ph) // s)
      if ohg <= 0:
        continue
      for pw in range(s):
        kw0 = (pw + pl) % s
        tj = len(range(kw0, self.KW, s))
        owg = -(-(self.W - pw) // s)
        if owg <= 0:
          continue
        if skip_empty and accumulate and ti * tj == 0:
          continue
        M = self.N * ohg * owg
        d = dict(N=self.N, H=self.OH, W=self.OW, Cpix=ct, OH=ohg, OW=owg,
                 OHfull=self.H, OWfull=self.W, Cout=self.Cin, ldc=self.Cin,
                 OSH=s, OSW=s, OOH=ph, OOW=pw, TI=ti, TJ=tj, Ctap=ct, ISH=1, ISW=1,
                 dh0=(ph + pt - kh0) // s, dhs=-1, dw0=(pw + pl - kw0) // s, dws=-1,
                 ldw=ldw, wk0=(kh0 * self.KW + kw0) * ct, wsi=s * self.KW * ct, wsj=s * ct,
                 cfg=pick_fwd_cfg(M, self.Cin) if cfg is None else cfg)
        d["in"] = dy
        d["wgt"] = wd
        d["out"] = dx
        if self.dt:
          d["dt"] = self.dt
        d["addend"] = dx if accumulate else addend
        if addend is not None and addend_mask is not None:
          d["addend_mask"] = addend_mask
        out.append(d)
    return out

  def has_empty_parities(self) -> bool:
    """True when some output parity of the data gradient is reached by no tap."""
    return self.stride > self.KH or self.stride > self.KW

  # -- weight gradient ------------------------------------------------------------------------
  def wgrad(self, x: int, dy: int, dw: int, cfg: Optional[int] = None,
            splits: Optional[int] = None) -> Dict:
    pt, _, pl, _ = self.pad
    d = dict(N=self.N, H=self.Hs, W=self.Ws, Cpix=self.Cs, OH=self.OH, OW=self.OW,
             Cout=self.Cout, ldy=self.ldc, ISH=self.stride, ISW=self.stride, dhs=1, dws=1)
    d["x"] = x
    d["dy"] = dy
    d["dw"] = dw
    if self.dt:
      d["dt"] = self.dt
    if self.mode == "merged":
      d.update(TI=self.KH, TJ=1, Ctap=32, dh0=0, dw0=0, Cpad=4, Creal=self.Cin, TJpad=8,
               TJreal=self.KW)
      ktotal = self.KH * 32
    else:
      d.update(TI=self.KH, TJ=self.KW, Ctap=self.Cs, dh0=-pt, dw0=-pl, Cpad=self.Cs,
               Creal=self.Cin, TJpad=self.KW, TJreal=self.KW)
      ktotal = self.KH * self.KW * self.Cs
    c, sp = pick_wgrad_cfg(ktotal, self.Cout, self.M)
    if self.dt == 2 and c >= 32:
      c, sp = c - 16, sp * 2        # fp32 tiles are twice the bytes: at most two wave groups
    d["cfg"] = c if cfg is None else cfg
    d["splits"] = sp if splits is None else splits
    return d
