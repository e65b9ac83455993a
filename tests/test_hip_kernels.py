"""Numerics of every hand-written HIP kernel against plain PyTorch fp32 references (GPU)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from distributed_tensorflow_amd.models.graph import same_padding
from distributed_tensorflow_amd.ops.conv_plan import ConvGeom, round_up


def _native():
  from distributed_tensorflow_amd.ops import native
  native.ext()
  return native


def _bf(t):
  return t.to(torch.bfloat16)


def _pack(native, g, w_hwio, which):
  """Runs the pack kernel for one weight; returns the packed bf16 tensor."""
  dev = w_hwio.device
  d = g.pack_fwd() if which == "fwd" else g.pack_dgrad()
  d.update(src_off=0, dst_off=0, first_block=0)
  out = torch.zeros(round_up(d["count"], 8), dtype=torch.bfloat16, device=dev)
  desc = torch.frombuffer(bytearray(native.pack_desc_bytes([d])), dtype=torch.uint8).to(dev)
  native.run_op("pack_weights", dict(params=w_hwio.contiguous().data_ptr(), packed=out.data_ptr(),
                                     descs=desc.data_ptr(), ndesc=1,
                                     nblocks=native.pack_blocks(d)))
  torch.cuda.synchronize()
  return out


def _ref_conv(x_nhwc, w_hwio, stride, pad):
  pt, pb, pl, pr = pad
  x = F.pad(x_nhwc.permute(0, 3, 1, 2), (pl, pr, pt, pb))
  return F.conv2d(x, w_hwio.permute(3, 2, 0, 1), None, stride).permute(0, 2, 3, 1)


CONV_CASES = [
    # N, H, W, Cin, K, Cout, stride, padding
    (2, 14, 14, 64, 1, 256, 1, "same"),      # 1x1 pointwise
    (2, 14, 14, 256, 1, 64, 1, "same"),      # Cout = 64 tile
    (3, 15, 13, 64, 3, 64, 1, "same"),       # 3x3, odd sizes (ragged M)
    (2, 16, 16, 128, 3, 128, 2, "same"),     # 3x3 stride 2, TF pad (0,1)
    (2, 16, 16, 256, 1, 512, 2, "same"),     # 1x1 stride 2 projection
    (2, 8, 8, 32, 3, 32, 1, "same"),         # CIFAR-size channels
    (2, 9, 9, 16, 3, 16, 1, "same"),         # Cout = 16 tile
    (2, 8, 8, 16, 3, 32, 2, "same"),
    (4, 7, 7, 512, 3, 512, 1, "same"),       # deep K = 4608
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case):
  native = _native()
  N, H, W, Cin, K, Cout, stride, padding = case
  dev = torch.device("cuda")
  gen = torch.Generator(device="cpu").manual_seed(hash(case) & 0xffff)
  pad = same_padding(H, K, stride) + same_padding(W, K, stride)
  g = ConvGeom(N, H, W, Cin, K, K, Cout, stride, pad)
  x = _bf(torch.randn(N, H, W, Cin, generator=gen)).to(dev)
  w = (torch.randn(K, K, Cin, Cout, generator=gen) * (1.0 / math.sqrt(K * K * Cin))).to(dev)
  wq = _bf(w).float()
  # ---- forward
  wf = _pack(native, g, w, "fwd")
  out = torch.full(g.output_shape, float("nan"), dtype=torch.bfloat16, device=dev)
  native.run_op("conv", g.fwd(x.data_ptr(), wf.data_ptr(), out.data_ptr()))
  torch.cuda.synchronize()
  ref = _ref_conv(x.float(), wq, stride, pad)
  assert out.shape[:3] == ref.shape[:3]
  err = (out.float() - ref).abs().max().item()
  assert err <= 0.02 * ref.abs().max().item() + 1e-2, err
  # pixel tiles handed out back to front (the launch plan's cache-aware direction): same result
  out_r = torch.full(g.output_shape, float("nan"), dtype=torch.bfloat16, device=dev)
  native.run_op("conv", dict(g.fwd(x.data_ptr(), wf.data_ptr(), out_r.data_ptr()), rev=1))
  torch.cuda.synchronize()
  assert torch.equal(out_r.view(torch.int16), out.view(torch.int16))
  # ---- data gradient (fp32 autograd reference on the same bf16-rounded operands)
  dy = _bf(torch.randn(N, g.OH, g.OW, Cout, generator=gen)).to(dev)
  xr = x.float().requires_grad_(True)
  wr = wq.clone().requires_grad_(True)
  _ref_conv(xr, wr, stride, pad).backward(dy.float())
  wd = _pack(native, g, w, "dgrad")
  dx = torch.full((N, H, W, Cin), float("nan"), dtype=torch.bfloat16, device=dev)
  for d in g.dgrad(dy.data_ptr(), wd.data_ptr(), dx.data_ptr()):
    native.run_op("conv", d)
  torch.cuda.synchronize()
  err = (dx.float() - xr.grad).abs().max().item()
  assert err <= 0.02 * xr.grad.abs().max().item() + 1e-2, err
  dx_r = torch.full((N, H, W, Cin), float("nan"), dtype=torch.bfloat16, device=dev)
  for d in g.dgrad(dy.data_ptr(), wd.data_ptr(), dx_r.data_ptr()):
    native.run_op("conv", dict(d, rev=1))
  torch.cuda.synchronize()
  assert torch.equal(dx_r.view(torch.int16), dx.view(torch.int16))
  # accumulate mode adds to the existing tensor
  dx2 = dx.clone()
  for d in g.dgrad(dy.data_ptr(), wd.data_ptr(), dx2.data_ptr(), accumulate=True):
    native.run_op("conv", d)
  torch.cuda.synchronize()
  assert torch.allclose(dx2.float(), 2 * dx.float(), rtol=2e-2, atol=2e-2)
  # ---- weight gradient (fp32 atomics into HWIO)
  dw = torch.zeros(K, K, Cin, Cout, dtype=torch.float32, device=dev)
  native.run_op("wgrad", g.wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr()))
  torch.cuda.synchronize()
  err = (dw - wr.grad).abs().max().item()
  assert err <= 0.01 * wr.grad.abs().max().item() + 1e-2, err
  dw_r = torch.zeros(K, K, Cin, Cout, dtype=torch.float32, device=dev)
  native.run_op("wgrad", dict(g.wgrad(x.data_ptr(), dy.data_ptr(), dw_r.data_ptr()), splits=3, rev=1))
  torch.cuda.synchronize()
  err = (dw_r - wr.grad).abs().max().item()
  assert err <= 0.01 * wr.grad.abs().max().item() + 1e-2, err


PIPE_CFGS = list(range(16, 32))


@pytest.mark.parametrize("cfg", PIPE_CFGS)
@pytest.mark.parametrize("case", [
    # N, H, W, Cin, K, Cout, stride: shapes the multi-stage pipeline serves itself (pointwise,
    # or channels per tap a multiple of the K-step), with ragged M / Cout / K and 1..72 K-steps
    (3, 15, 13, 64, 1, 200, 1),        # one or two K-steps, ragged Cout and M
    (2, 14, 14, 1024, 1, 256, 1),      # long K pointwise
    (5, 9, 9, 72, 1, 136, 1),          # K = 72: ragged last K-step
    (3, 15, 13, 64, 3, 64, 1),         # 3x3, out-of-image taps
    (2, 16, 16, 128, 3, 128, 2),       # 3x3 stride 2 (TF SAME pad 0, 1); parity data gradients
    (2, 16, 16, 256, 1, 512, 2),       # strided 1x1 projection (gather mode)
    (4, 7, 7, 512, 3, 512, 1),         # K = 4608
])
def test_conv_pipeline_cfgs(case, cfg):
  """Every tile configuration of the multi-stage kernel (conv_pipe.hip): forward, data gradient
  and accumulate mode against fp32 PyTorch on the same bf16-rounded operands."""
  _check_conv_cfg(case, cfg)


@pytest.mark.parametrize("cfg", [7, 8, 9, 10, 11, 12, 13, 14, 15, 32, 33, 34])
@pytest.mark.parametrize("case", [
    # ring variants of the general kernel: gathers whose K-steps straddle taps (the CIFAR
    # stages), tap loops, pointwise, strided; 1 .. 18 K-steps, ragged M and Cout
    (3, 32, 32, 16, 3, 16, 1), (5, 16, 16, 32, 3, 32, 1), (3, 8, 8, 64, 3, 64, 1),
    (3, 17, 17, 16, 3, 32, 2), (2, 15, 13, 24, 3, 40, 1), (4, 9, 9, 8, 1, 16, 1),
    (2, 16, 16, 32, 1, 64, 2), (3, 13, 11, 72, 1, 24, 1),
])
def test_conv_ring_cfgs(case, cfg):
  """`conv_igemm.hip` with a ring of 4 / 6 LDS stages (tile configurations 8-13)."""
  _check_conv_cfg(case, cfg)


def _check_conv_cfg(case, cfg):
  native = _native()
  N, H, W, Cin, K, Cout, stride = case
  dev = torch.device("cuda")
  gen = torch.Generator(device="cpu").manual_seed(sum(case) + cfg)
  pad = same_padding(H, K, stride) + same_padding(W, K, stride)
  g = ConvGeom(N, H, W, Cin, K, K, Cout, stride, pad)
  x = _bf(torch.randn(N, H, W, Cin, generator=gen)).to(dev)
  w = (torch.randn(K, K, Cin, Cout, generator=gen) * (1.0 / math.sqrt(K * K * Cin))).to(dev)
  wq = _bf(w).float()
  wf = _pack(native, g, w, "fwd")
  out = torch.full(g.output_shape, float("nan"), dtype=torch.bfloat16, device=dev)
  native.run_op("conv", g.fwd(x.data_ptr(), wf.data_ptr(), out.data_ptr(), cfg=cfg))
  torch.cuda.synchronize()
  ref = _ref_conv(x.float(), wq, stride, pad)
  err = (out.float()[..., :Cout] - ref).abs().max().item()
  assert err <= 0.02 * ref.abs().max().item() + 1e-2, err
  dy = _bf(torch.randn(N, g.OH, g.OW, g.ldc, generator=gen)).to(dev)
  dy[..., Cout:] = 0
  xr = x.float().requires_grad_(True)
  _ref_conv(xr, wq, stride, pad).backward(dy.float()[..., :Cout])
  wd = _pack(native, g, w, "dgrad")
  dx = torch.full((N, H, W, Cin), float("nan"), dtype=torch.bfloat16, device=dev)
  for d in g.dgrad(dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), cfg=cfg):
    native.run_op("conv", d)
  torch.cuda.synchronize()
  err = (dx.float() - xr.grad).abs().max().item()
  assert err <= 0.02 * xr.grad.abs().max().item() + 1e-2, err
  dx2 = dx.clone()
  for d in g.dgrad(dy.data_ptr(), wd.data_ptr(), dx2.data_ptr(), accumulate=True, cfg=cfg):
    native.run_op("conv", d)
  torch.cuda.synchronize()
  assert torch.allclose(dx2.float(), 2 * dx.float(), rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("case", [
    # N, H, W, Cin, K, Cout, stride, cfg
    (4, 28, 28, 64, 3, 128, 1, 0), (4, 28, 28, 64, 1, 64, 1, 1), (3, 17, 17, 16, 3, 32, 1, 2),
    (3, 17, 17, 16, 3, 16, 2, 3), (2, 14, 14, 256, 1, 512, 1, 4), (4, 28, 28, 128, 1, 64, 1, 5),
    (2, 9, 9, 32, 3, 72, 1, 0), (3, 28, 28, 64, 3, 128, 1, 6), (5, 14, 14, 256, 1, 200, 1, 6),
    (3, 28, 28, 64, 3, 128, 1, 16), (5, 14, 14, 256, 1, 200, 1, 17), (4, 28, 28, 128, 1, 64, 1, 21),
    (4, 28, 28, 64, 1, 64, 1, 23), (2, 14, 14, 256, 1, 512, 1, 25), (3, 28, 28, 64, 3, 128, 1, 20),
    (3, 17, 17, 16, 3, 16, 1, 8), (3, 17, 17, 16, 3, 32, 1, 9), (4, 16, 16, 32, 3, 64, 1, 10),
    (4, 16, 16, 32, 3, 64, 1, 11), (3, 17, 17, 16, 3, 16, 1, 12), (3, 17, 17, 32, 3, 32, 1, 13),
    (3, 56, 56, 64, 3, 64, 1, 48), (2, 28, 28, 128, 3, 128, 1, 49)])
def test_conv_fused_bn_statistics(case):
  """The epilogue's per-channel sum / sum of squares equal those of the stored bf16 output,
  followed by the BatchNorm finalize that consumes the replicas."""
  native = _native()
  N, H, W, Cin, K, Cout, stride, cfg = case
  dev = torch.device("cuda")
  gen = torch.Generator(device="cpu").manual_seed(sum(case))
  pad = same_padding(H, K, stride) + same_padding(W, K, stride)
  g = ConvGeom(N, H, W, Cin, K, K, Cout, stride, pad)
  x = _bf(torch.randn(N, H, W, Cin, generator=gen) + 0.3).to(dev)
  w = (torch.randn(K, K, Cin, Cout, generator=gen) * (1.0 / math.sqrt(K * K * Cin))).to(dev)
  wf = _pack(native, g, w, "fwd")
  R = 5
  out = torch.zeros(g.output_shape, dtype=torch.bfloat16, device=dev)
  stats = torch.zeros(R, 2, Cout, device=dev)
  native.run_op("conv", g.fwd(x.data_ptr(), wf.data_ptr(), out.data_ptr(), cfg=cfg,
                              stats=stats.data_ptr(), stats_replicas=R))
  torch.cuda.synchronize()
  o = out.float().reshape(-1, g.ldc)[:, :Cout]
  tot = stats.sum(0)
  assert torch.allclose(tot[0], o.sum(0), rtol=1e-3, atol=1e-2)
  assert torch.allclose(tot[1], (o * o).sum(0), rtol=1e-3, atol=1e-2)
  if Cout & (Cout - 1) == 0:
    P = o.shape[0]
    gamma = torch.ones(Cout, device=dev)
    beta = torch.zeros(Cout, device=dev)
    mm = torch.zeros(Cout, device=dev)
    mv = torch.ones(Cout, device=dev)
    saved = torch.zeros(4 * Cout, device=dev)
    z = torch.empty(P, Cout, dtype=torch.bfloat16, device=dev)
    native.run_op("bn_fwd_train", dict(
        y=out.data_ptr(), res=None, z=z.data_ptr(), gamma=gamma.data_ptr(), beta=beta.data_ptr(),
        mm=mm.data_ptr(), mv=mv.data_ptr(), save_mean=saved.data_ptr(),
        save_invstd=saved.data_ptr() + 4 * Cout, scale=saved.data_ptr() + 8 * Cout,
        shift=saved.data_ptr() + 12 * Cout, ws=stats.data_ptr(), fused_replicas=R, P=P, C=Cout,
        eps=1e-5, momentum=0.9, relu=0))
    torch.cuda.synchronize()
    ref = (o - o.mean(0)) * torch.rsqrt(o.var(0, unbiased=False) + 1e-5)
    assert torch.allclose(z.float(), ref, rtol=2e-2, atol=2e-2)
    # what the finalize step (a launch of its own, or the prologue of the sweep on narrow
    # layers) leaves for backward and for inference
    assert torch.allclose(saved[:Cout], o.mean(0), rtol=1e-3, atol=1e-3)
    assert torch.allclose(saved[Cout:2 * Cout], torch.rsqrt(o.var(0, unbiased=False) + 1e-5),
                          rtol=2e-3, atol=1e-3)
    assert torch.allclose(mm, 0.1 * o.mean(0), rtol=1e-3, atol=1e-4)
    assert torch.allclose(mv, 0.9 + 0.1 * o.var(0, unbiased=True), rtol=2e-3, atol=1e-4)


@pytest.mark.parametrize("case", [
    # N, H, W, Cin, K, Cout, stride, cfg, with_mask, with_addend
    (3, 20, 20, 64, 3, 72, 1, 0, True, False), (2, 14, 14, 256, 1, 128, 1, 6, True, True),
    (3, 19, 19, 128, 3, 64, 2, 5, True, True), (2, 16, 16, 64, 1, 96, 2, 1, False, False),
    (3, 17, 17, 16, 3, 32, 1, 3, True, False), (2, 12, 12, 512, 1, 64, 1, 4, True, True),
    (2, 56, 56, 64, 3, 64, 1, 48, True, True), (2, 28, 28, 128, 3, 128, 1, 51, True, False)])
def test_dgrad_fused_bn_backward_statistics(case):
  """Data gradient with the backward flavour of the fused statistics: per-channel sum g and
  sum g*y (g = stored bf16 gradient gated by a 1-bit mask) over every parity launch, and the
  BatchNorm backward that consumes the replicas instead of running its reduction sweep."""
  native = _native()
  N, H, W, Cin, K, Cout, stride, cfg, with_mask, with_addend = case
  dev = torch.device("cuda")
  gen = torch.Generator(device="cpu").manual_seed(sum(case[:8]))
  pad = same_padding(H, K, stride) + same_padding(W, K, stride)
  g = ConvGeom(N, H, W, Cin, K, K, Cout, stride, pad)
  w = (torch.randn(K, K, Cin, Cout, generator=gen) * (1.0 / math.sqrt(K * K * Cout))).to(dev)
  wd = _pack(native, g, w, "dgrad")
  dyc = _bf(torch.randn(N, g.OH, g.OW, Cout, generator=gen)).to(dev)
  y = _bf(torch.randn(N, H, W, Cin, generator=gen) * 1.5 + 0.7).to(dev)       # BatchNorm input
  bits = torch.rand(N * H * W, Cin, generator=gen) > 0.4
  mask = (bits.view(-1, Cin // 8, 8).to(torch.int32) << torch.arange(8)).sum(-1).to(torch.uint8)
  mask = torch.cat([mask.flatten(), torch.zeros(4, dtype=torch.uint8)]).to(dev)
  addend = _bf(torch.randn(N, H, W, Cin, generator=gen)).to(dev) if with_addend else None
  R = 32
  stats = torch.zeros(R, 2, Cin, device=dev)
  dx = torch.full((N, H, W, Cin), float("nan"), dtype=torch.bfloat16, device=dev)
  for d in g.dgrad(dyc.data_ptr(), wd.data_ptr(), dx.data_ptr(), cfg=cfg,
                   addend=addend.data_ptr() if with_addend else None):
    d.update(stats=stats.data_ptr(), stats_replicas=R, stats_y=y.data_ptr(),
             stats_mask=mask.data_ptr() if with_mask else None)
    native.run_op("conv", d)
  torch.cuda.synchronize()
  gq = dx.float().reshape(-1, Cin)
  assert torch.isfinite(gq).all()
  if with_mask:
    gq = gq * bits.to(dev)
  yq = y.float().reshape(-1, Cin)
  tot = stats.sum(0)
  scale = gq.abs().sum(0).max().item()
  assert torch.allclose(tot[0], gq.sum(0), rtol=1e-3, atol=1e-4 * scale + 1e-2)
  assert torch.allclose(tot[1], (gq * yq).sum(0), rtol=1e-3, atol=1e-4 * scale + 1e-2)
  if Cin & (Cin - 1) == 0:
    # BatchNorm backward from the replicas == BatchNorm backward with its own reduction
    P = gq.shape[0]
    mean, var = yq.mean(0), yq.var(0, unbiased=False)
    saved = torch.cat([mean, torch.rsqrt(var + 1e-5)]).contiguous()
    gamma = (torch.rand(Cin, generator=gen) + 0.5).to(dev)
    ws = torch.zeros(native.ext().bn_workspace_floats(), device=dev)
    outs = []
    for pre in (0, R):
      dgamma, dbeta = torch.zeros(Cin, device=dev), torch.zeros(Cin, device=dev)
      dyb = torch.empty(P, Cin, dtype=torch.bfloat16, device=dev)
      native.run_op("bn_bwd", dict(
          dz=dx.data_ptr(), y=y.data_ptr(), z=None, mask=mask.data_ptr() if with_mask else None,
          mean=saved.data_ptr(), invstd=saved.data_ptr() + 4 * Cin, gamma=gamma.data_ptr(),
          dgamma=dgamma.data_ptr(), dbeta=dbeta.data_ptr(), dy=dyb.data_ptr(), dres=None,
          ws=stats.data_ptr() if pre else ws.data_ptr(), pre_replicas=pre, P=P, C=Cin))
      torch.cuda.synchronize()
      outs.append((dgamma, dbeta, dyb.float()))
    tol = 2e-3 * max(outs[0][0].abs().max().item(), outs[0][1].abs().max().item()) + 1e-2
    assert torch.allclose(outs[0][0], outs[1][0], rtol=2e-3, atol=tol)
    assert torch.allclose(outs[0][1], outs[1][1], rtol=2e-3, atol=tol)
    assert torch.allclose(outs[0][2], outs[1][2], rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("cfg", [0, 1, 4, 5, 6])
@pytest.mark.parametrize("shape", [(3, 19, 19, 64, 3, 136, 2), (2, 15, 15, 96, 1, 264, 1),
                                   (2, 14, 14, 32, 3, 128, 1)])
def test_conv_every_tile_config(cfg, shape):
  """Forward and data gradient give the same result under every block tile (ragged M and
  Cout, strided parity launches, all three addressing modes)."""
  native = _native()
  N, H, W, Cin, K, Cout, stride = shape
  dev = torch.device("cuda")
  gen = torch.Generator(device="cpu").manual_seed(cfg * 100 + H)
  pad = same_padding(H, K, stride) + same_padding(W, K, stride)
  g = ConvGeom(N, H, W, Cin, K, K, Cout, stride, pad)
  x = _bf(torch.randn(N, H, W, Cin, generator=gen)).to(dev)
  w = (torch.randn(K, K, Cin, Cout, generator=gen) * (1.0 / math.sqrt(K * K * Cin))).to(dev)
  wq = _bf(w).float()
  out = torch.full(g.output_shape, float("nan"), dtype=torch.bfloat16, device=dev)
  native.run_op("conv", g.fwd(x.data_ptr(), _pack(native, g, w, "fwd").data_ptr(), out.data_ptr(),
                              cfg=cfg))
  torch.cuda.synchronize()
  ref = _ref_conv(x.float(), wq, stride, pad)
  assert (out.float()[..., :Cout] - ref).abs().max().item() <= 0.02 * ref.abs().max().item() + 1e-2
  dy = _bf(torch.randn(N, g.OH, g.OW, Cout, generator=gen)).to(dev)
  xr = x.float().requires_grad_(True)
  _ref_conv(xr, wq, stride, pad).backward(dy.float())
  wd = _pack(native, g, w, "dgrad")
  dx = torch.full((N, H, W, Cin), float("nan"), dtype=torch.bfloat16, device=dev)
  for d in g.dgrad(dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), cfg=cfg):
    native.run_op("conv", d)
  torch.cuda.synchronize()
  assert (dx.float() - xr.grad).abs().max().item() <= 0.02 * xr.grad.abs().max().item() + 1e-2


@pytest.mark.parametrize("cfg,splits", [(0, 5), (16, 3), (32, 2), (18, 4), (35, 1), (17, 7), (36, 2),
                                        (22, 3), (39, 2), (8, 2), (24, 3),
                                        # + 64 * (stages - 2): three- and four-stage LDS rings
                                        (64, 5), (80, 3), (99, 1), (65, 7), (66, 4), (67, 2), (83, 3),
                                        (72, 2), (88, 3), (128, 5), (144, 2), (129, 3), (131, 1),
                                        (147, 2), (163, 2), (136, 1)])
@pytest.mark.parametrize("shape", [(3, 19, 19, 64, 3, 136, 2), (2, 15, 15, 96, 1, 72, 1)])
def test_wgrad_every_tile_config_and_group_count(cfg, splits, shape):
  """Weight gradient under every tile / wave-group / pipeline-depth combination
  (cfg = tile + 16 * log2(groups) + 64 * (stages - 2))."""
  native = _native()
  N, H, W, Cin, K, Cout, stride = shape
  dev = torch.device("cuda")
  gen = torch.Generator(device="cpu").manual_seed(cfg + H)
  pad = same_padding(H, K, stride) + same_padding(W, K, stride)
  g = ConvGeom(N, H, W, Cin, K, K, Cout, stride, pad)
  x = _bf(torch.randn(N, H, W, Cin, generator=gen)).to(dev)
  dy = _bf(torch.randn(N, g.OH, g.OW, Cout, generator=gen)).to(dev)
  wr = torch.zeros(K, K, Cin, Cout, device=dev).requires_grad_(True)
  _ref_conv(x.float(), wr, stride, pad).backward(dy.float())
  dw = torch.zeros(K, K, Cin, Cout, dtype=torch.float32, device=dev)
  native.run_op("wgrad", g.wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), cfg=cfg,
                                 splits=splits))
  torch.cuda.synchronize()
  assert (dw - wr.grad).abs().max().item() <= 0.01 * wr.grad.abs().max().item() + 1e-2


@pytest.mark.parametrize("K,cfg,Cin", [(1, 0, 72), (3, 6, 72), (3, 5, 72), (3, 3, 16), (3, 2, 24),
                                       (1, 3, 8)])
def test_dgrad_masked_addend_and_mask_apply(K, cfg, Cin):
  """dX = dgrad(dY) + addend * bitmask (the deferred residual gradient), and the stand-alone
  mask kernel."""
  native = _native()
  dev = torch.device("cuda")
  gen = torch.Generator(device="cpu").manual_seed(K)
  N, H, Cout = 3, 13, 64
  pad = same_padding(H, K, 1) + same_padding(H, K, 1)
  g = ConvGeom(N, H, H, Cin, K, K, Cout, 1, pad)
  w = (torch.randn(K, K, Cin, Cout, generator=gen) * 0.1).to(dev)
  dy = _bf(torch.randn(N, H, H, Cout, generator=gen)).to(dev)
  add = _bf(torch.randn(N, H, H, Cin, generator=gen)).to(dev)
  keep = torch.rand(N, H, H, Cin, generator=gen).to(dev) > 0.4
  bits = (keep.view(-1, Cin // 8, 8).to(torch.int32) << torch.arange(8, device=dev)).sum(-1)
  mask = bits.to(torch.uint8).contiguous()
  xr = torch.zeros(N, H, H, Cin, device=dev).requires_grad_(True)
  _ref_conv(xr, _bf(w).float(), 1, pad).backward(dy.float())
  ref = xr.grad + add.float() * keep
  dx = torch.full((N, H, H, Cin), float("nan"), dtype=torch.bfloat16, device=dev)
  wd = _pack(native, g, w, "dgrad")
  for d in g.dgrad(dy.data_ptr(), wd.data_ptr(), dx.data_ptr(), cfg=cfg, addend=add.data_ptr(),
                   addend_mask=mask.data_ptr()):
    native.run_op("conv", d)
  torch.cuda.synchronize()
  assert (dx.float() - ref).abs().max().item() <= 0.02 * ref.abs().max().item() + 2e-2
  out = torch.empty_like(add)
  native.run_op("mask_apply", dict(x=add.data_ptr(), mask=mask.data_ptr(), out=out.data_ptr(),
                                   n=add.numel()))
  torch.cuda.synchronize()
  assert torch.equal(out.float(), add.float() * keep)


def test_conv_mfma_layout_identity():
  """A = I with an asymmetric B: catches a transposed / permuted accumulator mapping."""
  native = _native()
  dev = torch.device("cuda")
  C = 128
  g = ConvGeom(1, 16, 16, C, 1, 1, C)
  x = torch.zeros(1, 16, 16, C, dtype=torch.bfloat16, device=dev)
  idx = torch.arange(256, device=dev)
  x.view(256, C)[idx, idx % C] = 1.0                     # pixel p selects channel p % C
  w = (torch.arange(C * C, device=dev, dtype=torch.float32).reshape(1, 1, C, C) % 251) - 125
  wf = _pack(native, g, w, "fwd")
  out = torch.zeros(g.output_shape, dtype=torch.bfloat16, device=dev)
  native.run_op("conv", g.fwd(x.data_ptr(), wf.data_ptr(), out.data_ptr()))
  torch.cuda.synchronize()
  ref = w[0, 0][idx % C]                                 # row (p % C) of W for pixel p
  assert torch.equal(out.view(256, C).float(), _bf(ref).float())


def test_stem_merged_taps_and_dense():
  native = _native()
  dev = torch.device("cuda")
  gen = torch.Generator().manual_seed(3)
  # ImageNet stem: 7x7/2, explicit pad 3, 3 channels (merged-tap mode)
  N, H = 2, 32
  g = ConvGeom(N, H, H, 3, 7, 7, 64, 2, (3, 3, 3, 3), mode="merged")
  img = torch.randn(N, H, H, 3, generator=gen).to(dev) * 50 + 100
  xin = torch.zeros(g.input_shape, dtype=torch.bfloat16, device=dev)
  native.run_op("preprocess_pad", dict(N=N, H=H, W=H, Cin=3, pt=3, pl=3, OH=g.Hs, OW=g.Ws, Cpad=4,
                                       out=xin.data_ptr(), **{"in": img.data_ptr()}))
  w = (torch.randn(7, 7, 3, 64, generator=gen) * 0.05).to(dev)
  wf = _pack(native, g, w, "fwd")
  out = torch.zeros(g.output_shape, dtype=torch.bfloat16, device=dev)
  native.run_op("conv", g.fwd(xin.data_ptr(), wf.data_ptr(), out.data_ptr()))
  torch.cuda.synchronize()
  ref = _ref_conv(_bf(img).float(), _bf(w).float(), 2, (3, 3, 3, 3))
  assert (out.float() - ref).abs().max().item() <= 0.02 * ref.abs().max().item()
  dy = _bf(torch.randn(N, g.OH, g.OW, 64, generator=gen)).to(dev)
  dw = torch.zeros(7, 7, 3, 64, device=dev)
  native.run_op("wgrad", g.wgrad(xin.data_ptr(), dy.data_ptr(), dw.data_ptr()))
  torch.cuda.synchronize()
  wr = _bf(w).float().requires_grad_(True)
  _ref_conv(_bf(img).float(), wr, 2, (3, 3, 3, 3)).backward(dy.float())
  assert (dw - wr.grad).abs().max().item() <= 0.01 * wr.grad.abs().max().item() + 1e-2
  # dense layer 256 -> 1001 (ragged N, bias, pitch 1008)
  B, fin, fout = 16, 256, 1001
  gd = ConvGeom(B, 1, 1, fin, 1, 1, fout)
  xd = _bf(torch.randn(B, fin, generator=gen)).to(dev)
  wd_ = (torch.randn(1, 1, fin, fout, generator=gen) * 0.05).to(dev)
  bias = torch.randn(fout, generator=gen).to(dev)
  wfd = _pack(native, gd, wd_, "fwd")
  logits = torch.zeros(B, gd.ldc, dtype=torch.bfloat16, device=dev)
  native.run_op("conv", gd.fwd(xd.data_ptr(), wfd.data_ptr(), logits.data_ptr(), bias=bias.data_ptr()))
  torch.cuda.synchronize()
  ref = xd.float() @ _bf(wd_[0, 0]).float() + bias
  assert gd.ldc == 1008
  assert (logits[:, :fout].float() - ref).abs().max().item() <= 0.03 * ref.abs().max().item()
  assert torch.all(logits[:, fout:] == 0)
  dl = torch.zeros(B, gd.ldc, dtype=torch.bfloat16, device=dev)
  dl[:, :fout] = _bf(torch.randn(B, fout, generator=gen)).to(dev)
  wdd = _pack(native, gd, wd_, "dgrad")
  dxd = torch.zeros(B, fin, dtype=torch.bfloat16, device=dev)
  for d in gd.dgrad(dl.data_ptr(), wdd.data_ptr(), dxd.data_ptr()):
    native.run_op("conv", d)
  dwd = torch.zeros(fin, fout, device=dev)
  native.run_op("wgrad", gd.wgrad(xd.data_ptr(), dl.data_ptr(), dwd.data_ptr()))
  torch.cuda.synchronize()
  ref_dx = dl[:, :fout].float() @ _bf(wd_[0, 0]).float().t()
  ref_dw = xd.float().t() @ dl[:, :fout].float()
  assert (dxd.float() - ref_dx).abs().max().item() <= 0.03 * ref_dx.abs().max().item()
  assert (dwd - ref_dw).abs().max().item() <= 0.01 * ref_dw.abs().max().item() + 1e-2


@pytest.mark.parametrize("C,relu,residual,use_mask", [
    (64, True, False, False), (256, True, True, True), (16, False, False, False),
    (2048, True, True, False), (8, True, False, True), (512, True, False, True)])
def test_batchnorm_fwd_bwd(C, relu, residual, use_mask):
  native = _native()
  dev = torch.device("cuda")
  gen = torch.Generator().manual_seed(C)
  P = 64 * 7 * 7 + 3
  y = _bf(torch.randn(P, C, generator=gen) * 2 + 0.5).to(dev)
  res = _bf(torch.randn(P, C, generator=gen)).to(dev) if residual else None
  gamma = (torch.rand(C, generator=gen) + 0.5).to(dev)
  beta = (torch.randn(C, generator=gen) * 0.1).to(dev)
  mm = torch.zeros(C, device=dev)
  mv = torch.ones(C, device=dev)
  saved = torch.zeros(4 * C, device=dev)
  ws = torch.zeros(native.ext().bn_workspace_floats(), device=dev)
  z = torch.empty(P, C, dtype=torch.bfloat16, device=dev)
  mask = torch.full(((P * C // 8 + 3) // 4 * 4,), 0xAA, dtype=torch.uint8, device=dev) \
      if use_mask else None
  native.run_op("bn_fwd_train", dict(
      y=y.data_ptr(), res=res.data_ptr() if residual else None, z=z.data_ptr(),
      mask=mask.data_ptr() if use_mask else None,
      gamma=gamma.data_ptr(), beta=beta.data_ptr(), mm=mm.data_ptr(), mv=mv.data_ptr(),
      save_mean=saved.data_ptr(), save_invstd=saved.data_ptr() + 4 * C,
      scale=saved.data_ptr() + 8 * C, shift=saved.data_ptr() + 12 * C, ws=ws.data_ptr(),
      P=P, C=C, eps=1e-5, momentum=0.9, relu=int(relu)))
  torch.cuda.synchronize()
  yr = y.float().requires_grad_(True)
  gr = gamma.clone().requires_grad_(True)
  br = beta.clone().requires_grad_(True)
  rr = res.float().requires_grad_(True) if residual else None
  mean = yr.mean(0)
  var = yr.var(0, unbiased=False)
  zr = (yr - mean) * torch.rsqrt(var + 1e-5) * gr + br
  if residual:
    zr = zr + rr
  if relu:
    zr = torch.relu(zr)
  assert torch.allclose(z.float(), zr.detach(), rtol=2e-2, atol=2e-2)
  assert torch.allclose(mm, 0.1 * mean.detach(), rtol=1e-3, atol=1e-4)
  assert torch.allclose(mv, 0.9 + 0.1 * yr.detach().var(0, unbiased=True), rtol=1e-3, atol=1e-4)
  # backward
  dz = _bf(torch.randn(P, C, generator=gen)).to(dev)
  # reference mask must come from the bf16 output the kernel saved
  zq = z.float()
  g = dz.float() * (zq > 0) if relu else dz.float()
  xhat = ((yr - mean) * torch.rsqrt(var + 1e-5)).detach()
  dgamma_ref = (g * xhat).sum(0)
  dbeta_ref = g.sum(0)
  dy_ref = gamma * torch.rsqrt(var.detach() + 1e-5) * (g - dbeta_ref / P - xhat * dgamma_ref / P)
  dgamma = torch.zeros(C, device=dev)
  dbeta = torch.zeros(C, device=dev)
  dy = torch.empty(P, C, dtype=torch.bfloat16, device=dev)
  dres = torch.empty(P, C, dtype=torch.bfloat16, device=dev) if residual else None
  if use_mask:    # bit c of byte [row][chunk] <=> z[row, chunk*8 + c] > 0
    bits = (zq > 0).view(P, C // 8, 8).to(torch.int32)
    expect = (bits << torch.arange(8, device=dev)).sum(-1).to(torch.uint8)
    assert torch.equal(mask[:P * C // 8].view(P, C // 8), expect)
  native.run_op("bn_bwd", dict(dz=dz.data_ptr(), y=y.data_ptr(),
                               z=z.data_ptr() if relu and not use_mask else None,
                               mask=mask.data_ptr() if use_mask else None,
                               mean=saved.data_ptr(), invstd=saved.data_ptr() + 4 * C, gamma=gamma.data_ptr(),
                               dgamma=dgamma.data_ptr(), dbeta=dbeta.data_ptr(), dy=dy.data_ptr(),
                               dres=dres.data_ptr() if residual else None, ws=ws.data_ptr(), P=P, C=C))
  torch.cuda.synchronize()
  assert torch.allclose(dgamma, dgamma_ref, rtol=1e-2, atol=5e-2)
  assert torch.allclose(dbeta, dbeta_ref, rtol=1e-2, atol=5e-2)
  assert torch.allclose(dy.float(), dy_ref, rtol=3e-2, atol=3e-2)
  if residual:
    assert torch.equal(dres.float(), _bf(g).float())
  # inference mode uses the moving statistics
  z2 = torch.empty_like(z)
  native.run_op("bn_infer", dict(y=y.data_ptr(), res=None, z=z2.data_ptr(), gamma=gamma.data_ptr(),
                                 beta=beta.data_ptr(), mm=mm.data_ptr(), mv=mv.data_ptr(), P=P, C=C,
                                 eps=1e-5, relu=0))
  torch.cuda.synchronize()
  ref2 = (y.float() - mm) * torch.rsqrt(mv + 1e-5) * gamma + beta
  assert torch.allclose(z2.float(), ref2, rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("C,P", [(2048, 3139), (256, 40003), (64, 70001), (8, 300001),
                                 (256, 190001)])
def test_batchnorm_sweep_directions(C, P):
  """The BatchNorm sweeps walk their passes front to back or back to front (`rev`, `rev_reduce`,
  `rev_apply`: the launch plan picks the end of the tensor that is still cached): several passes
  with a ragged last one, every direction against the front-to-back launch.  The last case is a
  97 MB tensor: above 90 MB the sweeps switch to non-temporal loads and stores."""
  native = _native()
  dev = torch.device("cuda")
  gen = torch.Generator().manual_seed(C + P)
  y = _bf(torch.randn(P, C, generator=gen) * 2 + 0.5).to(dev)
  res = _bf(torch.randn(P, C, generator=gen)).to(dev)
  dz = _bf(torch.randn(P, C, generator=gen)).to(dev)
  gamma = (torch.rand(C, generator=gen) + 0.5).to(dev)
  beta = (torch.randn(C, generator=gen) * 0.1).to(dev)
  ws = torch.zeros(native.ext().bn_workspace_floats(), device=dev)
  nmask = (P * C // 8 + 3) // 4 * 4

  def run(rev_f, rev_r, rev_a):
    mm, mv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    saved = torch.zeros(4 * C, device=dev)
    z = torch.full((P, C), float("nan"), dtype=torch.bfloat16, device=dev)
    mask = torch.full((nmask,), 0xAA, dtype=torch.uint8, device=dev)
    native.run_op("bn_fwd_train", dict(
        y=y.data_ptr(), res=res.data_ptr(), z=z.data_ptr(), mask=mask.data_ptr(),
        gamma=gamma.data_ptr(), beta=beta.data_ptr(), mm=mm.data_ptr(), mv=mv.data_ptr(),
        save_mean=saved.data_ptr(), save_invstd=saved.data_ptr() + 4 * C,
        scale=saved.data_ptr() + 8 * C, shift=saved.data_ptr() + 12 * C, ws=ws.data_ptr(),
        P=P, C=C, eps=1e-5, momentum=0.9, relu=1, rev=rev_f))
    dgamma, dbeta = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dy = torch.full((P, C), float("nan"), dtype=torch.bfloat16, device=dev)
    dres = torch.full((P, C), float("nan"), dtype=torch.bfloat16, device=dev)
    native.run_op("bn_bwd", dict(
        dz=dz.data_ptr(), y=y.data_ptr(), z=None, mask=mask.data_ptr(), mean=saved.data_ptr(),
        invstd=saved.data_ptr() + 4 * C, gamma=gamma.data_ptr(), dgamma=dgamma.data_ptr(),
        dbeta=dbeta.data_ptr(), dy=dy.data_ptr(), dres=dres.data_ptr(), ws=ws.data_ptr(), P=P, C=C,
        rev_reduce=rev_r, rev_apply=rev_a))
    torch.cuda.synchronize()
    return z, mask[:P * C // 8].clone(), dgamma, dbeta, dy, dres

  z0, m0, dg0, db0, dy0, dr0 = run(0, 0, 0)
  zr = torch.relu((y.float() - y.float().mean(0)) * torch.rsqrt(y.float().var(0, unbiased=False) + 1e-5)
                  * gamma + beta + res.float())
  assert torch.allclose(z0.float(), zr, rtol=2e-2, atol=2e-2)
  for rev in ((1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 1)):
    z1, m1, dg1, db1, dy1, dr1 = run(*rev)
    assert torch.equal(z1.view(torch.int16), z0.view(torch.int16)), rev
    assert torch.equal(m1, m0), rev
    assert torch.equal(dr1.view(torch.int16), dr0.view(torch.int16)), rev
    # the reduction adds its rows in another order: sums agree to rounding, dy to a bf16 ulp
    assert torch.allclose(dg1, dg0, rtol=1e-4, atol=1e-2) and torch.allclose(db1, db0, rtol=1e-4, atol=1e-2)
    assert torch.allclose(dy1.float(), dy0.float(), rtol=1e-2, atol=1e-2), rev


@pytest.mark.parametrize("N,H,W,C", [(3, 12, 13, 64), (2, 9, 8, 8), (1, 16, 16, 256)])
def test_batchnorm_relu_maxpool_fused(N, H, W, C):
  """BatchNorm + ReLU + 3x3/2 max-pool (TensorFlow SAME) in one sweep each way against fp32
  autograd of the same composition (reference layers `resnet_model.py:264-270`)."""
  native = _native()
  dev = torch.device("cuda")
  gen = torch.Generator().manual_seed(N * 100 + C)
  k, st = 3, 2
  (pt, pb), (pl, pr) = same_padding(H, k, st), same_padding(W, k, st)
  OH, OW = -(-H // st), -(-W // st)
  P = N * H * W
  y = _bf(torch.randn(N, H, W, C, generator=gen) * 2 + 0.3).to(dev)
  gamma = (torch.rand(C, generator=gen) + 0.5).to(dev)
  gamma[::3] *= -1                      # negative scales: max(bn(y)) != bn(max(y))
  beta = (torch.randn(C, generator=gen) * 0.3).to(dev)
  mm = torch.zeros(C, device=dev)
  mv = torch.ones(C, device=dev)
  saved = torch.zeros(4 * C, device=dev)
  ws = torch.zeros(native.ext().bn_workspace_floats(), device=dev)
  zp = torch.empty(N, OH, OW, C, dtype=torch.bfloat16, device=dev)
  idx = torch.empty(N, OH, OW, C, dtype=torch.uint8, device=dev)
  geo = dict(N=N, H=H, W=W, C=C, OH=OH, OW=OW, k=k, stride=st, pt=pt, pl=pl, idx=idx.data_ptr(),
             y=y.data_ptr(), gamma=gamma.data_ptr(), ws=ws.data_ptr(),
             save_mean=saved.data_ptr(), save_invstd=saved.data_ptr() + 4 * C)
  native.run_op("bn_pool_fwd_train", dict(
      zp=zp.data_ptr(), beta=beta.data_ptr(), mm=mm.data_ptr(), mv=mv.data_ptr(),
      scale=saved.data_ptr() + 8 * C, shift=saved.data_ptr() + 12 * C, eps=1e-5, momentum=0.9,
      **geo))
  torch.cuda.synchronize()
  yr = y.float().requires_grad_(True)
  gr = gamma.clone().requires_grad_(True)
  br = beta.clone().requires_grad_(True)
  flat = yr.reshape(P, C)
  mean, var = flat.mean(0), flat.var(0, unbiased=False)
  zr = torch.relu((yr - mean) * torch.rsqrt(var + 1e-5) * gr + br)
  pr_ = F.max_pool2d(F.pad(zr.permute(0, 3, 1, 2), (pl, pr, pt, pb), value=float("-inf")), k, st)
  ref = pr_.permute(0, 2, 3, 1)
  assert torch.allclose(zp.float(), ref.detach(), rtol=2e-2, atol=2e-2)
  assert torch.allclose(mm, 0.1 * mean.detach(), rtol=1e-3, atol=1e-4)
  assert torch.equal(idx == 255, zp == 0)            # no positive tap <=> pooled value is zero
  dzp = _bf(torch.randn(N, OH, OW, C, generator=gen)).to(dev)
  ref.backward(dzp.float())
  dgamma = torch.zeros(C, device=dev)
  dbeta = torch.zeros(C, device=dev)
  dy = torch.full((N, H, W, C), float("nan"), dtype=torch.bfloat16, device=dev)
  native.run_op("bn_pool_bwd", dict(dzp=dzp.data_ptr(), dy=dy.data_ptr(), dgamma=dgamma.data_ptr(),
                                    dbeta=dbeta.data_ptr(), **geo))
  torch.cuda.synchronize()
  assert torch.allclose(dgamma, gr.grad, rtol=2e-2, atol=0.1)
  assert torch.allclose(dbeta, br.grad, rtol=2e-2, atol=0.1)
  assert torch.isfinite(dy.float()).all()
  bad = ~torch.isclose(dy.float(), yr.grad, rtol=3e-2, atol=3e-2)
  assert bad.float().mean().item() < 1e-3            # a near-tie may pick another tap


@pytest.mark.parametrize("C,P", [(64, 3001), (512, 777), (2048, 130)])
def test_batchnorm_backward_dual_matches_two_single_launches(C, P):
  """Two BatchNorm layers fed by the same masked gradient (a block's last layer and its
  projection shortcut's, reference `resnet_model.py:204-219`): the dual sweeps must give what
  two separate backward launches give."""
  native = _native()
  dev = torch.device("cuda")
  gen = torch.Generator().manual_seed(C + P)
  dz = _bf(torch.randn(P, C, generator=gen)).to(dev)
  bits = torch.rand(P, C, generator=gen) > 0.45
  mask = (bits.view(P, C // 8, 8).to(torch.int32) << torch.arange(8)).sum(-1).to(torch.uint8)
  mask = torch.cat([mask.flatten(), torch.zeros(4, dtype=torch.uint8)]).to(dev)
  ws = torch.zeros(native.ext().bn_workspace_floats(), device=dev)
  ws2 = torch.zeros_like(ws)
  layers = []
  for i in range(2):
    y = _bf(torch.randn(P, C, generator=gen) * (1.5 + i) + 0.4 * i).to(dev)
    mean = y.float().mean(0)
    invstd = torch.rsqrt(y.float().var(0, unbiased=False) + 1e-5)
    gamma = (torch.rand(C, generator=gen) + 0.5).to(dev)
    layers.append(dict(y=y, mean=mean.contiguous(), invstd=invstd.contiguous(), gamma=gamma))
  single = []
  for L in layers:
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dy = torch.empty(P, C, dtype=torch.bfloat16, device=dev)
    native.run_op("bn_bwd", dict(dz=dz.data_ptr(), y=L["y"].data_ptr(), z=None, mask=mask.data_ptr(),
                                 mean=L["mean"].data_ptr(), invstd=L["invstd"].data_ptr(),
                                 gamma=L["gamma"].data_ptr(), dgamma=dg.data_ptr(),
                                 dbeta=db.data_ptr(), dy=dy.data_ptr(), dres=None,
                                 ws=ws.data_ptr(), P=P, C=C))
    torch.cuda.synchronize()
    single.append((dg, db, dy))
  out = [(torch.zeros(C, device=dev), torch.zeros(C, device=dev),
          torch.full((P, C), float("nan"), dtype=torch.bfloat16, device=dev)) for _ in range(2)]
  A, B = layers
  native.run_op("bn_bwd_dual", dict(
      dz=dz.data_ptr(), mask=mask.data_ptr(), P=P, C=C, ws=ws.data_ptr(), ws2=ws2.data_ptr(),
      y=A["y"].data_ptr(), mean=A["mean"].data_ptr(), invstd=A["invstd"].data_ptr(),
      gamma=A["gamma"].data_ptr(), dgamma=out[0][0].data_ptr(), dbeta=out[0][1].data_ptr(),
      dy=out[0][2].data_ptr(),
      y2=B["y"].data_ptr(), mean2=B["mean"].data_ptr(), invstd2=B["invstd"].data_ptr(),
      gamma2=B["gamma"].data_ptr(), dgamma2=out[1][0].data_ptr(), dbeta2=out[1][1].data_ptr(),
      dy2=out[1][2].data_ptr()))
  torch.cuda.synchronize()
  for (dg, db, dy), (dg2, db2, dy2) in zip(single, out):
    assert torch.allclose(dg, dg2, rtol=1e-4, atol=1e-3)
    assert torch.allclose(db, db2, rtol=1e-4, atol=1e-3)
    assert torch.allclose(dy.float(), dy2.float(), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("C", [64, 1024])
def test_batchnorm_forward_with_lazily_normalised_residual(C):
  """`apply=0` leaves only the statistics / constants of a layer; a second layer then adds
  that layer's *input* as residual, normalising it on the fly (rscale / rshift)."""
  native = _native()
  dev = torch.device("cuda")
  gen = torch.Generator().manual_seed(C)
  P = 1500
  y1 = _bf(torch.randn(P, C, generator=gen) * 3 + 1).to(dev)
  y2 = _bf(torch.randn(P, C, generator=gen) * 2 - 0.5).to(dev)
  ws = torch.zeros(native.ext().bn_workspace_floats(), device=dev)
  pars = []
  for _ in range(2):
    pars.append(dict(gamma=(torch.rand(C, generator=gen) + 0.5).to(dev),
                     beta=(torch.randn(C, generator=gen) * 0.2).to(dev),
                     mm=torch.zeros(C, device=dev), mv=torch.ones(C, device=dev),
                     saved=torch.zeros(4 * C, device=dev)))
  def args(p, y, **kw):
    sv = p["saved"].data_ptr()
    return dict(y=y.data_ptr(), gamma=p["gamma"].data_ptr(), beta=p["beta"].data_ptr(),
                mm=p["mm"].data_ptr(), mv=p["mv"].data_ptr(), save_mean=sv, save_invstd=sv + 4 * C,
                scale=sv + 8 * C, shift=sv + 12 * C, ws=ws.data_ptr(), P=P, C=C, eps=1e-5,
                momentum=0.9, **kw)
  untouched = torch.full((P, C), 7.0, dtype=torch.bfloat16, device=dev)
  native.run_op("bn_fwd_train", args(pars[0], y1, res=None, z=untouched.data_ptr(), relu=0, apply=0))
  z = torch.empty(P, C, dtype=torch.bfloat16, device=dev)
  sv = pars[0]["saved"].data_ptr()
  native.run_op("bn_fwd_train", args(pars[1], y2, res=y1.data_ptr(), z=z.data_ptr(), relu=1,
                                     rscale=sv + 8 * C, rshift=sv + 12 * C))
  torch.cuda.synchronize()
  assert (untouched.float() == 7.0).all()                   # apply=0 wrote no activation
  def bn(y, p):
    y = y.float()
    return (y - y.mean(0)) * torch.rsqrt(y.var(0, unbiased=False) + 1e-5) * p["gamma"] + p["beta"]
  ref = torch.relu(bn(y2, pars[1]) + bn(y1, pars[0]))
  assert torch.allclose(z.float(), ref, rtol=2e-2, atol=2e-2)
  assert torch.allclose(pars[0]["mm"], 0.1 * y1.float().mean(0), rtol=1e-3, atol=1e-4)


def test_pooling():
  native = _native()
  dev = torch.device("cuda")
  gen = torch.Generator().manual_seed(5)
  N, H, W, C = 2, 12, 12, 64
  x = torch.relu(_bf(torch.randn(N, H, W, C, generator=gen))).to(dev)     # zeros -> ties
  OH = OW = 6
  out = torch.empty(N, OH, OW, C, dtype=torch.bfloat16, device=dev)
  idx = torch.empty(N, OH, OW, C, dtype=torch.uint8, device=dev)
  args = dict(N=N, H=H, W=W, C=C, OH=OH, OW=OW, k=3, stride=2, pt=0, pl=0, idx=idx.data_ptr())
  native.run_op("maxpool_fwd", dict(x=x.data_ptr(), out=out.data_ptr(), **args))
  xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
  ref = F.max_pool2d(F.pad(xr, (0, 1, 0, 1), value=float("-inf")), 3, 2)
  torch.cuda.synchronize()
  assert torch.equal(out.float(), ref.detach().permute(0, 2, 3, 1))
  dy = _bf(torch.randn(N, OH, OW, C, generator=gen)).to(dev)
  dx = torch.empty(N, H, W, C, dtype=torch.bfloat16, device=dev)
  native.run_op("maxpool_bwd", dict(dy=dy.data_ptr(), dx=dx.data_ptr(), **args))
  torch.cuda.synchronize()
  # gradient mass is conserved and lands on maxima
  assert torch.allclose(dx.float().sum((1, 2)), dy.float().sum((1, 2)), rtol=1e-2, atol=0.3)
  ref.backward(dy.float().permute(0, 3, 1, 2))
  nz = x.float() > 0          # away from the ReLU ties both implementations must agree
  assert torch.allclose(dx.float()[nz], xr.grad.permute(0, 2, 3, 1)[nz], rtol=2e-2, atol=2e-2)
  # global mean
  m = torch.empty(N, C, dtype=torch.bfloat16, device=dev)
  native.run_op("mean_fwd", dict(x=x.data_ptr(), out=m.data_ptr(), N=N, HW=H * W, C=C))
  torch.cuda.synchronize()
  assert torch.allclose(m.float(), x.float().mean((1, 2)), rtol=1e-2, atol=1e-2)
  dm = _bf(torch.randn(N, C, generator=gen)).to(dev)
  dxx = torch.empty(N, H, W, C, dtype=torch.bfloat16, device=dev)
  native.run_op("mean_bwd", dict(dy=dm.data_ptr(), dx=dxx.data_ptr(), N=N, HW=H * W, C=C))
  torch.cuda.synchronize()
  assert torch.allclose(dxx.float(), (dm.float() / (H * W))[:, None, None, :].expand(N, H, W, C),
                        rtol=1e-2, atol=1e-4)


@pytest.mark.parametrize("C,ld", [(1001, 1008), (10, 16)])
def test_softmax_xent(C, ld):
  native = _native()
  dev = torch.device("cuda")
  gen = torch.Generator().manual_seed(7)
  N = 32
  logits = torch.zeros(N, ld, dtype=torch.bfloat16, device=dev)
  logits[:, :C] = _bf(torch.randn(N, C, generator=gen) * 3).to(dev)
  labels = torch.randint(0, C, (N,), generator=gen).to(dev)
  labels[:4] = logits[:4, :C].float().argmax(1)          # some correct predictions
  lab32 = labels.to(torch.int32)
  dl = torch.full((N, ld), float("nan"), dtype=torch.bfloat16, device=dev)
  acc = torch.zeros(2, device=dev)
  scale = torch.full((1,), 128.0, device=dev)
  native.run_op("softmax_xent", dict(logits=logits.data_ptr(), labels=lab32.data_ptr(), dlogits=dl.data_ptr(),
                                     accum=acc.data_ptr(), N=N, C=C, ld=ld, grad_scale=1.0 / N,
                                     grad_scale_dev=scale.data_ptr()))
  torch.cuda.synchronize()
  lr = logits[:, :C].float().requires_grad_(True)
  loss = F.cross_entropy(lr, labels, reduction="sum")
  (loss * 128.0 / N).backward()
  assert acc[0].item() == pytest.approx(loss.item(), rel=1e-4)
  assert acc[1].item() == (lr.argmax(1) == labels).sum().item()
  assert torch.allclose(dl[:, :C].float(), lr.grad, rtol=2e-2, atol=2e-3)
  assert torch.all(dl[:, C:] == 0)
  probs = torch.empty(N, C, device=dev)
  native.run_op("softmax_probs", dict(logits=logits.data_ptr(), probs=probs.data_ptr(), N=N, C=C, ld=ld))
  torch.cuda.synchronize()
  assert torch.allclose(probs, torch.softmax(lr.detach(), 1), rtol=1e-4, atol=1e-6)
  cs = torch.zeros(C, device=dev)
  native.run_op("colsum", dict(x=dl.data_ptr(), out=cs.data_ptr(), N=N, C=C, ld=ld))
  torch.cuda.synchronize()
  assert torch.allclose(cs, dl[:, :C].float().sum(0), rtol=1e-4, atol=1e-4)


def test_fused_sgd_matches_keras_rule():
  native = _native()
  dev = torch.device("cuda")
  gen = torch.Generator().manual_seed(11)
  n = 100003 * 4
  p = torch.randn(n, generator=gen).to(dev)
  g = torch.randn(n, generator=gen).to(dev)
  v = torch.randn(n, generator=gen).to(dev) * 0.1
  l2 = torch.where(torch.rand(n, generator=gen) < 0.5, 2e-4, 0.0).to(dev)
  p0, v0 = p.clone(), v.clone()
  native.sgd_momentum_l2(p, g, v, l2, lr=0.1, momentum=0.9, grad_scale=0.125)
  torch.cuda.synchronize()
  gg = g * 0.125 + l2 * p0
  v_ref = 0.9 * v0 - 0.1 * gg
  assert torch.allclose(v, v_ref, rtol=1e-6, atol=1e-7)
  assert torch.allclose(p, p0 + v_ref, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("P,C,dual", [(50176, 256, False), (12544, 2048, False), (3000, 64, False),
                                      (12544, 1024, True), (777, 8, False)])
def test_batchnorm_backward_reduction_finalised_by_its_last_block(P, C, dual):
  """`tail`: the reduction sweep adds block sums to replica rows and its last block derives
  dgamma / dbeta (no finalize launch).  Same results as the deterministic three-launch path
  within summation-order differences, for the single and the dual (shared gradient) form; the
  replica rows and the ticket are left zeroed, so the second call sees a clean slate."""
  native = _native()
  dev = torch.device("cuda")
  gen = torch.Generator().manual_seed(P + C)
  ext = native.ext()
  tail = torch.zeros(ext.bn_tail_floats(), device=dev)
  ws = torch.empty(ext.bn_workspace_floats(), device=dev)
  ws2 = torch.empty(ext.bn_workspace_floats(), device=dev)

  def layer():
    y = _bf(torch.randn(P, C, generator=gen) * 2 + 0.5).to(dev)
    mean = y.float().mean(0)
    invstd = torch.rsqrt(y.float().var(0, unbiased=False) + 1e-5)
    gamma = (torch.rand(C, generator=gen) + 0.5).to(dev)
    return y, mean.contiguous(), invstd.contiguous(), gamma

  dz = _bf(torch.randn(P, C, generator=gen)).to(dev)
  mask = torch.randint(0, 256, (P * C // 8 + 4,), generator=gen, dtype=torch.uint8).to(dev)
  L1, L2 = layer(), layer()
  outs = {}
  for mode in ("plain", "tail", "tail_again"):
    t = None if mode == "plain" else tail.data_ptr()
    r = [torch.zeros(C, device=dev) for _ in range(4)]
    dy1 = torch.empty(P, C, dtype=torch.bfloat16, device=dev)
    dy2 = torch.empty(P, C, dtype=torch.bfloat16, device=dev)
    if dual:
      native.run_op("bn_bwd_dual", dict(
          dz=dz.data_ptr(), mask=mask.data_ptr(), P=P, C=C, ws=ws.data_ptr(), ws2=ws2.data_ptr(),
          y=L1[0].data_ptr(), mean=L1[1].data_ptr(), invstd=L1[2].data_ptr(), gamma=L1[3].data_ptr(),
          dgamma=r[0].data_ptr(), dbeta=r[1].data_ptr(), dy=dy1.data_ptr(),
          y2=L2[0].data_ptr(), mean2=L2[1].data_ptr(), invstd2=L2[2].data_ptr(), gamma2=L2[3].data_ptr(),
          dgamma2=r[2].data_ptr(), dbeta2=r[3].data_ptr(), dy2=dy2.data_ptr(), tail=t))
    else:
      native.run_op("bn_bwd", dict(
          dz=dz.data_ptr(), y=L1[0].data_ptr(), z=None, mask=mask.data_ptr(), mean=L1[1].data_ptr(),
          invstd=L1[2].data_ptr(), gamma=L1[3].data_ptr(), dgamma=r[0].data_ptr(), dbeta=r[1].data_ptr(),
          dy=dy1.data_ptr(), dres=None, ws=ws.data_ptr(), P=P, C=C, tail=t))
    torch.cuda.synchronize()
    outs[mode] = (r, dy1.float(), dy2.float())
    assert tail.abs().sum().item() == 0.0            # rows and ticket are clean again
  # fp32 reference of the sums
  bits = ((mask[:P * C // 8].view(P, C // 8, 1).to(torch.int32) >> torch.arange(8, device=dev)) & 1)
  g = dz.float() * bits.view(P, C).float()
  xhat = (L1[0].float() - L1[1]) * L1[2]
  assert torch.allclose(outs["plain"][0][0], (g * xhat).sum(0), rtol=1e-3, atol=1e-2)
  for mode in ("tail", "tail_again"):
    for a, b in zip(outs[mode][0], outs["plain"][0]):
      assert torch.allclose(a, b, rtol=1e-4, atol=1e-2 * max(1.0, (P / 1e4) ** 0.5)), mode
    assert torch.allclose(outs[mode][1], outs["plain"][1], rtol=2e-2, atol=2e-2)
    if dual:
      assert torch.allclose(outs[mode][2], outs["plain"][2], rtol=2e-2, atol=2e-2)


@pytest.mark.parametrize("shape,cfg,splits", [((16, 32, 32, 16, 3, 16, 1), 5, 4),
                                              ((16, 16, 16, 32, 3, 32, 1), 6, 2),
                                              ((8, 14, 14, 64, 1, 128, 1), 2, 2)])
def test_batched_weight_gradient_launch(shape, cfg, splits):
  """One launch, one grid z-slice per layer: the weight gradients of five layers of identical
  geometry from a device table of (x, dy, dw) pointers equal the five separate launches."""
  native = _native()
  N, H, W, Cin, K, Cout, stride = shape
  dev = torch.device("cuda")
  gen = torch.Generator(device="cpu").manual_seed(H + Cin)
  pad = same_padding(H, K, stride) + same_padding(W, K, stride)
  g = ConvGeom(N, H, W, Cin, K, K, Cout, stride, pad)
  L = 5
  xs = [_bf(torch.randn(N, H, W, Cin, generator=gen)).to(dev) for _ in range(L)]
  dys = [_bf(torch.randn(N, g.OH, g.OW, Cout, generator=gen)).to(dev) for _ in range(L)]
  one = [torch.zeros(K, K, Cin, Cout, dtype=torch.float32, device=dev) for _ in range(L)]
  for x, dy, dw in zip(xs, dys, one):
    native.run_op("wgrad", g.wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), cfg=cfg, splits=splits))
  many = [torch.zeros(K, K, Cin, Cout, dtype=torch.float32, device=dev) for _ in range(L)]
  table = torch.tensor([[x.data_ptr(), dy.data_ptr(), dw.data_ptr()] for x, dy, dw in zip(xs, dys, many)],
                       dtype=torch.int64, device=dev)
  d = g.wgrad(xs[0].data_ptr(), dys[0].data_ptr(), many[0].data_ptr(), cfg=cfg, splits=splits)
  d.update(batch=table.data_ptr(), nbatch=L)
  native.run_op("wgrad", d)
  torch.cuda.synchronize()
  for i, (a, b) in enumerate(zip(many, one)):
    wr = torch.zeros(K, K, Cin, Cout, device=dev).requires_grad_(True)
    _ref_conv(xs[i].float(), wr, stride, pad).backward(dys[i].float())
    assert (a - wr.grad).abs().max().item() <= 0.01 * wr.grad.abs().max().item() + 1e-2
    assert torch.allclose(a, b, rtol=1e-4, atol=1e-3 * b.abs().max().item())


@pytest.mark.parametrize("cfg", list(range(40, 48)))
@pytest.mark.parametrize("case", [
    # unit-stride layers of the CIFAR ResNet stages through the one-round-trip kernel
    # (conv_halo.hip: halo tile + resident weights); shapes a tile cannot hold (pixel rows that are
    # not whole image rows, weight slab too large) fall back to the two-buffer tile of the same
    # footprint, so every combination must give the right numbers
    (3, 32, 32, 16, 3, 16, 1), (5, 16, 16, 32, 3, 32, 1), (3, 8, 8, 64, 3, 64, 1),
    (2, 16, 8, 16, 3, 24, 1), (2, 8, 16, 32, 1, 40, 1), (3, 17, 17, 16, 3, 32, 2),
    (2, 15, 13, 24, 3, 40, 1),
])
def test_conv_halo_cfgs(case, cfg):
  """Forward, data gradient (taps walked backwards over the halo tile) and accumulate mode."""
  _check_conv_cfg(case, cfg)


@pytest.mark.parametrize("cfg", list(range(40, 48)))
@pytest.mark.parametrize("case", [
    # N, H, W, Cin, K, Cout, relu, residual: the consumer convolutions of the CIFAR ResNet stages
    (3, 32, 32, 16, 3, 16, 1, 0), (5, 16, 16, 32, 3, 32, 1, 1), (3, 8, 8, 64, 3, 64, 1, 1),
    (2, 16, 8, 16, 3, 24, 0, 0), (2, 8, 16, 32, 1, 40, 1, 0), (2, 32, 32, 16, 3, 16, 1, 1),
])
def test_conv_with_batchnorm_prologue(case, cfg):
  """Training BatchNorm (+ residual) + ReLU folded into the consumer convolution (reference layer
  pairs `resnet_cifar_model.py:75-92`, residual form `:94-96`): the launch takes the raw producer
  output y and the sums its epilogue accumulated, and must give what normalising first (fp32,
  rounded to bf16 like the separate sweep) and convolving second gives - output, forward
  statistics of the output, the normalised tensor and ReLU mask it writes on the side, saved and
  moving statistics.  A tile whose pixel rows are not whole image rows is refused on the host."""
  from distributed_tensorflow_amd.ops.conv_plan import FWD_CFGS
  native = _native()
  N, H, W, Cin, K, Cout, relu, residual = case
  dev = torch.device("cuda")
  gen = torch.Generator(device="cpu").manual_seed(hash(case) & 0xffff)
  pad = same_padding(H, K, 1) + same_padding(W, K, 1)
  g = ConvGeom(N, H, W, Cin, K, K, Cout, 1, pad)
  P = N * H * W
  y = _bf(torch.randn(N, H, W, Cin, generator=gen) * 1.5 + 0.4).to(dev)
  res = _bf(torch.randn(N, H, W, Cin, generator=gen)).to(dev) if residual else None
  w = (torch.randn(K, K, Cin, Cout, generator=gen) * (1.0 / math.sqrt(K * K * Cin))).to(dev)
  gamma = (torch.rand(Cin, generator=gen) + 0.5).to(dev)
  gamma[::5] *= -1
  beta = (torch.randn(Cin, generator=gen) * 0.3).to(dev)
  R = 32
  # the producer's epilogue: sums spread over replica rows
  flat = y.float().reshape(P, Cin)
  rows = torch.arange(P, device=dev) % R
  sums = torch.zeros(R, 2, Cin, device=dev)
  sums[:, 0].index_add_(0, rows, flat)
  sums[:, 1].index_add_(0, rows, flat * flat)
  mm = torch.full((Cin,), 0.25, device=dev)
  mv = torch.full((Cin,), 2.0, device=dev)
  saved = torch.zeros(4 * Cin, device=dev)
  z = torch.full((N, H, W, Cin), float("nan"), dtype=torch.bfloat16, device=dev)
  mask = torch.zeros(P * Cin // 8, dtype=torch.uint8, device=dev)
  out = torch.full(g.output_shape, float("nan"), dtype=torch.bfloat16, device=dev)
  ostats = torch.zeros(R, 2, Cout, device=dev)
  wf = _pack(native, g, w, "fwd")
  d = g.fwd(y.data_ptr(), wf.data_ptr(), out.data_ptr(), cfg=cfg,
            stats=ostats.data_ptr() if Cout % 8 == 0 else None, stats_replicas=R)
  d.update(pro_sums=sums.data_ptr(), pro_replicas=R, pro_gamma=gamma.data_ptr(),
           pro_beta=beta.data_ptr(), pro_mm=mm.data_ptr(), pro_mv=mv.data_ptr(),
           pro_save_mean=saved.data_ptr(), pro_save_invstd=saved.data_ptr() + 4 * Cin,
           pro_scale=saved.data_ptr() + 8 * Cin, pro_shift=saved.data_ptr() + 12 * Cin,
           pro_z=z.data_ptr(), pro_mask=mask.data_ptr() if relu else None, pro_P=P, pro_eps=1e-5,
           pro_momentum=0.9, pro_relu=relu, pro_res=res.data_ptr() if residual else None)
  BM, BN = FWD_CFGS[cfg]
  fits = BM % W == 0 and (H * W) % BM == 0 and (BM // W + K - 1) * (W + K - 1) * Cin * 2 <= 16384 \
      and -(-K * K * Cin // 32) * BN * 64 <= 40960
  if not fits:
    with pytest.raises(ValueError):
      native.run_op("conv", d)
    return
  native.run_op("conv", d)
  torch.cuda.synchronize()
  mean, var = flat.mean(0), flat.var(0, unbiased=False)
  invstd = torch.rsqrt(var + 1e-5)
  zr = (y.float() - mean) * invstd * gamma + beta
  if residual:
    zr = zr + res.float()
  if relu:
    zr = torch.relu(zr)
  assert torch.allclose(saved[:Cin], mean, rtol=1e-3, atol=1e-3)
  assert torch.allclose(saved[Cin:2 * Cin], invstd, rtol=2e-3, atol=1e-3)
  assert torch.allclose(mm, 0.25 * 0.9 + 0.1 * mean, rtol=1e-3, atol=1e-3)
  assert torch.allclose(mv, 2.0 * 0.9 + 0.1 * var * P / (P - 1), rtol=2e-3, atol=1e-3)
  assert torch.isfinite(z.float()).all()
  assert torch.allclose(z.float(), zr, rtol=2e-2, atol=3e-2)
  if relu:
    bits = ((mask.view(P, Cin // 8, 1).to(torch.int32) >> torch.arange(8, device=dev)) & 1).bool()
    assert torch.equal(bits.view(N, H, W, Cin), z > 0)
  # the convolution consumes exactly the tensor it wrote
  ref = _ref_conv(z.float(), _bf(w).float(), 1, pad)
  err = (out.float()[..., :Cout] - ref).abs().max().item()
  assert err <= 0.02 * ref.abs().max().item() + 1e-2, err
  if Cout % 8 == 0:
    o = out.float().reshape(P, -1)[:, :Cout]
    assert torch.allclose(ostats[:, 0].sum(0), o.sum(0), rtol=1e-2, atol=0.5)
    assert torch.allclose(ostats[:, 1].sum(0), (o * o).sum(0), rtol=1e-2, atol=0.5)


@pytest.mark.parametrize("N,HW,C", [(128, 64, 64), (5, 49, 2048), (3, 10, 24), (2, 7, 4096), (4, 1, 8)])
def test_global_mean_shapes(N, HW, C):
  """Global average pool (reference `resnet_cifar_model.py:235`, `resnet_model.py:372`): narrow
  tensors spread their pixels over thread lanes, wide ones their channels; more than 2048
  channels take several chunk groups."""
  native = _native()
  dev = torch.device("cuda")
  gen = torch.Generator().manual_seed(N + HW + C)
  x = _bf(torch.randn(N, HW, C, generator=gen)).to(dev)
  m = torch.full((N, C), float("nan"), dtype=torch.bfloat16, device=dev)
  native.run_op("mean_fwd", dict(x=x.data_ptr(), out=m.data_ptr(), N=N, HW=HW, C=C))
  torch.cuda.synchronize()
  assert torch.allclose(m.float(), x.float().mean(1), rtol=1e-2, atol=1e-2)


@pytest.mark.parametrize("cfg", list(range(40, 48)))
@pytest.mark.parametrize("case", [
    # N, H, W, Cin, K, Cout, relu
    (3, 32, 32, 16, 3, 16, 1), (5, 16, 16, 32, 3, 32, 1), (3, 8, 8, 64, 3, 64, 0),
    (2, 16, 8, 24, 3, 16, 1), (2, 8, 16, 40, 1, 32, 1),
])
def test_dgrad_with_batchnorm_backward_prologue(case, cfg):
  """BatchNorm backward folded into the data gradient of the layer in front of it (reference
  layer pairs `resnet_cifar_model.py:75-92`): the launch takes dz (gradient of the BatchNorm
  output), the layer's output y, the ReLU mask and the replica sums (sum g, sum g*y), and must
  produce what the separate sweep + data gradient produce: dgamma, dbeta, dy (written on the side)
  and dX = conv^T(dy)."""
  from distributed_tensorflow_amd.ops.conv_plan import FWD_CFGS
  native = _native()
  N, H, W, Cin, K, Cout, relu = case
  dev = torch.device("cuda")
  gen = torch.Generator(device="cpu").manual_seed(hash(case) & 0xffff)
  pad = same_padding(H, K, 1) + same_padding(W, K, 1)
  g = ConvGeom(N, H, W, Cin, K, K, Cout, 1, pad)
  P = N * H * W
  y = _bf(torch.randn(N, H, W, Cout, generator=gen) * 1.3 + 0.2).to(dev)      # the layer's output
  dz = _bf(torch.randn(N, H, W, Cout, generator=gen)).to(dev)
  bits = (torch.rand(N, H, W, Cout, generator=gen) > 0.4).to(dev) if relu else None
  gamma = (torch.rand(Cout, generator=gen) + 0.5).to(dev)
  w = (torch.randn(K, K, Cin, Cout, generator=gen) * (1.0 / math.sqrt(K * K * Cin))).to(dev)
  flat = y.float().reshape(P, Cout)
  mean, var = flat.mean(0), flat.var(0, unbiased=False)
  invstd = torch.rsqrt(var + 1e-5)
  gm = dz.float() * bits if relu else dz.float()
  gf = gm.reshape(P, Cout)
  R = 32
  rows = torch.arange(P, device=dev) % R
  sums = torch.zeros(R, 2, Cout, device=dev)
  sums[:, 0].index_add_(0, rows, gf)
  sums[:, 1].index_add_(0, rows, gf * flat)
  mask = None
  if relu:
    mask = (bits.view(P, Cout // 8, 8).to(torch.int32) << torch.arange(8, device=dev)).sum(-1).to(torch.uint8)
    mask = mask.flatten().contiguous()
  saved = torch.cat([mean, invstd]).contiguous()
  dgamma = torch.zeros(Cout, device=dev)
  dbeta = torch.zeros(Cout, device=dev)
  dy = torch.full((N, H, W, Cout), float("nan"), dtype=torch.bfloat16, device=dev)
  dx = torch.full((N, H, W, Cin), float("nan"), dtype=torch.bfloat16, device=dev)
  wd = _pack(native, g, w, "dgrad")
  descs = g.dgrad(dz.data_ptr(), wd.data_ptr(), dx.data_ptr(), cfg=cfg)
  assert len(descs) == 1
  d = descs[0]
  d.update(bpro_sums=sums.data_ptr(), bpro_replicas=R, bpro_y=y.data_ptr(),
           bpro_mask=mask.data_ptr() if relu else None, bpro_mean=saved.data_ptr(),
           bpro_invstd=saved.data_ptr() + 4 * Cout, bpro_gamma=gamma.data_ptr(),
           bpro_dgamma=dgamma.data_ptr(), bpro_dbeta=dbeta.data_ptr(), bpro_dy=dy.data_ptr(), bpro_P=P)
  BM, BN = FWD_CFGS[cfg]
  fits = BM % W == 0 and (H * W) % BM == 0 and (BM // W + K - 1) * (W + K - 1) * Cout * 2 <= 16384 \
      and -(-K * K * Cout // 32) * BN * 64 <= 40960
  if not fits:
    with pytest.raises(ValueError):
      native.run_op("conv", d)
    return
  native.run_op("conv", d)
  torch.cuda.synchronize()
  xhat = (flat - mean) * invstd
  dbeta_ref = gf.sum(0)
  dgamma_ref = (gf * xhat).sum(0)
  dy_ref = (gamma * invstd * (gf - dbeta_ref / P - xhat * dgamma_ref / P)).view(N, H, W, Cout)
  assert torch.allclose(dbeta, dbeta_ref, rtol=1e-2, atol=5e-2)
  assert torch.allclose(dgamma, dgamma_ref, rtol=1e-2, atol=5e-2)
  assert torch.isfinite(dy.float()).all()
  assert torch.allclose(dy.float(), dy_ref, rtol=3e-2, atol=3e-2)
  # the data gradient consumes exactly the dy it wrote
  xr = torch.zeros(N, H, W, Cin, device=dev, requires_grad=True)
  _ref_conv(xr, _bf(w).float(), 1, pad).backward(dy.float())
  err = (dx.float() - xr.grad).abs().max().item()
  assert err <= 0.02 * xr.grad.abs().max().item() + 1e-2, err


@pytest.mark.parametrize("case,cfg", [
    ((3, 56, 56, 64, 3, 64, 1), 48), ((2, 56, 56, 64, 3, 128, 1), 50), ((1, 56, 56, 64, 3, 64, 1), 50),
    ((3, 28, 28, 128, 3, 128, 1), 49), ((2, 28, 28, 128, 3, 256, 1), 51), ((5, 28, 28, 128, 3, 128, 1), 51),
    ((2, 56, 56, 64, 3, 64, 1), 53), ((61, 56, 56, 64, 3, 64, 1), 52)])
def test_conv_whole_rows_kernel(case, cfg):
  """conv_rows.hip (tile configurations 48..53: the large 3x3 unit-stride layers over whole
  image rows, every input pixel fetched once into a halo tile): forward, data gradient and
  accumulating data gradient against the PyTorch fp32 convolution - and the launches really are
  served by that kernel, not by a fallback tile."""
  native = _native()
  N, H, W, Cin, K, Cout, stride = case
  pad = same_padding(H, K, stride) + same_padding(W, K, stride)
  g = ConvGeom(N, H, W, Cin, K, K, Cout, stride, pad)
  probe = torch.zeros(16, device="cuda")
  fd = g.fwd(probe.data_ptr(), probe.data_ptr(), probe.data_ptr(), cfg=cfg)
  assert native.ext().conv_rows_serves(fd)
  for d in g.dgrad(probe.data_ptr(), probe.data_ptr(), probe.data_ptr(), cfg=cfg):
    assert native.ext().conv_rows_serves(d) == (Cout == Cin)     # its data gradient has Cin outputs
  assert not native.ext().conv_rows_serves(dict(fd, cfg=49 if Cin == 64 else 48))  # the other layer's tile
  _check_conv_cfg(case, cfg)


@pytest.mark.parametrize("N,dtype,HW,C", [(3, torch.bfloat16, 56, 64), (1, torch.bfloat16, 56, 64),
                                          (5, torch.float16, 56, 64), (3, torch.bfloat16, 28, 128),
                                          (37, torch.float16, 28, 128)])
def test_wgrad_whole_rows_kernel(N, dtype, HW, C):
  """conv_wgrad_rows.hip (cfg 192: the halo tile of X and the tile of dY of four output rows fetched
  once; a block owns the whole gradient of 64 input x 64 output channels: all of 56x56x64, a quarter of 28x28x128)
  against the PyTorch fp32 convolution's weight gradient; on top of a gradient that is already there
  (it accumulates); and a layer it cannot serve falls back to a tile of the general kernel."""
  native = _native()
  dev = torch.device("cuda")
  H = W = HW
  Cin = Cout = C
  gen = torch.Generator(device="cpu").manual_seed(N)
  pad = same_padding(H, 3, 1) + same_padding(W, 3, 1)
  g = ConvGeom(N, H, W, Cin, 3, 3, Cout, 1, pad)
  x = torch.randn(N, H, W, Cin, generator=gen).to(dtype).to(dev)
  dy = torch.randn(N, H, W, Cout, generator=gen).to(dtype).to(dev)
  wr = torch.zeros(3, 3, Cin, Cout, device=dev).requires_grad_(True)
  _ref_conv(x.float(), wr, 1, pad).backward(dy.float())
  before = torch.randn(3, 3, Cin, Cout, generator=gen).to(dev)
  dw = before.clone()
  d = dict(g.wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), cfg=192),
           dt={torch.bfloat16: 0, torch.float16: 1}[dtype])
  assert native.ext().wgrad_rows_serves(d)
  native.run_op("wgrad", d)                # partial gradients to the workspace, summed by a second launch
  torch.cuda.synchronize()
  err = (dw - before - wr.grad).abs().max().item()
  assert err <= 2e-3 * wr.grad.abs().max().item() + 1e-3, err
  # one split = the deterministic mode: a fixed summation order, bit for bit the same twice
  once, twice = torch.zeros_like(dw), torch.zeros_like(dw)
  for out in (once, twice):
    native.run_op("wgrad", dict(d, dw=out.data_ptr(), splits=1, rev=1))
  torch.cuda.synchronize()
  assert torch.equal(once, twice)
  assert (once - wr.grad).abs().max().item() <= 2e-3 * wr.grad.abs().max().item() + 1e-3
  native.run_op("wgrad", dict(d, atomic_flush=1))      # every block adds to the gradient itself
  torch.cuda.synchronize()
  err = (dw - before - 2 * wr.grad).abs().max().item()
  assert err <= 4e-3 * wr.grad.abs().max().item() + 2e-3, err
  # 14x14x256 is not this kernel's: same cfg, served by the general kernel
  g2 = ConvGeom(2, 14, 14, 256, 3, 3, 256, 1, same_padding(14, 3, 1) + same_padding(14, 3, 1))
  x2 = _bf(torch.randn(2, 14, 14, 256, generator=gen)).to(dev)
  dy2 = _bf(torch.randn(2, 14, 14, 256, generator=gen)).to(dev)
  wr2 = torch.zeros(3, 3, 256, 256, device=dev).requires_grad_(True)
  _ref_conv(x2.float(), wr2, 1, same_padding(14, 3, 1) + same_padding(14, 3, 1)).backward(dy2.float())
  dw2 = torch.zeros(3, 3, 256, 256, device=dev)
  d2 = g2.wgrad(x2.data_ptr(), dy2.data_ptr(), dw2.data_ptr(), cfg=192)
  assert not native.ext().wgrad_rows_serves(d2)
  native.run_op("wgrad", d2)
  torch.cuda.synchronize()
  assert (dw2 - wr2.grad).abs().max().item() <= 0.01 * wr2.grad.abs().max().item() + 1e-2


@pytest.mark.parametrize("N,dtype", [(3, torch.bfloat16), (2, torch.float16)])
def test_stem_whole_rows_kernel(N, dtype):
  """conv_rows.hip cfg 54 (the 7x7 stride-2 ImageNet stem over pairs of output rows, filter bank
  resident in LDS) against the PyTorch fp32 convolution, with the fused BatchNorm statistics of its
  output; a stem of another width is handed to the pipeline tile."""
  native = _native()
  dev = torch.device("cuda")
  gen = torch.Generator().manual_seed(N)
  H = 224
  dti = {torch.bfloat16: 0, torch.float16: 1}[dtype]
  g = ConvGeom(N, H, H, 3, 7, 7, 64, 2, (3, 3, 3, 3), mode="merged", dt=dti)
  img = torch.randn(N, H, H, 3, generator=gen).to(dev)
  xin = torch.zeros(g.input_shape, dtype=dtype, device=dev)
  native.run_op("preprocess_pad", dict(N=N, H=H, W=H, Cin=3, pt=3, pl=3, OH=g.Hs, OW=g.Ws, Cpad=4, dt=dti,
                                       out=xin.data_ptr(), **{"in": img.data_ptr()}))
  w = (torch.randn(7, 7, 3, 64, generator=gen) * 0.05).to(dev)
  pd = dict(g.pack_fwd(), src_off=0, dst_off=0, first_block=0)
  wf = torch.zeros(round_up(pd["count"], 8), dtype=dtype, device=dev)
  desc = torch.frombuffer(bytearray(native.pack_desc_bytes([pd])), dtype=torch.uint8).to(dev)
  native.run_op("pack_weights", dict(params=w.contiguous().data_ptr(), packed=wf.data_ptr(), descs=desc.data_ptr(),
                                     ndesc=1, nblocks=native.pack_blocks(pd), dt=dti))
  out = torch.full(g.output_shape, float("nan"), dtype=dtype, device=dev)
  stats = torch.zeros(32, 2, 64, device=dev)
  d = g.fwd(xin.data_ptr(), wf.data_ptr(), out.data_ptr(), cfg=54, stats=stats.data_ptr(), stats_replicas=32)
  assert native.ext().conv_rows_serves(d)
  native.run_op("conv", d)
  torch.cuda.synchronize()
  xr = xin[:, 3:3 + H, 3:3 + H, :3].float()
  ref = _ref_conv(xr, w.to(dtype).float(), 2, (3, 3, 3, 3))
  assert torch.isfinite(out.float()).all()
  assert (out.float() - ref).abs().max().item() <= 0.02 * ref.abs().max().item()
  got = stats.sum(0)
  o = out.float().reshape(-1, 64)
  assert torch.allclose(got[0], o.sum(0), rtol=1e-3, atol=1e-1)
  assert torch.allclose(got[1], (o * o).sum(0), rtol=1e-3, atol=1e-1)
  # the same launch with the tensors walked back to front
  out2 = torch.zeros_like(out)
  native.run_op("conv", dict(d, out=out2.data_ptr(), rev=1, stats=None))
  torch.cuda.synchronize()
  assert torch.equal(out2, out)
  small = ConvGeom(N, 64, 64, 3, 7, 7, 64, 2, (3, 3, 3, 3), mode="merged", dt=dti)
  probe = torch.zeros(16, device=dev)
  assert not native.ext().conv_rows_serves(small.fwd(probe.data_ptr(), probe.data_ptr(), probe.data_ptr(), cfg=54))


@pytest.mark.parametrize("N,H,W,dtype", [(3, 224, 224, torch.bfloat16), (2, 32, 30, torch.float16), (1, 9, 7, torch.bfloat16)])
def test_preprocess_pad_rgb(N, H, W, dtype):
  """Input staging: fp32 RGB images to the 16-bit 4-channel tensor with its zero border (the pair-of-pixels
  kernel for an even padded width, the per-pixel kernel otherwise), element for element."""
  native = _native()
  dev = torch.device("cuda")
  gen = torch.Generator().manual_seed(N * H)
  img = (torch.randn(N, H, W, 3, generator=gen) * 60 + 100).to(dev)
  pt, pl = 3, 3
  OH, OW = H + 6, W + (6 if W % 2 == 0 else 5)
  dti = {torch.bfloat16: 0, torch.float16: 1}[dtype]
  out = torch.full((N, OH, OW, 4), float("nan"), dtype=dtype, device=dev)
  native.run_op("preprocess_pad", dict(N=N, H=H, W=W, Cin=3, pt=pt, pl=pl, OH=OH, OW=OW, Cpad=4, dt=dti,
                                       out=out.data_ptr(), **{"in": img.data_ptr()}))
  torch.cuda.synchronize()
  want = torch.zeros(N, OH, OW, 4, dtype=dtype, device=dev)
  want[:, pt:pt + H, pl:pl + W, :3] = img.to(dtype)
  assert torch.equal(out, want)


@pytest.mark.parametrize("N,dtype", [(2, torch.bfloat16), (5, torch.float16), (19, torch.bfloat16)])
def test_stem_weight_gradient_whole_rows_kernel(N, dtype):
  """conv_wgrad_rows.hip on the 7x7 stride-2 stem (merged-tap form, cfg 192: nine input rows and two
  rows of dY in LDS, every window read out of them) against the PyTorch fp32 weight gradient and against
  the general kernel; it accumulates; a stem of another width goes to the general kernel."""
  native = _native()
  dev = torch.device("cuda")
  gen = torch.Generator().manual_seed(N)
  H = 224
  dti = {torch.bfloat16: 0, torch.float16: 1}[dtype]
  g = ConvGeom(N, H, H, 3, 7, 7, 64, 2, (3, 3, 3, 3), mode="merged", dt=dti)
  img = torch.randn(N, H, H, 3, generator=gen).to(dev)
  xin = torch.zeros(g.input_shape, dtype=dtype, device=dev)
  native.run_op("preprocess_pad", dict(N=N, H=H, W=H, Cin=3, pt=3, pl=3, OH=g.Hs, OW=g.Ws, Cpad=4, dt=dti,
                                       out=xin.data_ptr(), **{"in": img.data_ptr()}))
  dy = torch.randn(N, 112, 112, 64, generator=gen).to(dtype).to(dev)
  xr = xin[:, 3:3 + H, 3:3 + H, :3].float()
  wr = torch.zeros(7, 7, 3, 64, device=dev).requires_grad_(True)
  _ref_conv(xr, wr, 2, (3, 3, 3, 3)).backward(dy.float())
  before = torch.randn(7, 7, 3, 64, generator=gen).to(dev)
  dw = before.clone()
  d = g.wgrad(xin.data_ptr(), dy.data_ptr(), dw.data_ptr(), cfg=192)
  assert native.ext().wgrad_rows_serves(d)
  native.run_op("wgrad", d)
  torch.cuda.synchronize()
  scale = wr.grad.abs().max().item()
  assert (dw - before - wr.grad).abs().max().item() <= 2e-3 * scale + 1e-3
  dw2 = torch.zeros_like(dw)
  native.run_op("wgrad", g.wgrad(xin.data_ptr(), dy.data_ptr(), dw2.data_ptr(), cfg=1, splits=16))
  native.run_op("wgrad", dict(d, dw=dw.data_ptr(), rev=1))          # back to front, on top
  torch.cuda.synchronize()
  assert (dw2 - wr.grad).abs().max().item() <= 2e-3 * scale + 1e-3
  assert (dw - before - 2 * wr.grad).abs().max().item() <= 4e-3 * scale + 2e-3
  small = ConvGeom(N, 64, 64, 3, 7, 7, 64, 2, (3, 3, 3, 3), mode="merged", dt=dti)
  probe = torch.zeros(16, device=dev)
  assert not native.ext().wgrad_rows_serves(small.wgrad(probe.data_ptr(), probe.data_ptr(), probe.data_ptr(), cfg=192))
  # one split = a fixed summation order: not this kernel's (its blocks add with atomics); the general
  # kernel serves the launch and repeats bit for bit
  assert not native.ext().wgrad_rows_serves(dict(d, splits=1))
  once, twice = torch.zeros_like(dw), torch.zeros_like(dw)
  for out in (once, twice):
    native.run_op("wgrad", dict(d, dw=out.data_ptr(), splits=1))
  torch.cuda.synchronize()
  assert torch.equal(once, twice)
  assert (once - wr.grad).abs().max().item() <= 2e-3 * scale + 1e-3


@pytest.mark.parametrize("HW,C,cfgs", [(56, 64, (52, 53)), (28, 128, (49, 51))])
def test_whole_rows_kernels_equal_the_general_kernel_at_a_large_batch(HW, C, cfgs):
  """Every flavour of the whole-rows kernels against a pipeline tile at a batch with more tiles than
  resident blocks (timing-dependent faults need many blocks in flight): forward + statistics, data gradient plain / with backward statistics / with an addend and
  backward statistics, with an addend alone (the flavour whose sums the compiler once paired into a
  sequence that lost elements at large batches)."""
  native = _native()
  dev = torch.device("cuda")
  N = 72
  g = ConvGeom(N, HW, HW, C, 3, 3, C, 1, (1, 1, 1, 1))
  gen = torch.Generator(device="cpu").manual_seed(HW)
  x = _bf(torch.randn(N, HW, HW, C, generator=gen)).to(dev)
  w = (torch.randn(3, 3, C, C, generator=gen) / math.sqrt(9 * C)).to(dev)
  wf, wd = _pack(native, g, w, "fwd"), _pack(native, g, w, "dgrad")
  y = _bf(torch.randn(N, HW, HW, C, generator=gen) * 1.5 + 0.7).to(dev)
  add = _bf(torch.randn(N, HW, HW, C, generator=gen)).to(dev)
  bits = torch.rand(N * HW * HW, C, generator=gen) > 0.4
  mask = (bits.view(-1, C // 8, 8).to(torch.int32) << torch.arange(8)).sum(-1).to(torch.uint8)
  mask = torch.cat([mask.flatten(), torch.zeros(4, dtype=torch.uint8)]).to(dev)

  def run(cfg, kind, addend, stats):
    out = torch.zeros(N, HW, HW, C, dtype=torch.bfloat16, device=dev)
    st = torch.zeros(32, 2, C, device=dev)
    if kind == "fwd":
      descs = [g.fwd(x.data_ptr(), wf.data_ptr(), out.data_ptr(), cfg=cfg, stats=st.data_ptr(), stats_replicas=32)]
    else:
      descs = g.dgrad(x.data_ptr(), wd.data_ptr(), out.data_ptr(), cfg=cfg, addend=add.data_ptr() if addend else None)
      if stats:
        for d in descs:
          d.update(stats=st.data_ptr(), stats_replicas=32, stats_y=y.data_ptr(), stats_mask=mask.data_ptr())
    for d in descs:
      native.run_op("conv", d)
    torch.cuda.synchronize()
    return out, st.sum(0)

  for kind, addend, stats in (("fwd", False, True), ("dgrad", False, False), ("dgrad", False, True),
                              ("dgrad", True, True), ("dgrad", True, False)):
    ref, ref_st = run(30, kind, addend, stats)
    for cfg in cfgs:
      probe = (g.fwd(1, 2, 3, cfg=cfg) if kind == "fwd" else g.dgrad(1, 2, 3, cfg=cfg, addend=4 if addend else None)[0])
      if stats and kind == "dgrad":
        probe = dict(probe, stats=5, stats_y=6)
      assert native.ext().conv_rows_serves(probe), (cfg, kind, addend, stats)
      for _ in range(2):
        out, st = run(cfg, kind, addend, stats)
        tol = 2.0 ** -6 * ref.float().abs() + 2e-2
        assert int(((out.float() - ref.float()).abs() > tol).sum()) == 0, (cfg, kind, addend, stats)
        assert torch.allclose(st, ref_st, rtol=1e-3, atol=1e-3 * ref_st.abs().max().item() + 1e-2)



def _affine_relu(x, scale, shift):
  """What the BatchNorm apply sweep would have written: relu(fma(x, scale, shift)) in fp32, rounded
  (the product is not rounded on its own: computed in double and rounded once)."""
  return torch.relu((x.double() * scale.double() + shift.double()).float()).to(x.dtype)


@pytest.mark.parametrize("case,cfg", [((3, 56, 56, 64, 3, 64, 1), 52), ((2, 56, 56, 64, 3, 64, 1), 48),
                                      ((3, 28, 28, 128, 3, 128, 1), 49), ((2, 28, 28, 128, 3, 128, 1), 51),
                                      ((2, 56, 56, 64, 3, 64, 1), 53), ((59, 56, 56, 64, 3, 64, 1), 52),
                                      ((79, 28, 28, 128, 3, 128, 1), 49)])
def test_whole_rows_kernels_normalise_their_input_inside_the_tile(case, cfg):
  """conv_rows.hip with `in_scale` / `in_shift`: BatchNorm + ReLU of the layer in front applied to the halo
  tile in LDS.  The result equals, bit for bit, the same launch on a tensor normalised beforehand
  (padding pixels stay zero, not relu(shift)); other tiles refuse such a descriptor."""
  native = _native()
  N, H, W, Cin, K, Cout, stride = case
  dev = torch.device("cuda")
  gen = torch.Generator(device="cpu").manual_seed(cfg)
  pad = same_padding(H, K, stride) + same_padding(W, K, stride)
  g = ConvGeom(N, H, W, Cin, K, K, Cout, stride, pad)
  y = _bf(torch.randn(N, H, W, Cin, generator=gen)).to(dev)
  scale = (torch.rand(Cin, generator=gen) + 0.5).to(dev)
  shift = (torch.randn(Cin, generator=gen) * 0.5 + 0.3).to(dev)     # mostly positive: relu(shift) != 0
  a = _affine_relu(y, scale, shift)
  w = (torch.randn(K, K, Cin, Cout, generator=gen) * 0.05).to(dev)
  wf = _pack(native, g, w, "fwd")
  out_ref = torch.zeros(g.output_shape, dtype=torch.bfloat16, device=dev)
  out = torch.zeros_like(out_ref)
  stats_ref = torch.zeros(32, 2, Cout, device=dev)
  stats = torch.zeros_like(stats_ref)
  native.run_op("conv", g.fwd(a.data_ptr(), wf.data_ptr(), out_ref.data_ptr(), cfg=cfg,
                              stats=stats_ref.data_ptr(), stats_replicas=32))
  d = dict(g.fwd(y.data_ptr(), wf.data_ptr(), out.data_ptr(), cfg=cfg, stats=stats.data_ptr(), stats_replicas=32),
           in_scale=scale.data_ptr(), in_shift=shift.data_ptr())
  native.run_op("conv", d)
  torch.cuda.synchronize()
  assert torch.equal(out, out_ref)
  assert torch.allclose(stats.sum(0), stats_ref.sum(0), rtol=1e-4, atol=1e-2)
  ref = _ref_conv(a.float(), _bf(w).float(), stride, pad)
  assert (out.float() - ref).abs().max().item() <= 0.02 * ref.abs().max().item() + 1e-2
  with pytest.raises(ValueError):
    native.run_op("conv", dict(d, cfg=30))


@pytest.mark.parametrize("N,HW,C", [(3, 56, 64), (2, 56, 64), (5, 28, 128)])
def test_whole_rows_weight_gradient_normalises_x_inside_the_tile(N, HW, C):
  """conv_wgrad_rows.hip with `x_scale` / `x_shift` against the same launch on a normalised X."""
  native = _native()
  dev = torch.device("cuda")
  gen = torch.Generator(device="cpu").manual_seed(N)
  g = ConvGeom(N, HW, HW, C, 3, 3, C, 1, (1, 1, 1, 1))
  y = _bf(torch.randn(N, HW, HW, C, generator=gen)).to(dev)
  scale = (torch.rand(C, generator=gen) + 0.5).to(dev)
  shift = (torch.randn(C, generator=gen) * 0.5 + 0.3).to(dev)
  a = _affine_relu(y, scale, shift)
  dy = _bf(torch.randn(N, HW, HW, C, generator=gen)).to(dev)
  dw_ref = torch.zeros(3, 3, C, C, device=dev)
  dw = torch.zeros_like(dw_ref)
  native.run_op("wgrad", g.wgrad(a.data_ptr(), dy.data_ptr(), dw_ref.data_ptr(), cfg=192, splits=1))
  d = dict(g.wgrad(y.data_ptr(), dy.data_ptr(), dw.data_ptr(), cfg=192, splits=1),
           x_scale=scale.data_ptr(), x_shift=shift.data_ptr())
  native.run_op("wgrad", d)
  torch.cuda.synchronize()
  assert torch.equal(dw, dw_ref)               # one split: a fixed summation order in both launches
  wr = torch.zeros(3, 3, C, C, device=dev).requires_grad_(True)
  _ref_conv(a.float(), wr, 1, (1, 1, 1, 1)).backward(dy.float())
  assert (dw - wr.grad).abs().max().item() <= 2e-3 * wr.grad.abs().max().item() + 1e-3
  with pytest.raises(ValueError):
    native.run_op("wgrad", dict(d, cfg=1))


@pytest.mark.parametrize("case,cfg", [((3, 56, 56, 64, 256), 17), ((3, 56, 56, 64, 256), 31), ((2, 28, 28, 128, 512), 30),
                                      ((5, 14, 14, 256, 1024), 20), ((7, 7, 7, 512, 2048), 18), ((2, 28, 28, 96, 40), 16),
                                      ((33, 56, 56, 64, 256), 23)])
def test_pipeline_kernel_normalises_the_input_of_a_pointwise_layer_on_its_fragments(case, cfg):
  """conv_pipe.hip with `in_scale` / `in_shift` on a 1x1 layer: BatchNorm + ReLU of the layer in front applied
  to the pixel fragments between the LDS read and the matrix instructions.  Bit for bit the same launch on a
  tensor normalised beforehand, statistics included; a ragged K (96 channels) and a ragged pixel count too."""
  native = _native()
  N, H, W, Cin, Cout = case
  tdt = torch.bfloat16
  dev = torch.device("cuda")
  gen = torch.Generator(device="cpu").manual_seed(cfg + Cin)
  g = ConvGeom(N, H, W, Cin, 1, 1, Cout, 1, (0, 0, 0, 0))
  y = torch.randn(N, H, W, Cin, generator=gen).to(tdt).to(dev)
  scale = (torch.rand(Cin, generator=gen) + 0.5).to(dev)
  shift = (torch.randn(Cin, generator=gen) * 0.5 + 0.3).to(dev)
  a = _affine_relu(y, scale, shift)
  w = (torch.randn(1, 1, Cin, Cout, generator=gen) * 0.05).to(dev)
  wf = _pack(native, g, w, "fwd")
  out_ref = torch.zeros(g.output_shape, dtype=tdt, device=dev)
  out = torch.zeros_like(out_ref)
  stats_ref = torch.zeros(32, 2, Cout, device=dev)
  stats = torch.zeros_like(stats_ref)
  native.run_op("conv", g.fwd(a.data_ptr(), wf.data_ptr(), out_ref.data_ptr(), cfg=cfg,
                              stats=stats_ref.data_ptr(), stats_replicas=32))
  d = dict(g.fwd(y.data_ptr(), wf.data_ptr(), out.data_ptr(), cfg=cfg, stats=stats.data_ptr(), stats_replicas=32),
           in_scale=scale.data_ptr(), in_shift=shift.data_ptr())
  native.run_op("conv", d)
  torch.cuda.synchronize()
  assert torch.equal(out, out_ref)
  assert torch.allclose(stats.sum(0), stats_ref.sum(0), rtol=1e-4, atol=1e-2)
  ref = _ref_conv(a.float(), w.to(tdt).float(), 1, (0, 0, 0, 0))
  assert (out[..., :Cout].float() - ref).abs().max().item() <= 0.02 * ref.abs().max().item() + 1e-2
  with pytest.raises(ValueError):
    native.run_op("conv", dict(d, cfg=0))      # the two-stage kernel has no such operand path


@pytest.mark.parametrize("N,HW,Cin,Cout,cfg", [(3, 56, 64, 256, 1), (3, 56, 64, 256, 64 + 17), (2, 28, 128, 512, 0),
                                               (5, 14, 256, 1024, 128 + 0), (3, 7, 512, 2048, 16 + 8),
                                               (2, 28, 96, 40, 2)])
def test_weight_gradient_normalises_x_of_a_pointwise_layer_on_its_fragments(N, HW, Cin, Cout, cfg):
  """conv_wgrad.hip with `x_scale` / `x_shift` on a 1x1 layer against the same launch on a normalised X
  (one split: the same summation order)."""
  native = _native()
  dev = torch.device("cuda")
  gen = torch.Generator(device="cpu").manual_seed(N + cfg)
  g = ConvGeom(N, HW, HW, Cin, 1, 1, Cout, 1, (0, 0, 0, 0))
  y = _bf(torch.randn(N, HW, HW, Cin, generator=gen)).to(dev)
  scale = (torch.rand(Cin, generator=gen) + 0.5).to(dev)
  shift = (torch.randn(Cin, generator=gen) * 0.5 + 0.3).to(dev)
  a = _affine_relu(y, scale, shift)
  dy = _bf(torch.randn(g.output_shape, generator=gen)).to(dev)
  dw_ref = torch.zeros(1, 1, Cin, Cout, device=dev)
  dw = torch.zeros_like(dw_ref)
  native.run_op("wgrad", g.wgrad(a.data_ptr(), dy.data_ptr(), dw_ref.data_ptr(), cfg=cfg, splits=1))
  d = dict(g.wgrad(y.data_ptr(), dy.data_ptr(), dw.data_ptr(), cfg=cfg, splits=1),
           x_scale=scale.data_ptr(), x_shift=shift.data_ptr())
  native.run_op("wgrad", d)
  torch.cuda.synchronize()
  assert torch.equal(dw, dw_ref)
  wr = torch.zeros(1, 1, Cin, Cout, device=dev).requires_grad_(True)
  _ref_conv(a.float(), wr, 1, (0, 0, 0, 0)).backward(dy[..., :Cout].float())
  assert (dw - wr.grad).abs().max().item() <= 2e-3 * wr.grad.abs().max().item() + 1e-3


@pytest.mark.parametrize("P,C", [(70001, 64), (40003, 256)])
def test_batchnorm_backward_takes_its_relu_mask_from_the_recomputed_value(P, C):
  """bn_bwd with `relu_scale` / `relu_shift` (mask = y * scale + shift > 0, for a layer whose forward
  output was never written) equals bn_bwd with the mask bits of the same predicate."""
  native = _native()
  dev = torch.device("cuda")
  gen = torch.Generator(device="cpu").manual_seed(C)
  y = _bf(torch.randn(P, C, generator=gen)).to(dev)
  dz = _bf(torch.randn(P, C, generator=gen)).to(dev)
  gamma = (torch.rand(C, generator=gen) + 0.5).to(dev)
  beta = (torch.randn(C, generator=gen) * 0.2).to(dev)
  mean = y.float().mean(0)
  invstd = torch.rsqrt(y.float().var(0, unbiased=False) + 1e-5)
  saved = torch.cat([mean, invstd, gamma * invstd, beta - mean * gamma * invstd]).contiguous()
  scale, shift = saved[2 * C:3 * C], saved[3 * C:]
  open_ = (y.float() * scale + shift) > 0
  bits = open_.view(P, C // 8, 8).to(torch.int32)
  mask = (bits << torch.arange(8, device=dev)).sum(-1).to(torch.uint8).contiguous()
  ws = torch.zeros(native.ext().bn_workspace_floats(), device=dev)
  outs = []
  for kw in (dict(mask=mask.data_ptr()), dict(mask=None, relu_scale=saved.data_ptr() + 8 * C,
                                              relu_shift=saved.data_ptr() + 12 * C)):
    dgamma, dbeta = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dy = torch.empty(P, C, dtype=torch.bfloat16, device=dev)
    native.run_op("bn_bwd", dict(dz=dz.data_ptr(), y=y.data_ptr(), z=None, mean=saved.data_ptr(),
                                 invstd=saved.data_ptr() + 4 * C, gamma=gamma.data_ptr(), dgamma=dgamma.data_ptr(),
                                 dbeta=dbeta.data_ptr(), dy=dy.data_ptr(), dres=None, ws=ws.data_ptr(), P=P, C=C,
                                 **kw))
    torch.cuda.synchronize()
    outs.append((dgamma, dbeta, dy))
  for a, b in zip(*outs):
    assert torch.equal(a, b)
  g = dz.float() * open_
  assert torch.allclose(outs[1][1], g.sum(0), rtol=1e-2, atol=5e-2)
