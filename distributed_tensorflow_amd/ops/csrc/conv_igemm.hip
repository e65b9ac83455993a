// Implicit-GEMM NHWC convolution on MFMA (forward and data-gradient).
//
//   out[n, oh*OSH+OOH, ow*OSW+OOW, co] (+= addend) (+ bias[co]) =
//       sum over taps (i, j) and channels c of
//       in[n, oh*ISH + dh0 + i*dhs, ow*ISW + dw0 + j*dws, c] * wgt[co][wk0 + i*wsi + j*wsj + c]
//
// One kernel covers: forward convolution with TensorFlow SAME/VALID padding and
// any stride; stride-1 data gradient (taps walked backwards over dY); stride-2
// data gradient as four output-parity sub-problems (each a stride-1 gather with
// a tap subset, written with output stride 2); 1x1 convolutions and the dense
// layer (pure GEMM).  Out-of-image taps read zeros; `Cout` and `Ktotal` may be
// ragged (dense layer: 1001 outputs).
//
// GEMM view: M = N*OH*OW pixels, N = Cout, K = taps*Ctap.  Block tile BM pixels x
// BN channels x BK=32, 4 or 8 waves arranged WAVES_M x WAVES_N (a 128x128 tile moves 16 KiB
// from L2 per 1 MFLOP - 64 FLOP/B, which caps the chip near 1.1 PFLOP/s at the ~18 TB/s the
// L2s deliver; the 8-wave 256x128 tile reaches 96 FLOP/B at the same registers per wave).  The MFMA
// A operand is the *weight* tile and the B operand the *pixel* tile, so that a
// lane ends up holding 4*NR consecutive output channels of one pixel and stores
// them with 16-byte writes, no LDS transpose in the epilogue.
//
// Pipeline: double-buffered direct-to-LDS loads (global_load_lds_dwordx4).  While
// the MFMAs of K-step t run from LDS buffer t&1, the DMA of step t+1 fills the
// other buffer; one barrier (with its vmcnt(0) wait) closes the step.  There is
// no VGPR staging and no ds_write, whose 16-byte form moves only ~79 B/clk/CU and
// made the register-staged version of this kernel LDS-bound.  LDS rows are 64
// bytes (32 bf16 of K); the four 16-byte chunks of a row are XOR-swizzled so that
// the ds_read_b128 fragment reads of a 16-lane group land on 16 distinct 16-byte
// slots of the 256-byte bank row; with DMA writes the swizzle is applied to the
// source address (LDS is filled linearly).
//
// Blocks are numbered so that the BN-tiles of one pixel tile share `blockIdx % 8`
// (the label of the XCD they are dispatched to): they re-read the same input
// rows from that XCD's L2 instead of HBM.
#include <stdexcept>
#include <string>

#include "common.h"
#include "conv_params.h"
#include "conv_epilogue.h"

namespace {

// K-step: one 64-byte LDS row, i.e. 32 elements of a 16-bit type or 16 of fp32
constexpr int bk_of(int dt) { return dt == DT_F32 ? 16 : 32; }

// 16 bytes of zeros in global memory: the source of every masked direct-to-LDS load.
__device__ __attribute__((aligned(16))) uint32_t g_zero_page[4] = {0u, 0u, 0u, 0u};


// Addressing modes of the gather (template parameter MODE):
//   GENERIC    every K-step decodes its k index into (tap i, tap j, channel) per lane - needed
//              when a 32-wide K-step straddles taps (Ctap not a multiple of 32: CIFAR stages,
//              ragged K);
//   POINTWISE  1x1 / dense: a plain GEMM, the pixel index is the row;
//   TAPLOOP    Ctap % 32 == 0: a K-step never leaves its tap, so bounds checks and address
//              arithmetic run once per tap and a K-step only bumps one 64-bit pointer per
//              load (about 12 VALU instructions per step instead of 60: with 16 MFMAs per step
//              and four waves per SIMD the generic loop is VALU-issue bound).
constexpr int GENERIC = 0, POINTWISE_MODE = 1, TAPLOOP = 2;

// Second launch-bound argument = waves per SIMD the register allocation must allow: 4 blocks
// of 4 waves per CU (64 accumulator registers + ~60 others fit the 128-register budget).
// EPI selects the epilogue flavour at compile time, so that the extra loads and registers of the
// special ones stay out of the training-forward instantiations:
//   EPI_PLAIN   bias / addend / forward BatchNorm statistics (runtime options)
//   EPI_BSTATS  backward flavour of the fused statistics (sum g, sum g*y; see ConvParams)
//   EPI_INFER   folded BatchNorm for inference: acc * scale + bias (+ addend), optional ReLU
using conv_epi::EPI_PLAIN;
using conv_epi::EPI_ADD;
using conv_epi::EPI_BSTATS;
using conv_epi::EPI_INFER;

// DT: element type of activations and packed weights (common.h).  An LDS tile row is 64 bytes in
// every case - a K-step of 32 elements for the 16-bit types, 16 for fp32 - so the DMA, the swizzle
// and the fragment reads (one ds_read_b128 per row and lane group) are the same instructions;
// what differs is the matrix instruction: v_mfma_f32_16x16x32_{bf16,f16} takes the 8 elements a
// lane read as its 8 consecutive K values, fp32 runs four v_mfma_f32_16x16x4_f32 and feeds element
// e of the lane's chunk to step e (K index 4 * (lane / 16) + e for both operands, so the sum
// over a row's 16 elements is complete).  fp32 in, fp32 accumulate: exact fp32 products at
// 1/16 of the 16-bit rate.
//
// STAGES > 2: the two buffers become a ring and up to STAGES - 1 K-steps are in flight (counted
// vmcnt waits, one raw barrier per step, as in conv_pipe.hip) for *any* gather.  This is for
// the short reductions of the CIFAR ResNets (K = 144 .. 576 in 32-wide steps whose channels
// straddle taps): with two buffers such a block is a chain of 5 - 18 dependent DMA round trips
// and nothing else, with a ring it is two or three.
template <int DT, int BM, int BN, int WAVES_M, int WAVES_N, int MODE, int EPI = EPI_PLAIN,
          int STAGES = 2>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, STAGES == 2 ? 4 : 2)
void conv_igemm_kernel(const ConvParams p) {
  typedef typename Elem<DT>::T T;
  constexpr int EPC = 16 / (int)sizeof(T);          // elements per 16-byte chunk: 8 or 4
  constexpr int BK = 64 / (int)sizeof(T);           // elements per K-step (one 64-byte row)
  constexpr bool POINTWISE = MODE == POINTWISE_MODE;
  constexpr int NW = WAVES_M * WAVES_N;             // waves per block: 4, or 8 for the 256x128 tile
  constexpr int RPP = 16 * NW;                      // tile rows filled per pass (16 per wave)
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MR = WM / 16, NR = WN / 16;
  constexpr int XV = BM / RPP;                      // 16-byte pixel-tile vectors per thread
  constexpr int WV = (BN + RPP - 1) / RPP;          // 16-byte weight-tile vectors per thread
  constexpr int XBYTES = BM * 64, WBYTES = BN * 64;
  static_assert(NW == 4 || NW == 8, "4 or 8 waves per block");
  static_assert(BM % RPP == 0 && WM % 16 == 0 && WN % 16 == 0, "tile shape");

  static_assert(STAGES >= 2 && STAGES <= 6, "ring depth");
  __shared__ __attribute__((aligned(16))) uint8_t smem[STAGES * (XBYTES + WBYTES)];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // ---- block -> tile, XCD aware --------------------------------------------------
  const int tilesN = (p.Cout + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  int mt = (slot / tilesN) * 8 + xcd;
  const int nt = slot % tilesN;
  if (mt >= tilesM) return;
  if (p.rev) mt = tilesM - 1 - mt;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread load geometry ----------------------------------------------------
  // Tiles go global -> LDS directly (global_load_lds_dwordx4, no VGPR staging and no
  // ds_write): one wave instruction fills 1 KiB = 16 tile rows of 64 bytes, lane l landing
  // at row l/4, 16-byte slot l%4.  The XOR swizzle therefore moves to the *source*: the lane
  // that owns physical slot s of a row fetches logical K-chunk s ^ g(row).  Out-of-image
  // taps, rows past M and channels past Cout fetch from a 16-byte page of zeros.
  const int lrow = tid >> 2;       // tile row of this lane within a pass of RPP rows
  const int kvx = (tid & 3) ^ ((-(lrow >> 2)) & 3);            // logical K-chunk, pixel tile
  const int kvw = (tid & 3) ^ ((-(lrow / (4 * NR))) & 3);      // logical K-chunk, weight tile
  int xbase[XV];                   // POINTWISE: pixel row index; else pixel index base n*H*W
  // base input coordinates (before the tap offset), two signed 16-bit halves of one register
  // (row in the upper one): the tiles that run four waves per SIMD have no register to spare
  int xhw[XV];
#pragma unroll
  for (int r = 0; r < XV; ++r) {
    const int m = m0 + lrow + RPP * r;
    if (POINTWISE) {
      xbase[r] = (m < p.M) ? m : -1;
      xhw[r] = 0;
    } else {
      if (m < p.M) {
        const uint32_t t = fd_div((uint32_t)m, p.divOW);
        const int ow = m - (int)t * p.OW;
        const uint32_t n = fd_div(t, p.divOH);
        const int oh = (int)t - (int)n * p.OH;
        xbase[r] = (int)n * p.H * p.W;
        xhw[r] = ((oh * p.ISH + p.dh0) << 16) | ((ow * p.ISW + p.dw0) & 0xffff);
      } else {
        xbase[r] = 0;
        xhw[r] = (int)0xc0000000u;   // row -16384: fails every bounds check
      }
    }
  }
  auto xh_of = [&](int r) { return xhw[r] >> 16; };
  auto xw_of = [&](int r) { return (xhw[r] << 16) >> 16; };
  const T* __restrict__ gin = reinterpret_cast<const T*>(p.in);
  const T* __restrict__ gw = reinterpret_cast<const T*>(p.wgt);
  const T* zero_page = reinterpret_cast<const T*>(g_zero_page);

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  // ---- TAPLOOP state: one pointer (+ per-step increment: 64 bytes, or 0 on the zero page)
  // per load, recomputed when the wave-uniform tap (tli, tlj) changes --------------------
  const T* xptr[XV];
  const T* wptr[WV];
  uint32_t xlive = 0;              // bit r: pixel load r advances by a K-step (0: it sits on the zero page)
  int winc[WV];
  int tli = 0, tlj = 0, tlc = 0;                   // current tap and K-steps done inside it
  const int steps_per_tap = p.Ctap / BK;
  auto setup_tap = [&]() {
#pragma unroll
    for (int r = 0; r < XV; ++r) {
      const int hi = xh_of(r) + tli * p.dhs;
      const int wi = xw_of(r) + tlj * p.dws;
      const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      xptr[r] = ok ? gin + ((size_t)(xbase[r] + hi * p.W + wi) * p.Cpix + kvx * EPC) : zero_page;
      xlive = (xlive & ~(1u << r)) | ((ok ? 1u : 0u) << r);
    }
    const int wk = p.wk0 + tli * p.wsi + tlj * p.wsj + kvw * EPC;
#pragma unroll
    for (int r = 0; r < WV; ++r) {
      const int co = n0 + lrow + RPP * r;
      const bool ok = co < p.Cout;
      wptr[r] = ok ? gw + ((size_t)co * p.ldw + wk) : zero_page;
      winc[r] = ok ? BK : 0;
    }
  };
  auto issue_tile_tap = [&](int buf) {
    uint8_t* xs = smem + buf * (XBYTES + WBYTES);
    uint8_t* ws = xs + XBYTES;
#pragma unroll
    for (int r = 0; r < XV; ++r) {
      __builtin_amdgcn_global_load_lds((gptr_t)xptr[r], (lptr_t)(xs + r * (RPP * 64) + wave * 1024), 16, 0, 0);
      xptr[r] += ((xlive >> r) & 1u) * BK;
    }
#pragma unroll
    for (int r = 0; r < WV; ++r) {
      if (RPP * r + wave * 16 < BN) {               // wave-uniform: whole 16-row pieces only
        __builtin_amdgcn_global_load_lds((gptr_t)wptr[r], (lptr_t)(ws + r * (RPP * 64) + wave * 1024), 16, 0, 0);
        wptr[r] += winc[r];
      }
    }
    if (++tlc == steps_per_tap) {                  // next tap (uniform branch)
      tlc = 0;
      if (++tlj == p.TJ) { tlj = 0; ++tli; }
      if (tli < p.TI) setup_tap();
    }
  };

  // running decomposition of the two k indices into (tap i, tap j, channel c)
  int kkx = kvx * EPC, kcx = kvx * EPC, tix = 0, tjx = 0;
  int kkw = kvw * EPC, kcw = kvw * EPC, tiw = 0, tjw = 0;
  if (!POINTWISE) {
    while (kcx >= p.Ctap) { kcx -= p.Ctap; if (++tjx == p.TJ) { tjx = 0; ++tix; } }
    while (kcw >= p.Ctap) { kcw -= p.Ctap; if (++tjw == p.TJ) { tjw = 0; ++tiw; } }
  }

  auto issue_tile = [&](int buf) {
    uint8_t* xs = smem + buf * (XBYTES + WBYTES);
    uint8_t* ws = xs + XBYTES;
    const bool kvalx = kkx < p.Ktotal;
#pragma unroll
    for (int r = 0; r < XV; ++r) {
      const T* src;
      if (POINTWISE) {
        const bool ok = kvalx && xbase[r] >= 0;
        src = ok ? gin + ((size_t)xbase[r] * p.Cpix + kkx) : zero_page;
      } else {
        const int hi = xh_of(r) + tix * p.dhs;
        const int wi = xw_of(r) + tjx * p.dws;
        const bool ok = kvalx && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        src = ok ? gin + ((size_t)(xbase[r] + hi * p.W + wi) * p.Cpix + kcx) : zero_page;
      }
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(xs + r * (RPP * 64) + wave * 1024), 16, 0, 0);
    }
    const bool kvalw = kkw < p.Ktotal;
    const int wk = POINTWISE ? (p.wk0 + kkw) : (p.wk0 + tiw * p.wsi + tjw * p.wsj + kcw);
#pragma unroll
    for (int r = 0; r < WV; ++r) {
      if (RPP * r + wave * 16 < BN) {               // wave-uniform: whole 16-row pieces only
        const int co = n0 + lrow + RPP * r;
        const bool ok = kvalw && co < p.Cout;
        const T* src = ok ? gw + ((size_t)co * p.ldw + wk) : zero_page;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ws + r * (RPP * 64) + wave * 1024), 16, 0, 0);
      }
    }
  };
  auto advance_k = [&]() {
    kkx += BK;
    kkw += BK;
    if (!POINTWISE) {
      kcx += BK;
      while (kcx >= p.Ctap) { kcx -= p.Ctap; if (++tjx == p.TJ) { tjx = 0; ++tix; } }
      kcw += BK;
      while (kcw >= p.Ctap) { kcw -= p.Ctap; if (++tjw == p.TJ) { tjw = 0; ++tiw; } }
    }
  };

  // ---- fragment addresses (loop invariant) -------------------------------------------
  const int fr = lane & 15, fq = lane >> 4;
  int xoff[MR], woff[NR];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    const int row = wm * WM + mr * 16 + fr;
    xoff[mr] = row * 64 + ((fq ^ ((-(row >> 2)) & 3)) << 4);
  }
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
    // MFMA row fr = 4*q' + r' of tile nr carries output channel q'*(4*NR) + nr*4 + r'
    const int row = wn * WN + (fr >> 2) * (4 * NR) + nr * 4 + (fr & 3);
    woff[nr] = row * 64 + ((fq ^ ((-(row / (4 * NR))) & 3)) << 4);
  }

  f32x4 acc[NR][MR];
#pragma unroll
  for (int nr = 0; nr < NR; ++nr)
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) acc[nr][mr] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the addend of the lane's first pixel block does not depend on the product: a pointwise layer
  // (short K loop, no tap bookkeeping in registers) asks for it now - older than every tile load,
  // so the counted waits of the ring retire it first
  conv_epi::AddendAhead<NR, (DT == DT_F32 ? conv_epi::EPI_PLAIN : EPI)> ahead;
  if constexpr (DT != DT_F32 && POINTWISE)
    conv_epi::addend_fetch<NR, POINTWISE, EPI, DT>(p, ahead, m0 + wm * WM + (lane & 15),
                                                   n0 + wn * WN + (lane >> 4) * 4 * NR);
  const int KT = (p.Ktotal + BK - 1) / BK;
  if constexpr (STAGES > 2) {
    // ---- ring of STAGES buffers ----------------------------------------------------------
    // vector-memory operations this wave issues per stage: its pixel pieces plus the weight
    // pieces that fall to it (wave-uniform: WV, or WV - 1 when the tile has fewer pieces)
    const bool w_all = RPP * (WV - 1) + wave * 16 < BN;
    auto issue = [&](int buf) {
      if constexpr (MODE == TAPLOOP) {
        issue_tile_tap(buf);
      } else {
        issue_tile(buf);
        advance_k();
      }
    };
    auto wait_landed = [&](int ahead) {            // `ahead` later stages may stay in flight
      if (ahead <= 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
      if (w_all) {
        if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * (XV + WV)) : "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (XV + WV)) : "memory");
        else if (ahead == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (XV + WV)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (XV + WV)) : "memory");
      } else {
        if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1 * (XV + WV - 1)) : "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (XV + WV - 1)) : "memory");
        else if (ahead == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (XV + WV - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (XV + WV - 1)) : "memory");
      }
    };
    static_assert(4 * (XV + WV) < 64, "vmcnt range");
    if constexpr (MODE == TAPLOOP) setup_tap();
    {
      const int pre = KT < STAGES - 1 ? KT : STAGES - 1;
      for (int sgi = 0; sgi < pre; ++sgi) issue(sgi);
    }
    int rbuf = 0, ibuf = STAGES - 1;
    for (int kt = 0; kt < KT; ++kt) {
      const int left = KT - 1 - kt;
      wait_landed(left < STAGES - 2 ? left : STAGES - 2);
      // own LDS reads of the previous step retired, then the block meets: step kt has landed
      // for every wave and nobody still reads the buffer that is refilled next
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (kt + STAGES - 1 < KT) {
        issue(ibuf);
        if (++ibuf == STAGES) ibuf = 0;
      }
      const uint8_t* xs = smem + rbuf * (XBYTES + WBYTES);
      const uint8_t* ws = xs + XBYTES;
      if (++rbuf == STAGES) rbuf = 0;
      u32x4 wf[NR], xf[MR];
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) wf[nr] = *reinterpret_cast<const u32x4*>(ws + woff[nr]);
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) xf[mr] = *reinterpret_cast<const u32x4*>(xs + xoff[mr]);
#pragma unroll
      for (int nr = 0; nr < NR; ++nr)
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
          acc[nr][mr] = mfma_step<DT>(wf[nr], xf[mr], acc[nr][mr]);
    }
  } else {
  if constexpr (MODE == TAPLOOP) {
    setup_tap();
    if (KT > 0) issue_tile_tap(0);
  } else if (KT > 0) {
    issue_tile(0);
    advance_k();
  }
  dma_wait_all();
  __syncthreads();

  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < KT;
    if (more) {                                    // next tile streams into the other buffer
      if constexpr (MODE == TAPLOOP) {             // while this one is multiplied
        issue_tile_tap(buf ^ 1);
      } else {
        issue_tile(buf ^ 1);
        advance_k();
      }
    }
    const uint8_t* xs = smem + buf * (XBYTES + WBYTES);
    const uint8_t* ws = xs + XBYTES;
    u32x4 wf[NR], xf[MR];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) wf[nr] = *reinterpret_cast<const u32x4*>(ws + woff[nr]);
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) xf[mr] = *reinterpret_cast<const u32x4*>(xs + xoff[mr]);
#pragma unroll
    for (int nr = 0; nr < NR; ++nr)
#pragma unroll
      for (int mr = 0; mr < MR; ++mr)
        acc[nr][mr] = mfma_step<DT>(wf[nr], xf[mr], acc[nr][mr]);
    dma_wait_all();
    __syncthreads();
  }
  }

  if constexpr (DT != DT_F32)
    conv_epi::conv_epilogue<MR, NR, POINTWISE, EPI, DT>(p, acc, m0 + wm * WM,
                                                        n0 + wn * WN + fq * 4 * NR, mt, lane,
                                                        POINTWISE ? &ahead : nullptr);
  else
    conv_epi::conv_epilogue_plain<DT, MR, NR, POINTWISE>(p, acc, m0 + wm * WM,
                                                         n0 + wn * WN + fq * 4 * NR, mt, lane);
}

template <int DT, int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int STAGES = 2>
void launch_cfg_epi(const ConvParams& p, hipStream_t stream) {
  const int tilesN = (p.Cout + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const int grid = ((tilesM + 7) / 8) * 8 * tilesN;
  const bool pointwise = p.TI == 1 && p.TJ == 1 && p.ISH == 1 && p.ISW == 1 && p.dh0 == 0 &&
                         p.dw0 == 0 && p.OSH == 1 && p.OSW == 1 && p.OOH == 0 && p.OOW == 0 &&
                         p.OH == p.H && p.OW == p.W && p.OHfull == p.OH && p.OWfull == p.OW &&
                         p.Ctap <= p.Cpix;
  if (pointwise)
    hipLaunchKernelGGL((conv_igemm_kernel<DT, BM, BN, WAVES_M, WAVES_N, POINTWISE_MODE, EPI, STAGES>),
                       dim3(grid), dim3(64 * WAVES_M * WAVES_N), 0, stream, p);
  else if (p.Ctap % bk_of(DT) == 0)
    hipLaunchKernelGGL((conv_igemm_kernel<DT, BM, BN, WAVES_M, WAVES_N, TAPLOOP, EPI, STAGES>),
                       dim3(grid), dim3(64 * WAVES_M * WAVES_N), 0, stream, p);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<DT, BM, BN, WAVES_M, WAVES_N, GENERIC, EPI, STAGES>),
                       dim3(grid), dim3(64 * WAVES_M * WAVES_N), 0, stream, p);
}

// The fused epilogues of conv_epilogue (backward statistics, folded inference BatchNorm) exist for
// the 16-bit types; fp32 launches take the plain epilogue (bias, addend, forward statistics) and
// the planner schedules separate launches for the rest.
template <int DT, int BM, int BN, int WAVES_M, int WAVES_N>
void launch_cfg16(const ConvParams& p, hipStream_t stream) {
  if (p.scale) launch_cfg_epi<DT, BM, BN, WAVES_M, WAVES_N, EPI_INFER>(p, stream);
  else if (p.stats_y) launch_cfg_epi<DT, BM, BN, WAVES_M, WAVES_N, EPI_BSTATS>(p, stream);
  else if (p.addend && !p.stats) launch_cfg_epi<DT, BM, BN, WAVES_M, WAVES_N, EPI_ADD>(p, stream);
  else launch_cfg_epi<DT, BM, BN, WAVES_M, WAVES_N, EPI_PLAIN>(p, stream);
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
void launch_cfg(const ConvParams& p, hipStream_t stream) {
  if (p.dt == DT_F32) return launch_cfg_epi<DT_F32, BM, BN, WAVES_M, WAVES_N, EPI_PLAIN>(p, stream);
  if (p.dt == DT_F16) return launch_cfg16<DT_F16, BM, BN, WAVES_M, WAVES_N>(p, stream);
  launch_cfg16<DT_BF16, BM, BN, WAVES_M, WAVES_N>(p, stream);
}

// Ring variants (cfg 8..13): bfloat16 training kernels; other element types and the inference
// epilogue take the two-stage kernel of the same tile.
template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES>
void launch_ring(const ConvParams& p, hipStream_t stream) {
  if (p.dt != DT_BF16 || p.scale) return launch_cfg<BM, BN, WAVES_M, WAVES_N>(p, stream);
  if (p.stats_y) launch_cfg_epi<DT_BF16, BM, BN, WAVES_M, WAVES_N, EPI_BSTATS, STAGES>(p, stream);
  else if (p.addend && !p.stats) launch_cfg_epi<DT_BF16, BM, BN, WAVES_M, WAVES_N, EPI_ADD, STAGES>(p, stream);
  else launch_cfg_epi<DT_BF16, BM, BN, WAVES_M, WAVES_N, EPI_PLAIN, STAGES>(p, stream);
}

}  // namespace

// Validates the descriptor on the host (a kernel that indexes out of bounds can take the
// whole GPU down, so every shape assumption is checked before the launch).
void validate_conv_params(const ConvParams& p) {
  auto fail = [](const std::string& m) { throw std::invalid_argument("conv_igemm: " + m); };
  if (!p.in || !p.wgt || !p.out) fail("null tensor pointer");
  if (p.dt != DT_BF16 && p.dt != DT_F16 && p.dt != DT_F32) fail("unknown element type");
  if (p.dt == DT_F32 && (p.scale || p.stats_y || p.relu))
    fail("backward statistics / folded BatchNorm epilogues exist for the 16-bit element types only");
  if (p.N <= 0 || p.H <= 0 || p.W <= 0 || p.OH <= 0 || p.OW <= 0) fail("bad spatial size");
  // the kernels keep a pixel's base row / column as signed 16-bit halves of one register
  if (p.H > 8192 || p.W > 8192 || p.dh0 < -4096 || p.dh0 > 4096 || p.dw0 < -4096 || p.dw0 > 4096 ||
      (int64_t)(p.OH - 1) * p.ISH + p.dh0 > 12288 || (int64_t)(p.OW - 1) * p.ISW + p.dw0 > 12288)
    fail("spatial coordinates beyond the 16-bit range of the gather");
  if (p.M != p.N * p.OH * p.OW) fail("M != N*OH*OW");
  if ((int64_t)p.N * p.H * p.W >= (1ll << 31) / 8) fail("input too large for 32-bit pixel index");
  if (p.Ctap % 8 || p.Ctap <= 0) fail("Ctap must be a positive multiple of 8");
  if (p.Ktotal != p.TI * p.TJ * p.Ctap) fail("Ktotal != TI*TJ*Ctap");
  if (p.ldw % 8 || p.wk0 % 8 || p.wsi % 8 || p.wsj % 8) fail("weight offsets must be multiples of 8");
  if (p.TI * p.TJ > 0 && p.wk0 + (p.TI - 1) * p.wsi + (p.TJ - 1) * p.wsj + p.Ctap > p.ldw)
    fail("weight row overrun");
  if (p.Cout <= 0 || p.ldc < p.Cout) fail("ldc < Cout");
  if (p.addend_mask) {
    if (!p.addend) fail("addend_mask without addend");
    if (p.Cout % 8 || p.ldc % 8) fail("a masked addend needs Cout and ldc to be multiples of 8");
  }
  if (p.scale) {
    if (!p.bias) fail("the inference epilogue needs scale and bias (shift) together");
    if (p.stats || p.addend_mask) fail("the inference epilogue takes no statistics / masked addend");
  } else if (p.relu) {
    fail("relu belongs to the inference epilogue (scale)");
  }
  if (p.stats) {
    if (p.stats_replicas <= 0) fail("stats needs a positive replica count");
    if (p.bias) fail("fused statistics need a bias-free convolution");
    if (p.addend && !p.stats_y) fail("forward statistics need a plain convolution (no addend)");
    if (p.Cout % 8 || p.ldc % 8) fail("fused statistics need Cout and ldc to be multiples of 8");
  }
  if (p.pro.sums) {
    const int C = p.Cpix;
    if (p.scale || p.stats_y || p.addend || p.bias) fail("the BatchNorm prologue goes with the plain training epilogue");
    if (C > 128) fail("BatchNorm prologue: at most 128 input channels");
    if (p.pro.replicas <= 0 || p.pro.P != p.N * p.H * p.W) fail("BatchNorm prologue: bad replica count / sample count");
    if (!p.pro.gamma || !p.pro.beta || !p.pro.save_mean || !p.pro.save_invstd || !p.pro.scale || !p.pro.shift)
      fail("BatchNorm prologue: missing parameter / output vector");
    if ((p.pro.moving_mean == nullptr) != (p.pro.moving_var == nullptr)) fail("BatchNorm prologue: moving mean and variance come together");
    if (p.pro.mask && !p.pro.z) fail("BatchNorm prologue: the mask is written along with z");
    // the block's own rows must lie inside its halo tile: forward taps that cover the output pixel
    if (p.dhs != 1 || p.dws != 1 || p.dh0 > 0 || -p.dh0 >= p.TI || p.dw0 > 0 || -p.dw0 >= p.TJ)
      fail("BatchNorm prologue: needs forward taps that cover the output pixel");
  }
  if (p.bpro.sums) {
    if (p.pro.sums || p.scale || p.bias) fail("the backward BatchNorm prologue belongs to a training data gradient");
    if (p.Cpix > 128) fail("backward BatchNorm prologue: at most 128 channels");
    if (p.bpro.replicas <= 0 || p.bpro.P != p.N * p.H * p.W) fail("backward BatchNorm prologue: bad replica count / sample count");
    if (!p.bpro.y || !p.bpro.mean || !p.bpro.invstd || !p.bpro.gamma || !p.bpro.dgamma || !p.bpro.dbeta || !p.bpro.dy)
      fail("backward BatchNorm prologue: missing tensor");
    // the block's own rows must lie inside its halo tile: the taps cover the output pixel
    const int oh = p.dhs > 0 ? p.dh0 : p.dh0 - (p.TI - 1), ow = p.dws > 0 ? p.dw0 : p.dw0 - (p.TJ - 1);
    if (oh > 0 || -oh >= p.TI || ow > 0 || -ow >= p.TJ) fail("backward BatchNorm prologue: the taps must cover the output pixel");
  }
  if ((p.stats_y || p.stats_mask) && !p.stats) fail("stats_y / stats_mask without a stats buffer");
  if (p.stats_mask && !p.stats_y) fail("stats_mask belongs to the backward statistics (stats_y)");
  if ((p.OH - 1) * p.OSH + p.OOH >= p.OHfull || (p.OW - 1) * p.OSW + p.OOW >= p.OWfull)
    fail("output index overrun");
  // every in-bounds pixel read must stay inside the input tensor: pixel*Cpix + Ctap <= total
  if (p.Ctap > p.Cpix) {
    // a tap that spans several pixels (stem): the last pixel of a row must leave room
    const int span = (p.Ctap + p.Cpix - 1) / p.Cpix;
    const int max_w = (p.OW - 1) * p.ISW + p.dw0 + (p.TJ - 1) * p.dws;
    if (p.dws < 0 || p.dhs < 0 || p.dw0 < 0 || p.dh0 < 0) fail("multi-pixel taps need forward taps");
    if (max_w + span > p.W) fail("multi-pixel tap overruns the input row");
    if ((p.OH - 1) * p.ISH + p.dh0 + (p.TI - 1) * p.dhs >= p.H) fail("multi-pixel tap overruns H");
    if ((p.Cpix * p.ISW) % 8 || (p.Cpix * p.W) % 8) fail("multi-pixel taps need 16-byte aligned rows");
  } else if (p.Cpix % 8) {
    fail("Cpix must be a multiple of 8");
  }
}

// cfg 0..15 and 32..39: the kernel of this file; cfg 16..31: the multi-stage pipeline of
// conv_pipe.hip, which hands descriptors that need the generic gather back to the tile of
// this file with the same footprint; cfg 40..47: the one-round-trip kernel for small layers of
// conv_halo.hip (same fallback rule; the only one that runs a BatchNorm prologue); cfg 48..55: the
// whole-rows kernel for large 3x3 layers of conv_rows.hip (falls back to a pipeline tile).
void launch_conv_igemm(const ConvParams& p, int cfg, hipStream_t stream) {
  validate_conv_params(p);
  if (cfg >= 40 && cfg < 48) {
    if (launch_conv_halo(p, cfg, stream)) return;
    if (p.pro.sums || p.bpro.sums)
      throw std::invalid_argument("conv: this tile cannot run the layer with a BatchNorm prologue");
    cfg = halo_fallback_cfg(cfg);
  } else if (p.pro.sums || p.bpro.sums) {
    throw std::invalid_argument("conv: a BatchNorm prologue needs a tile configuration of conv_halo.hip (40..47)");
  }
  if ((p.in_scale == nullptr) != (p.in_shift == nullptr))
    throw std::invalid_argument("conv: in_scale / in_shift come together");
  if (cfg >= 48 && cfg < 56) {
    if (launch_conv_rows(p, cfg, stream)) return;
    cfg = rows_fallback_cfg(cfg);
  }
  if (cfg >= 16 && cfg < 32) {
    if (p.dt != DT_F32 && launch_conv_pipe(p, cfg, stream)) return;
    cfg = pipe_fallback_cfg(cfg);
  }
  if (p.in_scale)
    throw std::invalid_argument("conv: normalising the input inside the tile needs a whole-rows 3x3 configuration (48..53) or, for a pointwise layer of at most 512 channels, a pipeline configuration (16..31) that serves the layer");
  switch (cfg) {
    case 0: launch_cfg<128, 128, 2, 2>(p, stream); break;
    case 1: launch_cfg<256, 64, 4, 1>(p, stream); break;
    case 2: launch_cfg<256, 32, 4, 1>(p, stream); break;
    case 3: launch_cfg<256, 16, 4, 1>(p, stream); break;
    case 4: launch_cfg<64, 128, 1, 4>(p, stream); break;
    case 5: launch_cfg<128, 64, 2, 2>(p, stream); break;
    case 6: launch_cfg<256, 128, 4, 2>(p, stream); break;   // 8 waves: 1.5x the FLOP per L2 byte
    // small tiles: more blocks for the latency-bound layers of the CIFAR ResNets
    case 7: launch_cfg<128, 16, 4, 1>(p, stream); break;
    case 14: launch_cfg<64, 64, 2, 2>(p, stream); break;
    case 15: launch_cfg<128, 32, 4, 1>(p, stream); break;
    case 32: launch_cfg<64, 32, 4, 1>(p, stream); break;
    case 33: launch_cfg<64, 16, 4, 1>(p, stream); break;
    case 34: launch_cfg<64, 32, 2, 2>(p, stream); break;
    //                       BM   BN  WM WN  ring          LDS, blocks / CU
    case 8: launch_ring<256, 16, 4, 1, 4>(p, stream); break;    // 68 KiB, 2
    case 9: launch_ring<256, 32, 4, 1, 4>(p, stream); break;    // 72 KiB, 2
    case 10: launch_ring<128, 64, 2, 2, 4>(p, stream); break;   // 48 KiB, 3
    case 11: launch_ring<256, 64, 4, 1, 4>(p, stream); break;   // 80 KiB, 2
    case 12: launch_ring<128, 16, 4, 1, 6>(p, stream); break;   // 54 KiB, 2
    case 13: launch_ring<128, 32, 4, 1, 6>(p, stream); break;   // 60 KiB, 2
    default: throw std::invalid_argument("conv_igemm: unknown tile config");
  }
}
