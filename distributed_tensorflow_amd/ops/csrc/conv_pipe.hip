// Implicit-GEMM NHWC convolution on MFMA with a multi-stage direct-to-LDS pipeline.
//
// Same gather GEMM and descriptor as conv_igemm.hip (forward, data gradient, 1x1 / dense):
//
//   out[n, oh*OSH+OOH, ow*OSW+OOW, co] (+= addend) (+ bias[co]) =
//       sum over taps (i, j) and channels c of
//       in[n, oh*ISH + dh0 + i*dhs, ow*ISW + dw0 + j*dws, c] * wgt[co][wk0 + i*wsi + j*wsj + c]
//
// What differs is the K loop.  conv_igemm.hip keeps one K-step of 32 in flight and drains it
// (vmcnt(0)) at the barrier that ends every step, so a block alternates between waiting for
// memory and multiplying; only other blocks on the same CU hide that.  Here the tiles of
// STAGES - 1 K-steps stay in flight: the LDS holds a ring of STAGES tile pairs, every wave
// waits with a *counted* `s_waitcnt vmcnt(n)` only for the loads of the step that is about to
// be multiplied, and the block meets at a raw `s_barrier` (which, unlike `__syncthreads()`,
// does not drain the vector-memory counter).  One barrier per K-step does both jobs: after
// it every wave's part of step t has landed (each waited for its own loads before arriving),
// and every wave has finished reading step t-1, whose buffer the loads issued right after the
// barrier (step t + STAGES - 1) overwrite.
//
//   BK = 64: 128-byte tile rows (two rows per 256-byte LDS bank row), 32 MFMAs per wave and
//            barrier; BK = 32: 64-byte rows as in conv_igemm.hip, more stages per LDS byte.
//
// LDS image: rows of BK bf16, the 16-byte chunks of a row XOR-swizzled by a function of the
// row so that the ds_read_b128 fragment reads of every 16-lane group touch 16 distinct 16-byte
// slots of the 256-byte bank row.  The DMA writes LDS linearly (wave-uniform base + 16 * lane),
// so the swizzle is applied to the *source* address: the lane that owns physical chunk c of
// row r fetches logical K-chunk c ^ f(r).
//
// Out-of-image taps, rows past M, channels past Cout and K past Ktotal fetch a 16-byte page of
// zeros, so every wave issues the same number of loads per stage and the counted waits hold.
//
// Supported addressing modes: POINTWISE (1x1 stride 1 / dense) and TAPLOOP (Ctap % BK == 0);
// anything else is served by conv_igemm.hip's generic gather.
#include <stdexcept>
#include <string>

#include "common.h"
#include "conv_epilogue.h"
#include "conv_params.h"

namespace {

using conv_epi::EPI_ADD;
using conv_epi::EPI_BSTATS;
using conv_epi::EPI_INFER;
using conv_epi::EPI_PLAIN;

__device__ __attribute__((aligned(16))) uint32_t g_zero_page_p[4] = {0u, 0u, 0u, 0u};

constexpr int POINTWISE_MODE = 1, TAPLOOP = 2;

// XOR applied to the 16-byte chunk index of pixel-tile row `row`.
template <int BK>
__device__ __forceinline__ int swz_x(int row) {
  if constexpr (BK == 64) return (row >> 1) & 7;
  else return (-(row >> 2)) & 3;
}
// Same for the weight tile, whose MFMA row fr = 4*q + r of tile nr is tile row
// q*(4*NR) + nr*4 + r (so that a lane ends up with 4*NR consecutive output channels).
template <int BK, int NR>
__device__ __forceinline__ int swz_w(int row) {
  if constexpr (BK == 64) return ((row >> 1) & 1) | (((row / (4 * NR)) & 3) << 1);
  else return (-(row / (4 * NR))) & 3;
}

// waves per SIMD the register allocation has to allow: as many blocks as the LDS ring admits
constexpr int INBN_KMAX = 512;                      // channels of an input normalised in the fragments
constexpr int pipe_occ(int bm, int bn, int nw, int bk, int stages, bool inbn = false) {
  const int lds = stages * (bm + bn) * bk * 2 + (inbn ? INBN_KMAX * 8 : 0);
  int bpc = (160 * 1024) / lds;
  if (bpc < 1) bpc = 1;
  int w = bpc * nw / 4;
  return w > 4 ? 4 : (w < 1 ? 1 : w);
}

// max(x * scale + shift, 0) of the two 16-bit elements of a dword, rounded back to 16 bits.  bf16: one
// packed multiply-add for the pair and the ReLU on the packed result (a signed 16-bit max against
// zero clears every negative value, -0 included) - five vector instructions a pair instead of seven.
template <int DT>
__device__ __forceinline__ uint32_t bn_relu_pair(uint32_t w, float sc_lo, float sc_hi, float sh_lo, float sh_hi) {
  if constexpr (DT == DT_BF16) {
    typedef __attribute__((ext_vector_type(2))) short short2v;
    const f32x2 x = {__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)};
    const f32x2 s = {sc_lo, sc_hi}, h = {sh_lo, sh_hi};
    const f32x2 r = __builtin_elementwise_fma(x, s, h);
    const uint32_t pk = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
    const short2v z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(short2v, pk), z));
  } else {
    return pack2<DT>(fmaxf(lo16<DT>(w) * sc_lo + sh_lo, 0.f), fmaxf(hi16<DT>(w) * sc_hi + sh_hi, 0.f));
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// INBN (pointwise layers): the BatchNorm + ReLU of the layer in front runs on the pixel fragments
// between their LDS read and the matrix instructions, x := max(x * in_scale[k] + in_shift[k], 0)
// with the constants in an LDS table behind the ring - the normalised tensor is never written.  A
// fragment is transformed by each of the WAVES_N waves that multiply it; the layers this serves
// are bound by their memory traffic and leave the vector unit idle.
template <int BM, int BN, int WAVES_M, int WAVES_N, int BK, int STAGES, int MODE, int EPI,
          int DT = DT_BF16, bool INBN = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N,
                             pipe_occ(BM, BN, WAVES_M * WAVES_N, BK, STAGES, INBN))
void conv_pipe_kernel(const ConvParams p) {
  static_assert(DT == DT_BF16 || DT == DT_F16, "16-bit element types");
  constexpr bool POINTWISE = MODE == POINTWISE_MODE;
  static_assert(!INBN || POINTWISE, "fragment BatchNorm: pointwise layers");
  constexpr int NW = WAVES_M * WAVES_N;
  constexpr int CPR = BK / 8;                       // 16-byte chunks per tile row
  constexpr int ROWB = BK * 2;                      // bytes per tile row
  constexpr int RPW = 1024 / ROWB;                  // tile rows per 1 KiB piece (one DMA instruction)
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MR = WM / 16, NR = WN / 16;
  constexpr int XV = BM / RPW / NW, WV = BN / RPW / NW;   // DMA instructions per wave and stage
  constexpr int LPS = XV + WV;
  constexpr int XBYTES = BM * ROWB, WBYTES = BN * ROWB, SBYTES = XBYTES + WBYTES;
  static_assert(BK == 32 || BK == 64, "K-step");
  static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
  static_assert((BM / RPW) % NW == 0 && (BN / RPW) % NW == 0, "every wave issues the same loads");
  static_assert(WM % 16 == 0 && WN % 16 == 0, "tile shape");
  static_assert((STAGES - 1) * LPS < 64, "vmcnt range");

  // all LDS in one array (a second __shared__ object makes the compiler drain the DMA early)
  __shared__ __attribute__((aligned(16))) uint8_t smem[STAGES * SBYTES + (INBN ? INBN_KMAX * 8 : 0)];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // ---- block -> tile: the BN-tiles of one pixel tile share blockIdx % 8 (one XCD's L2) ----
  const int tilesN = (p.Cout + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  int mt = (slot / tilesN) * 8 + xcd;
  const int nt = slot % tilesN;
  if (mt >= tilesM) return;
  if (p.rev) mt = tilesM - 1 - mt;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-lane load geometry -----------------------------------------------------------
  const int prow = lane / CPR, pc = lane % CPR;     // row inside a piece, physical chunk
  int kvx[XV], kvw[WV];                             // logical K-chunk this lane fetches
  int xbase[XV], xh[XV], xw[XV];
#pragma unroll
  for (int r = 0; r < XV; ++r) {
    const int row = (r * NW + wave) * RPW + prow;
    kvx[r] = pc ^ swz_x<BK>(row);
    const int m = m0 + row;
    if (POINTWISE) {
      xbase[r] = (m < p.M) ? m : -1;
      xh[r] = xw[r] = 0;
    } else if (m < p.M) {
      const uint32_t t = fd_div((uint32_t)m, p.divOW);
      const int ow = m - (int)t * p.OW;
      const uint32_t n = fd_div(t, p.divOH);
      const int oh = (int)t - (int)n * p.OH;
      xbase[r] = (int)n * p.H * p.W;
      xh[r] = oh * p.ISH + p.dh0;
      xw[r] = ow * p.ISW + p.dw0;
    } else {
      xbase[r] = 0;
      xh[r] = -(1 << 28);                           // fails every bounds check
      xw[r] = 0;
    }
  }
  int wco[WV];
#pragma unroll
  for (int r = 0; r < WV; ++r) {
    const int row = (r * NW + wave) * RPW + prow;
    kvw[r] = pc ^ swz_w<BK, NR>(row);
    wco[r] = n0 + row;
  }
  const bf16_t* __restrict__ gin = reinterpret_cast<const bf16_t*>(p.in);
  const bf16_t* __restrict__ gw = reinterpret_cast<const bf16_t*>(p.wgt);
  const bf16_t* zero_page = reinterpret_cast<const bf16_t*>(g_zero_page_p);

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  // ---- source pointers: one per load, bumped by BK elements per K-step (0 on the zero page);
  // TAPLOOP recomputes them when the wave-uniform tap changes ---------------------------------
  const bf16_t* xptr[XV];
  const bf16_t* wptr[WV];
  int xinc[XV], winc[WV];
  int tli = 0, tlj = 0, tlc = 0;
  const int steps_per_tap = POINTWISE ? (1 << 30) : p.Ctap / BK;
  int kdone = 0;                                    // K elements already issued (POINTWISE: ragged K)
  auto setup_tap = [&]() {
#pragma unroll
    for (int r = 0; r < XV; ++r) {
      bool ok;
      size_t off;
      if (POINTWISE) {
        ok = xbase[r] >= 0 && kvx[r] * 8 < p.Ktotal;
        off = (size_t)(ok ? xbase[r] : 0) * p.Cpix + kvx[r] * 8;
      } else {
        const int hi = xh[r] + tli * p.dhs;
        const int wi = xw[r] + tlj * p.dws;
        ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
        off = (size_t)(xbase[r] + hi * p.W + wi) * p.Cpix + kvx[r] * 8;
      }
      xptr[r] = ok ? gin + off : zero_page;
      xinc[r] = ok ? BK : 0;
    }
#pragma unroll
    for (int r = 0; r < WV; ++r) {
      const int wk = p.wk0 + tli * p.wsi + tlj * p.wsj + kvw[r] * 8;
      const bool ok = wco[r] < p.Cout && (!POINTWISE || kvw[r] * 8 < p.Ktotal);
      wptr[r] = ok ? gw + ((size_t)wco[r] * p.ldw + wk) : zero_page;
      winc[r] = ok ? BK : 0;
    }
  };
  auto issue_stage = [&](int sbyte) {
    uint8_t* xs = smem + sbyte;
    uint8_t* ws = xs + XBYTES;
#pragma unroll
    for (int r = 0; r < XV; ++r) {
      __builtin_amdgcn_global_load_lds((gptr_t)xptr[r], (lptr_t)(xs + (r * NW + wave) * 1024), 16, 0, 0);
      xptr[r] += xinc[r];
    }
#pragma unroll
    for (int r = 0; r < WV; ++r) {
      __builtin_amdgcn_global_load_lds((gptr_t)wptr[r], (lptr_t)(ws + (r * NW + wave) * 1024), 16, 0, 0);
      wptr[r] += winc[r];
    }
    if constexpr (POINTWISE) {
      // ragged K (a multiple of 8, not of BK): chunks past Ktotal come from the zero page
      kdone += BK;
      if (kdone + BK > p.Ktotal) {
#pragma unroll
        for (int r = 0; r < XV; ++r)
          if (kdone + kvx[r] * 8 >= p.Ktotal) { xptr[r] = zero_page; xinc[r] = 0; }
#pragma unroll
        for (int r = 0; r < WV; ++r)
          if (kdone + kvw[r] * 8 >= p.Ktotal) { wptr[r] = zero_page; winc[r] = 0; }
      }
    } else {
      if (++tlc == steps_per_tap) {                 // next tap (wave-uniform branch)
        tlc = 0;
        if (++tlj == p.TJ) { tlj = 0; ++tli; }
        if (tli < p.TI) setup_tap();
      }
    }
  };

  // ---- fragment addresses (loop invariant, relative to the stage base) ----------------------
  const int fr = lane & 15, fq = lane >> 4;
  int xoff[MR], woff[NR];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    const int row = wm * WM + mr * 16 + fr;
    xoff[mr] = row * ROWB + ((fq ^ swz_x<BK>(row)) << 4);
  }
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
    const int row = wn * WN + (fr >> 2) * (4 * NR) + nr * 4 + (fr & 3);
    woff[nr] = XBYTES + row * ROWB + ((fq ^ swz_w<BK, NR>(row)) << 4);
  }

  f32x4 acc[NR][MR];
#pragma unroll
  for (int nr = 0; nr < NR; ++nr)
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) acc[nr][mr] = f32x4{0.f, 0.f, 0.f, 0.f};

  // the addend of the lane's first pixel block does not depend on the product: a pointwise layer
  // (short K loop, no tap bookkeeping in registers) asks for it now - older than every tile load,
  // so the counted waits of the ring retire it first
  conv_epi::AddendAhead<NR, EPI> ahead;
  if constexpr (POINTWISE)
    conv_epi::addend_fetch<NR, POINTWISE, EPI, DT>(p, ahead, m0 + wm * WM + (lane & 15),
                                                   n0 + wn * WN + (lane >> 4) * 4 * NR);
  const int KT = (p.Ktotal + BK - 1) / BK;
  float* const bntab = reinterpret_cast<float*>(smem + STAGES * SBYTES);   // [scale | shift]
  if constexpr (INBN) {
    // channels past Ktotal multiply weight chunks of zeros: any finite constant will do
    for (int k = tid; k < INBN_KMAX; k += 64 * NW) {
      bntab[k] = k < p.Ktotal ? p.in_scale[k] : 0.f;
      bntab[INBN_KMAX + k] = k < p.Ktotal ? p.in_shift[k] : 0.f;
    }
    // visible to every wave after the barrier of the first K-step (which waits for lgkmcnt(0))
  }
  setup_tap();
  // prologue: the first STAGES - 1 steps
  int issue_byte = 0;                               // ring position of the next stage to fill
  {
    const int pre = KT < STAGES - 1 ? KT : STAGES - 1;
    for (int s = 0; s < pre; ++s) {
      issue_stage(issue_byte);
      issue_byte += SBYTES;
    }
    // issue_byte is now pre * SBYTES <= (STAGES-1) * SBYTES: inside the ring
  }
  int read_byte = 0;
  for (int kt = 0; kt < KT; ++kt) {
    // stages issued after step kt that may stay in flight while it is multiplied
    const int ahead = (KT - 1 - kt) < (STAGES - 2) ? (KT - 1 - kt) : (STAGES - 2);
    if (STAGES >= 4 && ahead >= 2) wait_vmcnt<(STAGES >= 4 ? 2 : 0) * LPS>();
    else if (STAGES >= 3 && ahead >= 1) wait_vmcnt<(STAGES >= 3 ? 1 : 0) * LPS>();
    else wait_vmcnt<0>();
    // own LDS reads of the previous step retired, then the block meets: step kt has landed
    // for every wave and nobody still reads the buffer that is refilled next
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (kt + STAGES - 1 < KT) {
      issue_stage(issue_byte);
      issue_byte += SBYTES;
      if (issue_byte == STAGES * SBYTES) issue_byte = 0;
    }
    const uint8_t* st = smem + read_byte;
    read_byte += SBYTES;
    if (read_byte == STAGES * SBYTES) read_byte = 0;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      u32x4 wf[NR], xf[MR];
#pragma unroll
      for (int nr = 0; nr < NR; ++nr)
        wf[nr] = *reinterpret_cast<const u32x4*>(st + (woff[nr] ^ (ks << 6)));
#pragma unroll
      for (int mr = 0; mr < MR; ++mr)
        xf[mr] = *reinterpret_cast<const u32x4*>(st + (xoff[mr] ^ (ks << 6)));
      if constexpr (INBN) {
        // the lane's eight reduction channels of this step: kt * BK + ks * 32 + fq * 8 ...
        const float* tsc = bntab + kt * BK + ks * 32 + fq * 8;
        const f32x4 sa = *reinterpret_cast<const f32x4*>(tsc), sb = *reinterpret_cast<const f32x4*>(tsc + 4);
        const f32x4 ha = *reinterpret_cast<const f32x4*>(tsc + INBN_KMAX),
                    hb = *reinterpret_cast<const f32x4*>(tsc + INBN_KMAX + 4);
        const float sc[8] = {sa[0], sa[1], sa[2], sa[3], sb[0], sb[1], sb[2], sb[3]};
        const float sh[8] = {ha[0], ha[1], ha[2], ha[3], hb[0], hb[1], hb[2], hb[3]};
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) {
          const uint32_t w[4] = {xf[mr].x, xf[mr].y, xf[mr].z, xf[mr].w};
          uint32_t o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            o[j] = bn_relu_pair<DT>(w[j], sc[2 * j], sc[2 * j + 1], sh[2 * j], sh[2 * j + 1]);
          xf[mr] = u32x4{o[0], o[1], o[2], o[3]};
          // (a fragment in front of its own products: the vector instructions of fragment mr + 1
          // run while the matrix unit works on fragment mr)
#pragma unroll
          for (int nr = 0; nr < NR; ++nr)
            acc[nr][mr] = mfma_step<DT>(wf[nr], xf[mr], acc[nr][mr]);
        }
        continue;
      }
#pragma unroll
      for (int nr = 0; nr < NR; ++nr)
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
          acc[nr][mr] = mfma_step<DT>(wf[nr], xf[mr], acc[nr][mr]);
    }
  }

  conv_epi::conv_epilogue<MR, NR, POINTWISE, EPI, DT>(p, acc, m0 + wm * WM,
                                                      n0 + wn * WN + fq * 4 * NR, mt, lane,
                                                      POINTWISE ? &ahead : nullptr);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int BK, int STAGES, int EPI, int DT>
bool launch_pipe_epi(const ConvParams& p, hipStream_t stream) {
  const int tilesN = (p.Cout + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const int grid = ((tilesM + 7) / 8) * 8 * tilesN;
  const bool pointwise = p.TI == 1 && p.TJ == 1 && p.ISH == 1 && p.ISW == 1 && p.dh0 == 0 &&
                         p.dw0 == 0 && p.OSH == 1 && p.OSW == 1 && p.OOH == 0 && p.OOW == 0 &&
                         p.OH == p.H && p.OW == p.W && p.OHfull == p.OH && p.OWfull == p.OW &&
                         p.Ctap <= p.Cpix;
  if (p.in_scale) {
    // BatchNorm + ReLU of the input on the fragments: pointwise layers with a plain or
    // forward-statistics epilogue, at most INBN_KMAX channels, ring + table within the LDS
    if constexpr ((EPI == EPI_PLAIN) && STAGES * (BM + BN) * BK * 2 + INBN_KMAX * 8 <= 160 * 1024) {
      if (!pointwise || p.Ktotal > INBN_KMAX) return false;
      hipLaunchKernelGGL((conv_pipe_kernel<BM, BN, WAVES_M, WAVES_N, BK, STAGES, POINTWISE_MODE, EPI, DT, true>),
                         dim3(grid), dim3(64 * WAVES_M * WAVES_N), 0, stream, p);
      return true;
    } else {
      return false;
    }
  }
  if (pointwise)
    hipLaunchKernelGGL((conv_pipe_kernel<BM, BN, WAVES_M, WAVES_N, BK, STAGES, POINTWISE_MODE, EPI, DT>),
                       dim3(grid), dim3(64 * WAVES_M * WAVES_N), 0, stream, p);
  else if (p.Ctap % BK == 0)
    hipLaunchKernelGGL((conv_pipe_kernel<BM, BN, WAVES_M, WAVES_N, BK, STAGES, TAPLOOP, EPI, DT>),
                       dim3(grid), dim3(64 * WAVES_M * WAVES_N), 0, stream, p);
  else
    return false;
  return true;
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int BK, int STAGES, int DT>
bool launch_pipe_dt(const ConvParams& p, hipStream_t stream) {
  if (p.scale) return launch_pipe_epi<BM, BN, WAVES_M, WAVES_N, BK, STAGES, EPI_INFER, DT>(p, stream);
  if (p.stats_y) return launch_pipe_epi<BM, BN, WAVES_M, WAVES_N, BK, STAGES, EPI_BSTATS, DT>(p, stream);
  if (p.addend && !p.stats) return launch_pipe_epi<BM, BN, WAVES_M, WAVES_N, BK, STAGES, EPI_ADD, DT>(p, stream);
  return launch_pipe_epi<BM, BN, WAVES_M, WAVES_N, BK, STAGES, EPI_PLAIN, DT>(p, stream);
}

// F16: the configurations the tile table actually uses are instantiated for IEEE half as well;
// the others hand half-precision descriptors back to the two-stage kernel.
template <int BM, int BN, int WAVES_M, int WAVES_N, int BK, int STAGES, bool F16 = false>
bool launch_pipe(const ConvParams& p, hipStream_t stream) {
  if (p.dt == DT_F16) {
    if constexpr (F16) return launch_pipe_dt<BM, BN, WAVES_M, WAVES_N, BK, STAGES, DT_F16>(p, stream);
    else return false;
  }
  return launch_pipe_dt<BM, BN, WAVES_M, WAVES_N, BK, STAGES, DT_BF16>(p, stream);
}

}  // namespace

// Two-stage tile of conv_igemm.hip with the footprint of pipeline configuration `cfg`.
int pipe_fallback_cfg(int cfg) {
  switch (cfg) {
    case 16: case 17: case 26: case 28: case 29: case 30: case 31: return 6;
    case 18: case 19: case 20: case 27: case 25: return 0;
    case 21: return 5;
    case 22: case 23: return 1;
    case 24: return 4;
    default: throw std::invalid_argument("conv_pipe: unknown tile config");
  }
}

// Tile configurations 16.. of the convolution op.  Returns false when the descriptor needs the
// generic gather (the caller then uses conv_igemm.hip).
bool launch_conv_pipe(const ConvParams& p, int cfg, hipStream_t stream) {
  switch (cfg) {
    //                        BM   BN  WM WN  BK  STAGES
    case 16: return launch_pipe<256, 128, 4, 2, 64, 3>(p, stream);   // 144 KiB, 1 block / CU
    case 17: return launch_pipe<256, 128, 4, 2, 32, 3, true>(p, stream);   //  72 KiB, 2 blocks / CU
    case 18: return launch_pipe<128, 128, 2, 2, 64, 2, true>(p, stream);   //  64 KiB, 2 blocks / CU
    case 19: return launch_pipe<128, 128, 2, 2, 64, 3>(p, stream);   //  96 KiB, 1 block / CU
    case 20: return launch_pipe<128, 128, 2, 2, 32, 3, true>(p, stream);   //  48 KiB, 3 blocks / CU
    case 21: return launch_pipe<128, 64, 2, 2, 64, 3, true>(p, stream);    //  72 KiB, 2 blocks / CU
    case 22: return launch_pipe<256, 64, 4, 1, 64, 3>(p, stream);    // 120 KiB, 1 block / CU
    case 23: return launch_pipe<256, 64, 4, 1, 32, 3, true>(p, stream);    //  60 KiB, 2 blocks / CU
    case 24: return launch_pipe<64, 128, 1, 4, 64, 3, true>(p, stream);    //  72 KiB, 2 blocks / CU
    case 25: return launch_pipe<128, 256, 2, 4, 64, 3>(p, stream);   // 144 KiB, 1 block / CU
    case 26: return launch_pipe<256, 128, 4, 2, 32, 4>(p, stream);   //  96 KiB, 1 block / CU
    case 27: return launch_pipe<128, 128, 2, 2, 32, 4>(p, stream);   //  64 KiB, 2 blocks / CU
    // 128 x 64 per wave (12 fragment reads per 32 MFMAs instead of 8 per 16) and 256-wide blocks
    // (a third fewer bytes staged per multiply-add)
    case 28: return launch_pipe<256, 128, 2, 2, 32, 3>(p, stream);   //  72 KiB, 2 blocks / CU, 4 waves
    case 29: return launch_pipe<256, 256, 2, 4, 32, 3>(p, stream);   //  96 KiB, 1 block / CU, 8 waves
    case 30: return launch_pipe<256, 256, 4, 4, 32, 3, true>(p, stream);   //  96 KiB, 1 block / CU, 16 waves
    case 31: return launch_pipe<128, 256, 2, 4, 32, 3, true>(p, stream);   //  72 KiB, 2 blocks / CU
    default: throw std::invalid_argument("conv_pipe: unknown tile config");
  }
}
