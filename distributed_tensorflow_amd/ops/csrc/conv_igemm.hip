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
//
// PRO: training BatchNorm of the *input* folded into this launch (BnPrologue in conv_params.h;
// generic weight addressing, two buffers).  The block normalises its pixel rows and their border
// once into a halo tile in LDS and every tap reads its operand there; only the weight tile still
// streams per K-step.  This is for the small layers of the CIFAR ResNets, whose kernels are
// chains of a few dependent memory round trips: one launch instead of two saves about 2 us per
// layer there (ResNet-56 forward: 122 -> 73 launches), while the convolutions of large layers
// fill LDS by DMA and have no register stage to do arithmetic in.
template <int DT, int BM, int BN, int WAVES_M, int WAVES_N, int MODE, int EPI = EPI_PLAIN,
          int STAGES = 2, bool PRO = false>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, STAGES == 2 ? 4 : 2)
void conv_igemm_kernel(const ConvParams p) {
  static_assert(!PRO || (MODE == GENERIC && STAGES == 2 && DT != DT_F32),
                "the BatchNorm prologue is built on the generic two-buffer gather, 16-bit types");
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
  const int mt = (slot / tilesN) * 8 + xcd;
  const int nt = slot % tilesN;
  if (mt >= tilesM) return;
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
  int xh[XV], xw[XV];              // base input coordinates (before the tap offset)
#pragma unroll
  for (int r = 0; r < XV; ++r) {
    const int m = m0 + lrow + RPP * r;
    if (POINTWISE) {
      xbase[r] = (m < p.M) ? m : -1;
      xh[r] = xw[r] = 0;
    } else {
      if (m < p.M) {
        const uint32_t t = fd_div((uint32_t)m, p.divOW);
        const int ow = m - (int)t * p.OW;
        const uint32_t n = fd_div(t, p.divOH);
        const int oh = (int)t - (int)n * p.OH;
        xbase[r] = (int)n * p.H * p.W;
        xh[r] = oh * p.ISH + p.dh0;
        xw[r] = ow * p.ISW + p.dw0;
      } else {
        xbase[r] = 0;
        xh[r] = -(1 << 28);       // fails every bounds check
        xw[r] = 0;
      }
    }
  }
  const T* __restrict__ gin = reinterpret_cast<const T*>(p.in);
  const T* __restrict__ gw = reinterpret_cast<const T*>(p.wgt);
  const T* zero_page = reinterpret_cast<const T*>(g_zero_page);

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  // ---- TAPLOOP state: one pointer (+ per-step increment: 64 bytes, or 0 on the zero page)
  // per load, recomputed when the wave-uniform tap (tli, tlj) changes --------------------
  const T* xptr[XV];
  const T* wptr[WV];
  int xinc[XV], winc[WV];
  int tli = 0, tlj = 0, tlc = 0;                   // current tap and K-steps done inside it
  const int steps_per_tap = p.Ctap / BK;
  auto setup_tap = [&]() {
#pragma unroll
    for (int r = 0; r < XV; ++r) {
      const int hi = xh[r] + tli * p.dhs;
      const int wi = xw[r] + tlj * p.dws;
      const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      xptr[r] = ok ? gin + ((size_t)(xbase[r] + hi * p.W + wi) * p.Cpix + kvx * EPC) : zero_page;
      xinc[r] = ok ? BK : 0;
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
      xptr[r] += xinc[r];
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
        const int hi = xh[r] + tix * p.dhs;
        const int wi = xw[r] + tjx * p.dws;
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

  // ---- PRO: only the weight tile streams per K-step ------------------------------------------
  auto issue_w = [&](int buf) {      // the weight half of issue_tile
    uint8_t* ws = smem + buf * (XBYTES + WBYTES) + XBYTES;
    const bool kvalw = kkw < p.Ktotal;
    const int wk = p.wk0 + tiw * p.wsi + tjw * p.wsj + kcw;
#pragma unroll
    for (int r = 0; r < WV; ++r) {
      if (RPP * r + wave * 16 < BN) {
        const int co = n0 + lrow + RPP * r;
        const bool ok = kvalw && co < p.Cout;
        const T* src = ok ? gw + ((size_t)co * p.ldw + wk) : zero_page;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ws + r * (RPP * 64) + wave * 1024), 16, 0, 0);
      }
    }
  };

  // ---- BatchNorm prologue ----------------------------------------------------------------------
  // The block's pixels are BM / W whole rows of one image (host check).  Those rows plus the
  // border the taps reach go through registers once - y * scale + shift (+ residual), ReLU - into
  // a halo tile [rows + TI - 1][W + TJ - 1][C] in LDS, zeros outside the image; every tap of the
  // K loop reads its operand there.  The channel tile 0 blocks also store their own pixels'
  // normalised values (the input of this layer's weight gradient, a later layer's residual)
  // and ReLU bits.  Three global round trips are in flight together: the first weight tile
  // (DMA), the raw pixels of the halo tile (registers), the producer's replica sums.
  constexpr int HALO_BYTES = PRO ? 16384 : 16;
  __shared__ __attribute__((aligned(16))) uint8_t halo[HALO_BYTES];
  __shared__ __attribute__((aligned(16))) float pro_sc[PRO ? 128 : 4];
  __shared__ __attribute__((aligned(16))) float pro_sh[PRO ? 128 : 4];
  const int HW = p.W + p.TJ - 1;       // halo row, pixels
  const int KT = (p.Ktotal + BK - 1) / BK;
  if constexpr (PRO) {
    constexpr int NT = 64 * NW;
    constexpr int HCH = HALO_BYTES / 16 / NT;      // halo chunks per thread at most
    if (KT > 0) {
      issue_w(0);
      advance_k();
    }
    const int C = p.Cpix, cpr = C >> 3, cols = 2 * C;
    const int img = p.H * p.W;
    const int n_img = m0 / img;
    const int r0 = (m0 - n_img * img) / p.W;
    const int RB = BM / p.W;
    const int total = (RB + p.TI - 1) * HW * cpr;
    const T* gres = reinterpret_cast<const T*>(p.pro.res);
    u32x4 yv[HCH], rv[HCH];
    size_t hsrc[HCH];
    bool hok[HCH], hown[HCH];
#pragma unroll
    for (int u = 0; u < HCH; ++u) {
      const int idx = tid + u * NT;
      const int ch = idx % cpr, px = idx / cpr;
      const int wc = px % HW, hr = px / HW;
      const int hi = r0 + hr + p.dh0, wi = wc + p.dw0;
      hok[u] = idx < total && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
      hown[u] = (unsigned)(hr + p.dh0) < (unsigned)RB;       // one of the block's own rows
      hsrc[u] = hok[u] ? ((size_t)(n_img * p.H + hi) * p.W + wi) * C + ch * 8 : 0;
      yv[u] = hok[u] ? *reinterpret_cast<const u32x4*>(gin + hsrc[u]) : u32x4{0u, 0u, 0u, 0u};
      rv[u] = (hok[u] && gres != nullptr) ? *reinterpret_cast<const u32x4*>(gres + hsrc[u])
                                          : u32x4{0u, 0u, 0u, 0u};
    }
    // column sums of [replicas][2*C] (2*C <= 256 <= threads, checked on the host): thread
    // (row lane rl, column col) adds every RL-th replica row, one LDS round joins the row lanes
    const int RL = NT / cols;
    const int col = tid % cols, rl = tid / cols;
    float* part = reinterpret_cast<float*>(halo);           // the halo tile is filled afterwards
    if (rl < RL) {
      float sum = 0.f;
      for (int g = rl; g < p.pro.replicas; g += RL) sum += p.pro.sums[(size_t)g * cols + col];
      part[rl * cols + col] = sum;
    }
    __syncthreads();
    if (tid < C) {
      float s1 = 0.f, s2 = 0.f;
      for (int r = 0; r < RL; ++r) { s1 += part[r * cols + tid]; s2 += part[r * cols + C + tid]; }
      const float invP = 1.f / (float)p.pro.P;
      const float mean = s1 * invP;
      const float var = fmaxf(s2 * invP - mean * mean, 0.f);
      const float invstd = rsqrtf(var + p.pro.eps);
      const float k = p.pro.gamma[tid] * invstd;
      const float sh = p.pro.beta[tid] - mean * k;
      pro_sc[tid] = k;
      pro_sh[tid] = sh;
      if (bid == 0) {                  // one block publishes what backward / evaluation need
        p.pro.scale[tid] = k;
        p.pro.shift[tid] = sh;
        p.pro.save_mean[tid] = mean;
        p.pro.save_invstd[tid] = invstd;
        if (p.pro.moving_mean != nullptr) {
          const float unbiased = p.pro.P > 1 ? var * ((float)p.pro.P / (float)(p.pro.P - 1)) : var;
          p.pro.moving_mean[tid] = p.pro.moving_mean[tid] * p.pro.momentum + mean * (1.f - p.pro.momentum);
          p.pro.moving_var[tid] = p.pro.moving_var[tid] * p.pro.momentum + unbiased * (1.f - p.pro.momentum);
        }
      }
    }
    __syncthreads();                   // scale / shift visible; `part` has been read
    const bool side = p.pro.z != nullptr && nt == 0;
#pragma unroll
    for (int u = 0; u < HCH; ++u) {
      const int idx = tid + u * NT;
      if (idx >= total) continue;
      u32x4 ov = u32x4{0u, 0u, 0u, 0u};
      if (hok[u]) {
        const int ch = idx % cpr;
        const f32x4 sc0 = *reinterpret_cast<const f32x4*>(pro_sc + ch * 8);
        const f32x4 sc1 = *reinterpret_cast<const f32x4*>(pro_sc + ch * 8 + 4);
        const f32x4 sh0 = *reinterpret_cast<const f32x4*>(pro_sh + ch * 8);
        const f32x4 sh1 = *reinterpret_cast<const f32x4*>(pro_sh + ch * 8 + 4);
        const float sc[8] = {sc0[0], sc0[1], sc0[2], sc0[3], sc1[0], sc1[1], sc1[2], sc1[3]};
        const float sh[8] = {sh0[0], sh0[1], sh0[2], sh0[3], sh1[0], sh1[1], sh1[2], sh1[3]};
        const uint32_t w[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};
        const uint32_t rw[4] = {rv[u].x, rv[u].y, rv[u].z, rv[u].w};
        uint32_t o[4];
        uint32_t bits = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float lo = lo16<DT>(w[j]) * sc[2 * j] + sh[2 * j] + lo16<DT>(rw[j]);
          float hi2 = hi16<DT>(w[j]) * sc[2 * j + 1] + sh[2 * j + 1] + hi16<DT>(rw[j]);
          if (p.pro.relu) {
            bits |= (lo > 0.f ? 1u : 0u) << (2 * j);
            bits |= (hi2 > 0.f ? 1u : 0u) << (2 * j + 1);
            lo = fmaxf(lo, 0.f);
            hi2 = fmaxf(hi2, 0.f);
          }
          o[j] = pack2<DT>(lo, hi2);
        }
        ov = u32x4{o[0], o[1], o[2], o[3]};
        if (side && hown[u]) {           // hok: the column is inside the image as well
          *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.pro.z) + hsrc[u]) = ov;
          if (p.pro.mask != nullptr) p.pro.mask[hsrc[u] >> 3] = (uint8_t)bits;
        }
      }
      *reinterpret_cast<u32x4*>(halo + (size_t)idx * 16) = ov;
    }
  }

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

  // PRO: byte offset of the lane's pixel (mr, fr) in the halo tile, before the tap offset
  int hoff[MR];
  const int cshift = __ffs(p.Cpix) - 1;            // Cpix is a power of two in PRO launches
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    const int prow = wm * WM + mr * 16 + fr;
    hoff[mr] = PRO ? (((prow / p.W) * HW + (prow % p.W)) << (cshift + 1)) : 0;
  }

  f32x4 acc[NR][MR];
#pragma unroll
  for (int nr = 0; nr < NR; ++nr)
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) acc[nr][mr] = f32x4{0.f, 0.f, 0.f, 0.f};

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
  if constexpr (PRO) {
    // stage 0 of the weights was issued by the prologue
  } else if constexpr (MODE == TAPLOOP) {
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
      if constexpr (PRO) {                         // while this one is multiplied
        issue_w(buf ^ 1);
        advance_k();
      } else if constexpr (MODE == TAPLOOP) {
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
    if constexpr (PRO) {
      // this lane's 8 K values of the step: tap (ti, tj), channels chn .. chn + 7.  Past the last
      // tap the weights are zeros; the pixel operand only has to be finite: the last tap again
      const int kk = kt * BK + fq * EPC;
      int tap = kk >> cshift;
      const int chn = kk - (tap << cshift);
      tap = tap < p.TI * p.TJ ? tap : p.TI * p.TJ - 1;
      const int ti = tap / p.TJ, tj = tap - ti * p.TJ;
      const int toff = (((ti * HW + tj) << cshift) + chn) * 2;
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) xf[mr] = *reinterpret_cast<const u32x4*>(halo + hoff[mr] + toff);
    } else {
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) xf[mr] = *reinterpret_cast<const u32x4*>(xs + xoff[mr]);
    }
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
                                                        n0 + wn * WN + fq * 4 * NR, mt, lane);
  else
    conv_epi::conv_epilogue_plain<DT, MR, NR, POINTWISE>(p, acc, m0 + wm * WM,
                                                         n0 + wn * WN + fq * 4 * NR, lane);
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

// The fused epilogues (BatchNorm statistics, folded inference BatchNorm) exist for the 16-bit
// types; fp32 launches take the plain epilogue (bias, addend) and the planner schedules the
// separate statistics sweep / BatchNorm launch for them.
template <int DT, int BM, int BN, int WAVES_M, int WAVES_N>
void launch_cfg16(const ConvParams& p, hipStream_t stream) {
  if (p.scale) launch_cfg_epi<DT, BM, BN, WAVES_M, WAVES_N, EPI_INFER>(p, stream);
  else if (p.stats_y) launch_cfg_epi<DT, BM, BN, WAVES_M, WAVES_N, EPI_BSTATS>(p, stream);
  else launch_cfg_epi<DT, BM, BN, WAVES_M, WAVES_N, EPI_PLAIN>(p, stream);
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
void launch_cfg(const ConvParams& p, hipStream_t stream) {
  if (p.dt == DT_F32) return launch_cfg_epi<DT_F32, BM, BN, WAVES_M, WAVES_N, EPI_PLAIN>(p, stream);
  if (p.dt == DT_F16) return launch_cfg16<DT_F16, BM, BN, WAVES_M, WAVES_N>(p, stream);
  launch_cfg16<DT_BF16, BM, BN, WAVES_M, WAVES_N>(p, stream);
}

// BatchNorm prologue (p.pro.sums): generic gather, plain epilogue (with forward statistics).
template <int BM, int BN, int WAVES_M, int WAVES_N>
void launch_pro(const ConvParams& p, hipStream_t stream) {
  // a block's BM pixels must be whole rows of one image, and their halo tile must fit
  if (BM % p.W || (p.H * p.W) % BM ||
      (BM / p.W + p.TI - 1) * (p.W + p.TJ - 1) * p.Cpix * 2 > 16384)
    throw std::invalid_argument("conv_igemm: this tile cannot hold the layer's rows for the BatchNorm prologue");
  const int tilesN = (p.Cout + BN - 1) / BN;
  const int tilesM = (p.M + BM - 1) / BM;
  const int grid = ((tilesM + 7) / 8) * 8 * tilesN;
  if (p.dt == DT_F16)
    hipLaunchKernelGGL((conv_igemm_kernel<DT_F16, BM, BN, WAVES_M, WAVES_N, GENERIC, EPI_PLAIN, 2, true>),
                       dim3(grid), dim3(64 * WAVES_M * WAVES_N), 0, stream, p);
  else
    hipLaunchKernelGGL((conv_igemm_kernel<DT_BF16, BM, BN, WAVES_M, WAVES_N, GENERIC, EPI_PLAIN, 2, true>),
                       dim3(grid), dim3(64 * WAVES_M * WAVES_N), 0, stream, p);
}

// Ring variants (cfg 8..13): bfloat16 training kernels; other element types and the inference
// epilogue take the two-stage kernel of the same tile.
template <int BM, int BN, int WAVES_M, int WAVES_N, int STAGES>
void launch_ring(const ConvParams& p, hipStream_t stream) {
  if (p.dt != DT_BF16 || p.scale) return launch_cfg<BM, BN, WAVES_M, WAVES_N>(p, stream);
  if (p.stats_y) launch_cfg_epi<DT_BF16, BM, BN, WAVES_M, WAVES_N, EPI_BSTATS, STAGES>(p, stream);
  else launch_cfg_epi<DT_BF16, BM, BN, WAVES_M, WAVES_N, EPI_PLAIN, STAGES>(p, stream);
}

}  // namespace

// Validates the descriptor on the host (a kernel that indexes out of bounds can take the
// whole GPU down, so every shape assumption is checked before the launch).
void validate_conv_params(const ConvParams& p) {
  auto fail = [](const std::string& m) { throw std::invalid_argument("conv_igemm: " + m); };
  if (!p.in || !p.wgt || !p.out) fail("null tensor pointer");
  if (p.dt != DT_BF16 && p.dt != DT_F16 && p.dt != DT_F32) fail("unknown element type");
  if (p.dt == DT_F32 && (p.scale || p.stats || p.relu))
    fail("fused statistics / folded BatchNorm epilogues exist for the 16-bit element types only");
  if (p.N <= 0 || p.H <= 0 || p.W <= 0 || p.OH <= 0 || p.OW <= 0) fail("bad spatial size");
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
    if (p.dt == DT_F32) fail("the BatchNorm prologue exists for the 16-bit element types");
    if (p.scale || p.stats_y || p.addend || p.bias) fail("the BatchNorm prologue goes with the plain training epilogue");
    if (p.Ctap != C || C % 8 || C > 128 || (C & (C - 1))) fail("BatchNorm prologue: taps must be whole pixels of 8..128 (power of two) channels");
    if (p.pro.replicas <= 0 || p.pro.P != p.N * p.H * p.W) fail("BatchNorm prologue: bad replica count / sample count");
    if (!p.pro.gamma || !p.pro.beta || !p.pro.save_mean || !p.pro.save_invstd || !p.pro.scale || !p.pro.shift)
      fail("BatchNorm prologue: missing parameter / output vector");
    if ((p.pro.moving_mean == nullptr) != (p.pro.moving_var == nullptr)) fail("BatchNorm prologue: moving mean and variance come together");
    if (p.pro.mask && !p.pro.z) fail("BatchNorm prologue: the mask is written along with z");
    // unit-stride forward convolution over the same grid, taps centred on the output pixel
    if (p.ISH != 1 || p.ISW != 1 || p.dhs != 1 || p.dws != 1 || p.OH != p.H || p.OW != p.W ||
        p.OSH != 1 || p.OSW != 1 || p.OOH != 0 || p.OOW != 0 || p.OHfull != p.OH || p.OWfull != p.OW ||
        p.dh0 > 0 || -p.dh0 >= p.TI || p.dw0 > 0 || -p.dw0 >= p.TJ)
      fail("BatchNorm prologue: needs a unit-stride convolution whose taps cover the output pixel");
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

// cfg 0..15 and 32..: the kernel of this file; cfg 16..31: the multi-stage pipeline of
// conv_pipe.hip, which hands descriptors that need the generic gather back to the tile of
// this file with the same footprint.
void launch_conv_igemm(const ConvParams& p, int cfg, hipStream_t stream) {
  validate_conv_params(p);
  if (p.pro.sums) {
    switch (cfg) {
      case 1: launch_pro<256, 64, 4, 1>(p, stream); break;
      case 2: launch_pro<256, 32, 4, 1>(p, stream); break;
      case 3: launch_pro<256, 16, 4, 1>(p, stream); break;
      case 5: launch_pro<128, 64, 2, 2>(p, stream); break;
      case 7: launch_pro<128, 16, 4, 1>(p, stream); break;
      case 14: launch_pro<64, 64, 2, 2>(p, stream); break;
      case 15: launch_pro<128, 32, 4, 1>(p, stream); break;
      case 32: launch_pro<64, 32, 4, 1>(p, stream); break;
      case 33: launch_pro<64, 16, 4, 1>(p, stream); break;
      case 34: launch_pro<64, 32, 2, 2>(p, stream); break;
      default: throw std::invalid_argument("conv_igemm: this tile has no BatchNorm-prologue variant");
    }
    return;
  }
  if (cfg >= 16 && cfg < 32) {
    if (p.dt != DT_F32 && launch_conv_pipe(p, cfg, stream)) return;
    cfg = pipe_fallback_cfg(cfg);
  }
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
