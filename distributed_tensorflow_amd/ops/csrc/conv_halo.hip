// Small-layer convolution in one memory round trip (forward and unit-stride data gradient of the
// CIFAR ResNet stages: 32x32x16, 16x16x32, 8x8x64 at batch 128 - tensors of a few megabytes).
//
// Such a layer is not bound by arithmetic or bandwidth but by the chain of dependent memory
// round trips inside each block: the two-buffer kernel of conv_igemm.hip waits for one DMA per
// 32-wide K-step (5 - 18 per block), and measured 7 - 10 us per launch for 0.6 GFLOP.  Here a block
// asks for everything it will ever read at once -
//   * its weight slab [BN channels][K] for every K-step (DMA into dynamic LDS),
//   * its pixels: BM / W whole image rows plus the border the taps reach, as a halo tile
//     [rows + TI - 1][W + TJ - 1][C] in LDS (zeros outside the image),
//   * with PRO, the replica sums of the producer's BatchNorm statistics -
// waits once, and runs the K loop out of LDS without a barrier.  A tap is a shifted view of the
// halo tile, so each pixel is fetched once instead of once per tap.
//
// BPRO (BnBackwardPrologue): the mirror image for a data-gradient launch - the BatchNorm backward
// formula turns dz into dy on the way into the halo tile (constants from the replica sums of the
// launch that produced dz), dy is written on the side for the weight gradient.
//
// PRO (BnPrologue in conv_params.h): the training BatchNorm (+ residual, + ReLU) in front of the
// layer runs on the pixels' way through registers into the halo tile, x = relu?(y * scale + shift
// (+ res)); every block derives scale / shift from the sums, block 0 publishes saved and moving
// statistics, and the blocks of channel tile 0 write the normalised values and ReLU bits of their
// own rows (the operand of this layer's weight gradient, a later residual).  One launch instead of
// two per layer.
//
// GEMM view and register layout are those of conv_igemm.hip (weights are the MFMA A operand, a
// lane ends up with 4*NR consecutive channels of a pixel), so the epilogues of conv_epilogue.h
// apply unchanged.
#include <stdexcept>
#include <string>

#include "common.h"
#include "conv_params.h"
#include "conv_epilogue.h"

namespace {

using conv_epi::EPI_PLAIN;
using conv_epi::EPI_BSTATS;

__device__ __attribute__((aligned(16))) uint32_t g_halo_zero_page[4] = {0u, 0u, 0u, 0u};

constexpr int HALO_BYTES = 16384;

// PMODE: 0 plain, 1 BatchNorm prologue (forward), 2 BatchNorm backward prologue (data gradient)
template <int DT, int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int PMODE>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2)
void conv_halo_kernel(const ConvParams p) {
  constexpr bool PRO = PMODE == 1, BPRO = PMODE == 2;
  typedef typename Elem<DT>::T T;
  static_assert(DT != DT_F32, "16-bit element types");
  constexpr int EPC = 8, BK = 32;
  constexpr int NW = WAVES_M * WAVES_N, NT = 64 * NW;
  constexpr int RPP = 16 * NW;
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MR = WM / 16, NR = WN / 16;
  constexpr int WV = (BN + RPP - 1) / RPP;          // weight-tile DMA instructions per K-step
  constexpr int WBYTES = BN * 64;
  constexpr int HCH = HALO_BYTES / 16 / NT;         // halo chunks per thread at most
  static_assert(NW == 4, "four waves per block");
  static_assert(WM % 16 == 0 && WN % 16 == 0, "tile shape");

  __shared__ __attribute__((aligned(16))) uint8_t halo[HALO_BYTES];
  __shared__ __attribute__((aligned(16))) float pro_sc[PMODE ? 128 : 4];   // BPRO: gamma * invstd
  __shared__ __attribute__((aligned(16))) float pro_sh[PRO ? 128 : 4];
  // BPRO: mean, invstd, dbeta / P, dgamma / P per channel
  __shared__ __attribute__((aligned(16))) float bp_mean[BPRO ? 128 : 4];
  __shared__ __attribute__((aligned(16))) float bp_inv[BPRO ? 128 : 4];
  __shared__ __attribute__((aligned(16))) float bp_c1[BPRO ? 128 : 4];
  __shared__ __attribute__((aligned(16))) float bp_c2[BPRO ? 128 : 4];
  extern __shared__ __attribute__((aligned(16))) uint8_t wres[];     // [KT][BN rows of 64 bytes]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // block -> tile as in conv_igemm.hip: the channel tiles of a pixel tile share blockIdx % 8
  const int tilesN = (p.Cout + BN - 1) / BN;
  const int tilesM = p.M / BM;                      // whole tiles only (host check)
  const int bid = blockIdx.x;
  const int xcd = bid & 7, slot = bid >> 3;
  const int mt = (slot / tilesN) * 8 + xcd;
  const int nt = slot % tilesN;
  if (mt >= tilesM) return;
  const int m0 = mt * BM, n0 = nt * BN;

  const T* __restrict__ gin = reinterpret_cast<const T*>(p.in);
  const T* __restrict__ gw = reinterpret_cast<const T*>(p.wgt);
  const T* zero_page = reinterpret_cast<const T*>(g_halo_zero_page);
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int KT = (p.Ktotal + BK - 1) / BK;

  // ---- 1. every K-step of the weight slab, by DMA ----------------------------------------------
  // lane l of a wave instruction lands at row l/4, 16-byte slot l%4 of the wave's 16 rows; the
  // XOR swizzle of the fragment reads is applied to the source chunk (see conv_igemm.hip)
  {
    const int lrow = tid >> 2;
    const int kvw = (tid & 3) ^ ((-(lrow / (4 * NR))) & 3);
    int kkw = kvw * EPC, kcw = kvw * EPC, tiw = 0, tjw = 0;
    while (kcw >= p.Ctap) { kcw -= p.Ctap; if (++tjw == p.TJ) { tjw = 0; ++tiw; } }
    for (int kt = 0; kt < KT; ++kt) {
      const bool kvalw = kkw < p.Ktotal;
      const int wk = p.wk0 + tiw * p.wsi + tjw * p.wsj + kcw;
      uint8_t* ws = wres + kt * WBYTES;
#pragma unroll
      for (int r = 0; r < WV; ++r) {
        if (RPP * r + wave * 16 < BN) {             // wave-uniform: whole 16-row pieces only
          const int co = n0 + lrow + RPP * r;
          const bool ok = kvalw && co < p.Cout;
          const T* src = ok ? gw + ((size_t)co * p.ldw + wk) : zero_page;
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ws + r * (RPP * 64) + wave * 1024), 16, 0, 0);
        }
      }
      kkw += BK;
      kcw += BK;
      while (kcw >= p.Ctap) { kcw -= p.Ctap; if (++tjw == p.TJ) { tjw = 0; ++tiw; } }
    }
  }

  // ---- 2. the halo tile's pixels, into registers -------------------------------------------------
  // Taps run forwards (forward pass) or backwards (data gradient): tap (i, j) of output pixel
  // (r, c) reads input (r + dh0 + i*dhs, c + dw0 + j*dws); the halo tile starts at the smallest of
  // those offsets.
  const int C = p.Cpix, cpr = C >> 3;
  const int HW = p.W + p.TJ - 1;                    // halo row, pixels
  const int org_h = p.dhs > 0 ? p.dh0 : p.dh0 - (p.TI - 1);
  const int org_w = p.dws > 0 ? p.dw0 : p.dw0 - (p.TJ - 1);
  const int img = p.H * p.W;
  const int n_img = m0 / img;
  const int r0 = (m0 - n_img * img) / p.W;
  const int RB = BM / p.W;
  const int total = (RB + p.TI - 1) * HW * cpr;
  // second tensor read alongside the input: PRO the residual (optional), BPRO the BatchNorm input
  const T* gres = PRO ? reinterpret_cast<const T*>(p.pro.res)
                      : (BPRO ? reinterpret_cast<const T*>(p.bpro.y) : nullptr);
  u32x4 yv[HCH], rv[HCH];
  uint32_t mb[HCH];
  size_t hsrc[HCH];
  bool hok[HCH], hown[HCH];
#pragma unroll
  for (int u = 0; u < HCH; ++u) {
    const int idx = tid + u * NT;
    const int ch = idx % cpr, px = idx / cpr;
    const int wc = px % HW, hr = px / HW;
    const int hi = r0 + hr + org_h, wi = wc + org_w;
    hok[u] = idx < total && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
    hown[u] = (unsigned)(hr + org_h) < (unsigned)RB;         // one of the block's own rows
    hsrc[u] = hok[u] ? ((size_t)(n_img * p.H + hi) * p.W + wi) * C + ch * 8 : 0;
    yv[u] = hok[u] ? *reinterpret_cast<const u32x4*>(gin + hsrc[u]) : u32x4{0u, 0u, 0u, 0u};
    rv[u] = (hok[u] && gres != nullptr) ? *reinterpret_cast<const u32x4*>(gres + hsrc[u])
                                        : u32x4{0u, 0u, 0u, 0u};
    mb[u] = 0xffu;
    if constexpr (BPRO) {
      if (hok[u] && p.bpro.mask != nullptr) mb[u] = p.bpro.mask[hsrc[u] >> 3];
    }
  }

  // ---- 3. PRO: scale / shift of the input's channels from the producer's sums --------------------
  if constexpr (PRO) {
    // column sums of [replicas][2*C] (2*C <= 256 = threads, checked on the host): thread
    // (row lane rl, column col) adds every RL-th replica row, one LDS round joins the row lanes
    const int cols = 2 * C;
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
  }

  if constexpr (BPRO) {
    // the same column sums, of [sum g | sum g*y]; then the constants of the backward formula
    const int cols = 2 * C;
    const int RL = NT / cols;
    const int col = tid % cols, rl = tid / cols;
    float* part = reinterpret_cast<float*>(halo);
    if (rl < RL) {
      float sum = 0.f;
      for (int g = rl; g < p.bpro.replicas; g += RL) sum += p.bpro.sums[(size_t)g * cols + col];
      part[rl * cols + col] = sum;
    }
    __syncthreads();
    if (tid < C) {
      float sg = 0.f, sgy = 0.f;
      for (int r = 0; r < RL; ++r) { sg += part[r * cols + tid]; sgy += part[r * cols + C + tid]; }
      const float mean = p.bpro.mean[tid], invstd = p.bpro.invstd[tid];
      const float dgamma = (sgy - mean * sg) * invstd;
      const float invP = 1.f / (float)p.bpro.P;
      pro_sc[tid] = p.bpro.gamma[tid] * invstd;
      bp_mean[tid] = mean;
      bp_inv[tid] = invstd;
      bp_c1[tid] = sg * invP;
      bp_c2[tid] = dgamma * invP;
      if (bid == 0) {                  // plain stores into the flat gradient buffer
        p.bpro.dbeta[tid] = sg;
        p.bpro.dgamma[tid] = dgamma;
      }
    }
    __syncthreads();
  }

  // ---- 4. halo tile into LDS (normalised on the way with PRO) -------------------------------------
  const bool side = PRO && p.pro.z != nullptr && nt == 0;
#pragma unroll
  for (int u = 0; u < HCH; ++u) {
    const int idx = tid + u * NT;
    if (idx >= total) continue;
    u32x4 ov = yv[u];                  // zeros outside the image
    if constexpr (PRO) {
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
    }
    if constexpr (BPRO) {
      if (hok[u]) {
        const int c0 = (idx % cpr) * 8;
        auto ld8 = [&](const float* a, float (&v)[8]) {
          const f32x4 lo = *reinterpret_cast<const f32x4*>(a + c0);
          const f32x4 hi = *reinterpret_cast<const f32x4*>(a + c0 + 4);
          v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
          v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        };
        float sc[8], mean[8], inv[8], c1[8], c2[8];
        ld8(pro_sc, sc); ld8(bp_mean, mean); ld8(bp_inv, inv); ld8(bp_c1, c1); ld8(bp_c2, c2);
        const uint32_t w[4] = {yv[u].x, yv[u].y, yv[u].z, yv[u].w};       // dz
        const uint32_t yw[4] = {rv[u].x, rv[u].y, rv[u].z, rv[u].w};      // y
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float g0 = (mb[u] >> (2 * j)) & 1u ? lo16<DT>(w[j]) : 0.f;
          const float g1 = (mb[u] >> (2 * j + 1)) & 1u ? hi16<DT>(w[j]) : 0.f;
          const float x0 = (lo16<DT>(yw[j]) - mean[2 * j]) * inv[2 * j];
          const float x1 = (hi16<DT>(yw[j]) - mean[2 * j + 1]) * inv[2 * j + 1];
          o[j] = pack2<DT>(sc[2 * j] * (g0 - c1[2 * j] - x0 * c2[2 * j]),
                           sc[2 * j + 1] * (g1 - c1[2 * j + 1] - x1 * c2[2 * j + 1]));
        }
        ov = u32x4{o[0], o[1], o[2], o[3]};
        if (nt == 0 && hown[u])
          *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.bpro.dy) + hsrc[u]) = ov;
      }
    }
    *reinterpret_cast<u32x4*>(halo + (size_t)idx * 16) = ov;
  }

  // ---- 5. fragment addresses ------------------------------------------------------------------------
  const int fr = lane & 15, fq = lane >> 4;
  const int cshift = __ffs(C) - 1;                 // C is a power of two (host check)
  int woff[NR], hoff[MR];
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
    // MFMA row fr = 4*q' + r' of tile nr carries output channel q'*(4*NR) + nr*4 + r'
    const int row = wn * WN + (fr >> 2) * (4 * NR) + nr * 4 + (fr & 3);
    woff[nr] = row * 64 + ((fq ^ ((-(row / (4 * NR))) & 3)) << 4);
  }
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    const int prow = wm * WM + mr * 16 + fr;         // pixel of the tile -> (row, column)
    hoff[mr] = ((prow / p.W) * HW + (prow % p.W)) << (cshift + 1);
  }
  f32x4 acc[NR][MR];
#pragma unroll
  for (int nr = 0; nr < NR; ++nr)
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) acc[nr][mr] = f32x4{0.f, 0.f, 0.f, 0.f};

  dma_wait_all();                      // this wave's weight DMA has landed ...
  __syncthreads();                     // ... everybody's has, and the halo tile is written

  // ---- 6. K loop out of LDS: no barrier, no memory traffic -------------------------------------------
  const int ntaps = p.TI * p.TJ;
  for (int kt = 0; kt < KT; ++kt) {
    const uint8_t* ws = wres + kt * WBYTES;
    u32x4 wf[NR], xf[MR];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) wf[nr] = *reinterpret_cast<const u32x4*>(ws + woff[nr]);
    // this lane's 8 K values of the step: tap (ti, tj), channels chn .. chn + 7.  Past the last
    // tap the weights are zeros; the pixel operand only has to be finite: the last tap again
    const int kk = kt * BK + fq * EPC;
    int tap = kk >> cshift;
    const int chn = kk - (tap << cshift);
    tap = tap < ntaps ? tap : ntaps - 1;
    const int ti = tap / p.TJ, tj = tap - ti * p.TJ;
    const int trow = p.dh0 + ti * p.dhs - org_h, tcol = p.dw0 + tj * p.dws - org_w;
    const int toff = (((trow * HW + tcol) << cshift) + chn) * 2;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) xf[mr] = *reinterpret_cast<const u32x4*>(halo + hoff[mr] + toff);
#pragma unroll
    for (int nr = 0; nr < NR; ++nr)
#pragma unroll
      for (int mr = 0; mr < MR; ++mr)
        acc[nr][mr] = mfma_step<DT>(wf[nr], xf[mr], acc[nr][mr]);
  }

  conv_epi::conv_epilogue<MR, NR, false, EPI, DT>(p, acc, m0 + wm * WM, n0 + wn * WN + fq * 4 * NR,
                                                  mt, lane);
}

template <int DT, int BM, int BN, int WAVES_M, int WAVES_N>
void launch_halo_dt(const ConvParams& p, hipStream_t stream) {
  const int tilesN = (p.Cout + BN - 1) / BN;
  const int tilesM = p.M / BM;
  const int grid = ((tilesM + 7) / 8) * 8 * tilesN;
  const int KT = (p.Ktotal + 31) / 32;
  const size_t dyn = (size_t)KT * BN * 64;
  const dim3 block(64 * WAVES_M * WAVES_N);
  if (p.pro.sums)
    hipLaunchKernelGGL((conv_halo_kernel<DT, BM, BN, WAVES_M, WAVES_N, EPI_PLAIN, 1>), dim3(grid),
                       block, dyn, stream, p);
  else if (p.bpro.sums && p.stats_y)
    hipLaunchKernelGGL((conv_halo_kernel<DT, BM, BN, WAVES_M, WAVES_N, EPI_BSTATS, 2>), dim3(grid),
                       block, dyn, stream, p);
  else if (p.bpro.sums)
    hipLaunchKernelGGL((conv_halo_kernel<DT, BM, BN, WAVES_M, WAVES_N, EPI_PLAIN, 2>), dim3(grid),
                       block, dyn, stream, p);
  else if (p.stats_y)
    hipLaunchKernelGGL((conv_halo_kernel<DT, BM, BN, WAVES_M, WAVES_N, EPI_BSTATS, 0>), dim3(grid),
                       block, dyn, stream, p);
  else
    hipLaunchKernelGGL((conv_halo_kernel<DT, BM, BN, WAVES_M, WAVES_N, EPI_PLAIN, 0>), dim3(grid),
                       block, dyn, stream, p);
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
bool launch_halo(const ConvParams& p, hipStream_t stream) {
  // a block's BM pixels must be whole rows of one image; its halo tile and weight slab must fit
  const int KT = (p.Ktotal + 31) / 32;
  if (BM % p.W || (p.H * p.W) % BM ||
      (BM / p.W + p.TI - 1) * (p.W + p.TJ - 1) * p.Cpix * 2 > HALO_BYTES ||
      (size_t)KT * BN * 64 > 40960)
    return false;
  if (p.dt == DT_F16) launch_halo_dt<DT_F16, BM, BN, WAVES_M, WAVES_N>(p, stream);
  else launch_halo_dt<DT_BF16, BM, BN, WAVES_M, WAVES_N>(p, stream);
  return true;
}

}  // namespace

// True when the descriptor is one this kernel family serves at all (tile-independent part).
bool conv_halo_supports(const ConvParams& p) {
  const int C = p.Cpix;
  return p.dt != DT_F32 && !p.scale && p.Ctap == C && C >= 8 && C <= 128 && (C & (C - 1)) == 0 &&
         p.ISH == 1 && p.ISW == 1 && (p.dhs == 1 || p.dhs == -1) && (p.dws == 1 || p.dws == -1) &&
         p.OH == p.H && p.OW == p.W && p.OSH == 1 && p.OSW == 1 && p.OOH == 0 && p.OOW == 0 &&
         p.OHfull == p.OH && p.OWfull == p.OW && p.M == p.N * p.H * p.W;
}

// cfg 40..47.  False: the tile cannot serve this descriptor (the caller falls back, or fails when
// the launch carries a BatchNorm prologue, which only this kernel implements).
bool launch_conv_halo(const ConvParams& p, int cfg, hipStream_t stream) {
  if (!conv_halo_supports(p)) return false;
  switch (cfg) {
    //                           BM   BN  waves
    case 40: return launch_halo<256, 16, 4, 1>(p, stream);
    case 41: return launch_halo<128, 16, 4, 1>(p, stream);
    case 42: return launch_halo<64, 16, 4, 1>(p, stream);
    case 43: return launch_halo<256, 32, 4, 1>(p, stream);
    case 44: return launch_halo<128, 32, 4, 1>(p, stream);
    case 45: return launch_halo<64, 32, 4, 1>(p, stream);
    case 46: return launch_halo<64, 32, 2, 2>(p, stream);
    case 47: return launch_halo<64, 64, 2, 2>(p, stream);
    default: return false;
  }
}

int halo_fallback_cfg(int cfg) {       // two-buffer tile of the same footprint (conv_igemm.hip)
  switch (cfg) {
    case 40: return 3;
    case 41: return 7;
    case 42: return 33;
    case 43: return 2;
    case 44: return 15;
    case 45: return 32;
    case 46: return 34;
    default: return 14;
  }
}
