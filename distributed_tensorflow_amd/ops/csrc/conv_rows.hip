// 3x3 unit-stride convolution over whole image rows, every input pixel fetched once.
//
// The implicit GEMM of conv_igemm.hip / conv_pipe.hip gathers the pixel operand of a K-step from
// global memory - for a 3x3 layer each input pixel travels L2 -> LDS once per tap, nine times.
// The large 3x3 layers of ResNet-50 (56x56x64, 28x28x128) are bound by exactly that traffic: a
// 256x256x32 K-step moves 32 KB for 4.2 MFLOP, and the matrix cores sit at 37 % while the waves
// wait for operand tiles.  Here a block owns TR output rows of ONE image:
//
//   * its input rows plus the border the taps reach - (TR + 2) x (W + 2) pixels x C channels -
//     are fetched once, by DMA, into a halo tile that stays in LDS for the whole K loop (pixels
//     outside the image come from a page of zeros);
//   * a tap is a shifted view of that tile: the pixel fragment of K-step (tap, channel chunk)
//     is read at  halo[(row + ti) * (W + 2) + col + tj][chunk];
//   * only the weights stream: a ring of STAGES tiles [SPS][BN][32] (SPS K-steps per stage)
//     filled by DMA with counted vmcnt waits and one raw barrier per stage, as in conv_pipe.hip.
//     (Loading each wave's weight fragments straight into registers instead - no ring, no
//     barrier in the K loop, three blocks per CU - measured slower: 96 - 100 us against 76 us on
//     the 56x56x64 layer, whatever the prefetch depth.)
//
// L2 -> LDS traffic per block drops from 9 pixel tiles + weights to 1.3 pixel tiles
// + weights (56x56x64, 4 rows: 118 KB instead of 330 KB for 16.5 MFLOP).
//
// LDS image of the halo tile: a 256-byte bank row holds TWO adjacent 16-byte channel chunks of 8
// consecutive halo pixels, chunk-parity major -
//   offset(pixel, chunk) = ((pixel / 8) * C/16 + chunk / 2) * 256 + (chunk % 2) * 128 + (pixel % 8) * 16.
// A ds_read_b128 is served in lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...: with the
// MFMA operand layout (lane = 16 * chunk-of-K-step + pixel) a group holds pixel lanes 0-3 and 12-15
// with an even chunk and 4-11 with the odd one next to it (or the other way round).  Either set
// is 8 consecutive pixels modulo 16, hence 8 distinct residues modulo 8, and the chunk parity
// picks the half of the bank row: 16 distinct 16-byte slots whatever the tap shift.  (A
// pixel-major image with XOR-swizzled chunks is conflict free only for some alignments of the
// shifted tile.)  The chunk pair of a K-step is an immediate offset of the read.  The DMA writes
// LDS linearly, so the layout is applied to the source address; one DMA instruction still
// covers whole 128-byte lines of 8 pixels.
//
// GEMM view and register layout are those of conv_igemm.hip (weights are the MFMA A operand, a
// lane ends up with 4 * NR consecutive channels of a pixel).  A block's output pixels are
// contiguous in the NHWC tensor, so the epilogues of conv_epilogue.h apply in their pointwise
// form: bias-free training epilogue with fused BatchNorm statistics, masked addend, and the
// backward statistics of a data gradient.
//
// Serves: TI = TJ = 3, unit stride, SAME padding (forward taps or the reversed taps of a data
// gradient), Ctap == Cpix == C, OH == H, OW == W, H % TR == 0, 16-bit element types.
#include <stdexcept>
#include <string>

#include "common.h"
#include "conv_epilogue.h"
#include "conv_params.h"

namespace {

using conv_epi::EPI_BSTATS;
using conv_epi::EPI_PLAIN;

__device__ __attribute__((aligned(16))) uint32_t g_rows_zero_page[4] = {0u, 0u, 0u, 0u};

// halo pixels rounded up to groups of 8, C / 16 bank rows of 256 bytes per group; rounded to
// whole DMA rounds of 256 threads x 16 bytes
constexpr int rows_halo_bytes(int c, int w, int tr) {
  return ((((tr + 2) * (w + 2) + 7) / 8) * (c / 16) * 256 + 4095) / 4096 * 4096;
}

template <int N>
__device__ __forceinline__ void rows_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// At most N of this wave's DMA loads still in flight and its LDS reads retired, then the block
// meets - one statement, so that the scheduler puts nothing between the waits and the barrier.
template <int N>
__device__ __forceinline__ void rows_wait_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}

// STAGES: depth of the weight ring; SPS: K-steps (of 32) per stage, i.e. per barrier.
// OCC: waves per SIMD the register allocation has to allow (blocks per CU the LDS admits).
// EARLY (two stages of one K-step): every wave first moves its weight fragments of the stage into
// registers, the block meets a second time and the stage is refilled at once with the step after
// next - a refill then has two steps to arrive instead of one, with no third stage in LDS.
template <int DT, int C, int W, int TR, int BN, int WAVES_M, int WAVES_N, int STAGES, int SPS,
          int OCC, int EPI, bool EARLY = false>
__global__ __launch_bounds__(256, OCC)
void conv_rows_kernel(const ConvParams p) {
  static_assert(!EARLY || (STAGES == 2 && SPS == 1), "early refill: two stages of one K-step");
  static_assert(DT == DT_BF16 || DT == DT_F16, "16-bit element types");
  static_assert(WAVES_M * WAVES_N == 4, "four waves per block");
  static_assert(C == 64 || C == 128 || C == 256, "channels per pixel");
  constexpr int NW = 4;
  constexpr int BM = TR * W;                         // output pixels of the block
  constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
  constexpr int MR = WM / 16, NR = WN / 16;
  static_assert(BM % (16 * WAVES_M) == 0 && BN % (16 * WAVES_N) == 0, "tile shape");
  constexpr int HW = W + 2;                          // halo row, pixels
  constexpr int CPP = C / 8;                         // 16-byte chunks per pixel
  constexpr int HBYTES = rows_halo_bytes(C, W, TR);
  constexpr int HROUNDS = HBYTES / 4096;             // DMA instructions per wave for the halo tile
  constexpr int SBYTES = BN * 64;                    // one K-step of weights: BN rows of 32 K
  constexpr int WBYTES = SPS * SBYTES;               // one ring stage
  constexpr int WV = SPS * (BN / 64);                // weight DMA instructions per wave and stage
  constexpr int SPT = C / 32;                        // K-steps per tap
  constexpr int GPT = SPT / SPS;                     // stages per tap
  constexpr int NSTAGE = 9 * GPT;                    // stages in all
  static_assert(BN % 64 == 0, "every wave issues the same weight loads");
  static_assert(SPT % SPS == 0, "a stage never straddles taps");
  static_assert(HROUNDS + (STAGES - 1) * WV < 64, "vmcnt range");
  static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");

  // all LDS in one array (a second __shared__ object makes the compiler drain the DMA early)
  __shared__ __attribute__((aligned(16))) uint8_t smem[HBYTES + STAGES * WBYTES];
  uint8_t* const halo = smem;
  uint8_t* const wring = smem + HBYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WAVES_N, wn = wave % WAVES_N;

  // ---- block -> (row tile, channel tile): consecutive row tiles share an XCD's L2 (their halo
  // rows overlap), the channel tiles of a row tile sit next to each other -----------------------
  const int tilesN = p.Cout / BN;
  const int tilesM = p.M / BM;
  const int bid = blockIdx.x;
  int mt, nt;
  {
    const int total = tilesM * tilesN;
    int lin = bid;
    if (total % 8 == 0) lin = (bid & 7) * (total >> 3) + (bid >> 3);
    mt = lin / tilesN;
    nt = lin - mt * tilesN;
  }
  if (p.rev) mt = tilesM - 1 - mt;
  const int m0 = mt * BM, n0 = nt * BN;
  const int tiles_per_img = p.H / TR;
  const int n_img = mt / tiles_per_img;
  const int r0 = (mt - n_img * tiles_per_img) * TR;

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const bf16_t* __restrict__ gin = reinterpret_cast<const bf16_t*>(p.in);
  const bf16_t* __restrict__ gw = reinterpret_cast<const bf16_t*>(p.wgt);
  const bf16_t* zero_page = reinterpret_cast<const bf16_t*>(g_rows_zero_page);

  // taps run forwards (forward pass) or backwards (data gradient): tap (i, j) of output pixel
  // (r, c) reads input (r + dh0 + i*dhs, c + dw0 + j*dws); the halo tile starts at the smallest
  // of those offsets
  const int org_h = p.dhs > 0 ? p.dh0 : p.dh0 - 2;
  const int org_w = p.dws > 0 ? p.dw0 : p.dw0 - 2;

  // ---- 1. the halo tile, by DMA: 16-byte position q = bank row (q / 16) x slot (q % 16) holds
  // chunk 2 * (row % (CPP / 2)) + slot / 8 of halo pixel 8 * (row / (CPP / 2)) + slot % 8 -----------------
  constexpr int HPIX = (TR + 2) * HW;
#pragma unroll
  for (int it = 0; it < HROUNDS; ++it) {
    const int q = (it * NW + wave) * 64 + lane;
    const int brow = q >> 4, slot = q & 15;
    const int grp = brow / (CPP / 2), cp = brow - grp * (CPP / 2);
    const int hp = grp * 8 + (slot & 7), ch = cp * 2 + (slot >> 3);
    const int hr = hp / HW, hc = hp - hr * HW;
    const int hi = r0 + hr + org_h, wi = hc + org_w;
    const bool ok = hp < HPIX && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)W;
    const bf16_t* src = ok ? gin + (((size_t)(n_img * p.H + hi) * W + wi) * C + (ch << 3)) : zero_page;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(halo + (it * NW + wave) * 1024), 16, 0, 0);
  }

  // ---- 1b. BatchNorm + ReLU of the layer in front, inside the tile: every thread transforms the
  // 16-byte positions it fetched (always the same eight channels: the chunk index of a position
  // depends on lane and wave only), out-of-image pixels stay zero.  The ring's first loads are
  // issued behind this pass so that the counted wait below covers the halo loads alone.
  if (p.in_scale != nullptr) {
    constexpr int CH_PERIOD = CPP / 2;               // bank rows after which the chunk pair repeats
    static_assert((NW * 4) % CH_PERIOD == 0, "a thread's positions share their channel chunk");
    const int ch = ((((wave * 4 + (lane >> 4)) % CH_PERIOD) << 1) | ((lane & 15) >> 3));
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      sc[e] = p.in_scale[ch * 8 + e];
      sh[e] = p.in_shift[ch * 8 + e];
    }
    rows_wait_vmcnt<0>();                            // own halo loads have landed (own positions only)
#pragma unroll
    for (int it = 0; it < HROUNDS; ++it) {
      const int q = (it * NW + wave) * 64 + lane;
      const int brow = q >> 4, slot = q & 15;
      const int grp = brow / (CPP / 2);
      const int hp = grp * 8 + (slot & 7);
      const int hr = hp / HW, hc = hp - hr * HW;
      const int hi = r0 + hr + org_h, wi = hc + org_w;
      const bool ok = hp < HPIX && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)W;
      if (ok) {
        u32x4* cell = reinterpret_cast<u32x4*>(halo + (size_t)q * 16);
        const u32x4 v = *cell;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float lo = fmaxf(lo16<DT>(w[j]) * sc[2 * j] + sh[2 * j], 0.f);
          const float hi2 = fmaxf(hi16<DT>(w[j]) * sc[2 * j + 1] + sh[2 * j + 1], 0.f);
          o[j] = pack2<DT>(lo, hi2);
        }
        *cell = u32x4{o[0], o[1], o[2], o[3]};
      }
    }
  }

  // ---- 2. weight ring: lane l of a wave instruction lands at row l/4, slot l%4 of the wave's 16
  // rows; XOR swizzle of the fragment reads applied to the source chunk (conv_igemm.hip) ----------
  const int lrow = tid >> 2;                         // 0 .. 63
  const int kvw = (tid & 3) ^ ((-(lrow / (4 * NR))) & 3);
  const bf16_t* wsrc[BN / 64];
#pragma unroll
  for (int r = 0; r < BN / 64; ++r)
    wsrc[r] = gw + ((size_t)(n0 + lrow + 64 * r) * p.ldw + p.wk0 + kvw * 8);
  int wtap_i = 0, wtap_j = 0, wgrp = 0;              // tap and stage inside it of the next stage to issue
  auto issue_w = [&](int stage) {
    uint8_t* ws = wring + stage * WBYTES;
    const int koff = wtap_i * p.wsi + wtap_j * p.wsj + wgrp * (32 * SPS);
#pragma unroll
    for (int sub = 0; sub < SPS; ++sub)
#pragma unroll
      for (int r = 0; r < BN / 64; ++r)
        __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[r] + koff + sub * 32),
                                         (lptr_t)(ws + sub * SBYTES + r * 4096 + wave * 1024), 16, 0, 0);
    if (++wgrp == GPT) {
      wgrp = 0;
      if (++wtap_j == 3) { wtap_j = 0; ++wtap_i; }
    }
  };
#pragma unroll
  for (int s = 0; s < (EARLY ? STAGES : STAGES - 1); ++s) issue_w(s);

  // ---- 3. fragment addresses --------------------------------------------------------------------
  const int fr = lane & 15, fq = lane >> 4;
  int woff[NR];
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
    // MFMA row fr = 4*q' + r' of tile nr carries output channel q'*(4*NR) + nr*4 + r'
    const int row = wn * WN + (fr >> 2) * (4 * NR) + nr * 4 + (fr & 3);
    woff[nr] = row * 64 + ((fq ^ ((-(row / (4 * NR))) & 3)) << 4);
  }
  int hp0[MR];                                       // halo pixel of the lane's output pixel, tap (0, 0)
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    const int px = wm * WM + mr * 16 + fr;
    const int row = px / W, col = px - row * W;
    hp0[mr] = row * HW + col;
  }

  f32x4 acc[NR][MR];
#pragma unroll
  for (int nr = 0; nr < NR; ++nr)
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) acc[nr][mr] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- 4. K loop: taps outside, stages of SPS channel chunks inside --------------------------------
  int read_stage = 0, issue_stage = STAGES - 1, sidx = 0;
  const int fqoff = ((fq >> 1) << 8) + ((fq & 1) << 7);    // the lane's chunk inside a group of eight pixels
  for (int ti = 0; ti < 3; ++ti) {
    for (int tj = 0; tj < 3; ++tj) {
      const int trow = p.dh0 + ti * p.dhs - org_h, tcol = p.dw0 + tj * p.dws - org_w;
      const int shift = trow * HW + tcol;
      int hbase[MR];                   // the lane's pixel and its chunk fq of the tap's first K-step
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) {
        // (((hp >> 3) * (CPP / 2) + (fq >> 1)) << 8) + ((fq & 1) << 7) + ((hp & 7) << 4), in three operations
        const int hp = hp0[mr] + shift;
        hbase[mr] = (int)__umul24((unsigned)hp >> 3, CPP * 128 - 128) + ((hp << 4) + fqoff);     // hp >= 0
      }
#pragma unroll
      for (int gs = 0; gs < GPT; ++gs, ++sidx) {
        // stages issued after this one that may stay in flight while it is multiplied
        const int left = NSTAGE - 1 - sidx;
        if constexpr (EARLY) {
          // this stage (and, the first time, the halo tile) has landed for every wave; the one issued
          // after it may still be in flight
          if (left >= 1) rows_wait_barrier<WV>();
          else rows_wait_barrier<0>();
          const uint8_t* ws = wring + read_stage * WBYTES;
          u32x4 wf[NR], xf[MR];
#pragma unroll
          for (int nr = 0; nr < NR; ++nr) wf[nr] = *reinterpret_cast<const u32x4*>(ws + woff[nr]);
          // every wave holds its fragments of the stage: refill it with the step after next (the
          // vector-memory wait is the one above again: nothing new is in flight)
          if (left >= 1) rows_wait_barrier<WV>();
          else rows_wait_barrier<0>();
          if (sidx + 2 < NSTAGE) issue_w(read_stage);
          read_stage ^= 1;
#pragma unroll
          for (int mr = 0; mr < MR; ++mr)
            xf[mr] = *reinterpret_cast<const u32x4*>(halo + hbase[mr] + gs * 512);
#pragma unroll
          for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
              acc[nr][mr] = mfma_step<DT>(wf[nr], xf[mr], acc[nr][mr]);
          continue;
        }
        // own LDS reads of the previous stage retired, then the block meets: this stage (and, the
        // first time, the halo tile issued before it) has landed for every wave and nobody still
        // reads the stage that is refilled next
        if (STAGES >= 4 && left >= 2) rows_wait_barrier<(STAGES >= 4 ? 2 : 0) * WV>();
        else if (STAGES >= 3 && left >= 1) rows_wait_barrier<(STAGES >= 3 ? 1 : 0) * WV>();
        else rows_wait_barrier<0>();
        if (sidx + STAGES - 1 < NSTAGE) {
          issue_w(issue_stage);
          if (++issue_stage == STAGES) issue_stage = 0;
        }
        const uint8_t* ws = wring + read_stage * WBYTES;
        if (++read_stage == STAGES) read_stage = 0;
#pragma unroll
        for (int sub = 0; sub < SPS; ++sub) {
          u32x4 wf[NR], xf[MR];
#pragma unroll
          for (int nr = 0; nr < NR; ++nr)
            wf[nr] = *reinterpret_cast<const u32x4*>(ws + sub * SBYTES + woff[nr]);
#pragma unroll
          for (int mr = 0; mr < MR; ++mr)      // K-step cc of the tap: two bank rows further
            xf[mr] = *reinterpret_cast<const u32x4*>(halo + hbase[mr] + (gs * SPS + sub) * 512);
#pragma unroll
          for (int nr = 0; nr < NR; ++nr)
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
              acc[nr][mr] = mfma_step<DT>(wf[nr], xf[mr], acc[nr][mr]);
        }
      }
    }
  }

  conv_epi::conv_epilogue<MR, NR, true, EPI, DT>(p, acc, m0 + wm * WM, n0 + wn * WN + fq * 4 * NR,
                                                 mt, lane);
}

template <int DT, int C, int W, int TR, int BN, int WAVES_M, int WAVES_N, int STAGES, int SPS, int OCC,
          bool EARLY>
void launch_rows_dt(const ConvParams& p, hipStream_t stream) {
  const int grid = (p.M / (TR * W)) * (p.Cout / BN);
  if (p.stats_y)
    hipLaunchKernelGGL((conv_rows_kernel<DT, C, W, TR, BN, WAVES_M, WAVES_N, STAGES, SPS, OCC, EPI_BSTATS, EARLY>),
                       dim3(grid), dim3(256), 0, stream, p);
  else
    hipLaunchKernelGGL((conv_rows_kernel<DT, C, W, TR, BN, WAVES_M, WAVES_N, STAGES, SPS, OCC, EPI_PLAIN, EARLY>),
                       dim3(grid), dim3(256), 0, stream, p);
}

// `probe`: only answers whether the tile serves the descriptor
template <int C, int W, int TR, int BN, int WAVES_M, int WAVES_N, int STAGES, int SPS, int OCC, bool EARLY = false>
bool launch_rows(const ConvParams& p, hipStream_t stream, bool probe) {
  if (p.Cpix != C || p.W != W || p.H % TR || p.Cout % BN) return false;
  if (probe) return true;
  if (p.dt == DT_F16) launch_rows_dt<DT_F16, C, W, TR, BN, WAVES_M, WAVES_N, STAGES, SPS, OCC, EARLY>(p, stream);
  else launch_rows_dt<DT_BF16, C, W, TR, BN, WAVES_M, WAVES_N, STAGES, SPS, OCC, EARLY>(p, stream);
  return true;
}

// ---- the ImageNet stem: 7x7 stride 2 on a 4-channel image stored with its zero border ---------------
//
// In the merged-tap form of the stem (conv_plan.py) a kernel row is ONE K-step: the 8 x 4
// (column tap, channel) values of output pixel (r, c) are the 64 contiguous bytes at input pixel
// (2 r + i, 2 c).  Neighbouring output pixels share 48 of those 64 bytes and the seven kernel rows
// share input rows, so the generic gather moves 448 bytes through the vector memory path per
// output pixel for 16 bytes of new input (1.4 GB per launch at batch 256: the layer ran at the
// rate of the L2, 243 us where its 490 MB of HBM traffic take 100).  Here a block keeps the whole
// filter bank in LDS (7 K-steps x [64 channels][32] = 28 KB, fetched once) and walks over tiles
// of two output rows: the nine input rows of a tile are fetched once (18 KB), a lane's operand of
// K-step i is the 16 bytes at  row (2 r + i), byte 16 (c + fq)  - consecutive lanes read
// consecutive 16-byte slots, K-step and row are immediate offsets - and the tile is multiplied
// without a barrier.  The next tile's rows are requested before the epilogue of this one.
constexpr int STEM_OW = 112, STEM_BN = 64, STEM_PITCH = 2048, STEM_HROUNDS = 5, STEM_KS = 7;

template <int DT, int EPI>
__global__ __launch_bounds__(256, 3)
void conv_stem_rows_kernel(const ConvParams p, int tiles_per_block, int ntiles) {
  constexpr int MR = 7, NR = 2, WN = 32;
  constexpr int HBYTES = STEM_HROUNDS * 4096, SBYTES = STEM_BN * 64;
  __shared__ __attribute__((aligned(16))) uint8_t smem[HBYTES + STEM_KS * SBYTES];
  uint8_t* const halo = smem;
  uint8_t* const wts = smem + HBYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int t0 = blockIdx.x * tiles_per_block;
  const int t1 = min(t0 + tiles_per_block, ntiles);
  if (t0 >= t1) return;

  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const bf16_t* __restrict__ gin = reinterpret_cast<const bf16_t*>(p.in);
  const bf16_t* __restrict__ gw = reinterpret_cast<const bf16_t*>(p.wgt);
  const bf16_t* zero_page = reinterpret_cast<const bf16_t*>(g_rows_zero_page);
  const int tiles_per_img = p.OH >> 1;
  const int row_chunks = (p.W * 4) >> 3;             // 16-byte chunks of a stored input row

  // 16-byte position q of the tile image = chunk (q % 128) of input row 2 r0 + q / 128 (r0: the
  // tile's first output row)
  auto issue_halo = [&](int t) {
    const int tt = p.rev ? ntiles - 1 - t : t;
    const int n = tt / tiles_per_img, r0 = (tt - n * tiles_per_img) * 2;
#pragma unroll
    for (int it = 0; it < STEM_HROUNDS; ++it) {
      const int q = (it * 4 + wave) * 64 + lane;
      const int hr = q >> 7, ch = q & 127;
      const bool ok = hr < 9 && ch < row_chunks;
      const bf16_t* src = ok ? gin + (((size_t)(n * p.H + 2 * r0 + hr) * p.W) * 4 + ch * 8) : zero_page;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(halo + (it * 4 + wave) * 1024), 16, 0, 0);
    }
  };

  // ---- the filter bank, once: K-step i = kernel row i, rows = output channels, swizzled chunks ----
  {
    const int lrow = tid >> 2;
    const int kvw = (tid & 3) ^ ((-(lrow / (4 * NR))) & 3);
    const bf16_t* wsrc = gw + ((size_t)lrow * p.ldw + p.wk0 + kvw * 8);
#pragma unroll
    for (int i = 0; i < STEM_KS; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + i * p.wsi), (lptr_t)(wts + i * SBYTES + wave * 1024),
                                       16, 0, 0);
  }
  issue_halo(t0);

  const int fr = lane & 15, fq = lane >> 4;
  int woff[NR];
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
    const int row = wn * WN + (fr >> 2) * (4 * NR) + nr * 4 + (fr & 3);
    woff[nr] = row * 64 + ((fq ^ ((-(row / (4 * NR))) & 3)) << 4);
  }
  const int xoff = (2 * wm) * STEM_PITCH + ((fr + fq) << 4);       // + mr * 256 + i * STEM_PITCH

  for (int t = t0; t < t1; ++t) {
    rows_wait_vmcnt<0>();
    asm volatile("s_barrier" ::: "memory");            // this tile's rows (and the filters) have landed
    f32x4 acc[NR][MR];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr)
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) acc[nr][mr] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < STEM_KS; ++i) {
      u32x4 wf[NR], xf[MR];
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) wf[nr] = *reinterpret_cast<const u32x4*>(wts + i * SBYTES + woff[nr]);
#pragma unroll
      for (int mr = 0; mr < MR; ++mr)
        xf[mr] = *reinterpret_cast<const u32x4*>(halo + xoff + mr * 256 + i * STEM_PITCH);
#pragma unroll
      for (int nr = 0; nr < NR; ++nr)
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) acc[nr][mr] = mfma_step<DT>(wf[nr], xf[mr], acc[nr][mr]);
    }
    // every wave has read the tile: the next one may overwrite it while this one is stored
    rows_wait_vmcnt<0>();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (t + 1 < t1) issue_halo(t + 1);
    const int tt = p.rev ? ntiles - 1 - t : t;
    conv_epi::conv_epilogue<MR, NR, true, EPI, DT>(p, acc, tt * (2 * STEM_OW) + wm * STEM_OW,
                                                   wn * WN + fq * 4 * NR, tt, lane);
  }
}

bool stem_rows_supports(const ConvParams& p) {
  return p.dt != DT_F32 && !p.in_scale && !p.scale && !p.bias && !p.addend && !p.stats_y && !p.pro.sums && !p.bpro.sums &&
         p.TI == 7 && p.TJ == 1 && p.Ctap == 32 && p.Cpix == 4 && p.ISH == 2 && p.ISW == 2 &&
         p.dh0 == 0 && p.dw0 == 0 && p.dhs == 1 && p.wsi == 32 && p.OW == STEM_OW && p.OH % 2 == 0 &&
         p.OSH == 1 && p.OSW == 1 && p.OOH == 0 && p.OOW == 0 && p.OHfull == p.OH && p.OWfull == p.OW &&
         p.M == p.N * p.OH * p.OW && p.Cout == STEM_BN && p.ldc == STEM_BN && p.ldw % 8 == 0 &&
         p.W % 2 == 0 && p.W * 8 <= STEM_PITCH && p.W >= 2 * (STEM_OW - 1) + 8 &&
         p.H >= 2 * (p.OH - 1) + 7;
}

bool launch_stem_rows(const ConvParams& p, hipStream_t stream, bool probe) {
  if (!stem_rows_supports(p)) return false;
  if (probe) return true;
  const int ntiles = p.N * (p.OH / 2);
  int blocks = 768;                                  // three blocks on each of the 256 compute units
  if (blocks > ntiles) blocks = ntiles;
  const int tpb = (ntiles + blocks - 1) / blocks;
  blocks = (ntiles + tpb - 1) / tpb;
  if (p.dt == DT_F16)
    hipLaunchKernelGGL((conv_stem_rows_kernel<DT_F16, EPI_PLAIN>), dim3(blocks), dim3(256), 0, stream, p, tpb, ntiles);
  else
    hipLaunchKernelGGL((conv_stem_rows_kernel<DT_BF16, EPI_PLAIN>), dim3(blocks), dim3(256), 0, stream, p, tpb, ntiles);
  return true;
}

}  // namespace

// True when the descriptor is a 3x3 unit-stride SAME convolution (or its data gradient) over
// whole rows: the tile-independent part of what this kernel family serves.
bool conv_rows_supports(const ConvParams& p) {
  const bool fwd_taps = p.dhs == 1 && p.dws == 1 && p.dh0 == -1 && p.dw0 == -1;
  const bool bwd_taps = p.dhs == -1 && p.dws == -1 && p.dh0 == 1 && p.dw0 == 1;
  return p.dt != DT_F32 && !p.scale && !p.bias && !p.pro.sums && !p.bpro.sums &&
         p.TI == 3 && p.TJ == 3 && (fwd_taps || bwd_taps) && p.Ctap == p.Cpix &&
         p.ISH == 1 && p.ISW == 1 && p.OH == p.H && p.OW == p.W && p.OSH == 1 && p.OSW == 1 &&
         p.OOH == 0 && p.OOW == 0 && p.OHfull == p.OH && p.OWfull == p.OW &&
         p.M == p.N * p.H * p.W && p.ldc == p.Cout && p.Cout % 8 == 0 &&
         p.wsi % 8 == 0 && p.wsj % 8 == 0;
}

namespace {
bool rows_dispatch(const ConvParams& p, int cfg, hipStream_t stream, bool probe) {
  if (cfg == 54) return launch_stem_rows(p, stream, probe);       // the 7x7 stride-2 stem
  if (!conv_rows_supports(p)) return false;
  switch (cfg) {
    //                           C    W  rows  BN  waves  ring  steps / stage  waves / SIMD
    case 48: return launch_rows<64, 56, 4, 64, 2, 2, 4, 1, 2>(p, stream, probe);     //  60 KB LDS, 2 blocks / CU
    case 49: return launch_rows<128, 28, 4, 128, 1, 4, 3, 1, 2>(p, stream, probe);   //  72 KB LDS, 2 blocks / CU
    case 50: return launch_rows<64, 56, 4, 64, 2, 2, 3, 2, 2>(p, stream, probe);     //  68 KB: a tap per stage
    case 51: return launch_rows<128, 28, 4, 128, 1, 4, 2, 2, 2>(p, stream, probe);   //  80 KB: half a tap per stage
    case 52: return launch_rows<64, 56, 4, 64, 2, 2, 2, 1, 3>(p, stream, probe);     //  52 KB LDS, 3 blocks / CU
    case 53: return launch_rows<64, 56, 4, 64, 2, 2, 2, 1, 3, true>(p, stream, probe);   //  52 KB, 3 blocks / CU, early refill
    default: return false;
  }
}
}  // namespace

// cfg 48..55.  False: the tile cannot serve this descriptor (the caller falls back).
bool launch_conv_rows(const ConvParams& p, int cfg, hipStream_t stream) {
  return rows_dispatch(p, cfg, stream, false);
}

// True when tile configuration `cfg` of this file would run descriptor `p` (tests, the tuner).
bool conv_rows_serves(const ConvParams& p, int cfg) { return rows_dispatch(p, cfg, nullptr, true); }

int rows_fallback_cfg(int cfg) {        // the pipeline tile the tuner would otherwise pick
  return cfg == 54 ? 22 : 30;
}
