// Weight gradient of the large 3x3 unit-stride layers over whole image rows.
//
//   dW[tap][ci][co] += sum over pixels p of X[p + tap, ci] * dY[p, co]
//
// The general kernel (conv_wgrad.hip) gives every (k-tile, co-tile) of dW its own blocks, so the
// nine taps of a 3x3 layer fetch the same input pixels nine times and every k-tile fetches dY
// again: 1.8 KB through L2 per pixel where the tensors hold 256 bytes (56x56x64: 1.4 GB per
// launch - the layer runs at the rate of the L2, 146 us for 59 GFLOP).  Here a block owns the
// WHOLE gradient of 64 input x 64 output channels (64 x 576 fp32 = 144 accumulator registers per
// lane; a layer of 128 x 128 channels is four such slices, blockIdx.y / z) and walks over tiles
// of TR output rows: the halo tile [(TR + 2) rows][W + 2 pixels][64 channels] of X and the
// [TR * W pixels][64] tile of dY are fetched once by direct-to-LDS loads, all nine taps are
// multiplied out of LDS, and the accumulators leave the registers once, at the end of the
// block's tile range.  512 blocks adding 147 KB each with fp32 atomics take 44 us (atomics of
// different XCDs meet at the memory side, about 13 cache-line operations per clock whatever the
// address pattern - more than the multiplication itself), so the blocks store their partial
// gradients to a workspace with plain stores (in register order: 1 KB per store instruction)
// and a second launch sums them: 16 groups of partials per 4 KB piece, the group sums added to
// the HWIO gradient with 2.4 MB of atomics.
//
// Both MFMA operands need the reduction index (the pixel) contiguous per lane while both tensors
// are pixel-major: fragments are read with ds_read_b64_tr_b16 (conv_wgrad.hip).  Halo rows are
// PITCH pixels apart in LDS, so a tap's row shift is an immediate offset of the read and only its
// column shift (0, 1, 2 pixels) needs address arithmetic; the 32-byte channel chunks of a pixel
// are XOR-swizzled with bits 1 and 3 of the pixel index, which keeps the 8 pixels one 32-lane
// half reads (p .. p + 3, p + 8 .. p + 11) in 8 distinct bank groups for every shift.
// Wave w owns input channels 16 w .. 16 w + 15 of the slice, all nine taps and all 64 output channels.
#include <stdexcept>

#include "common.h"
#include "conv_params.h"

namespace {

__device__ __attribute__((aligned(16))) uint32_t g_zero_page_wr[4] = {0u, 0u, 0u, 0u};

typedef __attribute__((address_space(3))) short4v lds_short4;
typedef __attribute__((ext_vector_type(8))) short short8v;

// swizzle of the 32-byte chunk index inside a 128-byte row (pixel) of 64 16-bit channels
__device__ __forceinline__ int swz64(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

__device__ __forceinline__ bf16x8 tr_pair(const uint8_t* a0, const uint8_t* a1) {
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)a0);
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)a1);
  const short8v r = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return __builtin_bit_cast(bf16x8, r);
}

template <int DT>
__device__ __forceinline__ f32x4 mfma32(const bf16x8 a, const bf16x8 b, const f32x4 c) {
  if constexpr (DT == DT_F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

constexpr int WR_C = 64, WR_CO = 64;                // input / output channels of a block's slice

// CTOT, COUT: channels of the layer (multiples of 64).  W, TR: image width, output rows per tile
// (W * TR pixels, padded to whole steps of 32 with zero rows of dY).  PITCH: halo pixels per halo
// row (a multiple of 16 >= W + 2: a tap's row shift keeps the swizzle).  blockIdx.y / z: the 64
// input / output channels of the block.
template <int DT, int CTOT, int COUT, int W, int TR, int PITCH>
__global__ __launch_bounds__(256, 2) void conv_wgrad_rows_kernel(const WgradParams p, int tiles_per_block,
                                                                 int ntiles) {
  constexpr int C = WR_C, CO = WR_CO;
  constexpr int PB = C * 2;                          // bytes per halo pixel
  constexpr int HROWS = TR + 2;
  constexpr int HBYTES = HROWS * PITCH * PB;
  constexpr int PIX = TR * W;
  constexpr int KS = (PIX + 31) / 32;                // reduction steps of 32 pixels per tile
  constexpr int YBYTES = KS * 4096;                  // step tiles of [32 pixels][64 channels]
  constexpr int HROUNDS = HBYTES / 4096;             // DMA rounds: 256 threads x 16 bytes
  constexpr int TBYTES = 2 * C * 4;                  // scale / shift of an input normalised in the tile
  static_assert(HBYTES % 4096 == 0 && W + 2 <= PITCH && PITCH % 16 == 0, "tile shape");
  static_assert(2 * (HBYTES + YBYTES + TBYTES) <= 160 * 1024, "two blocks per compute unit");
  typedef uint16_t T;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  __shared__ __attribute__((aligned(16))) uint8_t smem[HBYTES + YBYTES + TBYTES];
  uint8_t* halo = smem;
  uint8_t* ys = smem + HBYTES;
  float* xtab = reinterpret_cast<float*>(smem + HBYTES + YBYTES);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ci0 = (int)blockIdx.y * C, co0 = (int)blockIdx.z * CO;
  const T* __restrict__ gx = reinterpret_cast<const T*>(p.x) + ci0;
  const T* __restrict__ gy = reinterpret_cast<const T*>(p.dy) + co0;
  const T* zero_page = reinterpret_cast<const T*>(g_zero_page_wr);

  f32x4 acc[9][4];
#pragma unroll
  for (int a = 0; a < 9; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment geometry of this lane: 16-lane group g reads pixels 8 g + qq and 8 g + 4 + qq of a
  // 32-pixel step, four channels (8 bytes) at channel offset 4 pp of a 16-channel block
  const int g = lane >> 4, idx = lane & 15, qq = idx >> 2, pp = idx & 3;
  // dY tile rows: this thread fills 16-byte slot (tid & 7) of row (tid >> 3) of every step's tile
  const int yrow = tid >> 3;
  const int ych = ((((tid & 7) >> 1) ^ swz64(yrow)) << 1) | (tid & 1);      // logical 16-byte chunk

  if (p.x_scale != nullptr) {                        // the forward launch normalised X inside its tile
    if (tid < C) {
      xtab[tid] = p.x_scale[ci0 + tid];
      xtab[C + tid] = p.x_shift[ci0 + tid];
    }
    dma_wait_all();
    __syncthreads();
  }
  const int tiles_h = p.H / TR;
  const int t0 = blockIdx.x * tiles_per_block;
  const int t1 = min(t0 + tiles_per_block, ntiles);
  for (int t = t0; t < t1; ++t) {
    const int tt = p.rev ? ntiles - 1 - t : t;       // the launch plan's sweep direction
    const int n = tt / tiles_h, r0 = (tt - n * tiles_h) * TR;
    if (t != t0) {                 // every wave has read the previous tile
      dma_wait_all();
      __syncthreads();
    }
    // ---- halo tile: 16-byte position u holds chunk (u % 8 / 2 ^ swizzle) of halo pixel u / 8 -------
#pragma unroll
    for (int it = 0; it < HROUNDS; ++it) {
      const int u = (it * 4 + wave) * 64 + lane;
      const int hp = u >> 3, s16 = u & 7;
      const int ch16 = ((((s16 >> 1) ^ swz64(hp)) << 1) | (s16 & 1));
      const int hr = hp / PITCH, hc = hp % PITCH;
      const int hi = r0 + hr - 1, wi = hc - 1;
      const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)W;
      const T* src = ok ? gx + ((size_t)((n * p.H + hi) * W + wi) * CTOT + ch16 * 8) : zero_page;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(halo + (it * 4 + wave) * 1024), 16, 0, 0);
    }
    // ---- dY tile: KS step tiles of [32 pixels][64 channels], zero rows past the tile's pixels ------
    const size_t pix0 = (size_t)(n * p.H + r0) * W;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const bool ok = PIX % 32 == 0 || s * 32 + yrow < PIX;
      const T* src = ok ? gy + ((pix0 + s * 32 + yrow) * (size_t)p.ldy + ych * 8) : zero_page;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ys + s * 4096 + wave * 1024), 16, 0, 0);
    }
    dma_wait_all();
    if (p.x_scale != nullptr) {
      // BatchNorm + ReLU of the layer in front on the positions this thread fetched itself
      // (out-of-image pixels stay zero), as the forward launch did in its halo tile.  A round moves
      // the thread 32 pixels on, which leaves the swizzle bits alone: the thread's eight channels
      // - its scale and shift - are the same in every round.
      const int u0 = wave * 64 + lane;
      const int ch0 = (((((u0 & 7) >> 1) ^ swz64(u0 >> 3)) << 1) | (u0 & 1));
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(xtab + ch0 * 8);
      const f32x4 s1 = *reinterpret_cast<const f32x4*>(xtab + ch0 * 8 + 4);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(xtab + C + ch0 * 8);
      const f32x4 h1 = *reinterpret_cast<const f32x4*>(xtab + C + ch0 * 8 + 4);
#pragma unroll 2
      for (int it = 0; it < HROUNDS; ++it) {
        const int u = (it * 4 + wave) * 64 + lane;
        const int hp = u >> 3;
        const int hr = hp / PITCH, hc = hp % PITCH;
        const int hi = r0 + hr - 1, wi = hc - 1;
        if ((unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)W) {
          u32x4* cell = reinterpret_cast<u32x4*>(halo + (size_t)u * 16);
          const u32x4 v = *cell;
          u32x4 o;
          o.x = pack2<DT>(fmaxf(lo16<DT>(v.x) * s0[0] + h0[0], 0.f), fmaxf(hi16<DT>(v.x) * s0[1] + h0[1], 0.f));
          o.y = pack2<DT>(fmaxf(lo16<DT>(v.y) * s0[2] + h0[2], 0.f), fmaxf(hi16<DT>(v.y) * s0[3] + h0[3], 0.f));
          o.z = pack2<DT>(fmaxf(lo16<DT>(v.z) * s1[0] + h1[0], 0.f), fmaxf(hi16<DT>(v.z) * s1[1] + h1[1], 0.f));
          o.w = pack2<DT>(fmaxf(lo16<DT>(v.w) * s1[2] + h1[2], 0.f), fmaxf(hi16<DT>(v.w) * s1[3] + h1[3], 0.f));
          *cell = o;
        }
      }
    }
    __syncthreads();

    // ---- 9 taps x KS steps out of LDS -----------------------------------------------------------------
#pragma unroll 1
    for (int s = 0; s < KS; ++s) {
      const uint8_t* yt = ys + s * 4096;
      bf16x8 yf[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        // channels 16 b + 4 pp .. of rows 8 g + qq and 8 g + 4 + qq (layout of conv_wgrad.hip's tiles)
        const int ra = 8 * g + qq, rb = ra + 4;
        yf[b] = tr_pair(yt + ra * 128 + (((b ^ swz64(ra)) << 5) | (pp << 3)),
                        yt + rb * 128 + (((b ^ swz64(rb)) << 5) | (pp << 3)));
      }
      // pixels past the tile (their dY rows are zero) read the tile's last pixel: finite values
      int q0 = s * 32 + 8 * g + qq, q1 = q0 + 4;
      if (PIX % 32 != 0) {
        q0 = min(q0, PIX - 1);
        q1 = min(q1, PIX - 1);
      }
      const int rr0 = q0 / W, rr1 = q1 / W;
      const int hb0 = rr0 * PITCH + (q0 - rr0 * W), hb1 = rr1 * PITCH + (q1 - rr1 * W);
#pragma unroll
      for (int dj = 0; dj < 3; ++dj) {
        const int h0 = hb0 + dj, h1 = hb1 + dj;
        const uint8_t* a0 = halo + h0 * PB + (((wave ^ swz64(h0)) << 5) | (pp << 3));
        const uint8_t* a1 = halo + h1 * PB + (((wave ^ swz64(h1)) << 5) | (pp << 3));
#pragma unroll
        for (int di = 0; di < 3; ++di) {
          const bf16x8 xf = tr_pair(a0 + di * (PITCH * PB), a1 + di * (PITCH * PB));
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[di * 3 + dj][b] = mfma32<DT>(xf, yf[b], acc[di * 3 + dj][b]);
        }
      }
    }
  }
  if (t0 >= t1) return;

  // ---- the block's gradient leaves the registers once -----------------------------------------------
  if (p.ws != nullptr) {
    // to the workspace in register order - [wave][tap][16 outputs][lane] of four floats: every store
    // instruction writes 1 KB in one piece; the summing launch knows the order
    const size_t part = ((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (9 * C * CO);
    f32x4* dst = reinterpret_cast<f32x4*>(p.ws + part) + (size_t)wave * (9 * 4 * 64) + lane;
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) dst[(a * 4 + b) * 64] = acc[a][b];
    return;
  }
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int a = 0; a < 9; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* row = p.dw + ((size_t)a * CTOT + ci0 + wave * 16 + fq * 4 + r) * COUT + co0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (p.debug_plain_store) row[b * 16 + fr] = acc[a][b][r];
        else atomicAdd(row + b * 16 + fr, acc[a][b][r]);
      }
    }
}

// dw slice (y, z) += sum over the partials g, g + G, ... of its blocks.  ws holds
// [z][y][part][wave][tap][16 outputs][lane][4] (the accumulator registers of the blocks as they
// were); the slice is input channels 64 y .., output channels 64 z .. of the [9][ctot][cout]
// gradient.  blockIdx = (256 accumulator registers, group, y + ny * z).
constexpr int RED_GROUPS = 16, RED_SLICE = 1024;
constexpr int WR_NF = 9 * WR_C * WR_CO;              // floats of a slice
// (gridDim.y groups; one group: the slice has a single owner, which adds in a fixed order without
// atomics - the deterministic mode)
__global__ __launch_bounds__(256) void wgrad_rows_reduce_kernel(const float* __restrict__ ws, int nparts,
                                                                float* __restrict__ dw, int ny, int ctot,
                                                                int cout) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  const int i = e * 4;
  const int G = gridDim.y;
  const int y = blockIdx.z % ny, z = blockIdx.z / ny;
  ws += (size_t)blockIdx.z * nparts * WR_NF;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int g = blockIdx.y;
  for (; g + 3 * G < nparts; g += 4 * G) {      // four loads in flight
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      v[u] = *reinterpret_cast<const f32x4*>(ws + (size_t)(g + u * G) * WR_NF + i);
#pragma unroll
    for (int u = 0; u < 4; ++u) { s[0] += v[u][0]; s[1] += v[u][1]; s[2] += v[u][2]; s[3] += v[u][3]; }
  }
  for (; g < nparts; g += G) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(ws + (size_t)g * WR_NF + i);
    s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
  }
  // which accumulator register this was: rows 4 (lane / 16) + r of its wave's 16 channels, column lane % 16
  const int lane = e & 63, b = (e >> 6) & 3, t = e >> 8;
  const int a = t % 9, wave = t / 9;
  const int krow = a * ctot + y * WR_C + wave * 16 + (lane >> 4) * 4;
  float* dst = dw + (size_t)krow * cout + z * WR_CO + b * 16 + (lane & 15);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (G == 1) dst[(size_t)r * cout] += s[r];
    else atomicAdd(dst + (size_t)r * cout, s[r]);
  }
}

constexpr int WR_MAX_BLOCKS = 512;                   // two on each of the 256 compute units
constexpr size_t WR_WS_FLOATS = (size_t)WR_MAX_BLOCKS * WR_NF;

template <int CTOT, int COUT, int W, int TR, int PITCH>
bool launch_rows(const WgradParams& p, hipStream_t stream, bool probe) {
  if (p.Cpix != CTOT || p.Cout != COUT || p.W != W || p.H % TR) return false;
  if (probe) return true;
  constexpr int ny = CTOT / WR_C, nz = COUT / WR_CO;
  const int ntiles = p.N * (p.H / TR);
  int blocks = WR_MAX_BLOCKS / (ny * nz);
  if (blocks > ntiles) blocks = ntiles;
  const int tpb = (ntiles + blocks - 1) / blocks;
  blocks = (ntiles + tpb - 1) / tpb;
  static_assert(WR_NF % RED_SLICE == 0, "pieces");
  const dim3 grid(blocks, ny, nz);
  if (p.dt == DT_F16)
    hipLaunchKernelGGL((conv_wgrad_rows_kernel<DT_F16, CTOT, COUT, W, TR, PITCH>), grid, dim3(256), 0, stream, p,
                       tpb, ntiles);
  else
    hipLaunchKernelGGL((conv_wgrad_rows_kernel<DT_BF16, CTOT, COUT, W, TR, PITCH>), grid, dim3(256), 0, stream, p,
                       tpb, ntiles);
  if (p.ws != nullptr) {
    // splits == 1 is how the engine asks for a summation order that does not depend on timing
    const int groups = p.splits == 1 ? 1 : (blocks < RED_GROUPS ? blocks : RED_GROUPS);
    hipLaunchKernelGGL(wgrad_rows_reduce_kernel, dim3(WR_NF / RED_SLICE, groups, ny * nz), dim3(256), 0, stream,
                       p.ws, blocks, p.dw, ny, CTOT, COUT);
  }
  return true;
}

// ---- the ImageNet stem: 7x7 stride 2 on the 4-channel image stored with its zero border -------------
//
//   dW[i][j][ci][co] += sum over output pixels (r, c) of X[2 r + i][2 c + j][ci] * dY[r][c][co]
//
// In the merged-tap form of the stem (conv_plan.py) a row of the [K][64] gradient is (i, j, ci) with
// j = 0 .. 7, ci = 0 .. 3 (tap 7 and channel 3 are padding): 7 x 32 = 224 rows, 14 MFMA tiles of 16.
// The general kernel fetches the 64-byte window of every (pixel, kernel row) from L2 - each input
// pixel 28 times, 1.4 GB per launch next to the 0.4 GB of dY, 191 us at the rate of the L2.  Here
// a block takes pairs of output rows (224 pixels = 7 reduction steps of 32): the 9 input rows they
// touch sit in LDS once (2 KB apart) next to the dY tile, and the fragment of (kernel row i, taps
// 4 h .. 4 h + 3) for output pixel c is the 32 bytes at row 2 r + i, pixel 2 c + 4 h: windows of
// neighbouring pixels overlap, so there is nothing to swizzle - lanes that meet at an address
// share the read.  Wave w owns the MFMA tiles w, w + 4, w + 8 (, w + 12) and all 64 output channels.
constexpr int SW_OW = 112, SW_PIX = 2 * SW_OW, SW_KS = SW_PIX / 32, SW_PITCH = 2048, SW_HROUNDS = 5;
constexpr int SW_MT = 14;                            // MFMA tiles of gradient rows: (kernel row, half of its taps)

template <int DT>
__global__ __launch_bounds__(256, 3) void conv_wgrad_stem_rows_kernel(const WgradParams p, int tiles_per_block,
                                                                      int ntiles) {
  constexpr int XBYTES = SW_HROUNDS * 4096, YBYTES = SW_KS * 4096;
  static_assert(9 * SW_PITCH <= XBYTES && SW_PIX % 32 == 0, "tile shape");
  typedef uint16_t T;
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  __shared__ __attribute__((aligned(16))) uint8_t smem[XBYTES + YBYTES];
  uint8_t* xs = smem;
  uint8_t* ys = smem + XBYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const T* __restrict__ gx = reinterpret_cast<const T*>(p.x);
  const T* __restrict__ gy = reinterpret_cast<const T*>(p.dy);
  const T* zero_page = reinterpret_cast<const T*>(g_zero_page_wr);

  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int g = lane >> 4, idx = lane & 15, qq = idx >> 2, pp = idx & 3;
  const int yrow = tid >> 3;
  const int ych = ((((tid & 7) >> 1) ^ swz64(yrow)) << 1) | (tid & 1);
  const int tiles_per_img = p.OH >> 1;
  const int row_chunks = (p.W * 4) >> 3;             // 16-byte chunks of a stored input row
  const int t0 = blockIdx.x * tiles_per_block;
  const int t1 = min(t0 + tiles_per_block, ntiles);
  for (int t = t0; t < t1; ++t) {
    const int tt = p.rev ? ntiles - 1 - t : t;
    const int n = tt / tiles_per_img, r0 = (tt - n * tiles_per_img) * 2;
    if (t != t0) {                 // every wave has read the previous tile
      dma_wait_all();
      __syncthreads();
    }
    // ---- 9 input rows: 16-byte position q = chunk (q % 128) of input row 2 r0 + q / 128 ----------------
#pragma unroll
    for (int it = 0; it < SW_HROUNDS; ++it) {
      const int q = (it * 4 + wave) * 64 + lane;
      const int hr = q >> 7, ch = q & 127;
      const bool ok = hr < 9 && ch < row_chunks;
      const T* src = ok ? gx + (((size_t)(n * p.H + 2 * r0 + hr) * p.W) * 4 + ch * 8) : zero_page;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(xs + (it * 4 + wave) * 1024), 16, 0, 0);
    }
    // ---- dY: two output rows = 224 consecutive pixels, 7 step tiles of [32 pixels][64 channels] --------
    const size_t pix0 = (size_t)(n * p.OH + r0) * SW_OW;
#pragma unroll
    for (int s = 0; s < SW_KS; ++s) {
      const T* src = gy + ((pix0 + s * 32 + yrow) * (size_t)p.ldy + ych * 8);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ys + s * 4096 + wave * 1024), 16, 0, 0);
    }
    dma_wait_all();
    __syncthreads();

#pragma unroll 1
    for (int s = 0; s < SW_KS; ++s) {
      const uint8_t* yt = ys + s * 4096;
      bf16x8 yf[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int ra = 8 * g + qq, rb = ra + 4;
        yf[b] = tr_pair(yt + ra * 128 + (((b ^ swz64(ra)) << 5) | (pp << 3)),
                        yt + rb * 128 + (((b ^ swz64(rb)) << 5) | (pp << 3)));
      }
      const int q0 = s * 32 + 8 * g + qq, q1 = q0 + 4;
      const int rr0 = q0 >= SW_OW ? 1 : 0, rr1 = q1 >= SW_OW ? 1 : 0;
      // input pixel 2 c + pp of input row 2 rr (kernel row and tap half are added per tile)
      const uint8_t* b0 = xs + (2 * rr0) * SW_PITCH + ((2 * (q0 - rr0 * SW_OW) + pp) << 3);
      const uint8_t* b1 = xs + (2 * rr1) * SW_PITCH + ((2 * (q1 - rr1 * SW_OW) + pp) << 3);
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int mt = wave + 4 * a;                 // MFMA tile: kernel row mt / 2, taps 4 (mt % 2) ..
        if (mt < SW_MT) {
          const int off = (mt >> 1) * SW_PITCH + ((mt & 1) << 5);
          const bf16x8 xf = tr_pair(b0 + off, b1 + off);
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = mfma32<DT>(xf, yf[b], acc[a][b]);
        }
      }
    }
  }
  if (t0 >= t1) return;

  // ---- the block's gradient: row (kernel row i, tap j, channel ci) of the real [7][7][3][64] tensor ---
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int mt = wave + 4 * a;
    if (mt >= SW_MT) continue;
    const int i = mt >> 1, j = ((mt & 1) << 2) + fq;
    if (j >= 7) continue;                            // the eighth tap is padding
#pragma unroll
    for (int r = 0; r < 3; ++r) {                    // (the fourth channel is padding)
      float* row = p.dw + (size_t)((i * 7 + j) * 3 + r) * 64;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (p.debug_plain_store) row[b * 16 + fr] = acc[a][b][r];
        else atomicAdd(row + b * 16 + fr, acc[a][b][r]);
      }
    }
  }
}

bool stem_rows_dispatch(const WgradParams& p, hipStream_t stream, bool probe) {
  if (p.batch != nullptr || p.x_scale != nullptr || (p.dt != DT_BF16 && p.dt != DT_F16)) return false;
  // one split is how the engine asks for a summation order that does not depend on timing: the
  // blocks of this kernel add their gradients with atomics, so that request goes to the general kernel
  if (p.splits == 1) return false;
  if (p.TI != 7 || p.TJ != 1 || p.Ctap != 32 || p.Cpix != 4 || p.Cpad != 4 || p.Creal != 3 || p.TJpad != 8 ||
      p.TJreal != 7 || p.ISH != 2 || p.ISW != 2 || p.dh0 != 0 || p.dw0 != 0 || p.dhs != 1)
    return false;
  if (p.OW != SW_OW || p.OH % 2 || p.Cout != 64 || p.ldy != 64 || p.W % 2 || p.W * 8 > SW_PITCH ||
      p.W < 2 * (SW_OW - 1) + 8 || p.H < 2 * (p.OH - 1) + 7)
    return false;
  if (probe) return true;
  const int ntiles = p.N * (p.OH / 2);
  int blocks = 768;                                  // three blocks on each of the 256 compute units
  if (blocks > ntiles) blocks = ntiles;
  const int tpb = (ntiles + blocks - 1) / blocks;
  blocks = (ntiles + tpb - 1) / tpb;
  if (p.dt == DT_F16)
    hipLaunchKernelGGL(conv_wgrad_stem_rows_kernel<DT_F16>, dim3(blocks), dim3(256), 0, stream, p, tpb, ntiles);
  else
    hipLaunchKernelGGL(conv_wgrad_stem_rows_kernel<DT_BF16>, dim3(blocks), dim3(256), 0, stream, p, tpb, ntiles);
  return true;
}

bool rows_dispatch(const WgradParams& p, int cfg, hipStream_t stream, bool probe) {
  if (cfg != WGRAD_ROWS_CFG) return false;
  if (p.TI == 7) return stem_rows_dispatch(p, stream, probe);
  if (p.batch != nullptr || (p.dt != DT_BF16 && p.dt != DT_F16)) return false;
  if (p.TI != 3 || p.TJ != 3 || p.ISH != 1 || p.ISW != 1 || p.dh0 != -1 || p.dw0 != -1 || p.dhs != 1 ||
      p.dws != 1 || p.OH != p.H || p.OW != p.W)
    return false;
  if (p.Ctap != p.Cpix || p.Creal != p.Cpix || p.Cpad != p.Cpix || p.ldy != p.Cout || p.TJpad != 3 ||
      p.TJreal != 3)
    return false;
  if ((int64_t)p.N * p.H * p.W * p.Cpix >= (1ll << 31)) return false;
  //           channels in, out   W rows pitch
  if (launch_rows<64, 64, 56, 4, 64>(p, stream, probe)) return true;        // 77 KB LDS: 224 pixels a tile
  if (launch_rows<128, 128, 28, 7, 32>(p, stream, probe)) return true;      // 65 KB LDS: 196 pixels, 7 steps
  return false;
}

}  // namespace

// 3x3, unit stride, padding 1, 16-bit elements, one layer per launch: 56x56x64 -> 64 and
// 28x28x128 -> 128; the 7x7 stride-2 stem in its merged-tap form
bool wgrad_rows_serves(const WgradParams& p, int cfg) { return rows_dispatch(p, cfg, nullptr, true); }

size_t wgrad_rows_workspace_floats() { return WR_WS_FLOATS; }

bool launch_conv_wgrad_rows(const WgradParams& p, int cfg, hipStream_t stream) {
  return rows_dispatch(p, cfg, stream, false);
}
