// Weight gradient of the NHWC convolution on MFMA, split-K over pixels.
//
//   dW[k][co] += sum over pixels p of X[p, k] * dY[p, co],   k = (tap i, tap j, channel c)
//
// where X[p, k] is gathered from the forward input exactly as in the forward kernel.
// GEMM view: M' = Ktotal (rows of dW), N' = Cout, reduction over P = N*OH*OW pixels,
// which is huge (up to 3.2 M) - so the pixel range is split over gridDim.y blocks and the
// partial tiles are combined with fp32 atomic adds into the (pre-zeroed) Keras-layout
// HWIO gradient buffer, which is directly a slice of the flat all-reduce bucket.
//
// Both MFMA operands need the *reduction* index (pixel) contiguous per lane, but both
// tensors are pixel-major in memory (channels contiguous).  The tiles are therefore staged
// in LDS as [32 pixels][channels] exactly as they arrive (direct global->LDS DMA, double
// buffered, no register staging) and read back with ds_read_b64_tr_b16, gfx950's transposing LDS read: each lane
// gets 4 consecutive pixels of one channel per instruction, two reads form one 8-deep
// MFMA operand.  The 32-byte column chunks of every LDS row are XOR-swizzled with a
// function of the row so that the 8 rows touched by one 32-lane half fall into 8
// distinct bank groups.
//
// The atomic epilogue is expensive (an fp32 tile per block and split: measured 25-45 % of the
// kernel with 20-90 splits), so a block can hold G groups of four waves that work on
// interleaved reduction steps of the same dW tile with their own LDS buffers, and add their
// accumulators through LDS before one group issues the atomics: G times fewer atomic bytes
// at the same number of resident waves.
#include <stdexcept>
#include <string>

#include "common.h"
#include "conv_params.h"

namespace {

constexpr int BP = 32;  // pixels per reduction step

// 16 bytes of zeros in global memory: the source of every masked direct-to-LDS load.
__device__ __attribute__((aligned(16))) uint32_t g_zero_page_w[4] = {0u, 0u, 0u, 0u};

typedef __attribute__((address_space(3))) short4v lds_short4;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8w;

// Byte offset of 16-byte chunk `ch` of pixel row `row` inside a [32][WIDTH] tile of ES-byte
// elements.  16-bit types: swizzled for the transposing reads.  fp32: fragments are read with
// ds_read_b32, 16 consecutive channels of pixel rows r and r + 1 per 32-lane half; rows are a
// multiple of 128 bytes apart, so odd rows swap their 64-byte halves to land on the other banks.
template <int WIDTH, int ES = 2>
__device__ __forceinline__ int tile_off(int row, int ch) {
  if constexpr (ES == 4) {
    if constexpr (WIDTH >= 32) return row * (WIDTH * 4) + ((ch ^ ((row & 1) << 2)) << 4);
    else return row * (WIDTH * 4) + ch * 16;
  } else if constexpr (WIDTH == 128 || WIDTH == 256) {
    // rows are a whole number of 256-byte bank rows: XOR the low three bits of the 32-byte
    // chunk index with a function that differs for the 8 rows one half-wave touches
    const int f = (row & 3) | (((row >> 3) & 1) << 2);
    return row * (WIDTH * 2) + ((((ch >> 1) ^ f) << 5) | ((ch & 1) << 4));
  } else if constexpr (WIDTH == 64) {
    const int f = ((row >> 1) & 1) | (((row >> 3) & 1) << 1);
    return row * 128 + ((((ch >> 1) ^ f) << 5) | ((ch & 1) << 4));
  } else {
    return row * (WIDTH * 2) + ch * 16;
  }
}

// Transposed fragment: for the 16 channels starting at c0, lane (g = lane>>4, idx = lane&15)
// receives pixels 8g..8g+7 of channel c0+idx.
template <int WIDTH>
__device__ __forceinline__ bf16x8 read_tr_frag(const uint8_t* tile, int c0, int lane) {
  const int g = lane >> 4, idx = lane & 15, q = idx >> 2, pp = idx & 3;
  const int ch = (c0 >> 3) + (pp >> 1);
  const int sub = (pp & 1) << 3;
  const uint8_t* a0 = tile + tile_off<WIDTH>(8 * g + q, ch) + sub;
  const uint8_t* a1 = tile + tile_off<WIDTH>(8 * g + 4 + q, ch) + sub;
  short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)a0);
  short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)a1);
  typedef __attribute__((ext_vector_type(8))) short short8v;
  short8v r = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  return __builtin_bit_cast(bf16x8, r);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// waves per SIMD the register allocation must allow: what the LDS ring admits, at most 4
constexpr int wgrad_occ(int bkk, int bco, int threads, int g, int stages, int es) {
  const int lds = g * stages * BP * (bkk + bco) * es;
  int bpc = (160 * 1024) / lds;
  if (bpc < 1) bpc = 1;
  const int w = bpc * threads / 256;
  return w > 4 ? 4 : (w < 1 ? 1 : w);
}

// STAGES = 2: one reduction step in flight, drained (vmcnt(0)) at the barrier that ends every
// step.  STAGES = 3, 4: the tiles of STAGES - 1 steps stay in flight; a wave waits with a counted
// `s_waitcnt vmcnt(n)` for the step it is about to multiply and the block meets at a raw
// `s_barrier`, which does not drain the vector-memory counter (same scheme as conv_pipe.hip).
//
// DT: element type of X and dY.  bf16 / fp16 use the transposing reads and
// v_mfma_f32_16x16x32_{bf16,f16}; fp32 reads single elements (a lane owns channel lane % 16 of
// pixel 4 * step + lane / 16) and runs v_mfma_f32_16x16x4_f32, eight per 32-pixel step.
//
// XBN (pointwise layers, 16-bit types): X is the raw input of a BatchNorm + ReLU that the forward
// launch applied on its own fragments; the same x := max(x * x_scale[k] + x_shift[k], 0) runs here
// on the transposed fragments (a lane's eight pixels belong to one channel: two constants per
// fragment row live in registers for the whole reduction).  Pixels past the end transform a zero
// into max(shift, 0) and meet a dY of zero.
template <int DT, int BKK, int BCO, int WAVES_K, int WAVES_C, int G, bool POINTWISE, int STAGES,
          bool XBN = false>
__global__ __launch_bounds__(64 * WAVES_K * WAVES_C * G,
                             wgrad_occ(BKK, BCO, 64 * WAVES_K * WAVES_C * G, G, STAGES,
                                       DT == DT_F32 ? 4 : 2))
void conv_wgrad_kernel(const WgradParams p) {
  static_assert(!XBN || (POINTWISE && DT != DT_F32), "fragment BatchNorm: pointwise, 16-bit");
  typedef typename Elem<DT>::T T;
  constexpr int ES = (int)sizeof(T);
  constexpr int EPC = 16 / ES;                                  // elements per 16-byte chunk
  constexpr int GT = 64 * WAVES_K * WAVES_C;                    // threads per wave group
  constexpr int WKK = BKK / WAVES_K, WCO = BCO / WAVES_C;
  constexpr int KR = WKK / 16, CR = WCO / 16;
  constexpr int XCPR = BKK / EPC, YCPR = BCO / EPC;             // 16-byte chunks per tile row
  constexpr int XRPP = GT / XCPR, YRPP = GT / YCPR;             // tile rows filled per pass
  constexpr int XV = (BP + XRPP - 1) / XRPP, YV = (BP + YRPP - 1) / YRPP;
  constexpr int XBYTES = BP * BKK * ES, YBYTES = BP * BCO * ES;
  static_assert(XCPR <= 64 && YCPR <= 64, "a tile row is at most one DMA instruction");
  static_assert(GT == 256 || GT == 512, "4 or 8 waves per group");
  static_assert(GT * G <= 1024, "block size");
  static_assert(WKK % 16 == 0 && WCO % 16 == 0, "tile shape");

  constexpr int SBYTES = XBYTES + YBYTES;                       // one stage of one group
  constexpr int GBYTES = STAGES * SBYTES;                       // LDS of one group
  // DMA instructions per wave and stage: identical for every wave when a pass fills at most
  // BP rows (the counted waits of the deep pipeline rely on it)
  constexpr int LPS = XV + YV;
  static_assert(STAGES == 2 || (XRPP <= BP && YRPP <= BP && BP % XRPP == 0 && BP % YRPP == 0),
                "deep pipeline needs the same number of loads in every wave");
  static_assert(STAGES >= 2 && STAGES <= 4 && (STAGES - 1) * LPS < 64, "ring depth");
  // accumulator rows (index a) exchanged per round through the block's LDS
  constexpr int ACH = (G * GBYTES) / (CR * 16 * GT) < KR ? (G * GBYTES) / (CR * 16 * GT) : KR;
  static_assert(G == 1 || ACH >= 1, "accumulator exchange must fit in LDS");

  __shared__ __attribute__((aligned(16))) uint8_t smem_all[G * GBYTES];

  const int grp = threadIdx.x / GT;                 // wave group
  const int tid = threadIdx.x % GT, lane = tid & 63, wave = tid >> 6;    // group-local ids
  const int wk = wave / WAVES_C, wc = wave % WAVES_C;
  uint8_t* smem = smem_all + grp * GBYTES;

  const int tilesC = (p.Cout + BCO - 1) / BCO;
  const int kt = blockIdx.x / tilesC, ct = blockIdx.x % tilesC;
  const int k0 = kt * BKK, c0 = ct * BCO;
  const int pbeg = (p.rev ? (int)gridDim.y - 1 - (int)blockIdx.y : (int)blockIdx.y) * p.pix_per_split;
  const int pend = min(pbeg + p.pix_per_split, p.P);
  if (pbeg >= pend) return;

  // Tiles go global -> LDS directly (global_load_lds_dwordx4): one wave instruction fills
  // 1 KiB of the linear LDS image, lane l at byte 16*l.  The lane that owns physical 16-byte
  // slot s of a row fetches the logical chunk that the XOR swizzle maps there, so the swizzle
  // is applied to the source address; masked elements fetch from a page of zeros.
  // ---- X (input) gather geometry: this thread always loads the same 8 k-columns ----------
  const int xrow = tid / XCPR, yrow = tid / YCPR;
  auto unswizzle = [](int width, int row, int slot) {
    if (ES == 4) return width >= 32 ? (slot ^ ((row & 1) << 2)) : slot;
    if (width == 128 || width == 256) return ((((slot >> 1) ^ ((row & 3) | (((row >> 3) & 1) << 2))) << 1) | (slot & 1));
    if (width == 64) return ((((slot >> 1) ^ (((row >> 1) & 1) | (((row >> 3) & 1) << 1))) << 1) | (slot & 1));
    return slot;
  };
  const int xch = unswizzle(BKK, xrow, tid % XCPR);     // same for every row this lane fills
  const int kcol = k0 + xch * EPC;
  const bool xk_ok = kcol < p.Ktotal;
  int kc = 0, ti = 0, tj = 0;
  if (!POINTWISE && xk_ok) {
    const int t = kcol / p.Ctap;
    kc = kcol - t * p.Ctap;
    ti = t / p.TJ;
    tj = t - ti * p.TJ;
  }
  const int tap_h = p.dh0 + ti * p.dhs, tap_w = p.dw0 + tj * p.dws;
  // ---- dY geometry -------------------------------------------------------------------------
  const int ych = unswizzle(BCO, yrow, tid % YCPR);
  const int ycol = c0 + ych * EPC;
  const bool yc_ok = ycol < p.Cout;

  const void* px = p.x;
  const void* pdy = p.dy;
  float* pdw = p.dw;
  if (p.batch != nullptr) {          // block-uniform: the layer this z-slice works on
    const WgradBatchItem it = p.batch[blockIdx.z];
    px = it.x; pdy = it.dy; pdw = it.dw;
  }
  const T* __restrict__ gx = reinterpret_cast<const T*>(px);
  const T* __restrict__ gy = reinterpret_cast<const T*>(pdy);
  const T* zero_page = reinterpret_cast<const T*>(g_zero_page_w);
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;

  auto issue_tile = [&](int pstep, int sbyte) {
    uint8_t* xs = smem + sbyte;
    uint8_t* ys = xs + XBYTES;
#pragma unroll
    for (int r = 0; r < XV; ++r) {
      if (XRPP * r + wave * (64 / XCPR) < BP) {      // wave-uniform: whole 1 KiB pieces only
        const int row = xrow + XRPP * r;
        const int pp = pstep + row;
        bool ok = xk_ok && pp < pend;
        const T* src = zero_page;
        if (POINTWISE) {
          if (ok) src = gx + ((size_t)pp * p.Cpix + kcol);
        } else {
          const uint32_t t = fd_div((uint32_t)(ok ? pp : 0), p.divOW);
          const int ow = (ok ? pp : 0) - (int)t * p.OW;
          const uint32_t n = fd_div(t, p.divOH);
          const int oh = (int)t - (int)n * p.OH;
          const int hi = oh * p.ISH + tap_h, wi = ow * p.ISW + tap_w;
          ok = ok && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
          if (ok) src = gx + ((size_t)(((int)n * p.H + hi) * p.W + wi) * p.Cpix + kc);
        }
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(xs + r * (GT * 16) + wave * 1024), 16, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < YV; ++r) {
      if (YRPP * r + wave * (64 / YCPR) < BP) {
        const int row = yrow + YRPP * r;
        const int pp = pstep + row;
        const bool ok = yc_ok && pp < pend;
        const T* src = ok ? gy + ((size_t)pp * p.ldy + ycol) : zero_page;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ys + r * (GT * 16) + wave * 1024), 16, 0, 0);
      }
    }
  };

  f32x4 acc[KR][CR];
#pragma unroll
  for (int a = 0; a < KR; ++a)
#pragma unroll
    for (int b = 0; b < CR; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  // group g takes reduction steps g, g + G, ...; the trip count is the same for every group
  // (a step past the end loads zeros) because all of them meet at the block barrier
  const int nsteps = ((pend - pbeg + BP - 1) / BP + G - 1) / G;
  float xsc[KR], xsh[KR];
  if constexpr (XBN) {
#pragma unroll
    for (int a = 0; a < KR; ++a) {
      const int k = k0 + wk * WKK + a * 16 + (lane & 15);
      xsc[a] = k < p.Ktotal ? p.x_scale[k] : 0.f;
      xsh[a] = k < p.Ktotal ? p.x_shift[k] : 0.f;
    }
  }
  auto multiply = [&](int sbyte) {
    const uint8_t* xs = smem + sbyte;
    const uint8_t* ys = xs + XBYTES;
    if constexpr (DT == DT_F32) {
      const int g = lane >> 4, idx = lane & 15;
#pragma unroll
      for (int st = 0; st < BP / 4; ++st) {
        const int row = 4 * st + g;
        float xa[KR], yb[CR];
#pragma unroll
        for (int a = 0; a < KR; ++a) {
          const int col = wk * WKK + a * 16 + idx;
          xa[a] = *reinterpret_cast<const float*>(xs + tile_off<BKK, 4>(row, col >> 2) + (col & 3) * 4);
        }
#pragma unroll
        for (int b = 0; b < CR; ++b) {
          const int col = wc * WCO + b * 16 + idx;
          yb[b] = *reinterpret_cast<const float*>(ys + tile_off<BCO, 4>(row, col >> 2) + (col & 3) * 4);
        }
#pragma unroll
        for (int a = 0; a < KR; ++a)
#pragma unroll
          for (int b = 0; b < CR; ++b)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[a], yb[b], acc[a][b], 0, 0, 0);
      }
    } else {
      bf16x8 xf[KR], yf[CR];
#pragma unroll
      for (int a = 0; a < KR; ++a) xf[a] = read_tr_frag<BKK>(xs, wk * WKK + a * 16, lane);
#pragma unroll
      for (int b = 0; b < CR; ++b) yf[b] = read_tr_frag<BCO>(ys, wc * WCO + b * 16, lane);
#pragma unroll
      for (int a = 0; a < KR; ++a) {
        if constexpr (XBN) {
          // one fragment row at a time, in front of its own products: the vector instructions of
          // row a + 1 run while the matrix unit works on row a
          const u32x4 v = __builtin_bit_cast(u32x4, xf[a]);
          const uint32_t w[4] = {v.x, v.y, v.z, v.w};
          uint32_t o[4];
#pragma unroll
          for (int j = 0; j < 4; ++j)
            o[j] = pack2<DT>(fmaxf(lo16<DT>(w[j]) * xsc[a] + xsh[a], 0.f),
                             fmaxf(hi16<DT>(w[j]) * xsc[a] + xsh[a], 0.f));
          xf[a] = __builtin_bit_cast(bf16x8, u32x4{o[0], o[1], o[2], o[3]});
        }
#pragma unroll
        for (int b = 0; b < CR; ++b) {
          if constexpr (DT == DT_F16)
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                __builtin_bit_cast(f16x8w, xf[a]), __builtin_bit_cast(f16x8w, yf[b]), acc[a][b], 0, 0, 0);
          else
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[a], yf[b], acc[a][b], 0, 0, 0);
        }
      }
    }
  };
  if constexpr (STAGES == 2) {
    issue_tile(pbeg + grp * BP, 0);
    dma_wait_all();
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
      const int buf = s & 1;
      if (s + 1 < nsteps) issue_tile(pbeg + ((s + 1) * G + grp) * BP, (buf ^ 1) * SBYTES);
      multiply(buf * SBYTES);
      dma_wait_all();
      __syncthreads();
    }
  } else {
    int issue_byte = 0, read_byte = 0;
    const int pre = nsteps < STAGES - 1 ? nsteps : STAGES - 1;
    for (int s = 0; s < pre; ++s) {
      issue_tile(pbeg + (s * G + grp) * BP, issue_byte);
      issue_byte += SBYTES;
    }
    for (int s = 0; s < nsteps; ++s) {
      // steps issued after s that may stay in flight while s is multiplied
      const int ahead = (nsteps - 1 - s) < (STAGES - 2) ? (nsteps - 1 - s) : (STAGES - 2);
      if (STAGES >= 4 && ahead >= 2) wait_vmcnt<(STAGES >= 4 ? 2 : 0) * LPS>();
      else if (ahead >= 1) wait_vmcnt<LPS>();
      else wait_vmcnt<0>();
      // own LDS reads of the previous step retired, then the block meets: step s has landed for
      // every wave and nobody still reads the buffer refilled next
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (s + STAGES - 1 < nsteps) {
        issue_tile(pbeg + ((s + STAGES - 1) * G + grp) * BP, issue_byte);
        issue_byte += SBYTES;
        if (issue_byte == GBYTES) issue_byte = 0;
      }
      multiply(read_byte);
      read_byte += SBYTES;
      if (read_byte == GBYTES) read_byte = 0;
    }
    if constexpr (G > 1)     // the accumulator exchange below reuses the ring
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }

  // ---- groups 1..G-1 hand their accumulators to group 0 through LDS ------------------------
  if constexpr (G > 1) {
    float* red = reinterpret_cast<float*>(smem_all);
    for (int r = 1; r < G; ++r) {
#pragma unroll
      for (int a0 = 0; a0 < KR; a0 += ACH) {
        if (grp == r) {
#pragma unroll
          for (int a = a0; a < a0 + ACH && a < KR; ++a)
#pragma unroll
            for (int b = 0; b < CR; ++b)
#pragma unroll
              for (int e = 0; e < 4; ++e)
                red[(((a - a0) * CR + b) * 4 + e) * GT + tid] = acc[a][b][e];
        }
        dma_wait_all();      // nothing is in flight here; keeps "no barrier without a wait"
        __syncthreads();     // checkable for the whole kernel (tests/test_kernel_isa.py)
        if (grp == 0) {
#pragma unroll
          for (int a = a0; a < a0 + ACH && a < KR; ++a)
#pragma unroll
            for (int b = 0; b < CR; ++b)
#pragma unroll
              for (int e = 0; e < 4; ++e)
                acc[a][b][e] += red[(((a - a0) * CR + b) * 4 + e) * GT + tid];
        }
        dma_wait_all();
        __syncthreads();
      }
    }
    if (grp != 0) return;
  }

  // ---- epilogue: atomic accumulation into the HWIO fp32 gradient ------------------------------
  const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
  for (int a = 0; a < KR; ++a) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kidx = k0 + wk * WKK + a * 16 + fq * 4 + r;
      if (kidx >= p.Ktotal) continue;
      // padded (tap, channel) -> real HWIO row
      const uint32_t t = fd_div((uint32_t)kidx, p.divCpad);
      const int c = kidx - (int)t * p.Cpad;
      const uint32_t tir = fd_div(t, p.divTJpad);
      const int tjr = (int)t - (int)tir * p.TJpad;
      if (c >= p.Creal || tjr >= p.TJreal) continue;
      float* row = pdw + (size_t)(((int)tir * p.TJreal + tjr) * p.Creal + c) * p.Cout;
#pragma unroll
      for (int b = 0; b < CR; ++b) {
        const int co = c0 + wc * WCO + b * 16 + fr;
        if (co < p.Cout) {
          if (p.debug_plain_store) row[co] = acc[a][b][r];
          else atomicAdd(row + co, acc[a][b][r]);
        }
      }
    }
  }
}

template <int DT, int BKK, int BCO, int WAVES_K, int WAVES_C, int G, int STAGES>
void launch_stg(const WgradParams& p, hipStream_t stream) {
  const int tilesK = (p.Ktotal + BKK - 1) / BKK;
  const int tilesC = (p.Cout + BCO - 1) / BCO;
  const bool pointwise = p.TI == 1 && p.TJ == 1 && p.ISH == 1 && p.ISW == 1 && p.dh0 == 0 &&
                         p.dw0 == 0 && p.OH == p.H && p.OW == p.W && p.Ctap <= p.Cpix;
  dim3 grid(tilesK * tilesC, p.splits, p.batch ? p.nbatch : 1);
  if (p.x_scale) {
    if constexpr (DT != DT_F32) {
      if (!pointwise || p.batch)
        throw std::invalid_argument("conv_wgrad: normalising X on the fragments needs a single pointwise layer");
      hipLaunchKernelGGL((conv_wgrad_kernel<DT, BKK, BCO, WAVES_K, WAVES_C, G, true, STAGES, true>), grid,
                         dim3(64 * WAVES_K * WAVES_C * G), 0, stream, p);
      return;
    } else {
      throw std::invalid_argument("conv_wgrad: normalising X on the fragments is for 16-bit element types");
    }
  }
  if (pointwise)
    hipLaunchKernelGGL((conv_wgrad_kernel<DT, BKK, BCO, WAVES_K, WAVES_C, G, true, STAGES>), grid,
                       dim3(64 * WAVES_K * WAVES_C * G), 0, stream, p);
  else
    hipLaunchKernelGGL((conv_wgrad_kernel<DT, BKK, BCO, WAVES_K, WAVES_C, G, false, STAGES>), grid,
                       dim3(64 * WAVES_K * WAVES_C * G), 0, stream, p);
}

template <int BKK, int BCO, int WAVES_K, int WAVES_C, int G>
void launch_grp(const WgradParams& p, int stages, hipStream_t stream) {
  // the deep pipeline exists for tiles whose every wave issues the same loads per step (all
  // tiles at least 64 wide) and whose ring fits the LDS
  constexpr int GT = 64 * WAVES_K * WAVES_C;
  constexpr bool uniform = GT / (BKK / 8) <= BP && GT / (BCO / 8) <= BP;
  if (p.dt != DT_BF16) {
    // fp16 / fp32: the two-stage kernel (these paths are bound by the matrix rate - fp32 - or
    // exist for parity with the reference's precision flags, not for the last few percent)
    if (stages != 2) throw std::invalid_argument("conv_wgrad: deep pipelines are bfloat16 only");
    if (p.dt == DT_F16) return launch_stg<DT_F16, BKK, BCO, WAVES_K, WAVES_C, G, 2>(p, stream);
    if constexpr (G * 2 * BP * (BKK + BCO) * 4 <= 160 * 1024 && BKK / 4 <= 64 && BCO / 4 <= 64) {
      return launch_stg<DT_F32, BKK, BCO, WAVES_K, WAVES_C, G, 2>(p, stream);
    } else {
      throw std::invalid_argument("conv_wgrad: tile / group combination too large for fp32");
    }
  }
  if (stages == 2) return launch_stg<DT_BF16, BKK, BCO, WAVES_K, WAVES_C, G, 2>(p, stream);
  if constexpr (uniform && G * 3 * BP * (BKK + BCO) * 2 <= 160 * 1024) {
    if (stages == 3) return launch_stg<DT_BF16, BKK, BCO, WAVES_K, WAVES_C, G, 3>(p, stream);
  }
  if constexpr (uniform && G * 4 * BP * (BKK + BCO) * 2 <= 160 * 1024) {
    if (stages == 4) return launch_stg<DT_BF16, BKK, BCO, WAVES_K, WAVES_C, G, 4>(p, stream);
  }
  throw std::invalid_argument("conv_wgrad: this tile has no pipeline of that depth");
}

template <int BKK, int BCO, int WAVES_K, int WAVES_C>
void launch_cfg(const WgradParams& p, int groups, int stages, hipStream_t stream) {
  switch (groups) {
    case 1: launch_grp<BKK, BCO, WAVES_K, WAVES_C, 1>(p, stages, stream); break;
    case 2: launch_grp<BKK, BCO, WAVES_K, WAVES_C, 2>(p, stages, stream); break;
    case 4:
      if constexpr (WAVES_K * WAVES_C == 4) {
        launch_grp<BKK, BCO, WAVES_K, WAVES_C, 4>(p, stages, stream);
        break;
      }
      [[fallthrough]];
    default: throw std::invalid_argument("conv_wgrad: unsupported number of wave groups");
  }
}

}  // namespace

void validate_wgrad_params(const WgradParams& p) {
  auto fail = [](const std::string& m) { throw std::invalid_argument("conv_wgrad: " + m); };
  if (p.batch == nullptr && (!p.x || !p.dy || !p.dw)) fail("null tensor pointer");
  if (p.batch != nullptr && (p.nbatch < 1 || p.nbatch > 1024)) fail("1..1024 layers per batched launch");
  if (p.dt != DT_BF16 && p.dt != DT_F16 && p.dt != DT_F32) fail("unknown element type");
  if (p.P != p.N * p.OH * p.OW || p.P <= 0) fail("P != N*OH*OW");
  if ((int64_t)p.N * p.H * p.W >= (1ll << 31) / 8) fail("input too large for 32-bit pixel index");
  if (p.Ctap % 8 || p.Ctap <= 0) fail("Ctap must be a positive multiple of 8");
  if (p.Ktotal != p.TI * p.TJ * p.Ctap) fail("Ktotal != TI*TJ*Ctap");
  if (p.ldy % 8 || p.ldy < ((p.Cout + 7) / 8) * 8) fail("ldy must be a multiple of 8 covering Cout");
  if (p.splits < 1 || p.pix_per_split % BP || (int64_t)p.splits * p.pix_per_split < p.P)
    fail("bad split-K partition");
  if (p.Cpad <= 0 || p.Creal <= 0 || p.Creal > p.Cpad || p.TJreal > p.TJpad) fail("bad k decode");
  if (p.Ctap > p.Cpix) {
    const int span = (p.Ctap + p.Cpix - 1) / p.Cpix;
    if (p.dws < 0 || p.dhs < 0 || p.dw0 < 0 || p.dh0 < 0) fail("multi-pixel taps need forward taps");
    if ((p.OW - 1) * p.ISW + p.dw0 + (p.TJ - 1) * p.dws + span > p.W)
      fail("multi-pixel tap overruns the input row");
    if ((p.OH - 1) * p.ISH + p.dh0 + (p.TI - 1) * p.dhs >= p.H) fail("multi-pixel tap overruns H");
    if ((p.Cpix * p.ISW) % 8 || (p.Cpix * p.W) % 8) fail("multi-pixel taps need 16-byte aligned rows");
  } else if (p.Cpix % 8) {
    fail("Cpix must be a multiple of 8");
  }
}

// cfg = tile id (0..15) + 16 * log2(groups) + 64 * (stages - 2): 1, 2 or 4 wave groups per block
// (tile 8, whose groups have eight waves, allows 1 or 2); LDS ring of 2, 3 or 4 stages.
// cfg 192: the whole-rows kernel of conv_wgrad_rows.hip.
void launch_conv_wgrad(const WgradParams& p, int cfg, hipStream_t stream) {
  validate_wgrad_params(p);
  if ((p.x_scale == nullptr) != (p.x_shift == nullptr))
    throw std::invalid_argument("conv_wgrad: x_scale / x_shift come together");
  if (cfg == WGRAD_ROWS_CFG) {
    if (launch_conv_wgrad_rows(p, cfg, stream)) return;
    cfg = 0;                   // a descriptor the whole-rows kernel cannot serve: the 128x128 tile
  }
  if (cfg < 0 || cfg >= 192 || (cfg % 64) >= 48)
    throw std::invalid_argument("conv_wgrad: unknown tile config");
  const int stages = 2 + cfg / 64;
  cfg %= 64;
  const int g = 1 << (cfg / 16);
  switch (cfg % 16) {
    case 0: launch_cfg<128, 128, 2, 2>(p, g, stages, stream); break;
    case 1: launch_cfg<128, 64, 2, 2>(p, g, stages, stream); break;
    case 2: launch_cfg<64, 128, 2, 2>(p, g, stages, stream); break;
    case 3: launch_cfg<64, 64, 2, 2>(p, g, stages, stream); break;
    case 4: launch_cfg<128, 32, 4, 1>(p, g, stages, stream); break;
    case 5: launch_cfg<128, 16, 4, 1>(p, g, stages, stream); break;
    case 6: launch_cfg<32, 32, 2, 2>(p, g, stages, stream); break;
    case 7: launch_cfg<64, 16, 4, 1>(p, g, stages, stream); break;
    case 8: launch_cfg<256, 128, 4, 2>(p, g, stages, stream); break;   // 8 waves: 96 FLOP per L2 byte
    default: throw std::invalid_argument("conv_wgrad: unknown tile config");
  }
}
