// Fused BatchNorm kernels for NHWC activations (bf16 / fp16 / fp32 storage, template DT) (training forward / backward, inference).
//
// Training forward is   z = relu?( gamma * (y - mean) * invstd + beta (+ residual) )
// with per-channel batch statistics over all P = N*H*W pixels, computed in fp32 from the
// bf16 convolution output.  Three launches: `bn_stats` sweeps the tensor once and leaves one
// row of partial sums per block in a workspace (plain stores - same-address atomics from a
// thousand blocks serialise in L2 and were 10x slower than the sweep itself);
// `bn_fwd_finalize` (one block per 32 channels) reduces the partials, derives scale / shift,
// stores mean / invstd for backward and updates the moving averages with the Keras/TensorFlow
// rule  moving = moving * m + batch * (1 - m)  (batch variance *unbiased* for the moving
// average); `bn_apply` does normalisation + residual add + ReLU in one sweep.
//
// Backward mirrors it: `bn_bwd_reduce` (ReLU mask from the saved output, dgamma / dbeta
// partials), `bn_bwd_finalize` (writes dgamma / dbeta into the flat fp32 gradient buffer),
// `bn_bwd_apply`
//   dy = gamma * invstd * (g - dbeta/P - xhat * dgamma/P),   g = dz masked by ReLU
// which also emits g itself as the gradient of the residual branch when asked.  The sums are
// deterministic (fixed partition, fixed order).
//
// Two traffic savers: the forward sweep can leave the ReLU mask as one bit per element
// (`mask`, a byte per 8 channels) so that both backward sweeps read 1/16 of what re-reading
// the bf16 output would cost; and the statistics sweep disappears when the producing
// convolution already accumulated sum / sum of squares in its epilogue (`fused_stats`: the
// workspace then holds `G` replicas filled with atomic adds).
//
// All kernels are bandwidth bound: every thread moves 16 bytes (8 channels) per access, a row
// of C channels is covered by C/8 consecutive lanes, and a block walks rows grid-strided.
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "common.h"
#include "elementwise.h"

namespace {

constexpr int TPB = 256;

// 8 consecutive per-channel constants with two 16-byte loads (the per-thread prologues used to
// issue up to 40 dependent 4-byte loads, which cost more than the sweep on short grids).
__device__ __forceinline__ void load8(const float* __restrict__ p, float (&v)[8]) {
  const f32x4 lo = *reinterpret_cast<const f32x4*>(p);
  const f32x4 hi = *reinterpret_cast<const f32x4*>(p + 4);
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
}
constexpr int MAX_PARTIAL_BLOCKS = 512;   // rows of the partial-sum workspace

// Per-block partial sums: threads that own the same channel chunk (same tid % tpr) are summed
// through LDS and the block writes one row [NV/8 quantities][C] of the workspace with plain
// stores.  No atomics: thousands of blocks adding to the same few addresses serialise in L2.
// ATOMIC: `ws_row` is one of a few replica rows shared by many blocks: the block's sums are
// added to it (the fused-statistics convolution epilogues accumulate the same way).
template <int NV, bool ATOMIC = false>
__device__ __forceinline__ void block_reduce_to_partials(float (&part)[NV], int tpr, int chunk,
                                                         int rowid, float* ws_row, int C) {
  __shared__ float red[TPB * NV];
  const int tid = threadIdx.x;
  const int rpb = TPB / tpr;
  if (tpr < 64) {
    // several lanes of a wave own the same chunk (lane % tpr): butterfly over those first,
    // then one LDS round over the four waves (a serial loop over the block's 32..256 rows made
    // this epilogue longer than the sweep itself on small tensors)
    const int lane = tid & 63, wave = tid >> 6;
    for (int off = tpr; off < 64; off <<= 1) {
#pragma unroll
      for (int i = 0; i < NV; ++i) part[i] += __shfl_xor(part[i], off);
    }
    if (lane < tpr) {
#pragma unroll
      for (int i = 0; i < NV; ++i) red[(i * 4 + wave) * 64 + lane] = part[i];
    }
    __syncthreads();
    if (tid < tpr) {                       // rowid == 0 and chunk == tid
#pragma unroll
      for (int i = 0; i < NV; ++i)
        part[i] = (red[(i * 4 + 0) * 64 + tid] + red[(i * 4 + 1) * 64 + tid]) +
                  (red[(i * 4 + 2) * 64 + tid] + red[(i * 4 + 3) * 64 + tid]);
    }
  } else if (rpb > 1) {
#pragma unroll
    for (int i = 0; i < NV; ++i) red[i * TPB + tid] = part[i];
    __syncthreads();
    if (rowid == 0) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        float s = 0.f;
        for (int r = 0; r < rpb; ++r) s += red[i * TPB + r * tpr + chunk];
        part[i] = s;
      }
    }
  }
  if (rowid == 0 && ATOMIC) {
#pragma unroll
    for (int qn = 0; qn < NV / 8; ++qn) {
#pragma unroll
      for (int c = 0; c < 8; ++c) atomicAdd(ws_row + qn * C + chunk * 8 + c, part[qn * 8 + c]);
    }
  } else if (rowid == 0) {
#pragma unroll
    for (int qn = 0; qn < NV / 8; ++qn) {
      f32x4 lo = {part[qn * 8 + 0], part[qn * 8 + 1], part[qn * 8 + 2], part[qn * 8 + 3]};
      f32x4 hi = {part[qn * 8 + 4], part[qn * 8 + 5], part[qn * 8 + 6], part[qn * 8 + 7]};
      *reinterpret_cast<f32x4*>(ws_row + qn * C + chunk * 8) = lo;
      *reinterpret_cast<f32x4*>(ws_row + qn * C + chunk * 8 + 4) = hi;
    }
  }
}

template <int DT>
__global__ __launch_bounds__(TPB) void bn_stats_kernel(const void* __restrict__ y,
                                                       float* __restrict__ ws, int P, int C) {
  const int tpr = C >> 3, rpb = TPB / tpr;
  const int chunk = threadIdx.x % tpr, rowid = threadIdx.x / tpr;
  float part[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) part[i] = 0.f;
  const int stride = gridDim.x * rpb;
  int row = blockIdx.x * rpb + rowid;
  for (; row + 3 * stride < P; row += 4 * stride) {     // four independent 16-byte loads in flight
    Raw8<DT> v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      v[u] = ldraw<DT>(y, (size_t)(row + u * stride) * C + chunk * 8);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float f[8];
      unpack8<DT>(v[u], f);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        part[c] += f[c];
        part[8 + c] += f[c] * f[c];
      }
    }
  }
  for (; row < P; row += stride) {
    const Raw8<DT> v = ldraw<DT>(y, (size_t)row * C + chunk * 8);
    float f[8];
    unpack8<DT>(v, f);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      part[c] += f[c];
      part[8 + c] += f[c] * f[c];
    }
  }
  block_reduce_to_partials<16>(part, tpr, chunk, rowid, ws + (size_t)blockIdx.x * 2 * C, C);
}

// Sums the G partial rows of the workspace: column j of [G][cols].  A finalize block has 1024
// threads = 32 columns x 32 row lanes (the loop is latency bound: a thread walks G/32 rows with
// 8 independent loads in flight).
constexpr int FIN_TPB = 1024, FIN_RL = FIN_TPB / 32;

__device__ __forceinline__ float column_sum(const float* __restrict__ ws, int G, int cols, int col,
                                            int rl, float* red) {
  float s = 0.f;
  if (col < cols) {
    const float* p = ws + col;
    int g = rl;
    for (; g + 7 * FIN_RL < G; g += 8 * FIN_RL) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(g + FIN_RL * u) * cols];
      s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; g < G; g += FIN_RL) s += p[(size_t)g * cols];
  }
  red[rl * 32 + (threadIdx.x & 31)] = s;
  __syncthreads();
  float t = 0.f;
  if (rl == 0) {
#pragma unroll
    for (int r = 0; r < FIN_RL; ++r) t += red[r * 32 + (threadIdx.x & 31)];
  }
  __syncthreads();
  return t;
}

struct BnFinalizeArgs {
  const float* ws; int G;
  const float* gamma; const float* beta; float* mm; float* mv;
  float* save_mean; float* save_invstd; float* scale; float* shift;
  int P, C; float eps, momentum;
};

// One block per 32 channels: mean / variance from the partial sums, scale / shift for the
// apply sweep, saved statistics for backward, moving-average update.
__global__ __launch_bounds__(FIN_TPB) void bn_fwd_finalize_kernel(const BnFinalizeArgs a) {
  __shared__ float red[FIN_TPB];
  const int ch = blockIdx.x * 32 + (threadIdx.x & 31), rl = threadIdx.x >> 5;
  const float s1 = column_sum(a.ws, a.G, 2 * a.C, ch, rl, red);
  const float s2 = column_sum(a.ws, a.G, 2 * a.C, a.C + ch, rl, red);
  if (rl == 0 && ch < a.C) {
    const float invP = 1.f / (float)a.P;
    const float mean = s1 * invP;
    const float var = fmaxf(s2 * invP - mean * mean, 0.f);
    const float invstd = rsqrtf(var + a.eps);
    const float sc = a.gamma[ch] * invstd;
    a.scale[ch] = sc;
    a.shift[ch] = a.beta[ch] - mean * sc;
    a.save_mean[ch] = mean;
    a.save_invstd[ch] = invstd;
    const float unbiased = a.P > 1 ? var * ((float)a.P / (float)(a.P - 1)) : var;
    a.mm[ch] = a.mm[ch] * a.momentum + mean * (1.f - a.momentum);
    a.mv[ch] = a.mv[ch] * a.momentum + unbiased * (1.f - a.momentum);
  }
}

struct BnApplyArgs {
  const void* y; const void* res; void* z; uint8_t* mask;
  const float* scale; const float* shift;        // training: produced by the finalize kernel
  const float* gamma; const float* beta; const float* mm; const float* mv;   // inference
  int P, C; float eps; int relu;
  // FOLD: the finalize step runs in the prologue of every block (see fold_columns)
  const float* fws; int fG; float momentum;
  float* umm; float* umv; float* save_mean; float* save_invstd; float* oscale; float* oshift;
  // Residual given as the *input* of another BatchNorm (the projection shortcut's, which has
  // no ReLU) plus that layer's scale / shift: res := res * rscale + rshift on the fly, so the
  // shortcut's normalised tensor is never written or read back.
  const float* rscale; const float* rshift;
  int rev;                  // 1: passes over the rows run back to front (set by the launch plan)
};

// Column sums of G rows of `cols` floats (cols = 2 * C <= 512, a power of two >= 16) by one
// block of 256 threads: float4 loads spread over (column group, row lane), one LDS round.
// Lets a sweep over a narrow layer derive its per-channel constants itself from the few
// replica rows the convolution epilogues filled, instead of waiting for a 5 us finalize
// launch; a thread issues at most G * cols / 1024 loads.  Result in sums[0 .. cols).
constexpr int FOLD_MAX_C = 256;
__device__ __forceinline__ void fold_columns(const float* __restrict__ ws, int G, int cols,
                                             float* part /* [1024] */, float* sums /* [512] */) {
  const int ncg = cols >> 2, RL = TPB / ncg;
  const int cg = threadIdx.x % ncg, rl = threadIdx.x / ncg;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int g = rl; g < G; g += RL) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(ws + (size_t)g * cols + cg * 4);
    s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
  }
  *reinterpret_cast<f32x4*>(part + rl * cols + cg * 4) = s;
  __syncthreads();
  for (int col = threadIdx.x; col < cols; col += TPB) {
    float t = 0.f;
    for (int r = 0; r < RL; ++r) t += part[r * cols + col];
    sums[col] = t;
  }
  __syncthreads();
}

// NTM: cache policy, bit 0 - the loads of y / res, bit 1 - the store of z are non-temporal
template <int DT, bool FOLD, int NTM = 0>
__global__ __launch_bounds__(TPB) void bn_apply_kernel(const BnApplyArgs a) {
  const int C = a.C, tpr = C >> 3, rpb = TPB / tpr;
  const int chunk = threadIdx.x % tpr, rowid = threadIdx.x / tpr;
  float scale[8], shift[8];
  if constexpr (FOLD) {
    __shared__ __attribute__((aligned(16))) float part[1024];
    __shared__ __attribute__((aligned(16))) float sums[2 * FOLD_MAX_C];
    fold_columns(a.fws, a.fG, 2 * C, part, sums);
    float* sc = part;                // reuse: [C] scale, [C] shift
    float* sh = part + FOLD_MAX_C;
    for (int ch = threadIdx.x; ch < C; ch += TPB) {
      const float invP = 1.f / (float)a.P;
      const float mean = sums[ch] * invP;
      const float var = fmaxf(sums[C + ch] * invP - mean * mean, 0.f);
      const float invstd = rsqrtf(var + a.eps);
      const float k = a.gamma[ch] * invstd;
      sc[ch] = k;
      sh[ch] = a.beta[ch] - mean * k;
      if (blockIdx.x == 0) {         // one block publishes what backward / inference need
        a.oscale[ch] = k;
        a.oshift[ch] = sh[ch];
        a.save_mean[ch] = mean;
        a.save_invstd[ch] = invstd;
        const float unbiased = a.P > 1 ? var * ((float)a.P / (float)(a.P - 1)) : var;
        a.umm[ch] = a.umm[ch] * a.momentum + mean * (1.f - a.momentum);
        a.umv[ch] = a.umv[ch] * a.momentum + unbiased * (1.f - a.momentum);
      }
    }
    __syncthreads();
    load8(sc + chunk * 8, scale);
    load8(sh + chunk * 8, shift);
  } else if (a.scale) {
    load8(a.scale + chunk * 8, scale);
    load8(a.shift + chunk * 8, shift);
  } else {
    float g[8], b[8], m[8], v[8];
    load8(a.gamma + chunk * 8, g);
    load8(a.beta + chunk * 8, b);
    load8(a.mm + chunk * 8, m);
    load8(a.mv + chunk * 8, v);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      scale[c] = g[c] * rsqrtf(v[c] + a.eps);
      shift[c] = b[c] - m[c] * scale[c];
    }
  }
  // The trip count is uniform per block (row0 is the block's first row of a pass) so that the
  // four lanes of a quad can merge their mask bytes into one 32-bit store: the byte of thread
  // t in a pass sits at  row0 * tpr + t,  and row0 * tpr is a multiple of 256.  Four passes
  // are loaded before any is used: with one 16-byte load per thread in flight the sweep is
  // latency bound at about half of what the memory system delivers.
  float rsc[8], rsh[8];
  const bool raff = a.rscale != nullptr;
  if (raff) {
    load8(a.rscale + chunk * 8, rsc);
    load8(a.rshift + chunk * 8, rsh);
  }
  constexpr int U = 4;
  const int stride = gridDim.x * rpb;
  const int npass = (a.P + stride - 1) / stride;
  for (int p0 = 0; p0 < npass; p0 += U) {
    Raw8<DT> yq[U], rq[U];
    bool valid[U];
    int prow[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = p0 + u;
      prow[u] = blockIdx.x * rpb + (a.rev ? npass - 1 - p : p) * stride;
      const int row = prow[u] + rowid;
      valid[u] = p < npass && row < a.P;
      const size_t off = (size_t)row * C + chunk * 8;
      yq[u] = valid[u] ? ldraw<DT, (NTM & 1) != 0>(a.y, off) : zero_raw<DT>();
      rq[u] = (valid[u] && a.res) ? ldraw<DT, (NTM & 1) != 0>(a.res, off) : zero_raw<DT>();
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int prow0 = prow[u];
      const size_t off = (size_t)(prow0 + rowid) * C + chunk * 8;
      float f[8];
      uint32_t bits = 0;
      if (valid[u]) {
        unpack8<DT>(yq[u], f);
#pragma unroll
        for (int c = 0; c < 8; ++c) f[c] = f[c] * scale[c] + shift[c];
        if (a.res) {
          float r[8];
          unpack8<DT>(rq[u], r);
          if (raff) {
#pragma unroll
            for (int c = 0; c < 8; ++c) r[c] = r[c] * rsc[c] + rsh[c];
          }
#pragma unroll
          for (int c = 0; c < 8; ++c) f[c] += r[c];
        }
        if (a.relu) {
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            bits |= (f[c] > 0.f ? 1u : 0u) << c;
            f[c] = fmaxf(f[c], 0.f);
          }
        }
        straw<DT, (NTM & 2) != 0>(a.z, off, f);
      }
      if (a.relu && a.mask && p0 + u < npass && prow0 < a.P) {   // block-uniform condition
        uint32_t w = bits << (8 * (threadIdx.x & 3));
        w |= __shfl_xor(w, 1);
        w |= __shfl_xor(w, 2);
        if ((threadIdx.x & 3) == 0 && valid[u])
          *reinterpret_cast<uint32_t*>(a.mask + (size_t)prow0 * tpr + threadIdx.x) = w;
      }
    }
  }
}

struct BnBwdArgs {
  const void* dz; const void* y;
  const void* z;          // forward output: ReLU mask = z > 0 (used when `mask` is null)
  const uint8_t* mask;      // bit mask written by the forward sweep; both null: no ReLU
  const float* mean; const float* invstd; const float* gamma;
  float* dgamma; float* dbeta;
  void* dy; void* dres;
  float* ws; int G;
  int P, C;
  int raw;                  // 1: `ws` rows hold [sum g | sum g*y] (from the data-gradient
                            // epilogues) instead of [dgamma | dbeta]
  // Second BatchNorm fed by the same gradient (NB == 2 instantiations): the projection
  // shortcut's layer next to the block's last one.  Both see g = dz gated by the mask, so one
  // sweep reads dz and the mask once for the two of them.  No ReLU of its own, no dres.
  const void* y2; const float* mean2; const float* invstd2; const float* gamma2;
  float* dgamma2; float* dbeta2; void* dy2; float* ws2;
  // TAIL instantiations of the reduction: `ws` (`ws2`) are TAIL_REPLICAS zeroed rows of
  // [dgamma | dbeta] that the blocks add to, `ticket` counts finished blocks; the last one sums
  // the rows, stores dgamma / dbeta and leaves rows and ticket zeroed for the next launch.
  unsigned int* ticket;
  int rev_reduce, rev_apply;   // passes of the reduction / apply sweep run back to front
  // ReLU mask from the sign of the recomputed value y * relu_scale[c] + relu_shift[c] (the forward
  // pass normalised this tensor inside its consumer's tile and wrote neither z nor mask bits)
  const float* relu_scale; const float* relu_shift;
};

constexpr int TAIL_REPLICAS = 8;

template <int DT, int NB, bool TAIL = false, int NTM = 0>
__global__ __launch_bounds__(TPB) void bn_bwd_reduce_kernel(const BnBwdArgs a) {
  const int C = a.C, tpr = C >> 3, rpb = TPB / tpr;
  const int chunk = threadIdx.x % tpr, rowid = threadIdx.x / tpr;
  float mean[8], invstd[8], mean2[8], invstd2[8];
  load8(a.mean + chunk * 8, mean);
  load8(a.invstd + chunk * 8, invstd);
  if constexpr (NB == 2) {
    load8(a.mean2 + chunk * 8, mean2);
    load8(a.invstd2 + chunk * 8, invstd2);
  }
  float part[16], part2[8];          // [dgamma | dbeta], dgamma of the second layer
#pragma unroll
  for (int i = 0; i < 16; ++i) part[i] = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) part2[i] = 0.f;
  const int mode = a.relu_scale ? 3 : (a.mask ? 2 : (a.z ? 1 : 0));
  float rsc[8], rsh[8];
  if (mode == 3) {
    load8(a.relu_scale + chunk * 8, rsc);
    load8(a.relu_shift + chunk * 8, rsh);
  }
  auto accumulate = [&](const Raw8<DT>& gv, const Raw8<DT>& yq, const Raw8<DT>& zq, uint32_t bits,
                        const Raw8<DT>& y2q) {
    float g[8], yv[8];
    unpack8<DT>(gv, g);
    unpack8<DT>(yq, yv);
    if (mode == 3) {
#pragma unroll
      for (int c = 0; c < 8; ++c) g[c] = yv[c] * rsc[c] + rsh[c] > 0.f ? g[c] : 0.f;
    } else if (mode == 2) {
#pragma unroll
      for (int c = 0; c < 8; ++c) g[c] = (bits >> c) & 1u ? g[c] : 0.f;
    } else if (mode == 1) {
      float zv[8];
      unpack8<DT>(zq, zv);
#pragma unroll
      for (int c = 0; c < 8; ++c) g[c] = zv[c] > 0.f ? g[c] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      part[c] += g[c] * ((yv[c] - mean[c]) * invstd[c]);   // dgamma
      part[8 + c] += g[c];                                  // dbeta
    }
    if constexpr (NB == 2) {
      float y2v[8];
      unpack8<DT>(y2q, y2v);
#pragma unroll
      for (int c = 0; c < 8; ++c) part2[c] += g[c] * ((y2v[c] - mean2[c]) * invstd2[c]);
    }
  };
  // Passes (one row per thread, `stride` rows per pass) in ascending or descending order: the
  // thread's rows are  r0 + p * stride  for p in [0, np).
  const int stride = gridDim.x * rpb;
  const int r0 = blockIdx.x * rpb + rowid;
  const int np = r0 < a.P ? (a.P - r0 + stride - 1) / stride : 0;
  const int first = a.rev_reduce ? r0 + (np - 1) * stride : r0;
  const int step = a.rev_reduce ? -stride : stride;
  int row = first, p = 0;
  for (; p + 3 < np; p += 4, row += 4 * step) {      // four rows of loads in flight
    Raw8<DT> gv[4], yq[4], zq[4], y2q[4];
    uint32_t bits[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t off = (size_t)(row + u * step) * C + chunk * 8;
      gv[u] = ldraw<DT, (NTM & 1) != 0>(a.dz, off);
      yq[u] = ldraw<DT, (NTM & 1) != 0>(a.y, off);
      zq[u] = mode == 1 ? ldraw<DT, (NTM & 1) != 0>(a.z, off) : zero_raw<DT>();
      bits[u] = mode == 2 ? a.mask[(size_t)(row + u * step) * tpr + chunk] : 0u;
      y2q[u] = NB == 2 ? ldraw<DT, (NTM & 1) != 0>(a.y2, off) : zero_raw<DT>();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) accumulate(gv[u], yq[u], zq[u], bits[u], y2q[u]);
  }
  for (; p < np; ++p, row += step) {
    const size_t off = (size_t)row * C + chunk * 8;
    const Raw8<DT> gv = ldraw<DT, (NTM & 1) != 0>(a.dz, off);
    const Raw8<DT> yq = ldraw<DT, (NTM & 1) != 0>(a.y, off);
    const Raw8<DT> zq = mode == 1 ? ldraw<DT, (NTM & 1) != 0>(a.z, off) : zero_raw<DT>();
    const uint32_t bits = mode == 2 ? a.mask[(size_t)row * tpr + chunk] : 0u;
    const Raw8<DT> y2q = NB == 2 ? ldraw<DT, (NTM & 1) != 0>(a.y2, off) : zero_raw<DT>();
    accumulate(gv, yq, zq, bits, y2q);
  }
  const size_t wrow = (size_t)(TAIL ? blockIdx.x % TAIL_REPLICAS : blockIdx.x) * 2 * C;
  if constexpr (NB == 2) {
    float second[16];                // [dgamma2 | dbeta] (same g: same dbeta), copied before the
#pragma unroll                       // first reduction overwrites `part` in the writing threads
    for (int c = 0; c < 8; ++c) { second[c] = part2[c]; second[8 + c] = part[8 + c]; }
    block_reduce_to_partials<16, TAIL>(part, tpr, chunk, rowid, a.ws + wrow, C);
    __syncthreads();                 // the LDS scratch of the reduction is reused
    block_reduce_to_partials<16, TAIL>(second, tpr, chunk, rowid, a.ws2 + wrow, C);
  } else {
    block_reduce_to_partials<16, TAIL>(part, tpr, chunk, rowid, a.ws + wrow, C);
  }
  if constexpr (TAIL) {
    // The block that takes the last ticket finds every other block's sums in the replica rows:
    // each block fences its atomics before it draws, the last one fences before it reads.
    __shared__ unsigned int is_last;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) is_last = atomicAdd(a.ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (is_last == 0u) return;
    __threadfence();
    for (int ch = threadIdx.x; ch < C; ch += TPB) {
      float dg = 0.f, db = 0.f, dg2 = 0.f;
#pragma unroll
      for (int r = 0; r < TAIL_REPLICAS; ++r) {
        float* row = a.ws + (size_t)r * 2 * C;
        dg += __hip_atomic_load(row + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        db += __hip_atomic_load(row + C + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        row[ch] = 0.f;
        row[C + ch] = 0.f;
        if constexpr (NB == 2) {
          float* row2 = a.ws2 + (size_t)r * 2 * C;
          dg2 += __hip_atomic_load(row2 + ch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          row2[ch] = 0.f;
          row2[C + ch] = 0.f;
        }
      }
      a.dgamma[ch] = dg;     // plain stores into the flat gradient buffer
      a.dbeta[ch] = db;
      if constexpr (NB == 2) {
        a.dgamma2[ch] = dg2;
        a.dbeta2[ch] = db;
      }
    }
    if (threadIdx.x == 0) *a.ticket = 0u;
  }
}

__global__ __launch_bounds__(FIN_TPB) void bn_bwd_finalize_kernel(const BnBwdArgs a) {
  __shared__ float red[FIN_TPB];
  const int ch = blockIdx.x * 32 + (threadIdx.x & 31), rl = threadIdx.x >> 5;
  float dg = column_sum(a.ws, a.G, 2 * a.C, ch, rl, red);
  float db = column_sum(a.ws, a.G, 2 * a.C, a.C + ch, rl, red);
  if (rl == 0 && ch < a.C) {
    if (a.raw) {           // dg = sum g, db = sum g*y  ->  dbeta, dgamma = sum g * xhat
      const float sg = dg, sgy = db;
      db = sg;
      dg = (sgy - a.mean[ch] * sg) * a.invstd[ch];
    }
    a.dgamma[ch] = dg;     // plain stores into the flat gradient buffer
    a.dbeta[ch] = db;
  }
}

// FOLD: `ws` holds G replica rows of [sum g | sum g*y] (left by the data-gradient epilogues) of
// a narrow layer; every block derives dbeta / dgamma itself (block 0 stores them into the
// gradient buffer) instead of waiting for a finalize launch.
template <int DT, bool FOLD, int NB = 1, int NTM = 0>
__global__ __launch_bounds__(TPB) void bn_bwd_apply_kernel(const BnBwdArgs a) {
  const int C = a.C, tpr = C >> 3, rpb = TPB / tpr;
  const int chunk = threadIdx.x % tpr, rowid = threadIdx.x / tpr;
  const float invP = 1.f / (float)a.P;
  float mean[8], invstd[8], scale[8], c1[8], c2[8];
  load8(a.mean + chunk * 8, mean);
  load8(a.invstd + chunk * 8, invstd);
  load8(a.gamma + chunk * 8, scale);
  if constexpr (FOLD) {
    __shared__ __attribute__((aligned(16))) float part[1024];
    __shared__ __attribute__((aligned(16))) float sums[2 * FOLD_MAX_C];
    fold_columns(a.ws, a.G, 2 * C, part, sums);
    float* db = part;                // reuse: [C] dbeta, [C] dgamma
    float* dg = part + FOLD_MAX_C;
    for (int ch = threadIdx.x; ch < C; ch += TPB) {
      const float sg = sums[ch], sgy = sums[C + ch];
      const float g2 = (sgy - a.mean[ch] * sg) * a.invstd[ch];
      db[ch] = sg;
      dg[ch] = g2;
      if (blockIdx.x == 0) {
        a.dbeta[ch] = sg;
        a.dgamma[ch] = g2;
      }
    }
    __syncthreads();
    load8(db + chunk * 8, c1);
    load8(dg + chunk * 8, c2);
  } else {
    load8(a.dbeta + chunk * 8, c1);
    load8(a.dgamma + chunk * 8, c2);
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    scale[c] *= invstd[c];
    c1[c] *= invP;
    c2[c] *= invP;
  }
  float mean2[8], invstd2[8], scale2[8], c12[8], c22[8];
  if constexpr (NB == 2) {
    load8(a.mean2 + chunk * 8, mean2);
    load8(a.invstd2 + chunk * 8, invstd2);
    load8(a.gamma2 + chunk * 8, scale2);
    load8(a.dbeta2 + chunk * 8, c12);
    load8(a.dgamma2 + chunk * 8, c22);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      scale2[c] *= invstd2[c];
      c12[c] *= invP;
      c22[c] *= invP;
    }
  }
  const int mode = a.relu_scale ? 3 : (a.mask ? 2 : (a.z ? 1 : 0));
  float rsc[8], rsh[8];
  if (mode == 3) {
    load8(a.relu_scale + chunk * 8, rsc);
    load8(a.relu_shift + chunk * 8, rsh);
  }
  auto finish = [&](const Raw8<DT>& gv, const Raw8<DT>& yq, const Raw8<DT>& zq, uint32_t bits, size_t off,
                    const Raw8<DT>& y2q) {
    float g[8], yv[8];
    unpack8<DT>(gv, g);
    unpack8<DT>(yq, yv);
    if (mode == 3) {
#pragma unroll
      for (int c = 0; c < 8; ++c) g[c] = yv[c] * rsc[c] + rsh[c] > 0.f ? g[c] : 0.f;
    } else if (mode == 2) {
#pragma unroll
      for (int c = 0; c < 8; ++c) g[c] = (bits >> c) & 1u ? g[c] : 0.f;
    } else if (mode == 1) {
      float zv[8];
      unpack8<DT>(zq, zv);
#pragma unroll
      for (int c = 0; c < 8; ++c) g[c] = zv[c] > 0.f ? g[c] : 0.f;
    }
    if (a.dres) straw<DT>(a.dres, off, g);
    float o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float xhat = (yv[c] - mean[c]) * invstd[c];
      o[c] = scale[c] * (g[c] - c1[c] - xhat * c2[c]);
    }
    straw<DT, (NTM & 2) != 0>(a.dy, off, o);
    if constexpr (NB == 2) {
      float y2v[8], o2[8];
      unpack8<DT>(y2q, y2v);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float xhat = (y2v[c] - mean2[c]) * invstd2[c];
        o2[c] = scale2[c] * (g[c] - c12[c] - xhat * c22[c]);
      }
      straw<DT, (NTM & 2) != 0>(a.dy2, off, o2);
    }
  };
  const int stride = gridDim.x * rpb;
  const int r0 = blockIdx.x * rpb + rowid;
  const int np = r0 < a.P ? (a.P - r0 + stride - 1) / stride : 0;
  const int first = a.rev_apply ? r0 + (np - 1) * stride : r0;
  const int step = a.rev_apply ? -stride : stride;
  int row = first, p = 0;
  for (; p + 3 < np; p += 4, row += 4 * step) {      // four rows of loads in flight
    Raw8<DT> gv[4], yq[4], zq[4], y2q[4];
    uint32_t bits[4];
    size_t off[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      off[u] = (size_t)(row + u * step) * C + chunk * 8;
      gv[u] = ldraw<DT, (NTM & 1) != 0>(a.dz, off[u]);
      yq[u] = ldraw<DT, (NTM & 1) != 0>(a.y, off[u]);
      zq[u] = mode == 1 ? ldraw<DT, (NTM & 1) != 0>(a.z, off[u]) : zero_raw<DT>();
      bits[u] = mode == 2 ? a.mask[(size_t)(row + u * step) * tpr + chunk] : 0u;
      y2q[u] = NB == 2 ? ldraw<DT, (NTM & 1) != 0>(a.y2, off[u]) : zero_raw<DT>();
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) finish(gv[u], yq[u], zq[u], bits[u], off[u], y2q[u]);
  }
  for (; p < np; ++p, row += step) {
    const size_t off = (size_t)row * C + chunk * 8;
    const Raw8<DT> gv = ldraw<DT, (NTM & 1) != 0>(a.dz, off);
    const Raw8<DT> yq = ldraw<DT, (NTM & 1) != 0>(a.y, off);
    const Raw8<DT> zq = mode == 1 ? ldraw<DT, (NTM & 1) != 0>(a.z, off) : zero_raw<DT>();
    const uint32_t bits = mode == 2 ? a.mask[(size_t)row * tpr + chunk] : 0u;
    const Raw8<DT> y2q = NB == 2 ? ldraw<DT, (NTM & 1) != 0>(a.y2, off) : zero_raw<DT>();
    finish(gv, yq, zq, bits, off, y2q);
  }
}

template <int DT>
__global__ __launch_bounds__(TPB) void mask_apply_kernel(const void* __restrict__ x,
                                                         const uint8_t* __restrict__ mask,
                                                         void* __restrict__ out, size_t nvec) {
  for (size_t i = (size_t)blockIdx.x * TPB + threadIdx.x; i < nvec; i += (size_t)gridDim.x * TPB) {
    float f[8];
    unpack8<DT>(ldraw<DT>(x, i * 8), f);
    const uint32_t b = mask[i];
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] = (b >> c) & 1u ? f[c] : 0.f;
    straw<DT>(out, i * 8, f);
  }
}

// ---- BatchNorm + ReLU feeding a max-pool (the ResNet-50 stem) --------------------------------
// Forward: the pooled tensor is produced straight from the convolution output; the full
// resolution activation is never written or re-read.  The winning tap is stored per pooled
// element as in the plain max-pool (first maximum in row-major tap order); 255 means that no
// tap was positive, i.e. the ReLU zeroed the whole window and no gradient flows.
// Backward: the gradient of a full-resolution element is gathered from the at most
// ceil(k/stride)^2 pooled elements whose window contains it (those it won), inside the two
// BatchNorm backward sweeps, so the scattered gradient tensor is never materialised either.
// Threads own a fixed 8-channel chunk (256 % (C/8) == 0), a block walks image rows.
struct BnPoolArgs {
  const void* y;             // [N,H,W,C] convolution output
  void* zp;                  // [N,OH,OW,C] pooled activation (forward)
  uint8_t* idx;                // [N,OH,OW,C] winning tap / 255
  const void* dzp;           // [N,OH,OW,C] gradient of the pooled activation (backward)
  void* dy;                  // [N,H,W,C] gradient of the convolution output (backward)
  const float* scale; const float* shift;                    // forward
  const float* mean; const float* invstd; const float* gamma;
  const float* dgamma; const float* dbeta;                   // backward apply
  float* ws;                                                 // backward reduce partial rows
  int N, H, W, C, OH, OW, k, stride, pt, pl;
};

template <int DT>
__global__ __launch_bounds__(TPB) void bn_pool_fwd_kernel(const BnPoolArgs a) {
  const int C = a.C;
  const uint32_t cpr = (uint32_t)C >> 3;
  const uint32_t chunk = threadIdx.x % cpr;
  float scale[8], shift[8];
  load8(a.scale + chunk * 8, scale);
  load8(a.shift + chunk * 8, shift);
  const uint32_t per_row = (uint32_t)a.OW * cpr;
  for (uint32_t row = blockIdx.x; row < (uint32_t)(a.N * a.OH); row += gridDim.x) {
    const uint32_t n = row / (uint32_t)a.OH, oh = row - n * (uint32_t)a.OH;
    for (uint32_t j = threadIdx.x; j < per_row; j += TPB) {
      const uint32_t ow = j / cpr;
      float best[8];
      uint32_t bi[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) { best[c] = 0.f; bi[c] = 255u; }
      for (int kh = 0; kh < a.k; ++kh) {
        const int hi = (int)oh * a.stride - a.pt + kh;
        if ((unsigned)hi >= (unsigned)a.H) continue;
        for (int kw = 0; kw < a.k; ++kw) {
          const int wi = (int)ow * a.stride - a.pl + kw;
          if ((unsigned)wi >= (unsigned)a.W) continue;
          float f[8];
          unpack8<DT>(ldraw<DT>(a.y, ((size_t)(n * a.H + hi) * a.W + wi) * C + chunk * 8), f);
          const uint32_t tap = (uint32_t)(kh * a.k + kw);
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const float v = f[c] * scale[c] + shift[c];
            if (v > best[c]) { best[c] = v; bi[c] = tap; }
          }
        }
      }
      const size_t o = ((size_t)row * a.OW + ow) * C + chunk * 8;
      straw<DT>(a.zp, o, best);
      u32x2 pk;
      pk.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
      pk.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
      *reinterpret_cast<u32x2*>(a.idx + o) = pk;
    }
  }
}

// APPLY == false: per-block partial sums of dgamma / dbeta;  true: dy.
// Windows overlap at most two deep per dimension (stride <= k <= 2 * stride, checked on the
// host): an input row h is tap kh0 = (h + pt) % stride of window (h + pt) / stride and tap
// kh0 + stride of the window before it.  The four candidate windows of an element are loaded
// unconditionally (addresses clamped) before any is used, and accumulated under predicates.
// S: compile-time stride (0 = read it from the arguments).
template <int DT, bool APPLY, int S>
__global__ __launch_bounds__(TPB) void bn_pool_bwd_kernel(const BnPoolArgs a) {
  const int C = a.C, k = a.k, stride = S ? S : a.stride;
  const uint32_t cpr = (uint32_t)C >> 3;
  const uint32_t chunk = threadIdx.x % cpr;
  float mean[8], invstd[8], sc[8], c1[8], c2[8];
  load8(a.mean + chunk * 8, mean);
  load8(a.invstd + chunk * 8, invstd);
  if constexpr (APPLY) {
    const float invP = 1.f / ((float)a.N * (float)a.H * (float)a.W);
    load8(a.gamma + chunk * 8, sc);
    load8(a.dbeta + chunk * 8, c1);
    load8(a.dgamma + chunk * 8, c2);
#pragma unroll
    for (int c = 0; c < 8; ++c) { sc[c] *= invstd[c]; c1[c] *= invP; c2[c] *= invP; }
  }
  float part[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) part[i] = 0.f;
  const uint32_t per_row = (uint32_t)a.W * cpr;
  for (uint32_t row = blockIdx.x; row < (uint32_t)(a.N * a.H); row += gridDim.x) {
    const uint32_t n = row / (uint32_t)a.H;
    const int h = (int)(row - n * (uint32_t)a.H);
    const int oh0 = (h + a.pt) / stride, kh0 = (h + a.pt) - oh0 * stride;
    // candidate window rows: {oh0, tap row kh0}, {oh0 - 1, tap row kh0 + stride}
    const bool vh[2] = {oh0 < a.OH, kh0 + stride < k && oh0 >= 1};
    const int ohc[2] = {oh0 < a.OH ? oh0 : a.OH - 1, oh0 >= 1 ? oh0 - 1 : 0};
    const int khc[2] = {kh0, kh0 + stride};
    for (uint32_t j = threadIdx.x; j < per_row; j += TPB) {
      const int w = (int)(j / cpr);
      const size_t off = ((size_t)row * a.W + w) * C + chunk * 8;
      const Raw8<DT> yq = ldraw<DT>(a.y, off);
      const int ow0 = (w + a.pl) / stride, kw0 = (w + a.pl) - ow0 * stride;
      const bool vw[2] = {ow0 < a.OW, kw0 + stride < k && ow0 >= 1};
      const int owc[2] = {ow0 < a.OW ? ow0 : a.OW - 1, ow0 >= 1 ? ow0 - 1 : 0};
      const int kwc[2] = {kw0, kw0 + stride};
      u32x2 pk[4];
      Raw8<DT> dq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t o = ((size_t)(n * a.OH + ohc[q >> 1]) * a.OW + owc[q & 1]) * C + chunk * 8;
        pk[q] = *reinterpret_cast<const u32x2*>(a.idx + o);
        dq[q] = ldraw<DT>(a.dzp, o);
      }
      float g[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) g[c] = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // 0x100 never equals a stored tap: an invalid candidate contributes nothing
        const unsigned tap = (vh[q >> 1] && vw[q & 1]) ? (unsigned)(khc[q >> 1] * k + kwc[q & 1])
                                                       : 0x100u;
        float d[8];
        unpack8<DT>(dq[q], d);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const unsigned t = ((c < 4 ? pk[q].x : pk[q].y) >> (8 * (c & 3))) & 0xffu;
          g[c] += t == tap ? d[c] : 0.f;
        }
      }
      float yv[8];
      unpack8<DT>(yq, yv);
      if constexpr (APPLY) {
        float o[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float xhat = (yv[c] - mean[c]) * invstd[c];
          o[c] = sc[c] * (g[c] - c1[c] - xhat * c2[c]);
        }
        straw<DT>(a.dy, off, o);
      } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          part[c] += g[c] * ((yv[c] - mean[c]) * invstd[c]);
          part[8 + c] += g[c];
        }
      }
    }
  }
  if constexpr (!APPLY)
    block_reduce_to_partials<16>(part, (int)cpr, (int)chunk, (int)(threadIdx.x / cpr),
                                 a.ws + (size_t)blockIdx.x * 2 * C, C);
}

// k == 3, stride == 2 (the ResNet stem): a thread owns the 2x2 patch of full-resolution elements
// whose (h + pt, w + pl) are (2a, 2a + 1) x (2b, 2b + 1).  The patch lies in the windows
// {a, a - 1} x {b, b - 1} and in no other, so four window loads serve four elements (the
// per-element kernel issues four for each: nine load instructions per 16 bytes of output, here
// three).  The even row (column) is tap 0 of window a (b) and tap 2 of window a - 1 (b - 1), the
// odd one is tap 1 of window a (b).
template <int DT, bool APPLY>
__global__ __launch_bounds__(TPB) void bn_pool_bwd_quad_kernel(const BnPoolArgs a) {
  const int C = a.C;
  const uint32_t cpr = (uint32_t)C >> 3;
  const uint32_t chunk = threadIdx.x % cpr;
  float mean[8], invstd[8], sc[8], c1[8], c2[8];
  load8(a.mean + chunk * 8, mean);
  load8(a.invstd + chunk * 8, invstd);
  if constexpr (APPLY) {
    const float invP = 1.f / ((float)a.N * (float)a.H * (float)a.W);
    load8(a.gamma + chunk * 8, sc);
    load8(a.dbeta + chunk * 8, c1);
    load8(a.dgamma + chunk * 8, c2);
#pragma unroll
    for (int c = 0; c < 8; ++c) { sc[c] *= invstd[c]; c1[c] *= invP; c2[c] *= invP; }
  }
  float part[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) part[i] = 0.f;
  const uint32_t QH = (uint32_t)(a.H + a.pt + 1) >> 1, QW = (uint32_t)(a.W + a.pl + 1) >> 1;
  const uint32_t per_row = QW * cpr;
  for (uint32_t qrow = blockIdx.x; qrow < (uint32_t)a.N * QH; qrow += gridDim.x) {
    const uint32_t n = qrow / QH;
    const int qa = (int)(qrow - n * QH);
    const bool vwr[2] = {qa < a.OH, qa >= 1 && qa - 1 < a.OH};
    const int ohc[2] = {qa < a.OH ? qa : a.OH - 1, vwr[1] ? qa - 1 : 0};
    const int h0 = 2 * qa - a.pt;
    const bool vh[2] = {(unsigned)h0 < (unsigned)a.H, (unsigned)(h0 + 1) < (unsigned)a.H};
    for (uint32_t j = threadIdx.x; j < per_row; j += TPB) {
      const int qb = (int)(j / cpr);
      const bool vwc[2] = {qb < a.OW, qb >= 1 && qb - 1 < a.OW};
      const int owc[2] = {qb < a.OW ? qb : a.OW - 1, vwc[1] ? qb - 1 : 0};
      const int w0 = 2 * qb - a.pl;
      const bool vw[2] = {(unsigned)w0 < (unsigned)a.W, (unsigned)(w0 + 1) < (unsigned)a.W};
      u32x2 pk[4];
      Raw8<DT> dq[4], yq[4];
      size_t off[4];
      bool ve[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const size_t o = ((size_t)(n * a.OH + ohc[q >> 1]) * a.OW + owc[q & 1]) * C + chunk * 8;
        pk[q] = *reinterpret_cast<const u32x2*>(a.idx + o);
        dq[q] = ldraw<DT>(a.dzp, o);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        ve[e] = vh[e >> 1] && vw[e & 1];
        const int h = vh[e >> 1] ? h0 + (e >> 1) : 0, w = vw[e & 1] ? w0 + (e & 1) : 0;
        off[e] = ((size_t)(n * a.H + h) * a.W + w) * C + chunk * 8;
        yq[e] = ldraw<DT>(a.y, off[e]);
      }
      float d[4][8];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        unpack8<DT>(dq[q], d[q]);
        if (!(vwr[q >> 1] && vwc[q & 1])) {
#pragma unroll
          for (int c = 0; c < 8; ++c) d[q][c] = 0.f;
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = e >> 1, cc = e & 1;
        float g[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) g[c] = 0.f;
#pragma unroll
        for (int wr = 0; wr < 2; ++wr) {
#pragma unroll
          for (int wc = 0; wc < 2; ++wc) {
            if ((wr == 1 && r == 1) || (wc == 1 && cc == 1)) continue;   // tap 3 does not exist
            const int q = wr * 2 + wc;
            const unsigned tap = (unsigned)((r + 2 * wr) * 3 + (cc + 2 * wc));
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              const unsigned t = ((c < 4 ? pk[q].x : pk[q].y) >> (8 * (c & 3))) & 0xffu;
              g[c] += t == tap ? d[q][c] : 0.f;
            }
          }
        }
        float yv[8];
        unpack8<DT>(yq[e], yv);
        if constexpr (APPLY) {
          if (ve[e]) {
            float o[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
              const float xhat = (yv[c] - mean[c]) * invstd[c];
              o[c] = sc[c] * (g[c] - c1[c] - xhat * c2[c]);
            }
            straw<DT>(a.dy, off[e], o);
          }
        } else {
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const float ge = ve[e] ? g[c] : 0.f;
            part[c] += ge * ((yv[c] - mean[c]) * invstd[c]);
            part[8 + c] += ge;
          }
        }
      }
    }
  }
  if constexpr (!APPLY)
    block_reduce_to_partials<16>(part, (int)cpr, (int)chunk, (int)(threadIdx.x / cpr),
                                 a.ws + (size_t)blockIdx.x * 2 * C, C);
}

// Inference: scale = gamma / sqrt(moving_var + eps), shift = beta - moving_mean * scale for every
// BatchNorm layer of a model in one launch (one block per layer), consumed by the folded
// convolution epilogue.  Descriptor: five int64 element offsets (gamma and beta in `params`,
// moving mean and variance in `state`, scale in `out`; shift follows at out + C), then C and eps.
struct BnFoldDesc {
  long long gamma, beta, mm, mv, out;
  int C;
  float eps;
};

__global__ __launch_bounds__(256) void bn_fold_kernel(const float* __restrict__ params,
                                                      const float* __restrict__ state,
                                                      float* __restrict__ out,
                                                      const BnFoldDesc* __restrict__ descs) {
  const BnFoldDesc d = descs[blockIdx.x];
  for (int c = threadIdx.x; c < d.C; c += blockDim.x) {
    const float sc = params[d.gamma + c] * rsqrtf(state[d.mv + c] + d.eps);
    out[d.out + c] = sc;
    out[d.out + d.C + c] = params[d.beta + c] - state[d.mm + c] * sc;
  }
}

void check_channels(int C, int P) {
  if (C < 8 || C > 2048 || (C & (C - 1)))
    throw std::invalid_argument("batch-norm kernels need a power-of-two channel count in [8, 2048]");
  if (P <= 0) throw std::invalid_argument("batch-norm kernels: empty tensor");
}

int rows_grid(int P, int C) {
  static const int cap = [] {
    const char* e = std::getenv("DTF_BN_GRID_CAP");
    return e ? std::atoi(e) : 512;
  }();
  const int rpb = TPB / (C / 8);
  const int blocks = (P + rpb - 1) / rpb;
  return blocks < cap ? blocks : cap;
}

// Reduction grid: enough blocks to pull full HBM bandwidth, at least 8 row passes per block,
// bounded by the rows of the workspace.
int reduce_grid(int P, int C) {
  const int rpb = TPB / (C / 8);
  int blocks = (P + rpb * 8 - 1) / (rpb * 8);
  if (blocks < 1) blocks = 1;
  return blocks < MAX_PARTIAL_BLOCKS ? blocks : MAX_PARTIAL_BLOCKS;
}

}  // namespace

int bn_fold_desc_size() { return (int)sizeof(BnFoldDesc); }

void launch_bn_fold(const float* params, const float* state, float* out, const void* descs,
                    int nlayers, hipStream_t s) {
  if (nlayers <= 0) return;
  hipLaunchKernelGGL(bn_fold_kernel, dim3(nlayers), dim3(256), 0, s, params, state, out,
                     reinterpret_cast<const BnFoldDesc*>(descs));
}

// Cache policy of a sweep over tensors of `bytes` each (DTF_NT_SWITCH): a tensor that does not
// fit the 256 MB memory-side cache next to its companions streams through with non-temporal
// accesses; smaller ones stay cacheable so that the next launch finds them.  `which` selects the
// bits of DTF_BN_NT (default 0x3f): 0 forward apply, 1 backward reduction, 2 backward apply; two
// bits each, loads / stores.
static int nt_mode(int which, size_t bytes) {
  static const size_t threshold = [] {
    const char* e = std::getenv("DTF_BN_NT_BYTES");
    return e ? (size_t)std::atoll(e) : (size_t)90 << 20;
  }();
  static const int mask = [] {
    const char* e = std::getenv("DTF_BN_NT");
    return e ? (int)std::strtol(e, nullptr, 0) : 0x3f;
  }();
  return bytes >= threshold ? (mask >> (2 * which)) & 3 : 0;
}

static size_t elem_bytes(int dt) { return dt == DT_F32 ? 4 : 2; }

size_t bn_workspace_floats() { return (size_t)MAX_PARTIAL_BLOCKS * 2 * 2048; }

void launch_bn_fwd_train(const void* y, const void* res, void* z, uint8_t* mask,
                         const float* gamma, const float* beta, float* mm, float* mv,
                         float* save_mean, float* save_invstd, float* scale, float* shift,
                         float* ws, int fused_replicas, int P, int C, float eps, float momentum,
                         int relu, const float* rscale, const float* rshift, int apply, int rev,
                         int dt, hipStream_t s) {
  check_channels(C, P);
  if ((rscale == nullptr) != (rshift == nullptr) || (rscale && !res))
    throw std::invalid_argument("bn_fwd_train: rscale / rshift come together, with res");
  int G = fused_replicas;             // > 0: `ws` already holds that many rows of [2][C] sums
  if (G <= 0) {
    G = reduce_grid(P, C);
    DTF_DT_SWITCH(dt, hipLaunchKernelGGL(bn_stats_kernel<DT>, dim3(G), dim3(TPB), 0, s, y, ws, P, C));
  }
  static const bool fold_ok = [] {
    const char* e = std::getenv("DTF_BN_FOLD_FINALIZE");
    return !(e && e[0] == '0');
  }();
  BnApplyArgs a{y, res, z, mask, scale, shift, gamma, beta, mm, mv, P, C, eps, relu,
                ws, G, momentum, mm, mv, save_mean, save_invstd, scale, shift, rscale, rshift,
                rev};
  if (apply && fold_ok && fused_replicas > 0 && fused_replicas * 2 * C <= 16384 &&
      C <= FOLD_MAX_C) {
    const int ntm = nt_mode(0, (size_t)P * C * elem_bytes(dt));
    DTF_NT_SWITCH(ntm, DTF_DT_SWITCH(dt, hipLaunchKernelGGL((bn_apply_kernel<DT, true, NTM>),
                                                            dim3(rows_grid(P, C)), dim3(TPB), 0, s, a)));
    return;
  }
  BnFinalizeArgs f{ws, G, gamma, beta, mm, mv, save_mean, save_invstd, scale, shift,
                   P, C, eps, momentum};
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((C + 31) / 32), dim3(FIN_TPB), 0, s, f);
  if (apply) {    // apply == 0: statistics only, a later layer normalises this tensor on the fly
    const int ntm = nt_mode(0, (size_t)P * C * elem_bytes(dt));
    DTF_NT_SWITCH(ntm, DTF_DT_SWITCH(dt, hipLaunchKernelGGL((bn_apply_kernel<DT, false, NTM>),
                                                            dim3(rows_grid(P, C)), dim3(TPB), 0, s, a)));
  }
}

namespace {
void check_bn_pool(int N, int H, int W, int C, int OH, int OW, int k, int stride, int pt, int pl) {
  check_channels(C, N * H * W);
  if (k < 1 || k * k > 254 || stride < 1 || stride > k || k > 2 * stride || pt < 0 || pl < 0 ||
      pt >= k || pl >= k)
    throw std::invalid_argument("bn_pool: unsupported window (needs stride <= k <= 2 * stride)");
  if (OH < 1 || OW < 1 || (OH - 1) * stride - pt >= H || (OW - 1) * stride - pl >= W)
    throw std::invalid_argument("bn_pool: pooled size does not match the input");
  if ((int64_t)N * H * W * (C / 8) >= (1ll << 31)) throw std::invalid_argument("bn_pool: tensor too large");
}
}  // namespace

void launch_bn_pool_fwd_train(const void* y, void* zp, uint8_t* idx, const float* gamma,
                              const float* beta, float* mm, float* mv, float* save_mean,
                              float* save_invstd, float* scale, float* shift, float* ws,
                              int fused_replicas, int N, int H, int W, int C, int OH, int OW, int k,
                              int stride, int pt, int pl, float eps, float momentum, int dt,
                              hipStream_t s) {
  check_bn_pool(N, H, W, C, OH, OW, k, stride, pt, pl);
  const int P = N * H * W;
  int G = fused_replicas;
  if (G <= 0) {
    G = reduce_grid(P, C);
    DTF_DT_SWITCH(dt, hipLaunchKernelGGL(bn_stats_kernel<DT>, dim3(G), dim3(TPB), 0, s, y, ws, P, C));
  }
  BnFinalizeArgs f{ws, G, gamma, beta, mm, mv, save_mean, save_invstd, scale, shift,
                   P, C, eps, momentum};
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((C + 31) / 32), dim3(FIN_TPB), 0, s, f);
  BnPoolArgs a{};
  a.y = y; a.zp = zp; a.idx = idx; a.scale = scale; a.shift = shift;
  a.N = N; a.H = H; a.W = W; a.C = C; a.OH = OH; a.OW = OW; a.k = k; a.stride = stride;
  a.pt = pt; a.pl = pl;
  const int rows = N * OH;
  DTF_DT_SWITCH(dt, hipLaunchKernelGGL(bn_pool_fwd_kernel<DT>, dim3(rows < 8192 ? rows : 8192),
                                       dim3(TPB), 0, s, a));
}

void launch_bn_pool_bwd(const void* dzp, const uint8_t* idx, const void* y, const float* mean,
                        const float* invstd, const float* gamma, float* dgamma, float* dbeta,
                        void* dy, float* ws, int N, int H, int W, int C, int OH, int OW, int k,
                        int stride, int pt, int pl, int dt, hipStream_t s) {
  check_bn_pool(N, H, W, C, OH, OW, k, stride, pt, pl);
  BnPoolArgs a{};
  a.y = y; a.idx = const_cast<uint8_t*>(idx); a.dzp = dzp; a.dy = dy; a.mean = mean; a.invstd = invstd; a.gamma = gamma;
  a.dgamma = dgamma; a.dbeta = dbeta; a.ws = ws;
  a.N = N; a.H = H; a.W = W; a.C = C; a.OH = OH; a.OW = OW; a.k = k; a.stride = stride;
  a.pt = pt; a.pl = pl;
  // partial rows are 2 * C floats: a narrow layer can use many more blocks than
  // MAX_PARTIAL_BLOCKS within the same workspace (the gathers are latency bound)
  const int cap = (int)(bn_workspace_floats() / (2 * (size_t)C));
  static const bool quad_ok = [] {
    const char* e = std::getenv("DTF_BN_POOL_QUAD");
    return !(e && e[0] == '0');
  }();
  const bool quad = quad_ok && k == 3 && stride == 2;
  const int rows = quad ? N * ((H + pt + 1) / 2) : N * H;
  int G = rows < cap ? rows : cap;
  if (G > 4096) G = 4096;
  if (quad) {
    DTF_DT_SWITCH(dt, hipLaunchKernelGGL((bn_pool_bwd_quad_kernel<DT, false>), dim3(G), dim3(TPB), 0, s, a));
  } else if (stride == 2) {
    DTF_DT_SWITCH(dt, hipLaunchKernelGGL((bn_pool_bwd_kernel<DT, false, 2>), dim3(G), dim3(TPB), 0, s, a));
  } else {
    DTF_DT_SWITCH(dt, hipLaunchKernelGGL((bn_pool_bwd_kernel<DT, false, 0>), dim3(G), dim3(TPB), 0, s, a));
  }
  BnBwdArgs b{nullptr, y, nullptr, nullptr, mean, invstd, gamma, dgamma, dbeta, dy, nullptr, ws, G,
              N * H * W, C, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
              nullptr, 0, 0};
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 31) / 32), dim3(FIN_TPB), 0, s, b);
  const int ga = rows < 8192 ? rows : 8192;
  if (quad) {
    DTF_DT_SWITCH(dt, hipLaunchKernelGGL((bn_pool_bwd_quad_kernel<DT, true>), dim3(ga), dim3(TPB), 0, s, a));
  } else if (stride == 2) {
    DTF_DT_SWITCH(dt, hipLaunchKernelGGL((bn_pool_bwd_kernel<DT, true, 2>), dim3(ga), dim3(TPB), 0, s, a));
  } else {
    DTF_DT_SWITCH(dt, hipLaunchKernelGGL((bn_pool_bwd_kernel<DT, true, 0>), dim3(ga), dim3(TPB), 0, s, a));
  }
}

void launch_mask_apply(const void* x, const uint8_t* mask, void* out, size_t n, int dt,
                       hipStream_t s) {
  if (n % 8) throw std::invalid_argument("mask_apply: element count must be a multiple of 8");
  const size_t nvec = n / 8;
  if (nvec == 0) return;
  const size_t blocks = (nvec + TPB - 1) / TPB;
  DTF_DT_SWITCH(dt, hipLaunchKernelGGL(mask_apply_kernel<DT>,
                                       dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(TPB),
                                       0, s, x, mask, out, nvec));
}

void launch_bn_infer(const void* y, const void* res, void* z, const float* gamma,
                     const float* beta, const float* mm, const float* mv, int P, int C, float eps,
                     int relu, int dt, hipStream_t s) {
  check_channels(C, P);
  BnApplyArgs a{y, res, z, nullptr, nullptr, nullptr, gamma, beta, mm, mv, P, C, eps, relu,
                nullptr, 0, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                nullptr, nullptr, 0};
  DTF_DT_SWITCH(dt, hipLaunchKernelGGL((bn_apply_kernel<DT, false>), dim3(rows_grid(P, C)),
                                       dim3(TPB), 0, s, a));
}

size_t bn_tail_floats() { return 16 + 2 * (size_t)TAIL_REPLICAS * 2 * 2048; }

void launch_bn_bwd(const void* dz, const void* y, const void* z, const uint8_t* mask,
                   const float* mean, const float* invstd, const float* gamma, float* dgamma,
                   float* dbeta, void* dy, void* dres, float* ws, int P, int C,
                   int pre_replicas, float* tail, int rev, const float* relu_scale,
                   const float* relu_shift, int dt, hipStream_t s) {
  check_channels(C, P);
  if (pre_replicas < 0) throw std::invalid_argument("bn_bwd: negative replica count");
  if ((relu_scale == nullptr) != (relu_shift == nullptr) || (relu_scale && (mask || z)))
    throw std::invalid_argument("bn_bwd: relu_scale / relu_shift come together, instead of mask / z");
  const int G = pre_replicas > 0 ? pre_replicas : reduce_grid(P, C);
  BnBwdArgs a{dz, y, z, mask, mean, invstd, gamma, dgamma, dbeta, dy, dres, ws, G, P, C,
              pre_replicas > 0 ? 1 : 0,
              nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
              rev & 1, (rev >> 1) & 1, relu_scale, relu_shift};
  static const bool fold_ok = [] {
    const char* e = std::getenv("DTF_BN_FOLD_FINALIZE");
    return !(e && e[0] == '0');
  }();
  if (fold_ok && pre_replicas > 0 && pre_replicas * 2 * C <= 16384 && C <= FOLD_MAX_C) {
    DTF_DT_SWITCH(dt, hipLaunchKernelGGL((bn_bwd_apply_kernel<DT, true>), dim3(rows_grid(P, C)),
                                         dim3(TPB), 0, s, a));
    return;
  }
  if (pre_replicas == 0 && tail != nullptr) {
    // the reduction's last block finalises: no separate launch between the two sweeps
    BnBwdArgs t = a;
    t.ticket = reinterpret_cast<unsigned int*>(tail);
    t.ws = tail + 16;
    DTF_DT_SWITCH(dt, hipLaunchKernelGGL((bn_bwd_reduce_kernel<DT, 1, true>), dim3(G), dim3(TPB), 0,
                                         s, t));
    DTF_DT_SWITCH(dt, hipLaunchKernelGGL((bn_bwd_apply_kernel<DT, false>), dim3(rows_grid(P, C)),
                                         dim3(TPB), 0, s, a));
    return;
  }
  const size_t bytes = (size_t)P * C * elem_bytes(dt);
  if (pre_replicas == 0) {
    const int ntr = nt_mode(1, bytes);
    DTF_NT_SWITCH(ntr, DTF_DT_SWITCH(dt, hipLaunchKernelGGL((bn_bwd_reduce_kernel<DT, 1, false, NTM>),
                                                            dim3(G), dim3(TPB), 0, s, a)));
  }
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 31) / 32), dim3(FIN_TPB), 0, s, a);
  const int nta = nt_mode(2, bytes);
  DTF_NT_SWITCH(nta, DTF_DT_SWITCH(dt, hipLaunchKernelGGL((bn_bwd_apply_kernel<DT, false, 1, NTM>),
                                                          dim3(rows_grid(P, C)), dim3(TPB), 0, s, a)));
}

// Backward of two BatchNorm layers that receive the same gradient g = dz gated by `mask`
// (a block's last BatchNorm and the one of its projection shortcut): dz and the mask are
// read once per sweep for both.  `ws` / `ws2`: a partial-sum workspace per layer.
void launch_bn_bwd_dual(const void* dz, const uint8_t* mask, int P, int C, float* ws, float* ws2,
                        const void* y1, const float* mean1, const float* invstd1,
                        const float* gamma1, float* dgamma1, float* dbeta1, void* dy1,
                        const void* y2, const float* mean2, const float* invstd2,
                        const float* gamma2, float* dgamma2, float* dbeta2, void* dy2,
                        float* tail, int rev, int dt, hipStream_t s) {
  check_channels(C, P);
  if (ws == ws2) throw std::invalid_argument("bn_bwd_dual: the two layers need separate workspaces");
  const int G = reduce_grid(P, C);
  BnBwdArgs a{dz, y1, nullptr, mask, mean1, invstd1, gamma1, dgamma1, dbeta1, dy1, nullptr, ws, G,
              P, C, 0, y2, mean2, invstd2, gamma2, dgamma2, dbeta2, dy2, ws2, nullptr,
              rev & 1, (rev >> 1) & 1};
  if (tail != nullptr) {
    BnBwdArgs t = a;
    t.ticket = reinterpret_cast<unsigned int*>(tail);
    t.ws = tail + 16;
    t.ws2 = tail + 16 + (size_t)TAIL_REPLICAS * 2 * 2048;
    DTF_DT_SWITCH(dt, hipLaunchKernelGGL((bn_bwd_reduce_kernel<DT, 2, true>), dim3(G), dim3(TPB), 0,
                                         s, t));
    DTF_DT_SWITCH(dt, hipLaunchKernelGGL((bn_bwd_apply_kernel<DT, false, 2>), dim3(rows_grid(P, C)),
                                         dim3(TPB), 0, s, a));
    return;
  }
  const size_t bytes = (size_t)P * C * elem_bytes(dt);
  const int ntr = nt_mode(1, bytes), nta = nt_mode(2, bytes);
  DTF_NT_SWITCH(ntr, DTF_DT_SWITCH(dt, hipLaunchKernelGGL((bn_bwd_reduce_kernel<DT, 2, false, NTM>),
                                                          dim3(G), dim3(TPB), 0, s, a)));
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 31) / 32), dim3(FIN_TPB), 0, s, a);
  BnBwdArgs b = a;                   // the finalize kernel reads the first layer's fields
  b.ws = ws2; b.dgamma = dgamma2; b.dbeta = dbeta2; b.mean = mean2; b.invstd = invstd2;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 31) / 32), dim3(FIN_TPB), 0, s, b);
  DTF_NT_SWITCH(nta, DTF_DT_SWITCH(dt, hipLaunchKernelGGL((bn_bwd_apply_kernel<DT, false, 2, NTM>),
                                                          dim3(rows_grid(P, C)), dim3(TPB), 0, s, a)));
}
