// Fused BatchNorm kernels for NHWC bf16 activations (training forward / backward, inference).
//
// Training forward is   z = relu?( gamma * (y - mean) * invstd + beta (+ residual) )
// with per-channel batch statistics over all P = N*H*W pixels, computed in fp32 from the
// bf16 convolution output.  The statistics pass (`bn_stats`) produces sum and sum of squares;
// `bn_apply` derives scale/shift in its prologue (so there is no separate "finalize" launch),
// applies normalisation + residual add + ReLU in one sweep, stores mean / invstd for the
// backward pass and lets block 0 update the moving averages with the Keras/TensorFlow rule
//   moving = moving * m + batch * (1 - m),   batch variance *unbiased* for the moving average.
//
// Backward is two sweeps: `bn_bwd_reduce` (ReLU mask from the saved output, dgamma / dbeta
// reductions straight into the flat fp32 gradient buffer) and `bn_bwd_apply`
//   dy = gamma * invstd * (g - dbeta/P - xhat * dgamma/P),   g = dz masked by ReLU
// which also emits g itself as the gradient of the residual branch when asked.
//
// All kernels are bandwidth bound: every thread moves 16 bytes (8 channels) per access, a row
// of C channels is covered by C/8 consecutive lanes, and a block walks rows grid-strided.
#include <stdexcept>
#include <string>

#include "common.h"
#include "elementwise.h"

namespace {

constexpr int TPB = 256;

// Sum the per-thread partials of threads that own the same channel chunk (same tid % tpr) and
// add the block result to global memory. part[] has NV values per thread.
template <int NV>
__device__ __forceinline__ void block_reduce_to_global(float (&part)[NV], int tpr, int chunk,
                                                       int rowid, float* const* dst, int C) {
  __shared__ float red[TPB * NV];
  const int tid = threadIdx.x;
  const int rpb = TPB / tpr;
  if (rpb > 1) {
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
  if (rowid == 0) {
    // NV = 8 channels x (NV/8) quantities
#pragma unroll
    for (int qn = 0; qn < NV / 8; ++qn)
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int ch = chunk * 8 + c;
        if (ch < C) atomicAdd(dst[qn] + ch, part[qn * 8 + c]);
      }
  }
}

__global__ __launch_bounds__(TPB) void bn_stats_kernel(const bf16_t* __restrict__ y, float* sum,
                                                       float* sqsum, int P, int C) {
  const int tpr = C >> 3, rpb = TPB / tpr;
  const int chunk = threadIdx.x % tpr, rowid = threadIdx.x / tpr;
  float part[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) part[i] = 0.f;
  for (int row = blockIdx.x * rpb + rowid; row < P; row += gridDim.x * rpb) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(y + (size_t)row * C + chunk * 8);
    float f[8];
    unpack8(v, f);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      part[c] += f[c];
      part[8 + c] += f[c] * f[c];
    }
  }
  float* dst[2] = {sum, sqsum};
  block_reduce_to_global<16>(part, tpr, chunk, rowid, dst, C);
}

struct BnApplyArgs {
  const bf16_t* y; const bf16_t* res; bf16_t* z;
  const float* sum; const float* sqsum; const float* gamma; const float* beta;
  float* mm; float* mv; float* save_mean; float* save_invstd;
  int P, C; float eps, momentum; int relu, training;
};

__global__ __launch_bounds__(TPB) void bn_apply_kernel(const BnApplyArgs a) {
  const int C = a.C, tpr = C >> 3, rpb = TPB / tpr;
  const int chunk = threadIdx.x % tpr, rowid = threadIdx.x / tpr;
  float scale[8], shift[8];
  const float invP = 1.f / (float)a.P;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int ch = chunk * 8 + c;
    float mean, var;
    if (a.training) {
      mean = a.sum[ch] * invP;
      var = fmaxf(a.sqsum[ch] * invP - mean * mean, 0.f);
    } else {
      mean = a.mm[ch];
      var = a.mv[ch];
    }
    const float invstd = rsqrtf(var + a.eps);
    scale[c] = a.gamma[ch] * invstd;
    shift[c] = a.beta[ch] - mean * scale[c];
    if (a.training && blockIdx.x == 0 && rowid == 0) {
      a.save_mean[ch] = mean;
      a.save_invstd[ch] = invstd;
      const float unbiased = a.P > 1 ? var * ((float)a.P / (float)(a.P - 1)) : var;
      a.mm[ch] = a.mm[ch] * a.momentum + mean * (1.f - a.momentum);
      a.mv[ch] = a.mv[ch] * a.momentum + unbiased * (1.f - a.momentum);
    }
  }
  for (int row = blockIdx.x * rpb + rowid; row < a.P; row += gridDim.x * rpb) {
    const size_t off = (size_t)row * C + chunk * 8;
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(a.y + off), f);
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] = f[c] * scale[c] + shift[c];
    if (a.res) {
      float r[8];
      unpack8(*reinterpret_cast<const u32x4*>(a.res + off), r);
#pragma unroll
      for (int c = 0; c < 8; ++c) f[c] += r[c];
    }
    if (a.relu) {
#pragma unroll
      for (int c = 0; c < 8; ++c) f[c] = fmaxf(f[c], 0.f);
    }
    *reinterpret_cast<u32x4*>(a.z + off) = pack8(f);
  }
}

struct BnBwdArgs {
  const bf16_t* dz; const bf16_t* y; const bf16_t* z;   // z == nullptr: no ReLU mask
  const float* mean; const float* invstd; const float* gamma;
  float* dgamma; float* dbeta;
  bf16_t* dy; bf16_t* dres;
  int P, C;
};

__global__ __launch_bounds__(TPB) void bn_bwd_reduce_kernel(const BnBwdArgs a) {
  const int C = a.C, tpr = C >> 3, rpb = TPB / tpr;
  const int chunk = threadIdx.x % tpr, rowid = threadIdx.x / tpr;
  float mean[8], invstd[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    mean[c] = a.mean[chunk * 8 + c];
    invstd[c] = a.invstd[chunk * 8 + c];
  }
  float part[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) part[i] = 0.f;
  for (int row = blockIdx.x * rpb + rowid; row < a.P; row += gridDim.x * rpb) {
    const size_t off = (size_t)row * C + chunk * 8;
    float g[8], yv[8];
    unpack8(*reinterpret_cast<const u32x4*>(a.dz + off), g);
    unpack8(*reinterpret_cast<const u32x4*>(a.y + off), yv);
    if (a.z) {
      float zv[8];
      unpack8(*reinterpret_cast<const u32x4*>(a.z + off), zv);
#pragma unroll
      for (int c = 0; c < 8; ++c) g[c] = zv[c] > 0.f ? g[c] : 0.f;
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      part[c] += g[c] * ((yv[c] - mean[c]) * invstd[c]);   // dgamma
      part[8 + c] += g[c];                                  // dbeta
    }
  }
  float* dst[2] = {a.dgamma, a.dbeta};
  block_reduce_to_global<16>(part, tpr, chunk, rowid, dst, C);
}

__global__ __launch_bounds__(TPB) void bn_bwd_apply_kernel(const BnBwdArgs a) {
  const int C = a.C, tpr = C >> 3, rpb = TPB / tpr;
  const int chunk = threadIdx.x % tpr, rowid = threadIdx.x / tpr;
  const float invP = 1.f / (float)a.P;
  float mean[8], invstd[8], scale[8], c1[8], c2[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int ch = chunk * 8 + c;
    mean[c] = a.mean[ch];
    invstd[c] = a.invstd[ch];
    scale[c] = a.gamma[ch] * invstd[c];
    c1[c] = a.dbeta[ch] * invP;
    c2[c] = a.dgamma[ch] * invP;
  }
  for (int row = blockIdx.x * rpb + rowid; row < a.P; row += gridDim.x * rpb) {
    const size_t off = (size_t)row * C + chunk * 8;
    float g[8], yv[8];
    unpack8(*reinterpret_cast<const u32x4*>(a.dz + off), g);
    unpack8(*reinterpret_cast<const u32x4*>(a.y + off), yv);
    if (a.z) {
      float zv[8];
      unpack8(*reinterpret_cast<const u32x4*>(a.z + off), zv);
#pragma unroll
      for (int c = 0; c < 8; ++c) g[c] = zv[c] > 0.f ? g[c] : 0.f;
    }
    if (a.dres) *reinterpret_cast<u32x4*>(a.dres + off) = pack8(g);
    float o[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float xhat = (yv[c] - mean[c]) * invstd[c];
      o[c] = scale[c] * (g[c] - c1[c] - xhat * c2[c]);
    }
    *reinterpret_cast<u32x4*>(a.dy + off) = pack8(o);
  }
}

void check_channels(int C, int P) {
  if (C < 8 || C > 2048 || (C & (C - 1)))
    throw std::invalid_argument("batch-norm kernels need a power-of-two channel count in [8, 2048]");
  if (P <= 0) throw std::invalid_argument("batch-norm kernels: empty tensor");
}

int rows_grid(int P, int C) {
  const int rpb = TPB / (C / 8);
  const int blocks = (P + rpb - 1) / rpb;
  return blocks < 2048 ? blocks : 2048;
}

}  // namespace

void launch_bn_stats(const bf16_t* y, float* sum, float* sqsum, int P, int C, hipStream_t s) {
  check_channels(C, P);
  int grid = rows_grid(P, C);
  if (grid > 1024) grid = 1024;     // bounds the atomics per channel
  hipLaunchKernelGGL(bn_stats_kernel, dim3(grid), dim3(TPB), 0, s, y, sum, sqsum, P, C);
}

void launch_bn_apply(const bf16_t* y, const bf16_t* res, bf16_t* z, const float* sum,
                     const float* sqsum, const float* gamma, const float* beta, float* mm,
                     float* mv, float* save_mean, float* save_invstd, int P, int C, float eps,
                     float momentum, int relu, int training, hipStream_t s) {
  check_channels(C, P);
  BnApplyArgs a{y, res, z, sum, sqsum, gamma, beta, mm, mv, save_mean, save_invstd,
                P, C, eps, momentum, relu, training};
  hipLaunchKernelGGL(bn_apply_kernel, dim3(rows_grid(P, C)), dim3(TPB), 0, s, a);
}

void launch_bn_bwd(const bf16_t* dz, const bf16_t* y, const bf16_t* z, const float* mean,
                   const float* invstd, const float* gamma, float* dgamma, float* dbeta,
                   bf16_t* dy, bf16_t* dres, int P, int C, hipStream_t s) {
  check_channels(C, P);
  BnBwdArgs a{dz, y, z, mean, invstd, gamma, dgamma, dbeta, dy, dres, P, C};
  int grid = rows_grid(P, C);
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(grid > 1024 ? 1024 : grid), dim3(TPB), 0, s, a);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid), dim3(TPB), 0, s, a);
}
