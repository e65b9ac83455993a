// Dense layers that do not fit the tiled GEMM path: any feature counts, fp32 master weights read
// directly ([in, out], the Keras layout; no packed copy).
//
// The convolution engine serves a dense layer as a 1x1 convolution when its input width is a
// multiple of 8 and the output is wide enough to fill MFMA tiles (the ResNet heads).  The
// reference's "trivial" model (`trivial_model.py:26-41`: Flatten -> Dense(1) -> Dense(classes),
// used to measure the input pipeline and the framework overhead) is the opposite shape: a
// 150 528 -> 1 reduction and a 1 -> 1001 outer product.  Both are bandwidth problems with no
// reuse to tile for; these kernels stream them:
//
//   forward, long K    one block per (row n, 4 outputs): threads stride over K with coalesced
//                      reads of x, block reduction
//   forward, short K   one thread per output
//   weight gradient    one thread per (k, co): dW[k, co] = sum_n x[n, k] * dy[n, co]
//                      (consecutive threads: consecutive co, then consecutive k - both operand
//                      reads coalesce or broadcast); plain stores, no atomics
//   data gradient      one thread per (n, k): dx[n, k] = sum_co dy[n, co] * W[k, co]
#include <stdexcept>

#include "common.h"
#include "elementwise.h"

namespace {

constexpr int CO_T = 4;      // outputs per block of the long-K forward kernel

template <int DT>
__global__ __launch_bounds__(256) void dense_fwd_rows_kernel(
    const void* __restrict__ xv, const float* __restrict__ w, const float* __restrict__ bias,
    void* __restrict__ outv, int N, int K, int Cout, int ldx, int ldo) {
  typedef typename Elem<DT>::T T;
  const T* x = reinterpret_cast<const T*>(xv);
  T* out = reinterpret_cast<T*>(outv);
  const int tiles = (Cout + CO_T - 1) / CO_T;
  const int n = blockIdx.x / tiles, co0 = (blockIdx.x % tiles) * CO_T;
  float s[CO_T];
#pragma unroll
  for (int j = 0; j < CO_T; ++j) s[j] = 0.f;
  const T* xr = x + (size_t)n * ldx;
  for (int k = threadIdx.x; k < K; k += 256) {
    const float xv1 = to_f<DT>(xr[k]);
#pragma unroll
    for (int j = 0; j < CO_T; ++j)
      if (co0 + j < Cout) s[j] += xv1 * w[(size_t)k * Cout + co0 + j];
  }
  __shared__ float red[4][CO_T];
#pragma unroll
  for (int j = 0; j < CO_T; ++j) s[j] = wave_sum(s[j]);
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int j = 0; j < CO_T; ++j) red[threadIdx.x >> 6][j] = s[j];
  }
  __syncthreads();
  if (threadIdx.x < CO_T && co0 + (int)threadIdx.x < Cout) {
    const int j = threadIdx.x;
    const float v = red[0][j] + red[1][j] + red[2][j] + red[3][j] + (bias ? bias[co0 + j] : 0.f);
    out[(size_t)n * ldo + co0 + j] = from_f<DT>(v);
  }
}

template <int DT>
__global__ __launch_bounds__(256) void dense_fwd_small_kernel(
    const void* __restrict__ xv, const float* __restrict__ w, const float* __restrict__ bias,
    void* __restrict__ outv, int N, int K, int Cout, int ldx, int ldo) {
  typedef typename Elem<DT>::T T;
  const T* x = reinterpret_cast<const T*>(xv);
  T* out = reinterpret_cast<T*>(outv);
  const size_t total = (size_t)N * ldo;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int n = (int)(i / ldo), co = (int)(i % ldo);
    float s = 0.f;
    if (co < Cout) {
      s = bias ? bias[co] : 0.f;
      for (int k = 0; k < K; ++k) s += to_f<DT>(x[(size_t)n * ldx + k]) * w[(size_t)k * Cout + co];
    }
    out[i] = from_f<DT>(s);          // pitch padding columns are written as zeros
  }
}

template <int DT>
__global__ __launch_bounds__(256) void dense_wgrad_kernel(
    const void* __restrict__ xv, const void* __restrict__ dyv, float* __restrict__ dw, int N,
    int K, int Cout, int ldx, int ldy) {
  typedef typename Elem<DT>::T T;
  const T* x = reinterpret_cast<const T*>(xv);
  const T* dy = reinterpret_cast<const T*>(dyv);
  const size_t total = (size_t)K * Cout;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t k = i / Cout;
    const int co = (int)(i % Cout);
    float s = 0.f;
    for (int n = 0; n < N; ++n)
      s += to_f<DT>(x[(size_t)n * ldx + k]) * to_f<DT>(dy[(size_t)n * ldy + co]);
    dw[i] = s;
  }
}

template <int DT>
__global__ __launch_bounds__(256) void dense_dgrad_kernel(
    const void* __restrict__ dyv, const float* __restrict__ w, void* __restrict__ dxv, int N,
    int K, int Cout, int ldy, int ldx) {
  typedef typename Elem<DT>::T T;
  const T* dy = reinterpret_cast<const T*>(dyv);
  T* dx = reinterpret_cast<T*>(dxv);
  const size_t total = (size_t)N * ldx;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int n = (int)(i / ldx);
    const size_t k = i % ldx;
    float s = 0.f;
    if (k < (size_t)K)
      for (int co = 0; co < Cout; ++co) s += to_f<DT>(dy[(size_t)n * ldy + co]) * w[k * Cout + co];
    dx[i] = from_f<DT>(s);
  }
}

// ---- gradient finiteness (dynamic loss scaling) --------------------------------------------------
// flag[0] = 1 when any element of g is inf / nan (the flag is zeroed by the state kernel below
// after it has been consumed).  One pass over the flat gradient buffer, no host synchronisation.
__global__ __launch_bounds__(256) void nonfinite_flag_kernel(const float* __restrict__ g, size_t n,
                                                             unsigned int* __restrict__ flag) {
  bool bad = false;
  for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float v = g[i];
    bad |= !(fabsf(v) <= 3.402823466e38f);       // false for inf and nan
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

// Dynamic loss-scale state machine on the device (`resnet_imagenet_main.py:182-187`,
// LossScaleOptimizer with "dynamic"): state = [scale, clean steps, skipped steps, skip this step,
// 1 / scale of this step's gradients].  After the finiteness pass: overflow -> divide the scale
// (not below 1) and skip the update; otherwise count the step and multiply the scale after
// `growth` clean steps.  The fused optimizer reads state[3..4]; the next step's loss gradient
// reads the new scale from `loss_scale_dev`.
__global__ void loss_scale_update_kernel(float* __restrict__ state, unsigned int* __restrict__ flag,
                                         float* __restrict__ loss_scale_dev, float growth,
                                         float multiplier) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float scale_used = state[0];
  if (flag[0]) {
    state[0] = fmaxf(scale_used / multiplier, 1.f);
    state[1] = 0.f;
    state[2] += 1.f;
    state[3] = 1.f;
  } else {
    state[3] = 0.f;
    state[1] += 1.f;
    if (state[1] >= growth) {
      state[0] = scale_used * multiplier;
      state[1] = 0.f;
    }
  }
  flag[0] = 0u;
  state[4] = 1.f / scale_used;
  if (loss_scale_dev) loss_scale_dev[0] = state[0];
}

inline int grid_for(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b < 8192 ? (b ? b : 1) : 8192);
}

}  // namespace

void launch_dense_fwd(const void* x, const float* w, const float* bias, void* out, int N, int K,
                      int Cout, int ldx, int ldo, int dt, hipStream_t s) {
  if (N <= 0 || K <= 0 || Cout <= 0 || ldx < K || ldo < Cout)
    throw std::invalid_argument("dense_fwd: bad shape");
  if (K >= 1024 && ldo == Cout) {
    const int tiles = (Cout + CO_T - 1) / CO_T;
    DTF_DT_SWITCH(dt, hipLaunchKernelGGL(dense_fwd_rows_kernel<DT>, dim3(N * tiles), dim3(256), 0, s,
                                         x, w, bias, out, N, K, Cout, ldx, ldo));
  } else {
    DTF_DT_SWITCH(dt, hipLaunchKernelGGL(dense_fwd_small_kernel<DT>, dim3(grid_for((size_t)N * ldo)),
                                         dim3(256), 0, s, x, w, bias, out, N, K, Cout, ldx, ldo));
  }
}

void launch_dense_wgrad(const void* x, const void* dy, float* dw, int N, int K, int Cout, int ldx,
                        int ldy, int dt, hipStream_t s) {
  if (N <= 0 || K <= 0 || Cout <= 0 || ldx < K || ldy < Cout)
    throw std::invalid_argument("dense_wgrad: bad shape");
  DTF_DT_SWITCH(dt, hipLaunchKernelGGL(dense_wgrad_kernel<DT>, dim3(grid_for((size_t)K * Cout)),
                                       dim3(256), 0, s, x, dy, dw, N, K, Cout, ldx, ldy));
}

void launch_dense_dgrad(const void* dy, const float* w, void* dx, int N, int K, int Cout, int ldy,
                        int ldx, int dt, hipStream_t s) {
  if (N <= 0 || K <= 0 || Cout <= 0 || ldx < K || ldy < Cout)
    throw std::invalid_argument("dense_dgrad: bad shape");
  DTF_DT_SWITCH(dt, hipLaunchKernelGGL(dense_dgrad_kernel<DT>, dim3(grid_for((size_t)N * ldx)),
                                       dim3(256), 0, s, dy, w, dx, N, K, Cout, ldy, ldx));
}

void launch_nonfinite_flag(const float* g, size_t n, unsigned int* flag, hipStream_t s) {
  hipLaunchKernelGGL(nonfinite_flag_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, g, n, flag);
}

void launch_loss_scale_update(float* state, unsigned int* flag, float* loss_scale_dev, float growth,
                              float multiplier, hipStream_t s) {
  hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, s, state, flag,
                     loss_scale_dev, growth, multiplier);
}
