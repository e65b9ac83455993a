// Fused softmax + cross-entropy forward/backward over bf16 / fp16 / fp32 logits (fp32 math).
//
// One 256-thread block per row.  The row is read once into registers (up to 16 values per
// thread, C <= 4096), then: block max -> block sum of exp -> loss = log(sum) + max - logit[y],
// gradient (softmax - onehot) * grad_scale written in the logits' type (padding columns up to the row
// pitch are zero-filled so that the dense-layer gradient GEMMs can read whole vectors),
// and the first-argmax == label test for the accuracy counter.
#include <stdexcept>

#include "common.h"
#include "elementwise.h"

namespace {

constexpr int TPB = 256;
constexpr int MAXV = 16;

__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  return v;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  v = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return v;
}

template <int DT, bool PROBS>
__global__ __launch_bounds__(TPB) void softmax_xent_kernel(
    const void* __restrict__ logits_v, const int32_t* __restrict__ labels,
    void* __restrict__ dlogits_v, float* __restrict__ probs, float* accum, int N, int C, int ld,
    float grad_scale, const float* __restrict__ grad_scale_dev, float loss_weight) {
  typedef typename Elem<DT>::T T;
  const T* logits = reinterpret_cast<const T*>(logits_v);
  T* dlogits = reinterpret_cast<T*>(dlogits_v);
  if (grad_scale_dev) grad_scale = grad_scale * grad_scale_dev[0];
  __shared__ float red[4];
  __shared__ int redi[4];
  const int row = blockIdx.x;
  const T* lr = logits + (size_t)row * ld;
  float v[MAXV];
  float mx = -INFINITY;
  int amax = 0x7fffffff;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int c = threadIdx.x + i * TPB;
    v[i] = c < C ? to_f<DT>(lr[c]) : -INFINITY;
    if (v[i] > mx) { mx = v[i]; amax = c; }
  }
  const float bmax = block_max(mx, red);
  // first index attaining the maximum
  int cand = (mx == bmax) ? amax : 0x7fffffff;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_xor(cand, o, 64));
  if ((threadIdx.x & 63) == 0) redi[threadIdx.x >> 6] = cand;
  float se = 0.f;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    v[i] = __expf(v[i] - bmax);   // exp(-inf) = 0 for the padding lanes
    se += v[i];
  }
  const float sum = block_sum(se, red);      // contains the barrier that publishes redi
  const int arg = min(min(redi[0], redi[1]), min(redi[2], redi[3]));
  const float inv = 1.f / sum;
  if (PROBS) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = threadIdx.x + i * TPB;
      if (c < C) probs[(size_t)row * C + c] = v[i] * inv;
    }
    return;
  }
  const int label = labels[row];
  if (dlogits) {
    T* dr = dlogits + (size_t)row * ld;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int c = threadIdx.x + i * TPB;
      if (c < ld) {
        float g = 0.f;
        if (c < C) g = (v[i] * inv - (c == label ? 1.f : 0.f)) * grad_scale;
        dr[c] = from_f<DT>(g);
      }
    }
  }
  if (threadIdx.x == 0) {
    const float ly = (label >= 0 && label < C) ? to_f<DT>(lr[label]) : 0.f;
    atomicAdd(accum + 0, (logf(sum) + bmax - ly) * loss_weight);   // 1: loss sum, 1 / N: the mean
    if (arg == label) atomicAdd(accum + 1, 1.f);
  }
}

template <int DT>
__global__ void colsum_kernel(const void* __restrict__ xv, float* __restrict__ out, int N, int C,
                              int ld) {
  const typename Elem<DT>::T* x = reinterpret_cast<const typename Elem<DT>::T*>(xv);
  // blockIdx.y owns a slice of the rows: a handful of columns-wide blocks walking all rows one
  // dependent load after the other took 99 us on [256 x 1001]
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int per = (N + gridDim.y - 1) / gridDim.y;
  const int n0 = blockIdx.y * per, n1 = min(N, n0 + per);
  float s = 0.f;
#pragma unroll 4
  for (int n = n0; n < n1; ++n) s += to_f<DT>(x[(size_t)n * ld + c]);
  if (n1 > n0) atomicAdd(out + c, s);
}

}  // namespace

void launch_softmax_xent(const void* logits, const int32_t* labels, void* dlogits,
                         float* accum, int N, int C, int ld, float grad_scale,
                         const float* grad_scale_dev, float loss_weight, int dt, hipStream_t s) {
  if (C > TPB * MAXV || ld > TPB * MAXV || ld < C)
    throw std::invalid_argument("softmax_xent: at most 4096 classes, ld >= C");
  DTF_DT_SWITCH(dt, hipLaunchKernelGGL((softmax_xent_kernel<DT, false>), dim3(N), dim3(TPB), 0, s,
                                       logits, labels, dlogits, (float*)nullptr, accum, N, C, ld,
                                       grad_scale, grad_scale_dev, loss_weight));
}

void launch_softmax_probs(const void* logits, float* probs, int N, int C, int ld, int dt,
                          hipStream_t s) {
  if (C > TPB * MAXV || ld < C) throw std::invalid_argument("softmax_probs: at most 4096 classes");
  DTF_DT_SWITCH(dt, hipLaunchKernelGGL((softmax_xent_kernel<DT, true>), dim3(N), dim3(TPB), 0, s,
                                       logits, (const int32_t*)nullptr, (void*)nullptr, probs,
                                       (float*)nullptr, N, C, ld, 1.f, (const float*)nullptr, 1.f));
}

void launch_colsum(const void* x, float* out, int N, int C, int ld, int dt, int slices_req,
                   hipStream_t s) {
  // slices_req > 0: that many row slices (1: a fixed summation order, one block per column group)
  const int slices = slices_req > 0 ? slices_req : (N >= 64 ? 64 : (N > 0 ? N : 1));
  DTF_DT_SWITCH(dt, hipLaunchKernelGGL(colsum_kernel<DT>, dim3((C + 255) / 256, slices), dim3(256),
                                       0, s, x, out, N, C, ld));
}
