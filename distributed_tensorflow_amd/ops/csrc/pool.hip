// Max-pool (TensorFlow SAME padding) and global mean pool for NHWC activations (bf16 / fp16 / fp32 storage).
//
// Forward max-pool stores the winning tap (0 .. k*k-1, first maximum in row-major tap order)
// as one byte per output element; backward is then a gather: every input pixel checks the at
// most ceil(k/stride)^2 windows that contain it and sums the gradients of those it won.  No
// atomics, deterministic, one 16-byte access per 8 channels.
#include <stdexcept>

#include "common.h"
#include "elementwise.h"

namespace {

// Both max-pool kernels walk the tensor one image row per block iteration: the (n, row) split is
// one 32-bit division per row instead of three 64-bit divisions per 16-byte access.
template <int DT>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(
    const void* __restrict__ x, void* __restrict__ out, uint8_t* __restrict__ idx, int N,
    int H, int W, int C, int OH, int OW, int k, int stride, int pt, int pl) {
  const uint32_t cpr = (uint32_t)C >> 3;
  const uint32_t per_row = (uint32_t)OW * cpr;
  for (uint32_t row = blockIdx.x; row < (uint32_t)(N * OH); row += gridDim.x) {
    const uint32_t n = row / (uint32_t)OH, oh = row - n * (uint32_t)OH;
    for (uint32_t j = threadIdx.x; j < per_row; j += blockDim.x) {
      const uint32_t ow = j / cpr, chunk = j - ow * cpr;
      float best[8];
      int bi[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) { best[c] = -INFINITY; bi[c] = 0; }
      for (int kh = 0; kh < k; ++kh) {
        const int hi = (int)oh * stride - pt + kh;
        if ((unsigned)hi >= (unsigned)H) continue;
        for (int kw = 0; kw < k; ++kw) {
          const int wi = (int)ow * stride - pl + kw;
          if ((unsigned)wi >= (unsigned)W) continue;
          float f[8];
          unpack8<DT>(ldraw<DT>(x, ((size_t)(n * H + hi) * W + wi) * C + chunk * 8), f);
          const int tap = kh * k + kw;
#pragma unroll
          for (int c = 0; c < 8; ++c)
            if (f[c] > best[c]) { best[c] = f[c]; bi[c] = tap; }
        }
      }
      const size_t o = ((size_t)row * OW + ow) * C + chunk * 8;
      straw<DT>(out, o, best);
      u32x2 pk;
      pk.x = bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24);
      pk.y = bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24);
      *reinterpret_cast<u32x2*>(idx + o) = pk;
    }
  }
}

template <int DT>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(
    const void* __restrict__ dy, const uint8_t* __restrict__ idx, void* __restrict__ dx, int N,
    int H, int W, int C, int OH, int OW, int k, int stride, int pt, int pl) {
  const uint32_t cpr = (uint32_t)C >> 3;
  const uint32_t per_row = (uint32_t)W * cpr;
  for (uint32_t row = blockIdx.x; row < (uint32_t)(N * H); row += gridDim.x) {
    const uint32_t n = row / (uint32_t)H;
    const int h = (int)(row - n * (uint32_t)H);
    // windows oh with 0 <= h + pt - oh*stride < k  (block-uniform)
    int oh_lo = (h + pt - k + 1 + stride - 1);
    oh_lo = oh_lo > 0 ? oh_lo / stride : 0;
    int oh_hi = (h + pt) / stride; if (oh_hi > OH - 1) oh_hi = OH - 1;
    for (uint32_t j = threadIdx.x; j < per_row; j += blockDim.x) {
      const uint32_t wu = j / cpr, chunk = j - wu * cpr;
      const int w = (int)wu;
      float g[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) g[c] = 0.f;
      int ow_lo = (w + pl - k + 1 + stride - 1);
      ow_lo = ow_lo > 0 ? ow_lo / stride : 0;
      int ow_hi = (w + pl) / stride; if (ow_hi > OW - 1) ow_hi = OW - 1;
      for (int oh = oh_lo; oh <= oh_hi; ++oh) {
        const int kh = h + pt - oh * stride;
        for (int ow = ow_lo; ow <= ow_hi; ++ow) {
          const int kw = w + pl - ow * stride;
          const unsigned tap = (unsigned)(kh * k + kw);
          const size_t o = ((size_t)(n * OH + oh) * OW + ow) * C + chunk * 8;
          const u32x2 pk = *reinterpret_cast<const u32x2*>(idx + o);
          float d[8];
          unpack8<DT>(ldraw<DT>(dy, o), d);
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const unsigned t = ((c < 4 ? pk.x : pk.y) >> (8 * (c & 3))) & 0xffu;
            if (t == tap) g[c] += d[c];
          }
        }
      }
      straw<DT>(dx, ((size_t)row * W + w) * C + chunk * 8, g);
    }
  }
}

// One block per sample.  The 256 threads are (pixel lane, 8-channel chunk): a narrow tensor
// (CIFAR: 64 channels = 8 chunks) spreads its pixels over 32 lanes that meet in LDS - with one
// thread per chunk walking all pixels the launch was four blocks of 64 dependent loads (22 us).
template <int DT>
__global__ __launch_bounds__(256) void global_mean_fwd_kernel(const void* __restrict__ x,
                                                              void* __restrict__ out, int N,
                                                              int HW, int C) {
  __shared__ float red[256 * 8];
  const int cpr = C >> 3;
  const int n = blockIdx.x, tid = threadIdx.x;
  const float inv = 1.f / (float)HW;
  for (int c0 = 0; c0 < cpr; c0 += 256) {           // groups of 256 chunks (C > 2048: several)
    const int cw = cpr - c0 < 256 ? cpr - c0 : 256;
    const int PL = 256 / cw;                        // pixel lanes
    const int chunk = tid % cw, pl = tid / cw;
    float s[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) s[c] = 0.f;
    if (pl < PL) {
      const size_t base = (size_t)n * HW * C + (size_t)(c0 + chunk) * 8;
      int p = pl;
      for (; p + 3 * PL < HW; p += 4 * PL) {        // four independent loads in flight
        Raw8<DT> v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ldraw<DT>(x, base + (size_t)(p + u * PL) * C);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float f[8];
          unpack8<DT>(v[u], f);
#pragma unroll
          for (int c = 0; c < 8; ++c) s[c] += f[c];
        }
      }
      for (; p < HW; p += PL) {
        float f[8];
        unpack8<DT>(ldraw<DT>(x, base + (size_t)p * C), f);
#pragma unroll
        for (int c = 0; c < 8; ++c) s[c] += f[c];
      }
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) red[tid * 8 + c] = s[c];
    __syncthreads();
    if (tid < cw) {
      float t[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) t[c] = 0.f;
      for (int l = 0; l < PL; ++l) {
#pragma unroll
        for (int c = 0; c < 8; ++c) t[c] += red[(l * cw + tid) * 8 + c];
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) t[c] *= inv;
      straw<DT>(out, (size_t)n * C + (size_t)(c0 + tid) * 8, t);
    }
    __syncthreads();
  }
}

template <int DT>
__global__ __launch_bounds__(256) void global_mean_bwd_kernel(const void* __restrict__ dy,
                                                              void* __restrict__ dx, int N,
                                                              int HW, int C) {
  const int cpr = C >> 3;
  const size_t total = (size_t)N * HW * cpr;
  const float inv = 1.f / (float)HW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int chunk = (int)(i % cpr);
    const size_t pix = i / cpr;
    const int n = (int)(pix / HW);
    float f[8];
    unpack8<DT>(ldraw<DT>(dy, (size_t)n * C + chunk * 8), f);
#pragma unroll
    for (int c = 0; c < 8; ++c) f[c] *= inv;
    straw<DT>(dx, pix * C + chunk * 8, f);
  }
}

inline int grid_for(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b < 2048 * 4 ? (b ? b : 1) : 2048 * 4);
}

}  // namespace

void launch_maxpool_fwd(const void* x, void* out, uint8_t* idx, int N, int H, int W, int C,
                        int OH, int OW, int k, int stride, int pt, int pl, int dt, hipStream_t s) {
  if (C % 8 || k * k > 255) throw std::invalid_argument("maxpool: C %% 8 != 0 or window too large");
  if ((int64_t)N * H * W * (C / 8) >= (1ll << 31)) throw std::invalid_argument("maxpool: tensor too large");
  const int rows = N * OH;
  DTF_DT_SWITCH(dt, hipLaunchKernelGGL(maxpool_fwd_kernel<DT>, dim3(rows < 8192 ? rows : 8192),
                                       dim3(256), 0, s, x, out, idx, N, H, W, C, OH, OW, k, stride,
                                       pt, pl));
}

void launch_maxpool_bwd(const void* dy, const uint8_t* idx, void* dx, int N, int H, int W,
                        int C, int OH, int OW, int k, int stride, int pt, int pl, int dt,
                        hipStream_t s) {
  if (C % 8) throw std::invalid_argument("maxpool: C %% 8 != 0");
  if ((int64_t)N * H * W * (C / 8) >= (1ll << 31)) throw std::invalid_argument("maxpool: tensor too large");
  const int rows = N * H;
  DTF_DT_SWITCH(dt, hipLaunchKernelGGL(maxpool_bwd_kernel<DT>, dim3(rows < 8192 ? rows : 8192),
                                       dim3(256), 0, s, dy, idx, dx, N, H, W, C, OH, OW, k, stride,
                                       pt, pl));
}

void launch_global_mean_fwd(const void* x, void* out, int N, int HW, int C, int dt,
                            hipStream_t s) {
  if (C % 8) throw std::invalid_argument("global_mean: C %% 8 != 0");
  DTF_DT_SWITCH(dt, hipLaunchKernelGGL(global_mean_fwd_kernel<DT>, dim3(N), dim3(256), 0, s, x, out,
                                       N, HW, C));
}

void launch_global_mean_bwd(const void* dy, void* dx, int N, int HW, int C, int dt,
                            hipStream_t s) {
  if (C % 8) throw std::invalid_argument("global_mean: C %% 8 != 0");
  DTF_DT_SWITCH(dt, hipLaunchKernelGGL(global_mean_bwd_kernel<DT>,
                                       dim3(grid_for((size_t)N * HW * (C / 8))), dim3(256), 0, s,
                                       dy, dx, N, HW, C));
}
