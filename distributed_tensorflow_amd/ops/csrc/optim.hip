// Optimizer, weight packing and input preprocessing kernels.
//
// * sgd_momentum_l2: the whole Keras SGD-momentum update over the flat fp32 buffers in one
//   launch:  g = grad*grad_scale + l2x2*w;  v = momentum*v - lr*g;  w += v.
//   grad_scale folds 1/replicas and 1/loss_scale; l2x2 holds 2*lambda per element (zero for
//   BatchNorm scale/offset, which the reference does not regularise).
// * pack_weights: one multi-tensor launch that converts every fp32 HWIO master weight into
//   the bf16 layouts the MFMA kernels read: [Cout][taps][Cin] (K contiguous) for forward and
//   [Cin][taps][Cout] for the data gradient, with zero padding of channels / taps where a
//   kernel wants aligned K.
// * preprocess_pad: fp32 NHWC images -> bf16, zero spatial padding and channel padding (the
//   explicit ZeroPadding2D in front of the stem convolutions is materialised here, once).
#include <stdexcept>

#include "common.h"
#include "elementwise.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float f4;

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ v,
                                                  const float* __restrict__ l2x2, size_t n4,
                                                  size_t n, float lr, float momentum,
                                                  float grad_scale, int use_momentum,
                                                  const float* __restrict__ hyper,
                                                  const float* __restrict__ skip) {
  if (hyper) {   // per-step scalars live in device memory so a captured graph sees new values
    lr = hyper[0];
    momentum = hyper[1];
    grad_scale = hyper[2];
  }
  // dynamic loss scaling, decided on the device (the host never looks): skip[0] != 0 - a
  // non-finite gradient was found, the step is left out; skip[1] = 1 / (scale the gradients
  // were produced with)
  if (skip) {
    if (skip[0] != 0.f) return;
    grad_scale *= skip[1];
  }
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += stride) {
    f4 w = reinterpret_cast<const f4*>(p)[i];
    const f4 gr = reinterpret_cast<const f4*>(g)[i];
    const f4 l2 = reinterpret_cast<const f4*>(l2x2)[i];
    f4 gg = gr * grad_scale + l2 * w;
    if (use_momentum) {
      f4 vel = reinterpret_cast<const f4*>(v)[i];
      vel = vel * momentum - gg * lr;
      reinterpret_cast<f4*>(v)[i] = vel;
      w += vel;
    } else {
      w -= gg * lr;
    }
    reinterpret_cast<f4*>(p)[i] = w;
  }
  // tail (n not a multiple of 4)
  const size_t t = n4 * 4 + blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (t < n) {
    float w = p[t];
    const float gg = g[t] * grad_scale + l2x2[t] * w;
    if (use_momentum) {
      const float vel = v[t] * momentum - gg * lr;
      v[t] = vel;
      w += vel;
    } else {
      w -= gg * lr;
    }
    p[t] = w;
  }
}

constexpr int PACK_PER_BLOCK = 256 * 8;   // packed elements handled by one block

template <int DT>
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* __restrict__ params,
                                                           void* __restrict__ packed_v,
                                                           const PackDesc* __restrict__ descs,
                                                           int ndesc) {
  // binary search: last descriptor whose first_block <= blockIdx.x
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (descs[mid].first_block <= (int64_t)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const PackDesc d = descs[lo];
  const float* src = params + d.src_off;
  typename Elem<DT>::T* dst = reinterpret_cast<typename Elem<DT>::T*>(packed_v) + d.dst_off;
  if (d.mode == 0) {
    // dst[co][ti][tjp][cp] from src[ti][tj][ci][co]: a transpose between the two innermost
    // dimensions.  One block moves a tile of 32 cp x 64 co of one tap through LDS so that both
    // the fp32 reads (co contiguous) and the bf16 writes (cp contiguous) are coalesced.
    __shared__ float tile[32][65];
    int64_t b = (int64_t)blockIdx.x - d.first_block;
    const int cot = (d.Cout + 63) / 64, cpt = (d.Cpad + 31) / 32;
    const int co0 = (int)(b % cot) * 64; b /= cot;
    const int cp0 = (int)(b % cpt) * 32; b /= cpt;
    const int tjp = (int)(b % d.TJpad);
    const int ti = (int)(b / d.TJpad);
    {
      const int co = co0 + (threadIdx.x & 63);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = (threadIdx.x >> 6) + 4 * k, cp = cp0 + r;
        float v = 0.f;
        if (cp < d.Cin && tjp < d.TJ && co < d.Cout)
          v = src[((int64_t)(ti * d.TJ + tjp) * d.Cin + cp) * d.Cout + co];
        tile[r][threadIdx.x & 63] = v;
      }
    }
    __syncthreads();
    if (DT != DT_F32 && d.Cpad % 8 == 0 && d.dst_off % 8 == 0) {
      // eight consecutive cp of one output channel per thread: one 16-byte store
      const int r8 = (threadIdx.x & 3) * 8, c = threadIdx.x >> 2, co = co0 + c, cp = cp0 + r8;
      if (cp < d.Cpad && co < d.Cout) {
        u32x4 o;
        o.x = pack2<DT>(tile[r8 + 0][c], tile[r8 + 1][c]); o.y = pack2<DT>(tile[r8 + 2][c], tile[r8 + 3][c]);
        o.z = pack2<DT>(tile[r8 + 4][c], tile[r8 + 5][c]); o.w = pack2<DT>(tile[r8 + 6][c], tile[r8 + 7][c]);
        *reinterpret_cast<u32x4*>(dst + ((((int64_t)co * d.TI + ti) * d.TJpad + tjp) * d.Cpad + cp)) = o;
      }
      return;
    }
    {
      const int r = threadIdx.x & 31, cp = cp0 + r;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int c = (threadIdx.x >> 5) + 8 * k, co = co0 + c;
        if (cp < d.Cpad && co < d.Cout)
          dst[(((int64_t)co * d.TI + ti) * d.TJpad + tjp) * d.Cpad + cp] = from_f<DT>(tile[r][c]);
      }
    }
    return;
  }
  // mode 1: dst[ci][ti][tj][cop], cop contiguous in both tensors
  const int64_t base = ((int64_t)blockIdx.x - d.first_block) * PACK_PER_BLOCK;
  if (DT != DT_F32 && d.CoutPad % 8 == 0 && d.Cout % 4 == 0 && d.src_off % 4 == 0 && d.dst_off % 8 == 0 &&
      d.count < (1ll << 31)) {
    // eight consecutive output channels per thread: one index decode in 32-bit arithmetic, two
    // 16-byte loads, one 16-byte store (the per-element form spends its time in 64-bit divisions)
    const int i = (int)base + (int)threadIdx.x * 8;
    if (i >= (int)d.count) return;
    int r = i;
    const int cop = r % d.CoutPad; r /= d.CoutPad;
    const int tj = r % d.TJ; r /= d.TJ;
    const int ti = r % d.TI;
    const int ci = r / d.TI;
    const float* row = src + ((int64_t)(ti * d.TJ + tj) * d.Cin + ci) * d.Cout + cop;
    f4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
    if (cop + 4 <= d.Cout) lo = *reinterpret_cast<const f4*>(row);
    if (cop + 8 <= d.Cout) hi = *reinterpret_cast<const f4*>(row + 4);
    u32x4 o;
    o.x = pack2<DT>(lo[0], lo[1]); o.y = pack2<DT>(lo[2], lo[3]);
    o.z = pack2<DT>(hi[0], hi[1]); o.w = pack2<DT>(hi[2], hi[3]);
    *reinterpret_cast<u32x4*>(dst + i) = o;
    return;
  }
  for (int e = 0; e < 8; ++e) {
    const int64_t i = base + e * 256 + threadIdx.x;
    if (i >= d.count) return;
    int64_t r = i;
    const int cop = (int)(r % d.CoutPad); r /= d.CoutPad;
    const int tj = (int)(r % d.TJ); r /= d.TJ;
    const int ti = (int)(r % d.TI);
    const int ci = (int)(r / d.TI);
    float val = 0.f;
    if (cop < d.Cout) val = src[((int64_t)(ti * d.TJ + tj) * d.Cin + ci) * d.Cout + cop];
    dst[i] = from_f<DT>(val);
  }
}

template <int DT>
__global__ __launch_bounds__(256) void preprocess_pad_kernel(const float* __restrict__ in,
                                                             void* __restrict__ out_v, int N,
                                                             int H, int W, int Cin, int pt, int pl,
                                                             int OH, int OW, int Cpad) {
  const size_t total = (size_t)N * OH * OW;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int ow = (int)(r % OW); r /= OW;
    const int oh = (int)(r % OH);
    const int n = (int)(r / OH);
    const int h = oh - pt, w = ow - pl;
    const bool inside = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
    const float* src = in + ((size_t)(n * H + (inside ? h : 0)) * W + (inside ? w : 0)) * Cin;
    typename Elem<DT>::T* dst = reinterpret_cast<typename Elem<DT>::T*>(out_v) + i * Cpad;
    for (int c = 0; c < Cpad; ++c) dst[c] = from_f<DT>((inside && c < Cin) ? src[c] : 0.f);
  }
}

// The ImageNet case - three fp32 channels to four 16-bit ones - with a thread per PAIR of output
// pixels: six coalesced float loads and one 16-byte store instead of four 2-byte stores per pixel
// (the staging of a batch sits alone at the head of the step: 86 -> 6x us for 256 x 224 x 224).
template <int DT>
__global__ __launch_bounds__(256) void preprocess_pad_rgb_kernel(const float* __restrict__ in,
                                                                 u32x4* __restrict__ out, int N, int H,
                                                                 int W, int pt, int pl, int OH, int OW) {
  const int pairs = OW >> 1;
  const size_t total = (size_t)N * OH * pairs;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    size_t r = i;
    const int op = (int)(r % pairs); r /= pairs;
    const int oh = (int)(r % OH);
    const int n = (int)(r / OH);
    const int h = oh - pt, w0 = 2 * op - pl;
    const bool row = (unsigned)h < (unsigned)H;
    const bool in0 = row && (unsigned)w0 < (unsigned)W, in1 = row && (unsigned)(w0 + 1) < (unsigned)W;
    const float* src = in + ((size_t)(n * H + (row ? h : 0)) * W) * 3;
    float a[3] = {0.f, 0.f, 0.f}, b[3] = {0.f, 0.f, 0.f};
    if (in0) { a[0] = src[w0 * 3]; a[1] = src[w0 * 3 + 1]; a[2] = src[w0 * 3 + 2]; }
    if (in1) { b[0] = src[w0 * 3 + 3]; b[1] = src[w0 * 3 + 4]; b[2] = src[w0 * 3 + 5]; }
    u32x4 o;
    o.x = pack2<DT>(a[0], a[1]); o.y = pack2<DT>(a[2], 0.f);
    o.z = pack2<DT>(b[0], b[1]); o.w = pack2<DT>(b[2], 0.f);
    out[i] = o;
  }
}

// Zero fill as a kernel (not hipMemsetAsync): inside a captured hipGraph a large memset node
// was observed to race with the atomics of the kernel that follows it; a kernel node is
// ordered like every other launch.
__global__ __launch_bounds__(256) void zero_kernel(u32x4* __restrict__ p, size_t n16,
                                                   uint8_t* __restrict__ tail, int ntail) {
  const u32x4 z = {0u, 0u, 0u, 0u};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16;
       i += (size_t)gridDim.x * blockDim.x)
    p[i] = z;
  if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}

template <int DT>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ in,
                                                   void* __restrict__ out_v, size_t n) {
  typename Elem<DT>::T* out = reinterpret_cast<typename Elem<DT>::T*>(out_v);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x)
    out[i] = from_f<DT>(in[i]);
}

inline int grid_for(size_t total) {
  size_t b = (total + 255) / 256;
  return (int)(b < 8192 ? (b ? b : 1) : 8192);
}

}  // namespace

void launch_sgd_momentum_l2(float* p, const float* g, float* v, const float* l2x2, size_t n,
                            float lr, float momentum, float grad_scale, int use_momentum,
                            const float* hyper, const float* skip, hipStream_t s) {
  if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)v | (uintptr_t)l2x2) & 15)
    throw std::invalid_argument("sgd: buffers must be 16-byte aligned");
  const size_t n4 = n / 4;
  hipLaunchKernelGGL(sgd_kernel, dim3(grid_for(n4 ? n4 : 1)), dim3(256), 0, s, p, g, v, l2x2, n4, n,
                     lr, momentum, grad_scale, use_momentum, hyper, skip);
}

void launch_pack_weights(const float* params, void* packed, const PackDesc* descs, int ndesc,
                         int nblocks, int dt, hipStream_t s) {
  if (ndesc <= 0 || nblocks <= 0) return;
  DTF_DT_SWITCH(dt, hipLaunchKernelGGL(pack_weights_kernel<DT>, dim3(nblocks), dim3(256), 0, s,
                                       params, packed, descs, ndesc));
}

void launch_preprocess_pad(const float* in, void* out, int N, int H, int W, int Cin, int pt,
                           int pl, int OH, int OW, int Cpad, int dt, hipStream_t s) {
  if (Cpad < Cin) throw std::invalid_argument("preprocess_pad: Cpad < Cin");
  if (Cin == 3 && Cpad == 4 && OW % 2 == 0 && dt != DT_F32 && ((uintptr_t)out & 15) == 0) {
    const size_t pairs = (size_t)N * OH * (OW / 2);
    if (dt == DT_F16)
      hipLaunchKernelGGL(preprocess_pad_rgb_kernel<DT_F16>, dim3(grid_for(pairs)), dim3(256), 0, s, in,
                         reinterpret_cast<u32x4*>(out), N, H, W, pt, pl, OH, OW);
    else
      hipLaunchKernelGGL(preprocess_pad_rgb_kernel<DT_BF16>, dim3(grid_for(pairs)), dim3(256), 0, s, in,
                         reinterpret_cast<u32x4*>(out), N, H, W, pt, pl, OH, OW);
    return;
  }
  DTF_DT_SWITCH(dt, hipLaunchKernelGGL(preprocess_pad_kernel<DT>,
                                       dim3(grid_for((size_t)N * OH * OW)), dim3(256), 0, s, in,
                                       out, N, H, W, Cin, pt, pl, OH, OW, Cpad));
}

void launch_zero(void* ptr, size_t nbytes, hipStream_t s) {
  if ((uintptr_t)ptr & 15) throw std::invalid_argument("zero fill: pointer must be 16-byte aligned");
  const size_t n16 = nbytes / 16;
  const int ntail = (int)(nbytes - n16 * 16);
  hipLaunchKernelGGL(zero_kernel, dim3(grid_for(n16 ? n16 : 1)), dim3(256), 0, s,
                     reinterpret_cast<u32x4*>(ptr), n16,
                     reinterpret_cast<uint8_t*>(ptr) + n16 * 16, ntail);
}

void launch_cast_from_f32(const float* in, void* out, size_t n, int dt, hipStream_t s) {
  DTF_DT_SWITCH(dt, hipLaunchKernelGGL(cast_kernel<DT>, dim3(grid_for(n)), dim3(256), 0, s, in, out, n));
}
