// Shared device helpers for the gfx950 (MI355X, CDNA4) kernels.
//
// Conventions used by every kernel in this directory:
//  * activations are NHWC bf16 (raw uint16_t bits), accumulation and statistics are fp32;
//  * a wavefront has 64 lanes; blocks are 256 threads = 4 waves, one per SIMD of a CU;
//  * matrix work goes through v_mfma_f32_16x16x32_bf16 (A: lane l holds row l&15,
//    k = 8*(l>>4)..+7; B: lane l holds column l&15, same k; C/D: column l&15,
//    rows 4*(l>>4)..+3 in the four result registers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdexcept>
#include <string>

typedef uint16_t bf16_t;  // raw bfloat16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;   // 16-byte load/store unit
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define DTF_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) {
  return __uint_as_float(((uint32_t)v) << 16);
}

// Round-to-nearest-even conversion; lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaNs.
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(bf16_t, h);
}

// Two values in one v_cvt_pk_bf16_f32 (converted one by one they cost two conversions, a shift and
// an or).
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ void unpack8(const u32x4 v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

__device__ __forceinline__ u32x4 pack8(const float* f) {
  u32x4 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
  v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}

// ---- element types of activations / packed weights ---------------------------------------------
// Kernels that move activations are templates over DT: bfloat16 (the default of the engine),
// IEEE half (the reference's fp16 mixed precision, `resnet_imagenet_main.py:97-100`) and fp32
// (the reference's default precision).  Arithmetic is fp32 in all three; only storage differs.
// Tensor arguments cross the launcher interface as `void*` + DT.
constexpr int DT_BF16 = 0, DT_F16 = 1, DT_F32 = 2;

template <int DT> struct Elem { typedef uint16_t T; };
template <> struct Elem<DT_F32> { typedef float T; };

__device__ __forceinline__ float h2f(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }

template <int DT>
__device__ __forceinline__ float to_f(typename Elem<DT>::T v) {
  if constexpr (DT == DT_BF16) return bf2f(v);
  else if constexpr (DT == DT_F16) return h2f(v);
  else return v;
}
template <int DT>
__device__ __forceinline__ typename Elem<DT>::T from_f(float f) {
  if constexpr (DT == DT_BF16) return f2bf(f);
  else if constexpr (DT == DT_F16) return f2h(f);
  else return f;
}

// Eight consecutive elements as they sit in memory: 16 bytes for the 16-bit types, 32 for fp32.
template <int DT> struct Raw8 { u32x4 v; };
template <> struct Raw8<DT_F32> { f32x4 lo, hi; };

// NT: non-temporal access (streaming cache policy) for data that is not needed again soon - a
// sweep over a tensor larger than the caches runs 10 - 15 % faster when its last-use operands do
// not displace what the next launch will read.
template <int DT, bool NT = false>
__device__ __forceinline__ Raw8<DT> ldraw(const void* base, size_t off) {
  Raw8<DT> r;
  if constexpr (DT == DT_F32) {
    const float* p = reinterpret_cast<const float*>(base) + off;
    if constexpr (NT) {
      r.lo = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
      r.hi = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 4));
    } else {
      r.lo = *reinterpret_cast<const f32x4*>(p);
      r.hi = *reinterpret_cast<const f32x4*>(p + 4);
    }
  } else {
    const u32x4* p = reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(base) + off);
    if constexpr (NT) r.v = __builtin_nontemporal_load(p);
    else r.v = *p;
  }
  return r;
}
template <int DT>
__device__ __forceinline__ Raw8<DT> zero_raw() {
  Raw8<DT> r;
  if constexpr (DT == DT_F32) { r.lo = f32x4{0.f, 0.f, 0.f, 0.f}; r.hi = r.lo; }
  else r.v = u32x4{0u, 0u, 0u, 0u};
  return r;
}
template <int DT>
__device__ __forceinline__ void unpack8(const Raw8<DT>& r, float* f) {
  if constexpr (DT == DT_F32) {
    f[0] = r.lo[0]; f[1] = r.lo[1]; f[2] = r.lo[2]; f[3] = r.lo[3];
    f[4] = r.hi[0]; f[5] = r.hi[1]; f[6] = r.hi[2]; f[7] = r.hi[3];
  } else if constexpr (DT == DT_F16) {
    const uint32_t w[4] = {r.v.x, r.v.y, r.v.z, r.v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f[2 * j] = h2f((uint16_t)(w[j] & 0xffffu));
      f[2 * j + 1] = h2f((uint16_t)(w[j] >> 16));
    }
  } else {
    unpack8(r.v, f);
  }
}
// Stores eight fp32 values as eight consecutive elements at base + off.
template <int DT, bool NT = false>
__device__ __forceinline__ void straw(void* base, size_t off, const float* f) {
  if constexpr (DT == DT_F32) {
    float* p = reinterpret_cast<float*>(base) + off;
    const f32x4 lo = f32x4{f[0], f[1], f[2], f[3]}, hi = f32x4{f[4], f[5], f[6], f[7]};
    if constexpr (NT) {
      __builtin_nontemporal_store(lo, reinterpret_cast<f32x4*>(p));
      __builtin_nontemporal_store(hi, reinterpret_cast<f32x4*>(p + 4));
    } else {
      *reinterpret_cast<f32x4*>(p) = lo;
      *reinterpret_cast<f32x4*>(p + 4) = hi;
    }
  } else {
    u32x4 v;
    if constexpr (DT == DT_F16) {
      v.x = (uint32_t)f2h(f[0]) | ((uint32_t)f2h(f[1]) << 16);
      v.y = (uint32_t)f2h(f[2]) | ((uint32_t)f2h(f[3]) << 16);
      v.z = (uint32_t)f2h(f[4]) | ((uint32_t)f2h(f[5]) << 16);
      v.w = (uint32_t)f2h(f[6]) | ((uint32_t)f2h(f[7]) << 16);
    } else {
      v = pack8(f);
    }
    u32x4* p = reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(base) + off);
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
  }
}
// Two fp32 values packed into one 32-bit word of 16-bit elements (16-bit types only).
// the two 16-bit elements packed in a 32-bit word, as floats (bf16 / fp16 only)
template <int DT>
__device__ __forceinline__ float lo16(uint32_t w) {
  if constexpr (DT == DT_BF16) return __uint_as_float(w << 16);
  else return to_f<DT>((uint16_t)(w & 0xffffu));
}
template <int DT>
__device__ __forceinline__ float hi16(uint32_t w) {
  if constexpr (DT == DT_BF16) return __uint_as_float(w & 0xffff0000u);
  else return to_f<DT>((uint16_t)(w >> 16));
}

template <int DT>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  if constexpr (DT == DT_F16) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));      // v_cvt_pk_f16_f32
  } else {
    return pack2bf(lo, hi);
  }
}

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

// One K-step of a 16x16 output tile from the 16 bytes each lane read of the A (weights) and B
// (pixels) rows.
template <int DT>
__device__ __forceinline__ f32x4 mfma_step(const u32x4& a, const u32x4& b, f32x4 c) {
  if constexpr (DT == DT_BF16) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a),
                                                   __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  } else if constexpr (DT == DT_F16) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a),
                                                  __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  } else {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    return c;
  }
}


// Runs `...` with the compile-time constant DT bound to the run-time element type `dt`.
// Cache policy of a sweep kernel: bit 0 - loads, bit 1 - stores are non-temporal.
#define DTF_NT_SWITCH(ntm, ...)                                                         \
  switch ((ntm) & 3) {                                                                  \
    case 0: { constexpr int NTM = 0; __VA_ARGS__; } break;                              \
    case 1: { constexpr int NTM = 1; __VA_ARGS__; } break;                              \
    case 2: { constexpr int NTM = 2; __VA_ARGS__; } break;                              \
    default: { constexpr int NTM = 3; __VA_ARGS__; } break;                             \
  }

#define DTF_DT_SWITCH(dt, ...)                                                          \
  switch (dt) {                                                                         \
    case DT_BF16: { constexpr int DT = DT_BF16; __VA_ARGS__; } break;                   \
    case DT_F16: { constexpr int DT = DT_F16; __VA_ARGS__; } break;                     \
    case DT_F32: { constexpr int DT = DT_F32; __VA_ARGS__; } break;                     \
    default: throw std::invalid_argument("unknown element type (0: bf16, 1: fp16, 2: fp32)"); \
  }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Direct-to-LDS loads (global_load_lds) are ordered for a following ds_read only by the issuing
// wave's own vector-memory wait followed by a barrier; the compiler's wait insertion does not
// reliably track them (it has been seen to omit the wait across a loop back edge), so every
// kernel that reads a DMA-filled tile states the wait itself before the barrier.
__device__ __forceinline__ void dma_wait_all() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Unsigned division by a run-time constant with a precomputed multiplier
// (q = (n * mul) >> 32 >> shift); exact for n < 2^31. Built on the host by make_fastdiv().
struct FastDiv {
  uint32_t d, mul, shift;
};

__host__ __device__ __forceinline__ uint32_t fd_div(uint32_t n, const FastDiv& f) {
#if defined(__HIP_DEVICE_COMPILE__)
  return f.d == 1 ? n : (__umulhi(n, f.mul) >> f.shift);
#else
  return f.d == 1 ? n : (uint32_t)((((uint64_t)n * f.mul) >> 32) >> f.shift);
#endif
}

static inline FastDiv make_fastdiv(uint32_t d) {
  // l = ceil(log2 d), mul = ceil(2^(31+l) / d) < 2^32, q = umulhi(n, mul) >> (l - 1).
  // mul*d - 2^(31+l) < d <= 2^l, which makes the quotient exact for every n < 2^31.
  FastDiv f;
  f.d = d;
  if (d <= 1) { f.mul = 0; f.shift = 0; f.d = 1; return f; }
  uint32_t l = 0;
  while ((1ull << l) < d) ++l;
  uint64_t p2 = 1ull << (31 + l);
  f.mul = (uint32_t)((p2 + d - 1) / d);
  f.shift = l - 1;
  return f;
}

#define HIP_CHECK(expr)                                                            \
  do {                                                                             \
    hipError_t _e = (expr);                                                        \
    if (_e != hipSuccess)                                                          \
      throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)
