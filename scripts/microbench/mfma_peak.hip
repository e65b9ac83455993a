// Known-peak yardstick for the matrix-core counters: a kernel that does nothing but bf16 MFMAs
// (no memory, no LDS), 4 waves per SIMD on every CU, 8 independent accumulator tiles per wave.
// Run under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
// SQ_INSTS_VALU_MFMA_MOPS_BF16 -- ./mfma_peak` it pins what the derived "MFMA busy share" of
// profiles/conv_pmc_lds_current.md reads when the matrix cores are saturated, and how many FLOP
// one count of SQ_INSTS_VALU_MFMA_MOPS_BF16 stands for.  Also prints the sustained rate of the
// two bf16 shapes (16x16x32, 32x32x16): the chip may hold a different clock on each.
//
//   hipcc --offload-arch=gfx950 -O3 -o scripts/microbench/mfma_peak scripts/microbench/mfma_peak.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int SHAPE>
__global__ __launch_bounds__(256, 4) void mfma_loop(float* out, int iters, unsigned seed) {
  // operands from the lane id (never zero: zero-filled operands run at a higher clock)
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)(0.5f + 0.001f * (float)((threadIdx.x * 7 + i + seed) % 61));
    b[i] = (__bf16)(-0.25f + 0.002f * (float)((threadIdx.x * 13 + i * 3 + seed) % 53));
  }
  float r = 0.f;
  if constexpr (SHAPE == 16) {
    f32x4 acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t], 0, 0, 0);
    for (int t = 0; t < 8; ++t) r += acc[t][0] + acc[t][3];
  } else {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
    for (int t = 0; t < 4; ++t) r += acc[t][0] + acc[t][15];
  }
  if (r == 12345.678f) out[0] = r;                  // keeps the loop alive
}

template <int SHAPE>
double run(int iters, float* out) {
  const int blocks = 256 * 4;                       // 4 blocks of 4 waves per CU: 4 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, iters / 8, 1u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(mfma_loop<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, iters, 2u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  // per wave and iteration: 8 x (16*16*32*2) or 4 x (32*32*16*2) = 131072 FLOP
  const double flop = (double)blocks * 4 * iters * 131072.0;
  return flop / (ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  float* out = nullptr;
  if (hipMalloc(&out, 64) != hipSuccess) return 1;
  const double t16 = run<16>(iters, out);
  const double t32 = run<32>(iters, out);
  printf("mfma_f32_16x16x32_bf16: %.0f TFLOP/s   mfma_f32_32x32x16_bf16: %.0f TFLOP/s   "
         "(%d iterations, %.3g FLOP per launch)\n", t16, t32, iters, (double)1024 * 4 * iters * 131072.0);
  return 0;
}
