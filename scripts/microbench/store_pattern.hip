// Microbenchmark: HBM write bandwidth of the conv epilogue's store pattern vs fully coalesced.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench/store_pattern.hip -o /tmp/store_pattern && /tmp/store_pattern
// Measured on an MI355X (411 MB tensor): epilogue-like 6.63 TB/s, coalesced 5.45 TB/s, 64-byte contiguous 5.49 TB/s:
// the lane-per-pixel 16-byte stores of conv_igemm are not what limits the bandwidth-bound layers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// Epilogue-like: block = 256 threads (4 waves as 2x2), tile 128 pixels x 128 channels, ldc = C.
// wave (wm, wn): pixels wm*64 + mr*16 + fr, channels wn*64 + fq*16 + h*8
__global__ void epi_like(unsigned short* out, int M, int C) {
  const int tilesN = C / 128;
  const int mt = blockIdx.x / tilesN, nt = blockIdx.x % tilesN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
  u32x4 v = {1u, 2u, 3u, (unsigned)lane};
  for (int mr = 0; mr < 4; ++mr) {
    const size_t m = (size_t)mt * 128 + wm * 64 + mr * 16 + fr;
    if (m >= (size_t)M) continue;
    unsigned short* p = out + m * C + nt * 128 + wn * 64 + fq * 16;
    for (int h = 0; h < 2; ++h) *reinterpret_cast<u32x4*>(p + h * 8) = v;
  }
}
// Same tile ownership, but a wave writes 4 full pixel rows (4 x 256 B) per instruction:
// lane -> pixel (lane / 16), 16-byte chunk (lane % 16) of the 128-channel row segment
__global__ void coalesced(unsigned short* out, int M, int C) {
  const int tilesN = C / 128;
  const int mt = blockIdx.x / tilesN, nt = blockIdx.x % tilesN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4 v = {1u, 2u, 3u, (unsigned)lane};
  for (int it = 0; it < 8; ++it) {
    const size_t m = (size_t)mt * 128 + wave * 32 + it * 4 + (lane >> 4);
    if (m >= (size_t)M) continue;
    *reinterpret_cast<u32x4*>(out + m * C + nt * 128 + (lane & 15) * 8) = v;
  }
}
// epilogue-like, but the two 16-byte halves of a lane are written by one 32-byte... (not possible) -> swap h and fq roles:
// lane writes chunk (fq + 4*h)?? keep channels contiguous per instruction: instruction h covers channels h*32 + fq*8
__global__ void epi_fix(unsigned short* out, int M, int C) {
  const int tilesN = C / 128;
  const int mt = blockIdx.x / tilesN, nt = blockIdx.x % tilesN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, fr = lane & 15, fq = lane >> 4;
  u32x4 v = {1u, 2u, 3u, (unsigned)lane};
  for (int mr = 0; mr < 4; ++mr) {
    const size_t m = (size_t)mt * 128 + wm * 64 + mr * 16 + fr;
    if (m >= (size_t)M) continue;
    unsigned short* p = out + m * C + nt * 128 + wn * 64;
    for (int h = 0; h < 2; ++h) *reinterpret_cast<u32x4*>(p + h * 32 + fq * 8) = v;
  }
}
int main() {
  const int M = 802816, C = 256;
  unsigned short* out;
  CK(hipMalloc(&out, (size_t)M * C * 2));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int grid = (M / 128) * (C / 128);
  for (int k = 0; k < 3; ++k) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(a));
      for (int i = 0; i < 10; ++i) {
        if (k == 0) hipLaunchKernelGGL(epi_like, dim3(grid), dim3(256), 0, 0, out, M, C);
        if (k == 1) hipLaunchKernelGGL(coalesced, dim3(grid), dim3(256), 0, 0, out, M, C);
        if (k == 2) hipLaunchKernelGGL(epi_fix, dim3(grid), dim3(256), 0, 0, out, M, C);
      }
      CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      if (rep) printf("%s: %.1f us, %.2f TB/s\n", k == 0 ? "epilogue-like" : k == 1 ? "coalesced" : "epilogue 64B-contig",
                      ms * 100, (double)M * C * 2 / (ms / 10 * 1e-3) / 1e12);
    }
  }
  return 0;
}
