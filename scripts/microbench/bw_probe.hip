// Memory-system probe for the sweep kernels (BatchNorm apply / backward): how fast can 256 CUs
// stream large tensors, as a function of blocks per CU, loads in flight per thread, the access
// pattern (grid-stride passes vs one contiguous slab per block) and the cache policy of the
// accesses.  Build: hipcc --offload-arch=gfx950 -O3 -o bw_probe bw_probe.hip ; run: ./bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MODE 0: read a (sum)          1 stream
// MODE 1: b = f(a)              1 read + 1 write
// MODE 2: c = f(a, b)           2 reads + 1 write   (BatchNorm apply with residual)
// MODE 3: d = f(a, b), read c   3 reads + 1 write
template <int MODE, int U, bool NT, bool SLAB>
__global__ __launch_bounds__(256) void probe(const u32x4* __restrict__ a, const u32x4* __restrict__ b,
                                             u32x4* __restrict__ c, unsigned int* sink, size_t n) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * 256;
  u32x4 acc = {0u, 0u, 0u, 0u};
  size_t i, step, end;
  if (SLAB) {            // block b owns [b*per, (b+1)*per): passes of 256*U vectors inside it
    const size_t per = (n + gridDim.x - 1) / gridDim.x;
    i = (size_t)blockIdx.x * per + threadIdx.x;
    end = (size_t)(blockIdx.x + 1) * per;
    if (end > n) end = n;
    step = 256;
  } else {
    i = tid; step = nthreads; end = n;
  }
  for (; i + (U - 1) * step < end; i += U * step) {
    u32x4 va[U], vb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      va[u] = NT ? __builtin_nontemporal_load(a + i + u * step) : a[i + u * step];
      if (MODE >= 2) vb[u] = NT ? __builtin_nontemporal_load(b + i + u * step) : b[i + u * step];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      u32x4 v = va[u];
      if (MODE >= 2) v ^= vb[u];
      if (MODE == 0) acc ^= v;
      else if (MODE == 1) { if (NT) __builtin_nontemporal_store(v, const_cast<u32x4*>(b) + i + u * step); else const_cast<u32x4*>(b)[i + u * step] = v; }
      else { if (NT) __builtin_nontemporal_store(v, c + i + u * step); else c[i + u * step] = v; }
    }
  }
  for (; i < end; i += step) {
    u32x4 v = a[i];
    if (MODE >= 2) v ^= b[i];
    if (MODE == 0) acc ^= v;
    else if (MODE == 1) const_cast<u32x4*>(b)[i] = v;
    else c[i] = v;
  }
  if (MODE == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}

template <int MODE, int U, bool NT, bool SLAB>
float run(const u32x4* a, const u32x4* b, u32x4* c, unsigned int* sink, size_t n, int grid) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<MODE, U, NT, SLAB>), dim3(grid), dim3(256), 0, 0, a, b, c, sink, n);
  CHECK(hipEventRecord(e0));
  const int reps = 10;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((probe<MODE, U, NT, SLAB>), dim3(grid), dim3(256), 0, 0, a, b, c, sink, n);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
  return ms / reps;
}

template <int MODE>
void sweep(const char* label, int streams, const u32x4* a, const u32x4* b, u32x4* c, unsigned int* sink, size_t n) {
  const double gb = (double)n * 16 * streams / 1e9;
  const int grids[] = {256, 512, 1024, 2048, 4096, 8192};
  printf("%s, %.0f MB per tensor, %d streams: TB/s by (variant, blocks)\n", label, n * 16 / 1e6, streams);
  printf("%-28s", "variant \\ blocks");
  for (int g : grids) printf("%8d", g);
  printf("\n");
#define ROW(U, NT, SLAB, name) { printf("%-28s", name); for (int g : grids) { \
    float ms = run<MODE, U, NT, SLAB>(a, b, c, sink, n, g); printf("%8.2f", gb / ms); } printf("\n"); }
  ROW(4, false, false, "stride U4")
  ROW(8, false, false, "stride U8")
  ROW(4, true, false, "stride U4 nontemporal")
  ROW(8, true, false, "stride U8 nontemporal")
  ROW(4, false, true, "slab U4")
  ROW(8, false, true, "slab U8")
  ROW(8, true, true, "slab U8 nontemporal")
#undef ROW
  fflush(stdout);
}

int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? atoi(argv[1]) : 411;
  const size_t n = mb * 1000000 / 16;
  u32x4 *a, *b, *c; unsigned int* sink;
  CHECK(hipMalloc(&a, n * 16)); CHECK(hipMalloc(&b, n * 16)); CHECK(hipMalloc(&c, n * 16));
  CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(a, 1, n * 16)); CHECK(hipMemset(b, 2, n * 16)); CHECK(hipMemset(c, 3, n * 16));
  sweep<0>("read", 1, a, b, c, sink, n);
  sweep<1>("copy", 2, a, b, c, sink, n);
  sweep<2>("2 reads + 1 write", 3, a, b, c, sink, n);
  return 0;
}
