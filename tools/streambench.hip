// What does the HBM deliver to plain streaming kernels of the shapes the conv stack has (1 - 3 read streams, 1 write stream,
// 16-byte accesses, 100 - 800 MB per stream)?  The practical ceiling to hold the layer tables against, next to the 8 TB/s peak.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/streambench.hip -o tools/streambench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
using f4 = __attribute__((ext_vector_type(4))) float;

template <int NR, bool NT, int UNROLL>
__global__ __launch_bounds__(256) void stream_kernel(const f4* __restrict__ a, const f4* __restrict__ b, const f4* __restrict__ c,
                                                     f4* __restrict__ o, long long n) {
  const long long stride = (long long)gridDim.x * 256 * UNROLL;
  for (long long i0 = (long long)blockIdx.x * 256 * UNROLL + threadIdx.x; i0 < n; i0 += stride) {
    f4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long long i = i0 + 256 * u;
      if (i < n) {
        v[u] = NT ? __builtin_nontemporal_load(a + i) : a[i];
        if (NR >= 2) v[u] += NT ? __builtin_nontemporal_load(b + i) : b[i];
        if (NR >= 3) v[u] *= NT ? __builtin_nontemporal_load(c + i) : c[i];
      }
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      const long long i = i0 + 256 * u;
      if (i < n) { if (NT) __builtin_nontemporal_store(v[u], o + i); else o[i] = v[u]; }
    }
  }
}

int main() {
  const size_t max_bytes = 822083584ull;             // 64 x 224 x 224 x 16 x 4 B x 4: the largest map of a pass, four of them
  f4 *a, *b, *c, *o;
  CK(hipMalloc(&a, max_bytes)); CK(hipMalloc(&b, max_bytes)); CK(hipMalloc(&c, max_bytes)); CK(hipMalloc(&o, max_bytes));
  CK(hipMemset(a, 0, max_bytes)); CK(hipMemset(b, 0, max_bytes)); CK(hipMemset(c, 0, max_bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (size_t mb : {103, 205, 411, 822}) {
    const long long n = (long long)mb * 1000000 / 16;
    printf("%4zu MB per stream:", mb);
    auto run = [&](const char* name, int nr, auto kern, int wg_per_cu) {
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(256 * wg_per_cu), dim3(256), 0, nullptr, a, b, c, o, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      printf("  %s %.2f", name, (double)(nr + 1) * n * 16 / best / 1e9);
    };
    run("1R1W", 1, stream_kernel<1, false, 4>, 8);
    run("1R1W/nt", 1, stream_kernel<1, true, 4>, 8);
    run("2R1W", 2, stream_kernel<2, false, 4>, 8);
    run("2R1W/nt", 2, stream_kernel<2, true, 4>, 8);
    run("3R1W", 3, stream_kernel<3, false, 4>, 8);
    run("3R1W/nt", 3, stream_kernel<3, true, 4>, 8);
    run("3R1W/4wg", 3, stream_kernel<3, false, 4>, 4);
    run("3R1W/u8", 3, stream_kernel<3, false, 8>, 8);
    run("3R1W/2wg,u8", 3, stream_kernel<3, false, 8>, 2);
    printf("  TB/s\n");
  }
  return 0;
}
