// What does an in-kernel grid barrier cost on this chip?  (The question behind a persistent "stage kernel" that would run
// several BatchNorm layers in one launch: VERDICT round 2, item 4a.)  NWG co-resident workgroups run R rounds of
// { write one value per workgroup, barrier, read the value the NEXT workgroup wrote in this round }, with the barrier
// built three ways:
//   atomics : a monotonic device-scope counter (atomicAdd, then spin on an atomic load), no fences;
//   fenced  : the same with __threadfence() (release) before the arrive and (acquire) after the wait - what correct
//             hand-over of ordinary global stores between workgroups on different XCDs needs;
//   nt      : atomics only, but the payload itself is written / read with device-scope atomics (no fence needed).
// Every spin is bounded (the kernel drains even if the barrier never completes); mismatches of the hand-over are counted.
// The same R rounds as R dependent launches of an equivalent kernel give the launch-based reference.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gridsyncbench.hip -o tools/gridsyncbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void rounds_kernel(unsigned* counter, int* payload, int rounds, int* mismatches, int* gave_up) {
  const int wg = blockIdx.x, nwg = gridDim.x;
  __shared__ int ok;
  for (int r = 0; r < rounds; ++r) {
    if (threadIdx.x == 0) {
      if (MODE == 2) atomicExch(&payload[wg], r * 1000 + wg); else payload[wg] = r * 1000 + wg;
      if (MODE == 1) __threadfence();
      atomicAdd(counter, 1u);
      const unsigned target = (unsigned)(r + 1) * (unsigned)nwg;
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && spins < 2000000) ++spins;
      if (spins >= 2000000) atomicAdd(gave_up, 1);
      if (MODE == 1) __threadfence();
      const int nb = (wg + 1) % nwg;
      const int got = MODE == 2 ? atomicAdd(&payload[nb], 0) : payload[nb];
      ok = got == r * 1000 + nb;
      if (!ok) atomicAdd(mismatches, 1);
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void one_round_kernel(int* payload, int r, int* mismatches) {
  const int wg = blockIdx.x, nwg = gridDim.x;
  if (threadIdx.x == 0) {
    const int nb = (wg + 1) % nwg;
    if (r > 0 && payload[nwg + ((r - 1) & 1) * nwg + nb] != (r - 1) * 1000 + nb) atomicAdd(mismatches, 1);
    payload[nwg + (r & 1) * nwg + wg] = r * 1000 + wg;
  }
}

int main() {
  unsigned* counter; int *payload, *mism, *gave;
  CK(hipMalloc(&counter, 4)); CK(hipMalloc(&payload, 3 * 1024 * 4)); CK(hipMalloc(&mism, 4)); CK(hipMalloc(&gave, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int R = 200;
  for (int nwg : {196, 256, 512}) {
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e30f; int hm = 0, hg = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(counter, 0, 4)); CK(hipMemset(mism, 0, 4)); CK(hipMemset(gave, 0, 4)); CK(hipMemset(payload, 0xff, 3 * 1024 * 4));
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(rounds_kernel<0>, dim3(nwg), dim3(256), 0, nullptr, counter, payload, R, mism, gave);
        if (mode == 1) hipLaunchKernelGGL(rounds_kernel<1>, dim3(nwg), dim3(256), 0, nullptr, counter, payload, R, mism, gave);
        if (mode == 2) hipLaunchKernelGGL(rounds_kernel<2>, dim3(nwg), dim3(256), 0, nullptr, counter, payload, R, mism, gave);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        int a, b; CK(hipMemcpy(&a, mism, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&b, gave, 4, hipMemcpyDeviceToHost));
        hm += a; hg += b;
      }
      printf("%3d workgroups, %-8s: %6.2f us per round (hand-over mismatches %d, spins given up %d)\n", nwg,
             mode == 0 ? "atomics" : mode == 1 ? "fenced" : "nt", best * 1e3f / R, hm, hg);
    }
    CK(hipMemset(mism, 0, 4));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(one_round_kernel, dim3(nwg), dim3(256), 0, nullptr, payload, 0, mism);
    CK(hipMemset(mism, 0, 4));
    CK(hipEventRecord(e0));
    for (int r = 0; r < R; ++r) hipLaunchKernelGGL(one_round_kernel, dim3(nwg), dim3(256), 0, nullptr, payload, r, mism);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    int a; CK(hipMemcpy(&a, mism, 4, hipMemcpyDeviceToHost));
    printf("%3d workgroups, launches: %6.2f us per round (hand-over mismatches %d)\n", nwg, ms * 1e3f / R, a);
  }
  return 0;
}
