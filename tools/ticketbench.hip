// Does a "last-arriver ticket" beat the per-layer bn_finalize launch?  (after its fp64
// atomics every producer workgroup bumps a ticket; the LAST arriver reads the sums with device-scope atomic loads and
// writes the (scale, shift) table — no release fence, nobody waits.)
//
// The chain of a train-mode forward pass, R times back to back on one stream:
//     producer   G persistent workgroups stream a tile set (read, fma, write: a ~10-30 us "layer") and end with ONE set of
//                2 C fp64 atomics per workgroup into replica (wg % NREP) of the layer's statistics
//     finalize   batch mean / variance from the replicas -> table [3][C]           (variant L: its own launch, as today)
//     consumer   G workgroups read the table with plain loads, transform the producer's output, write     (next layer)
// Variant T: no finalize launch; the producer ends with  s_waitcnt vmcnt(0) -> barrier -> ticket = atomicAdd(+1, returning)
// -> the workgroup that drew G - 1 sums the replicas with agent-scope atomic loads, writes the table with plain stores (the
// kernel boundary publishes them) and re-arms the ticket.  Correctness is checked: the consumer verifies every table entry
// against the value the statistics imply (all inputs are constants, so the expected entry is known).
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ticketbench.hip -o tools/ticketbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NREP = 32, C = 32;

__device__ __forceinline__ void finalize_channel(const double* st, int nrep, int c, double count, float* tab, bool atomic_loads) {
  double s1 = 0.0, s2 = 0.0;
  for (int r = 0; r < nrep; ++r) {
    const double* p = st + (size_t)r * 2 * C + 2 * c;
    if (atomic_loads) {
      s1 += __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s2 += __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      s1 += p[0]; s2 += p[1];
    }
  }
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + 1e-3));
  tab[c] = invstd; tab[C + c] = -(float)mean * invstd; tab[2 * C + c] = 1.0f;
}

// x [M][C] -> y [M][C] (y = x * sc + sh of the INPUT table), statistics of y; persistent over 64-pixel tiles
template <bool TICKET>
__global__ __launch_bounds__(256) void producer_kernel(const float* __restrict__ x, float* __restrict__ y, long long M,
                                                       const float* __restrict__ in_tab, double* __restrict__ stats,
                                                       unsigned* __restrict__ ticket, float* __restrict__ out_tab, double count,
                                                       int* __restrict__ bad) {
  __shared__ float red[2 * C];
  __shared__ int s_last;
  const int tid = threadIdx.x, c4 = (tid % (C / 4)) * 4, r0 = tid / (C / 4);
  if (tid < 2 * C) red[tid] = 0.0f;
  const float4 sc = *reinterpret_cast<const float4*>(in_tab + c4), sh = *reinterpret_cast<const float4*>(in_tab + C + c4);
  // the consumer role of this kernel: the input table must be what the previous layer's statistics imply (constants)
  if (tid == 0 && !(in_tab[2 * C] == 1.0f)) atomicAdd(bad, 1);
  float4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
  const long long tiles = (M + 63) / 64;
  for (long long t = blockIdx.x; t < tiles; t += gridDim.x) {
    for (int r = r0; r < 64; r += 256 / (C / 4)) {
      const long long m = t * 64 + r;
      if (m >= M) continue;
      const float4 v = *reinterpret_cast<const float4*>(x + m * C + c4);
      float4 o = {fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w)};
      *reinterpret_cast<float4*>(y + m * C + c4) = o;
      s1.x += o.x; s1.y += o.y; s1.z += o.z; s1.w += o.w;
      s2.x += o.x * o.x; s2.y += o.y * o.y; s2.z += o.z * o.z; s2.w += o.w * o.w;
    }
  }
  __syncthreads();
  atomicAdd(&red[2 * c4 + 0], s1.x); atomicAdd(&red[2 * c4 + 1], s2.x); atomicAdd(&red[2 * c4 + 2], s1.y); atomicAdd(&red[2 * c4 + 3], s2.y);
  atomicAdd(&red[2 * c4 + 4], s1.z); atomicAdd(&red[2 * c4 + 5], s2.z); atomicAdd(&red[2 * c4 + 6], s1.w); atomicAdd(&red[2 * c4 + 7], s2.w);
  __syncthreads();
  if (tid < 2 * C) atomicAdd(&stats[(size_t)(blockIdx.x % NREP) * 2 * C + tid], (double)red[tid]);
  if (TICKET) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's statistics atomics have been performed
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    __syncthreads();
    if (s_last) {
      if (tid < C) finalize_channel(stats, NREP, tid, count, out_tab, true);
      if (tid == 0) atomicExch(ticket, 0u);
    }
  }
}

__global__ void finalize_kernel(const double* __restrict__ stats, double count, float* __restrict__ tab) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) finalize_channel(stats, NREP, c, count, tab, false);
}

__global__ void check_kernel(const float* __restrict__ tab, int layers, int* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= layers * C) return;
  const float* t = tab + (size_t)(i / C) * 3 * C;
  const int c = i % C;
  // every layer normalises its input: from the third table on (two normalisations behind it) invstd ~ 1, shift ~ 0
  if (i / C >= 3 && !(fabsf(t[c] - 1.0f) < 2e-3f && fabsf(t[C + c]) < 2e-3f && t[2 * C + c] == 1.0f)) atomicAdd(bad, 1);
}

int main(int argc, char** argv) {
  const int R = 60;
  for (long long M : {64LL * 3136, 64LL * 12544}) {
    const double count = (double)M;
    float *x, *y, *tab; double* stats; unsigned* ticket; int* bad;
    CK(hipMalloc(&x, M * C * 4)); CK(hipMalloc(&y, M * C * 4)); CK(hipMalloc(&tab, (size_t)(R + 1) * 3 * C * 4));
    CK(hipMalloc(&stats, (size_t)R * NREP * 2 * C * 8)); CK(hipMalloc(&ticket, 4)); CK(hipMalloc(&bad, 4));
    std::vector<float> h((size_t)M * C);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.0f + 0.25f * (float)(i % C);
    CK(hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> t0(3 * C, 0.0f);
    for (int c = 0; c < C; ++c) { t0[c] = 1.0f; t0[2 * C + c] = 1.0f; }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int G : {1280, 512}) {
      std::vector<float> tab_launch((size_t)(R + 1) * 3 * C), tab_ticket((size_t)(R + 1) * 3 * C);
      for (int variant = 0; variant < 2; ++variant) {
        float best = 1e30f; int nbad = 0;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipMemset(stats, 0, (size_t)R * NREP * 2 * C * 8)); CK(hipMemset(ticket, 0, 4)); CK(hipMemset(bad, 0, 4));
          CK(hipMemcpy(tab, t0.data(), 3 * C * 4, hipMemcpyHostToDevice));
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0));
          for (int l = 0; l < R; ++l) {
            const float* src = (l & 1) ? y : x; float* dst = (l & 1) ? x : y;
            float* it = tab + (size_t)l * 3 * C; float* ot = tab + (size_t)(l + 1) * 3 * C;
            double* st = stats + (size_t)l * NREP * 2 * C;
            if (variant == 0) {
              hipLaunchKernelGGL(producer_kernel<false>, dim3(G), dim3(256), 0, nullptr, src, dst, M, it, st, ticket, ot, count, bad);
              hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(64), 0, nullptr, st, count, ot);
            } else {
              hipLaunchKernelGGL(producer_kernel<true>, dim3(G), dim3(256), 0, nullptr, src, dst, M, it, st, ticket, ot, count, bad);
            }
          }
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
          hipLaunchKernelGGL(check_kernel, dim3((R * C + 255) / 256), dim3(256), 0, nullptr, tab, R, bad);
          int b; CK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost));
          nbad += b;
          CK(hipMemcpy((variant == 0 ? tab_launch : tab_ticket).data(), tab, tab_launch.size() * 4, hipMemcpyDeviceToHost));
        }
        if (variant == 1) {      // the ticket's tables against the launch's (the atomics' order moves the last bits only)
          double worst = 0.0;
          for (size_t i = 0; i < tab_launch.size(); ++i) worst = std::max(worst, (double)fabsf(tab_launch[i] - tab_ticket[i]));
          printf("    max |table(ticket) - table(launch)| over %d layers: %.2e\n", R, worst);
        }
        printf("M = %8lld pixels x %d channels (%5.1f MB per layer), %4d workgroups, %-28s: %7.2f us per layer (bad table entries %d)\n",
               M, C, 2.0 * M * C * 4 / 1e6, G, variant == 0 ? "finalize launch (as today)" : "last-arriver ticket", best * 1e3f / R, nbad);
      }
    }
    CK(hipFree(x)); CK(hipFree(y)); CK(hipFree(tab)); CK(hipFree(stats)); CK(hipFree(ticket)); CK(hipFree(bad));
  }
  return 0;
}
