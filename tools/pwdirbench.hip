// Where does the weight-stationary 1x1 kernel (pw_dir_kernel, kernels_pwres.hip) spend its time on the wide layers?
// Times it on the data-gradient launch of the backward (20 glimpse steps per launch, transposed weights) and on the
// forward launch (one step, statistics on), in the build given by the macros:
//   (none)            the production kernel
//   -DJN_PWDIR_HOT    every pixel fetch from the same eight tiles (cache-resident): no HBM / L2 latency
//   -DJN_PWDIR_NOMFMA the same loads, transform and stores, matrix instructions replaced by one VALU op
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 [-D...] tools/pwdirbench.hip -o tools/pwdirbench[_hot|_nomfma]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../jolineedle_amd/csrc/kernels_conv.hip"
#include "../jolineedle_amd/csrc/kernels_pwres.hip"
#include "../jolineedle_amd/csrc/kernels_pwxs.hip"

using namespace jnr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  struct Shape { int hw, K, N; };
  const Shape shapes[] = {{14, 256, 256}, {28, 128, 128}, {14, 128, 128}, {28, 256, 128}, {14, 512, 256}, {56, 128, 64}};
  const int B = 64, S = 20, iters = 10;
  float *x, *w, *out, *tab; double* stats;
  const size_t maxe = (size_t)S * B * 56 * 56 * 128;
  CK(hipMalloc(&x, maxe * 4)); CK(hipMalloc(&out, maxe * 4)); CK(hipMalloc(&w, 512 * 512 * 4)); CK(hipMalloc(&tab, 3 * 2048 * 4));
  CK(hipMalloc(&stats, 32 * 2 * 4096 * 8));
  {
    std::vector<float> h((size_t)64 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;
    for (size_t o = 0; o < maxe; o += h.size()) CK(hipMemcpy(x + o, h.data(), std::min(h.size(), maxe - o) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, h.data(), 512 * 512 * 4, hipMemcpyHostToDevice));
    std::vector<float> t(3 * 2048, 0.0f);
    for (int i = 0; i < 2048; ++i) t[i] = 1.0f;                   // identity table (flag 0), as the gradient views have
    CK(hipMemcpy(tab, t.data(), t.size() * 4, hipMemcpyHostToDevice));
  }
  CK(hipMemset(stats, 0, 32 * 2 * 4096 * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipStream_t s = nullptr;
  auto time_it = [&](auto&& fn) {
    fn(); fn();
    hipEventRecord(e0, s);
    for (int i = 0; i < iters; ++i) fn();
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / iters;
  };
  for (const Shape& sh : shapes) {
    const long long M = (long long)B * sh.hw * sh.hw;
    ConvArgs a{};
    a.in = x; a.in_ld = sh.K; a.in_dtype = JN_F32; a.itab = ChanTab{tab, tab + 2048, tab + 4096}; a.w = w; a.out = out; a.out_ld = sh.N;
    a.out_dtype = JN_F32; a.N = B; a.H = sh.hw; a.W = sh.hw; a.OH = sh.hw; a.OW = sh.hw; a.cin = sh.K; a.cout = sh.N; a.stride = 1;
    a.act = ACT_NONE;
    const double gf1 = 2.0 * M * sh.K * sh.N / 1e9;
    // backward data gradient: 20 steps per launch, transposed weights, no statistics
    ConvArgs b = a;
    b.w_transposed = 1; b.in_identity = std::getenv("TFON") ? 0 : 1; b.n_slots = S; b.in_slot_stride = M * sh.K; b.out_slot_stride = M * sh.N; b.tab_slot_stride = 0;
    printf("%2dx%2d K=%3d N=%3d |", sh.hw, sh.hw, sh.K, sh.N);
    for (int pf : {2, 3, 4}) {
      char v[8]; snprintf(v, sizeof v, "%d", pf); setenv("JN_PW_WT_PF", v, 1);   // (round 3: the library fixed the depth at 4; the switch is gone)
      // (launch_pw_dir reads the variable once: the first value wins inside one process, so run one pf per process)
      if (pf != (std::getenv("PF") ? atoi(std::getenv("PF")) : 3)) continue;
      if (launch_pw_dir(b, 0, 0, s) != 0 || hipDeviceSynchronize() != hipSuccess) { printf(" bwd pf%d n/a", pf); continue; }
      const float tt = time_it([&] { launch_pw_dir(b, 0, 0, s); });
      printf(" bwd x20 pf%d %7.1f us = %5.1f TF/s |", pf, tt, gf1 * S / tt * 1e-3 * 1e3);
    }
    // forward: one step, statistics on
    ConvArgs f = a;
    f.stats = stats; f.stats_rep_stride = 2 * 4096; f.stats_nrep = 8;
    if (launch_pw_dir(f, 0, 0, s) == 0 && hipDeviceSynchronize() == hipSuccess) {
      const float tt = time_it([&] { launch_pw_dir(f, 0, 0, s); });
      printf(" fwd x1 %6.1f us = %5.1f TF/s", tt, gf1 / tt * 1e-3 * 1e3);
    }
    printf("\n");
  }
  return 0;
}
