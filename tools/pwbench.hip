// Stand-alone timing of the 1x1-conv GEMM kernels on the shapes of the nano PAFPN (B = 64, 448 px): the resident-weight
// kernel (kernels_pwres.hip) in several tile configurations against pw_mfma_kernel (kernels_conv.hip) and a pure
// streaming kernel that moves the same bytes.  With -DJN_PWRES_STAMPS the resident kernel also records wall-clock stamps
// of its phases for a few workgroups.   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/pwbench.hip -o tools/pwbench
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../jolineedle_amd/csrc/kernels_conv.hip"
#include "../jolineedle_amd/csrc/kernels_pwres.hip"
#include "../jolineedle_amd/csrc/kernels_pwxs.hip"
#include "pwres_legacy.hip"

using namespace jnr;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void stream_kernel(const float4* __restrict__ in, float4* __restrict__ out, long long nin, long long nout) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x, st = (long long)gridDim.x * 256;
  float4 acc = {0, 0, 0, 0};
  for (long long j = i; j < nin; j += st) { const float4 v = in[j]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
  for (long long j = i; j < nout; j += st) out[j] = acc;
}

int main(int argc, char** argv) {
  struct Shape { int hw, K, N; };
  const Shape shapes[] = {{28, 128, 128}, {28, 64, 64}, {28, 256, 128}, {28, 64, 128}, {28, 128, 64}, {14, 128, 128}, {14, 256, 256},
                          {14, 512, 256}, {14, 256, 128}, {14, 128, 256}, {56, 128, 64}, {56, 64, 64}};
  const int B = 64, iters = 40;
  float *x, *w, *out, *tab; double* stats;
  const size_t maxx = (size_t)B * 56 * 56 * 512, maxo = (size_t)B * 56 * 56 * 256;
  CK(hipMalloc(&x, maxx * 4)); CK(hipMalloc(&out, maxo * 4)); CK(hipMalloc(&w, 512 * 512 * 4)); CK(hipMalloc(&tab, 3 * 2048 * 4));
  CK(hipMalloc(&stats, 32 * 2 * 4096 * 8));
  std::vector<float> h(maxx);
  for (size_t i = 0; i < maxx; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;
  CK(hipMemcpy(x, h.data(), maxx * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, h.data(), 512 * 512 * 4, hipMemcpyHostToDevice));
  std::vector<float> t(3 * 2048, 0.0f);
  for (int i = 0; i < 2048; ++i) { t[i] = 1.0f; t[4096 + i] = 1.0f; }
  CK(hipMemcpy(tab, t.data(), t.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(stats, 0, 32 * 2 * 4096 * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipStream_t s = nullptr;
  auto time_it = [&](auto&& fn) {
    for (int i = 0; i < 3; ++i) fn();
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) fn();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / iters;
  };
  for (const Shape& sh : shapes) {
    const long long M = (long long)B * sh.hw * sh.hw;
    ConvArgs a{};
    a.in = x; a.in_ld = sh.K; a.in_dtype = JN_F32; a.itab = ChanTab{tab, tab + 2048, tab + 4096}; a.w = w; a.out = out; a.out_ld = sh.N;
    a.out_dtype = JN_F32; a.N = B; a.H = sh.hw; a.W = sh.hw; a.OH = sh.hw; a.OW = sh.hw; a.cin = sh.K; a.cout = sh.N; a.stride = 1;
    a.act = ACT_NONE; a.stats = stats; a.stats_rep_stride = 2 * 4096; a.stats_nrep = 8;
    const double mb = (double)M * (sh.K + sh.N) * 4 / 1e6, gf = 2.0 * M * sh.K * sh.N / 1e9;
    const float t_stream = time_it([&] { hipLaunchKernelGGL(stream_kernel, dim3(2048), dim3(256), 0, s, (const float4*)x, (float4*)out, M * sh.K / 4, M * sh.N / 4); });
    setenv("JN_NO_PW_RES", "1", 1);
    // (pw_res_supported caches the env on first use: call the old path directly)
    const float t_old = time_it([&] { launch_pw_types<float, float, false>(a, s); });
    ConvArgs a2 = a; a2.stats = nullptr;
    const float t_old_ns = time_it([&] { launch_pw_types<float, float, false>(a2, s); });
    printf("%2dx%2d K=%3d N=%3d  %6.1f MB %5.2f GF | stream %6.1f us | pw_mfma %6.1f (no stats %6.1f)", sh.hw, sh.hw, sh.K, sh.N, mb, gf, t_stream, t_old, t_old_ns);
    const char* cfgs[] = {"0,0,0,2,0", "0,0,0,2,1"};
    for (const char* c : cfgs) {
      int ct, kc, bm, pd, sp; sscanf(c, "%d,%d,%d,%d,%d", &ct, &kc, &bm, &pd, &sp);
      if (ct && 16 * ct > sh.N) { printf(" | %s  skip", c); continue; }
      g_pw_res_force = c;
      hipError_t pre = hipGetLastError(); (void)pre;
      int rc = launch_pw_res(a, s);
      hipError_t e = hipDeviceSynchronize();
      if (rc != 0 || e != hipSuccess || hipGetLastError() != hipSuccess) { printf(" | %s  n/a", c); continue; }
      const float tt = time_it([&] { launch_pw_res(a, s); });
      printf(" | %s %6.1f", c, tt);
    }
    for (int sp = 0; sp < 2; ++sp)
      for (int ctw : {8, 4, 2}) {
        if (16 * ctw > sh.N) { printf(" | dir%d/%d skip", ctw, sp); continue; }
        int rc = launch_pw_dir(a, ctw, sp, s);
        hipError_t e = hipDeviceSynchronize();
        if (rc != 0 || e != hipSuccess || hipGetLastError() != hipSuccess) { printf(" | dir%d/%d  n/a", ctw, sp); continue; }
        const float tt = time_it([&] { launch_pw_dir(a, ctw, sp, s); });
        printf(" | dir%d/%d %6.1f", ctw, sp, tt);
      }
    for (int pf : {3, 4}) {
      g_pw_dir_pf = pf;
      if (launch_pw_dir(a, 4, 0, s) != 0 || hipDeviceSynchronize() != hipSuccess) { printf(" | dir4/0/pf%d n/a", pf); continue; }
      const float tt = time_it([&] { launch_pw_dir(a, 4, 0, s); });
      printf(" | dir4/0/pf%d %6.1f", pf, tt);
    }
    g_pw_dir_pf = 0;
    {
      std::vector<float> ref((size_t)M * sh.N), got((size_t)M * sh.N);
      launch_pw_types<float, float, false>(a2, s); CK(hipDeviceSynchronize());
      CK(hipMemcpy(ref.data(), out, ref.size() * 4, hipMemcpyDeviceToHost));
      for (int sp = 0; sp < 2; ++sp) {
        CK(hipMemset(out, 0, ref.size() * 4));
        if (launch_pw_dir(a2, 0, sp, s) != 0) { printf(" | dir-err[%d] n/a", sp); continue; }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
        double md = 0;
        for (size_t i = 0; i < ref.size(); ++i) md = std::max(md, (double)std::fabs(ref[i] - got[i]));
        printf(" | dir-err[%d] %.2e", sp, md);
      }
    }
    {   // accuracy of the automatic configurations against pw_mfma_kernel (exact fp32 fma chain)
      std::vector<float> ref((size_t)M * sh.N), got((size_t)M * sh.N);
      launch_pw_types<float, float, false>(a2, s); CK(hipDeviceSynchronize());
      CK(hipMemcpy(ref.data(), out, ref.size() * 4, hipMemcpyDeviceToHost));
      for (const char* c : {"0,0,0,2,0", "0,0,0,2,1"}) {
        g_pw_res_force = c;
        CK(hipMemset(out, 0, ref.size() * 4));
        launch_pw_res(a2, s); CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
        double md = 0, mr = 0;
        for (size_t i = 0; i < ref.size(); ++i) { md = std::max(md, (double)std::fabs(ref[i] - got[i])); mr = std::max(mr, (double)std::fabs(ref[i])); }
        printf(" | err[%s] %.2e (max |ref| %.2f)", c, md, mr);
      }
    }
    g_pw_res_force = nullptr;
    printf("\n");
    fflush(stdout);
  }
#ifdef JN_PWRES_STAMPS
  {   // phase stamps of the 28x28 128 -> 128 layer: entry, prologue issued, barrier, then per chunk (staged, barrier, MFMAs done[, tile stored])
    long long* dbg; const int NW = 256;
    CK(hipMalloc(&dbg, NW * 32 * 8));
    for (const char* c : {"4,0,32,2,0", "4,0,32,2,1", "8,0,32,2,1"}) {
      for (int hw : {28, 14}) {
        CK(hipMemset(dbg, 0, NW * 32 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_pwres_dbg), &dbg, sizeof(dbg)));
        ConvArgs a{};
        a.in = x; a.in_ld = 128; a.in_dtype = JN_F32; a.itab = ChanTab{tab, tab + 2048, tab + 4096}; a.w = w; a.out = out; a.out_ld = 128;
        a.out_dtype = JN_F32; a.N = B; a.H = hw; a.W = hw; a.OH = hw; a.OW = hw; a.cin = 128; a.cout = 128; a.stride = 1;
        a.act = ACT_NONE; a.stats = stats; a.stats_rep_stride = 2 * 4096; a.stats_nrep = 8;
        g_pw_res_force = c;
        launch_pw_res(a, s); CK(hipDeviceSynchronize());
        launch_pw_res(a, s); CK(hipDeviceSynchronize());
        std::vector<long long> hd(NW * 32);
        CK(hipMemcpy(hd.data(), dbg, NW * 32 * 8, hipMemcpyDeviceToHost));
        long long t0 = hd[0];
        for (int wg = 0; wg < NW; ++wg) if (hd[wg * 32] && hd[wg * 32] < t0) t0 = hd[wg * 32];
        printf("stamps cfg %s hw %d (us since the first workgroup's entry)\n", c, hw);
        for (int wg : {0, 1, 7, 64, 130, 200, 255}) {
          printf("  wg %3d:", wg);
          for (int i = 0; i < 32; ++i) if (hd[wg * 32 + i]) printf(" [%d]%.2f", i, (hd[wg * 32 + i] - t0) * 0.01);
          printf("\n");
        }
      }
    }
    long long* nul = nullptr;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_pwres_dbg), &nul, sizeof(nul)));
  }
#endif
  return 0;
}
