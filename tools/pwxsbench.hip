// Stand-alone timing of the pixel-stationary 1x1 kernel (kernels_pwxs.hip) against the weight-stationary pw_dir_kernel and
// the generic pw_mfma_kernel on the small-map shapes of the nano PAFPN (B = 64, 448 px), with a plain table and with a
// deferred (consumer-side BatchNorm) table as the train-mode pass uses it; results are compared bit for bit.
// Round 3: the same shapes on the bf16 pipe with three-way split operands (pw_x3_kernel), timing and error against fp64.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/pwxsbench.hip -o tools/pwxsbench
#define JN_X3_ALL_SHAPES
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../jolineedle_amd/csrc/kernels_conv.hip"
#include "../jolineedle_amd/csrc/kernels_pwres.hip"
#include "../jolineedle_amd/csrc/kernels_pwxs.hip"

using namespace jnr;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void null_kernel(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
// one dependent global round trip + one barrier per workgroup: the shape of every kernel's prologue
__global__ void touch_kernel(const float* __restrict__ in, float* __restrict__ out) {
  __shared__ float s;
  if (threadIdx.x == 0) s = in[blockIdx.x * 64];
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x * 64] = s + 1.0f;
}

int main(int argc, char** argv) {
  struct Shape { int hw, K, N; };
  const Shape shapes[] = {{28, 64, 64}, {28, 64, 128}, {28, 128, 64}, {28, 128, 128}, {28, 256, 128}, {14, 128, 128}, {14, 128, 256},
                          {14, 256, 128}, {14, 256, 256}, {14, 512, 256}, {56, 64, 64}, {56, 128, 64}};
  const int B = 64, iters = 50;
  float *x, *w, *out, *out2, *tab, *params; double *stats, *dstats; void* w3;
  const size_t maxx = (size_t)B * 56 * 56 * 128 + (size_t)B * 14 * 14 * 512, maxo = (size_t)B * 56 * 56 * 64 + (size_t)B * 28 * 28 * 256;
  CK(hipMalloc(&x, maxx * 4)); CK(hipMalloc(&out, maxo * 4)); CK(hipMalloc(&out2, maxo * 4)); CK(hipMalloc(&w, 512 * 512 * 4)); CK(hipMalloc(&w3, 512 * 512 * 6));
  CK(hipMalloc(&tab, 3 * 2048 * 4)); CK(hipMalloc(&params, 4096 * 4));
  CK(hipMalloc(&stats, 32 * 2 * 4096 * 8)); CK(hipMalloc(&dstats, 8 * 2 * 4096 * 8));
  std::vector<float> h(maxx);
  for (size_t i = 0; i < maxx; ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;
  CK(hipMemcpy(x, h.data(), maxx * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, h.data(), 512 * 512 * 4, hipMemcpyHostToDevice));
  launch_w_split3(w, w3, 512 * 512, nullptr); CK(hipDeviceSynchronize());
  std::vector<float> t(3 * 2048, 0.0f);
  for (int i = 0; i < 2048; ++i) { t[i] = 1.0f + 0.001f * (i % 7); t[2048 + i] = 0.01f * (i % 5); t[4096 + i] = 1.0f; }
  CK(hipMemcpy(tab, t.data(), t.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> pr(4096);
  for (int i = 0; i < 4096; ++i) pr[i] = i < 2048 ? 1.0f + 0.01f * (i % 3) : 0.02f * (i % 4);    // gamma | beta
  CK(hipMemcpy(params, pr.data(), pr.size() * 4, hipMemcpyHostToDevice));
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
  // floor of a dependent launch on this box: in-order stream, each kernel waits for the previous one (the forward pass
  // is ~85 such launches); "touch" adds what every real kernel has at least — one global round trip and a barrier
  for (int wgs : {196, 784}) {
    const float t_null = time_it([&] { hipLaunchKernelGGL(null_kernel, dim3(wgs), dim3(256), 0, s, (int*)nullptr); });
    const float t_touch = time_it([&] { hipLaunchKernelGGL(touch_kernel, dim3(wgs), dim3(256), 0, s, (const float*)x, out); });
    printf("launch floor, %d workgroups: empty kernel %.2f us per dependent launch, one load + barrier + store %.2f us\n", wgs, t_null, t_touch);
  }
  for (const Shape& sh : shapes) {
    const long long M = (long long)B * sh.hw * sh.hw;
    // deferred table: channel c of the input = BatchNorm channel c of a producer with the same pixel count; sums that
    // give mean 0.1 and variance ~1
    std::vector<double> ds(8 * 2 * 4096, 0.0);
    for (int r = 0; r < 8; ++r)
      for (int c = 0; c < sh.K; ++c) { ds[(size_t)r * 2 * 4096 + 2 * c] = 0.1 * M / 8.0; ds[(size_t)r * 2 * 4096 + 2 * c + 1] = (1.0 + 0.01 * (c % 9)) * M / 8.0; }
    CK(hipMemcpy(dstats, ds.data(), ds.size() * 8, hipMemcpyHostToDevice));
    ChanTab plain{tab, tab + 2048, tab + 4096};
    ChanTab defer = plain;
    defer.dparams = params; defer.dstats = dstats; defer.drep_stride = 2 * 4096; defer.dN = B; defer.nseg = 1;
    defer.r0 = ChanTab::Run{0, sh.K, 0, 0, 2048, (float)(sh.hw * sh.hw)};
    ConvArgs a{};
    a.in = x; a.in_ld = sh.K; a.in_dtype = JN_F32; a.itab = plain; a.w = w; a.out = out; a.out_ld = sh.N;
    a.out_dtype = JN_F32; a.N = B; a.H = sh.hw; a.W = sh.hw; a.OH = sh.hw; a.OW = sh.hw; a.cin = sh.K; a.cout = sh.N; a.stride = 1;
    a.w_x3 = w3; a.act = ACT_NONE; a.stats = stats; a.stats_rep_stride = 2 * 4096; a.stats_nrep = 8;
    const double mb = (double)M * (sh.K + sh.N) * 4 / 1e6, gf = 2.0 * M * sh.K * sh.N / 1e9;
    printf("%2dx%2d K=%3d N=%3d %6.1f MB %5.2f GF (fp32 MFMA floor %5.1f us)", sh.hw, sh.hw, sh.K, sh.N, mb, gf, gf / 122.0 * 1e3);
    for (int dt = 0; dt < 2; ++dt) {
      a.itab = dt ? defer : plain;
      printf(" | %s:", dt ? "deferred" : "plain");
      const float t_old = time_it([&] { launch_pw_types<float, float, false>(a, s); });
      printf(" mfma %5.1f", t_old);
      if (launch_pw_dir(a, 0, 0, s) == 0 && hipDeviceSynchronize() == hipSuccess) printf(" dir %5.1f", time_it([&] { launch_pw_dir(a, 0, 0, s); }));
      for (int pt : {2, 4})
        for (int wg : {1, 2, 3}) {
          if (dt == 0 && wg != 2) continue;
          if (launch_pw_xs(a, pt, s, wg) != 0) { printf(" xs%d n/a", pt); continue; }
          hipError_t e = hipDeviceSynchronize();
          if (e != hipSuccess) { printf(" xs%d %s\n", pt, hipGetErrorString(e)); return 1; }
          printf(" xs%d/%d %5.1f", pt, wg, time_it([&] { launch_pw_xs(a, pt, s, wg); }));
        }
      for (int pt : {2, 4})
        for (int wg : {1, 2, 3}) {
          if (dt == 0 && wg != 2) continue;
          if (sh.K == 512 && pt == 4) continue;
          if (launch_pw_x3(a, pt, s, wg) != 0) { printf(" x3 n/a"); continue; }
          hipError_t e = hipDeviceSynchronize();
          if (e != hipSuccess) { printf(" x3_%d %s\n", pt, hipGetErrorString(e)); return 1; }
          printf(" x3_%d/%d %5.1f", pt, wg, time_it([&] { launch_pw_x3(a, pt, s, wg); }));
        }
    }
    // correctness: outputs bit for bit against pw_mfma_kernel (deferred table), statistics to fp64 rounding
    a.itab = defer;
    std::vector<float> ref((size_t)M * sh.N), got((size_t)M * sh.N);
    std::vector<double> sref(8 * 2 * 4096), sgot(8 * 2 * 4096);
    CK(hipMemset(stats, 0, 32 * 2 * 4096 * 8));
    launch_pw_types<float, float, false>(a, s); CK(hipDeviceSynchronize());
    CK(hipMemcpy(ref.data(), out, ref.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(sref.data(), stats, sref.size() * 8, hipMemcpyDeviceToHost));
    for (int pt : {2, 4}) {
      ConvArgs b = a; b.out = out2;
      CK(hipMemset(out2, 0, ref.size() * 4)); CK(hipMemset(stats, 0, 32 * 2 * 4096 * 8));
      if (launch_pw_xs(b, pt, s) != 0) continue;
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(got.data(), out2, got.size() * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(sgot.data(), stats, sgot.size() * 8, hipMemcpyDeviceToHost));
      double md = 0, sd = 0, sm = 0;
      for (size_t i = 0; i < ref.size(); ++i) md = std::max(md, (double)std::fabs(ref[i] - got[i]));
      for (int c = 0; c < 2 * sh.N; ++c) {
        double ar = 0, ag = 0;
        for (int r = 0; r < 8; ++r) { ar += sref[(size_t)r * 2 * 4096 + c]; ag += sgot[(size_t)r * 2 * 4096 + c]; }
        sd = std::max(sd, std::fabs(ar - ag)); sm = std::max(sm, std::fabs(ar));
      }
      printf(" | xs%d err %.1e stats %.1e/%.1e", pt, md, sd, sm);
    }
    // accuracy of both pipes against an fp64 sum of the SAME transformed operand (plain table: T is the identity flag 1 ->
    // silu(sc z + sh) computed on the host in fp64 from the fp32 inputs): 4096 sampled outputs
    {
      a.itab = plain;
      std::vector<float> o32((size_t)M * sh.N), o3((size_t)M * sh.N);
      ConvArgs b = a; b.stats = nullptr;
      b.out = out; launch_pw_xs(b, 0, s); CK(hipDeviceSynchronize());
      CK(hipMemcpy(o32.data(), out, o32.size() * 4, hipMemcpyDeviceToHost));
      b.out = out2; CK(hipMemset(out2, 0, o3.size() * 4));
      if (launch_pw_x3(b, 0, s) == 0) {
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(o3.data(), out2, o3.size() * 4, hipMemcpyDeviceToHost));
        double e32 = 0, e3 = 0, d = 0, nrm = 0; 
        for (int smp = 0; smp < 4096; ++smp) {
          const long long m = ((long long)smp * 2654435761LL) % M; const int n = (smp * 37) % sh.N;
          double ref = 0, absum = 0;
          for (int k = 0; k < sh.K; ++k) {
            // the kernel's own fp32 transform, so that only the contraction differs
            const float zf = h[(size_t)m * sh.K + k];
            const float y = fmaf(zf, t[k], t[2048 + k]);
            const float av = y * (1.0f / (1.0f + expf(-y)));
            ref += (double)av * (double)h[(size_t)n * sh.K + k]; absum += std::fabs((double)av * (double)h[(size_t)n * sh.K + k]);
          }
          e32 = std::max(e32, std::fabs(o32[(size_t)m * sh.N + n] - ref) / absum);
          e3 = std::max(e3, std::fabs(o3[(size_t)m * sh.N + n] - ref) / absum);
          d = std::max(d, std::fabs((double)o3[(size_t)m * sh.N + n] - o32[(size_t)m * sh.N + n]) / absum); nrm += 1;
        }
        printf(" | vs fp64 / sum|ab|: fp32 pipe %.1e, x3 %.1e, x3 - fp32 %.1e", e32, e3, d);
      }
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
