// The data gradient of the wide 1x1 layers in the step-batched backward (20 slots of 64 x 14 x 14 pixels), outside the
// engine: pw_dir_kernel<.., WT> on the fp32 pipe against pw_x3_kernel<.., SLOTS> on the bf16 pipe in several tilings.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/x3bwdbench.hip -o tools/x3bwdbench
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../jolineedle_amd/csrc/kernels_conv.hip"
#include "../jolineedle_amd/csrc/kernels_pwres.hip"
#include "../jolineedle_amd/csrc/kernels_pwxs.hip"

using namespace jnr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

int main() {
  const int S = 20, B = 64;
  struct Shape { int hw, cout, cin; };     // of the LAYER: the gradient conv has K = cout, N = cin
  const Shape shapes[] = {{14, 256, 256}, {14, 256, 128}, {14, 128, 256}, {28, 128, 256}};
  float *gz, *gx, *gx2, *w, *tab; void* w3t;
  const size_t maxe = (size_t)S * B * 28 * 28 * 256;
  CK(hipMalloc(&gz, maxe * 4)); CK(hipMalloc(&gx, maxe * 4)); CK(hipMalloc(&gx2, maxe * 4)); CK(hipMalloc(&w, 512 * 512 * 4));
  CK(hipMalloc(&w3t, 512 * 512 * 6)); CK(hipMalloc(&tab, 3 * 2048 * 4));
  std::vector<float> h((size_t)B * 28 * 28 * 256);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f - 0.5f;
  for (int sl = 0; sl < S; ++sl) CK(hipMemcpy(gz + (size_t)sl * h.size(), h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, h.data(), 512 * 512 * 4, hipMemcpyHostToDevice));
  std::vector<float> t(3 * 2048, 0.0f);
  for (int i = 0; i < 2048; ++i) t[i] = 1.0f;            // identity table: scale 1, shift 0, no activation
  CK(hipMemcpy(tab, t.data(), t.size() * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipStream_t s = nullptr;
  auto time_it = [&](auto&& fn) {
    for (int i = 0; i < 2; ++i) fn();
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 10; ++i) fn();
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3f / 10;
  };
  for (const Shape& sh : shapes) {
    const long long Ms = (long long)B * sh.hw * sh.hw;
    ConvArgs a{};
    a.in = gz; a.in_ld = sh.cout; a.in_dtype = JN_F32; a.itab = ChanTab{tab, tab + 2048, tab + 4096}; a.w = w;
    a.out = gx; a.out_ld = sh.cin; a.out_dtype = JN_F32; a.N = B; a.H = sh.hw; a.W = sh.hw; a.OH = sh.hw; a.OW = sh.hw;
    a.cin = sh.cout; a.cout = sh.cin; a.stride = 1; a.act = ACT_NONE; a.accumulate = 0; a.w_transposed = 1; a.in_identity = 1;
    a.n_slots = S; a.in_slot_stride = Ms * sh.cout; a.out_slot_stride = Ms * sh.cin;
    const double gf = 2.0 * S * Ms * sh.cout * sh.cin / 1e9, mb = (double)S * Ms * (sh.cout + sh.cin) * 4 / 1e6;
    printf("%2dx%2d x %d slots, layer %3d -> %3d: %6.1f MB, %5.1f GF (fp32 pipe floor %5.1f us, bf16 x 6 products %5.1f us)\n", sh.hw, sh.hw, S,
           sh.cin, sh.cout, mb, gf, gf / 157.3 * 1e3, 6 * gf / 2500.0 * 1e3);
    printf("    pw_dir_kernel<WT> (fp32 pipe): %7.1f us\n", time_it([&] { launch_pw(a, s); }));
    CK(hipDeviceSynchronize());
    std::vector<float> ref((size_t)Ms * sh.cin), got(ref.size());
    CK(hipMemcpy(ref.data(), (float*)gx + (size_t)(S - 1) * Ms * sh.cin, ref.size() * 4, hipMemcpyDeviceToHost));
    ConvArgs b = a; b.out = gx2; b.w_x3 = w3t;
    hipLaunchKernelGGL(w_split3_t_kernel, dim3((unsigned)((sh.cin * (sh.cout / 8) + 255) / 256)), dim3(256), 0, s, w, (bf16_t*)w3t, sh.cout, sh.cin);
    auto run = [&](const char* name, auto&& fn) {
      CK(hipMemsetAsync(gx2, 0, (size_t)S * Ms * sh.cin * 4, s));
      fn(); 
      hipError_t e = hipDeviceSynchronize();
      if (e != hipSuccess) { printf("    %s: %s\n", name, hipGetErrorString(e)); exit(1); }
      CK(hipMemcpy(got.data(), (float*)gx2 + (size_t)(S - 1) * Ms * sh.cin, got.size() * 4, hipMemcpyDeviceToHost));
      double md = 0, mx = 0;
      for (size_t i = 0; i < ref.size(); ++i) { md = std::max(md, (double)std::fabs(ref[i] - got[i])); mx = std::max(mx, (double)std::fabs(ref[i])); }
      printf("    %-44s: %7.1f us   (max |x3 - fp32| / max |fp32| %.1e)\n", name, time_it(fn), md / mx);
    };
    if (sh.cout == 256 && sh.cin == 256) {
      run("x3 <256, 4 ch tiles, 64 px, D 2> 1 wg/CU", [&] { launch_pw_x3_t<256, 4, 4, 2, 3, float, false, true>(b, Ms, 1, s); });
      run("x3 <256, 4 ch tiles, 64 px, D 4> 1 wg/CU", [&] { launch_pw_x3_t<256, 4, 4, 4, 3, float, false, true>(b, Ms, 1, s); });
      run("x3 <256, 4 ch tiles, 32 px, D 2> 2 wg/CU", [&] { launch_pw_x3_t<256, 4, 2, 2, 3, float, false, true>(b, Ms, 2, s); });
      run("x3 <256, 4 ch tiles, 32 px, D 4> 2 wg/CU", [&] { launch_pw_x3_t<256, 4, 2, 4, 3, float, false, true>(b, Ms, 2, s); });
      run("x3 <256, 2 ch tiles, 32 px, D 2> 2 wg/CU x 2 halves", [&] {
        for (int n0 = 0; n0 < 256; n0 += 128) {
          ConvArgs c = b; c.w_x3 = (const bf16_t*)w3t + (long long)n0 * 256 * 3; c.out = gx2 + n0;
          launch_pw_x3_t<256, 2, 2, 2, 3, float, false, true>(c, Ms, 2, s);
        } });
    } else if (sh.cout == 256) {
      run("x3 <256, 2 ch tiles, 32 px, D 2> 2 wg/CU", [&] { launch_pw_x3_t<256, 2, 2, 2, 3, float, false, true>(b, Ms, 2, s); });
      run("x3 <256, 2 ch tiles, 32 px, D 4> 2 wg/CU", [&] { launch_pw_x3_t<256, 2, 2, 4, 3, float, false, true>(b, Ms, 2, s); });
      run("x3 <256, 2 ch tiles, 64 px, D 2> 1 wg/CU", [&] { launch_pw_x3_t<256, 2, 4, 2, 3, float, false, true>(b, Ms, 1, s); });
    } else {
      run("x3 <128, 4 ch tiles, 64 px, D 2> 1 wg/CU", [&] { launch_pw_x3_t<128, 4, 4, 2, 3, float, false, true>(b, Ms, 1, s); });
      run("x3 <128, 4 ch tiles, 64 px, D 4> 1 wg/CU", [&] { launch_pw_x3_t<128, 4, 4, 4, 3, float, false, true>(b, Ms, 1, s); });
      run("x3 <128, 4 ch tiles, 32 px, D 4> 2 wg/CU", [&] { launch_pw_x3_t<128, 4, 2, 4, 3, float, false, true>(b, Ms, 2, s); });
      run("x3 <128, 4 ch tiles, 64 px, D 4> 2 wg/CU", [&] { launch_pw_x3_t<128, 4, 4, 4, 3, float, false, true>(b, Ms, 2, s); });
    }
    fflush(stdout);
  }
  return 0;
}
