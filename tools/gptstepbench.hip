// One decision step of gpt-mini (192 wide, 6 layers, 6 heads) for B agents mid-trajectory, outside the engine: the VALU
// kernel (one workgroup per agent) against the MFMA kernel (4 / 16 agents per workgroup), with the wall-clock stamps of the
// MFMA kernel's phases in layer 0.  Weights are constants: only the timing means anything.
//   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DJN_GM_STAMPS tools/gptstepbench.hip -o tools/gptstepbench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../jolineedle_amd/csrc/kernels_gpt.hip"
#include "../jolineedle_amd/csrc/kernels_gptmfma.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
using namespace jnr;

int main(int argc, char** argv) {
  const int C = argc > 1 ? atoi(argv[1]) : 192, L = argc > 2 ? atoi(argv[2]) : 6, NH = argc > 3 ? atoi(argv[3]) : 6, nA = 5, Tmax = 33, len0 = 16;
  for (int B : {16, 64}) {
    const size_t per_layer = (size_t)12 * C * C + 13 * C;
    const size_t n_w = L * per_layer + (size_t)JN_N_CLASS_ROWS * C + (size_t)C * nA + 4 * C;
    float* w; CK(hipMalloc(&w, n_w * 4));
    std::vector<float> hw(n_w);
    for (size_t i = 0; i < n_w; ++i) hw[i] = 0.02f * (float)((int)((i * 2654435761u >> 10) & 0xff) - 128) / 128.0f;
    CK(hipMemcpy(w, hw.data(), n_w * 4, hipMemcpyHostToDevice));
    std::vector<GptLayerPtrs> hl(L);
    float* p = w;
    auto take = [&](size_t n) { float* r = p; p += n; return r; };
    for (int l = 0; l < L; ++l) {
      hl[l].ln1_w = take(C); hl[l].ln1_b = take(C); hl[l].qkv_wt = take((size_t)3 * C * C); hl[l].qkv_b = take(3 * C);
      hl[l].proj_wt = take((size_t)C * C); hl[l].proj_b = take(C); hl[l].ln2_w = take(C); hl[l].ln2_b = take(C);
      hl[l].fc_wt = take((size_t)4 * C * C); hl[l].fc_b = take(4 * C); hl[l].fc2_wt = take((size_t)4 * C * C); hl[l].fc2_b = take(C);
    }
    GptLayerPtrs* dl; CK(hipMalloc(&dl, L * sizeof(GptLayerPtrs)));
    CK(hipMemcpy(dl, hl.data(), L * sizeof(GptLayerPtrs), hipMemcpyHostToDevice));
    GptStepArgs a{};
    a.C = C; a.n_head = NH; a.n_layer = L; a.nA = nA; a.Tmax = Tmax; a.B = B; a.T = 32;
    a.embed_class = take((size_t)JN_N_CLASS_ROWS * C); a.head_wt = take((size_t)C * nA); a.lnf_w = take(C); a.lnf_b = take(C);
    a.layers = dl;
    const size_t kv = (size_t)L * B * Tmax * C;
    CK(hipMalloc(&a.kcache, kv * 4)); CK(hipMalloc(&a.vcache, kv * 4));
    CK(hipMemset(a.kcache, 0, kv * 4)); CK(hipMemset(a.vcache, 0, kv * 4));
    CK(hipMalloc(&a.cache_len, B * 4));
    std::vector<int32_t> hlen(B, len0);
    a.src_mode = GPT_SRC_CLASS; a.emb_stride = Tmax;
    CK(hipMalloc(&a.logits_rows, (size_t)B * nA * 4)); a.logits_stride = nA;
    std::vector<float> ref((size_t)B * nA), got((size_t)B * nA);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < (C % 64 == 0 ? 3 : 1); ++variant) {
      float best = 1e30f;
      for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemcpy(a.cache_len, hlen.data(), B * 4, hipMemcpyHostToDevice));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        if (variant == 0) {
          const int nt = C <= 64 ? 256 : 1024;
          const size_t smem = (size_t)(9 * C + NH * Tmax + nt / 64 + 16 + 4 + 4 * nt) * sizeof(float);
          if (nt == 256) hipLaunchKernelGGL(gpt_step_kernel<256>, dim3(B), dim3(256), smem, nullptr, a);
          else hipLaunchKernelGGL(gpt_step_kernel<1024>, dim3(B), dim3(1024), smem, nullptr, a);
        } else if (variant == 1) {
          gm_launch<4>(a, nullptr);
        } else {
          gm_launch<16>(a, nullptr);
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      CK(hipMemcpy((variant == 0 ? ref : got).data(), a.logits_rows, ref.size() * 4, hipMemcpyDeviceToHost));
      double worst = 0.0;
      if (variant) for (size_t i = 0; i < ref.size(); ++i) worst = fmax(worst, fabs((double)ref[i] - got[i]));
      printf("n_embd %d, B = %2d, %-34s: %7.1f us per step   (max |logit - VALU kernel's| %.1e)\n", C, B,
             variant == 0 ? "gpt_step_kernel (VALU)" : variant == 1 ? "MFMA, 4 agents per workgroup" : "MFMA, 16 agents per workgroup",
             best * 1e3f, worst);
      if (variant) {
        long long st[32];
        CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(gm_stamps), sizeof(st)));
        const char* names[] = {"embedding", "ln_1", "c_attn GEMM", "k/v store + q.k", "softmax", "att.v", "c_proj GEMM", "residual + ln_2",
                               "c_fc GEMM + GELU", "mlp.c_proj GEMM", "layers 1.. + residual", "ln_f + head"};
        printf("    layer 0 of workgroup 0 [us]:");
        for (int i = 0; i < 12; ++i) printf(" %s %.1f |", names[i], (st[i + 1] - st[i]) * 0.01);
        printf("\n");
      }
    }
    CK(hipFree(w)); CK(hipFree(dl)); CK(hipFree(a.kcache)); CK(hipFree(a.vcache)); CK(hipFree(a.cache_len)); CK(hipFree(a.logits_rows));
  }
  return 0;
}
