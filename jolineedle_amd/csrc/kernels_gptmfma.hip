// The glimpse step of the decision model (kernels_gpt.hip) for WIDE models, batched over the agents on the matrix pipe —
// a measured "no" that stays in the tree, opt-in (JN_GPT_MFMA=1) and under test.
//
// gpt_step_kernel gives every agent a workgroup and runs each Linear as a split-K matrix-vector product on the VALU.  For
// gpt-mini (192 wide, 6 layers: 2.65 M weights = 10.6 MB per step) that is 0.34 ms per step alone on the chip, 0.5 ms inside
// the configs[4] iteration, and the north star asks for the attention / MLP GEMMs on MFMA.  Here ONE workgroup owns AG agents
// (4 or 16: the columns of v_mfma_f32_16x16x4_f32) and every Linear is a real GEMM  Y[agents][N] = X[agents][K] . Wt[K][N]:
//   * 16 waves; the first wave of an agent does its vector work (token embedding, LayerNorm, softmax, head, sampling, env
//     step), its 16 / AG waves share the attention products over its KV cache, all waves share the 64-column macro tiles;
//   * a macro tile is four MFMA tiles whose columns interleave (tile t = columns n0 + 4 i + t): a lane's ONE 16-byte load of
//     the transposed weight row feeds the A operands of all four, and its 16 results are 16 consecutive columns of its agent;
//   * layers with few macro tiles (N = C: 3 of them at C = 192) split K over waves, partials through free LDS, summed in
//     fixed order (no atomics: the logits do not depend on scheduling);
//   * residual stream, LN output, qkv, MLP activations and attention rows of the agents stay in LDS (9 C + 32 + heads x
//     Tmax floats per agent: 125 KB at C = 192, T = 32, 16 agents), rows padded by 4 floats (conflict-free B operands).
// Exact fp32 (fp32 MFMA, fp32 accumulate): logits within 1.2e-9 of gpt_step_kernel's; same arguments, modes and outputs.
//
// What the measurement says (tools/gptstepbench.hip with wall-clock stamps per phase, profiles/r04_gptstepbench*.txt):
//   * one step takes 446 us at 4 agents per workgroup and 568 us at 16, against 343 us of gpt_step_kernel (B = 16 and 64);
//   * the four GEMMs of a layer take 53 us, and they take 45 us when the whole model fits the L2 (one layer instead of six):
//     not weight streaming.  They are bound by the fp32 matrix pipe of ONE CU: 256 flop / clock / CU, 12 C^2 x 2 x 16 =
//     14.2 Mflop per layer and 16-agent tile = 23 us at peak; 9 - 12 busy waves of 16 reach 45 % of that.  gpt_step_kernel
//     spends the same flops on 16 CUs.  The matrix pipe only wins where one workgroup's tile is deep enough to need it: M = 16
//     agents is a matrix-VECTOR shape, and on MI355X the fp32 VALU peak equals the fp32 MFMA peak;
//   * what would beat gpt_step_kernel is the weights split over many CUs with a hand-off after every Linear (3 - 4 us each by
//     tools/gridsyncbench.hip, 5 per layer: the saving is spent), or the three-way bf16 split on the bf16 pipe (2.7 x the fp32
//     rate after six products; needs bf16 weight planes for the decision model): DESIGN.md section 10.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "jn_device.h"
#include "jn_types.h"
#include "jn_kernels.h"

namespace jnr {

namespace {

#ifdef JN_GM_STAMPS          // tools/gptstepbench.hip: wall-clock stamps (10 ns) of workgroup 0 at the phase boundaries of layer 0
__device__ long long gm_stamps[32];
#define GM_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) gm_stamps[i] = wall_clock64(); } while (0)
#define GM_STAMP0(i) do { if (l == 0) GM_STAMP(i); } while (0)
#else
#define GM_STAMP(i)
#define GM_STAMP0(i)
#endif
constexpr int GM_NT = 1024;                     // threads per workgroup

__device__ __forceinline__ float gm_gelu(float x) {   // NewGELU, src/models/gpt.py:37-47
  return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Y[agent][n] = act(bias[n] + sum_k X[agent][k] Wt[k][n]) for the workgroup's AG agents.  N % 64 == 0, K / KS % 16 == 0.
// KS K-slices (waves per macro tile); KS > 1: partials go to `scratch` [KS][AG][N] and are summed in slice order.
// MODE 0: store, 1: GELU then store.  Ends with a barrier: Y is complete.
// AG < 16: MFMA column j stands for agent j % AG (LDS broadcasts the duplicate reads), only columns < AG are stored.
template <int D>
__device__ __forceinline__ void gm_ksteps(f32x4 (&acc)[4], const float*& wp, const float*& xp, long long N) {
  f32x4 av[D]; float bv[D];
#pragma unroll
  for (int u = 0; u < D; ++u) { av[u] = *reinterpret_cast<const f32x4*>(wp + 4LL * u * N); bv[u] = xp[4 * u]; }
  wp += 4LL * D * N; xp += 4 * D;
#pragma unroll
  for (int u = 0; u < D; ++u)
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][t], bv[u], acc[t], 0, 0, 0);
}

template <int MODE, int AG>
__device__ __noinline__ void gm_gemm(float* __restrict__ Y, int ldy, const float* __restrict__ X, int ldx,
                                     const float* __restrict__ wt, const float* __restrict__ bias, int K, int N, int KS,
                                     float* __restrict__ scratch) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lm = lane & 15, g = lane >> 4;
  const int ag = lm & (AG - 1);
  const int n_mt = N >> 6;
  const int kper = K / KS;
  for (int job = wave; job < n_mt * KS; job += GM_NT / 64) {
    const int mt = job / KS, sl = job - mt * KS;
    const int n0 = mt << 6, kb = sl * kper;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* wp = wt + (long long)(kb + g) * N + n0 + 4 * lm;       // A: rows k, this lane's 4 interleaved columns
    const float* xp = X + ag * ldx + kb + g;                            // B: agent, k
    int rem = kper;
    for (; rem >= 48; rem -= 48) gm_ksteps<12>(acc, wp, xp, N);         // 12 independent 16-byte loads per lane in flight
    for (; rem >= 16; rem -= 16) gm_ksteps<4>(acc, wp, xp, N);
    // D_t[i = 4 g + r][agent lm] = column n0 + 4 (4 g + r) + t: the lane holds columns n0 + 16 g .. + 15 of agent lm
    if (lm < AG) {
      float* yr = (KS > 1 ? scratch + ((long long)sl * AG + lm) * N : Y + lm * ldy) + n0 + 16 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
        if (KS == 1) {
          const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + n0 + 16 * g + 4 * r) : f32x4{0.f, 0.f, 0.f, 0.f};
          v += bv;
          if (MODE == 1) { v[0] = gm_gelu(v[0]); v[1] = gm_gelu(v[1]); v[2] = gm_gelu(v[2]); v[3] = gm_gelu(v[3]); }
        }
        *reinterpret_cast<f32x4*>(yr + 4 * r) = v;
      }
    }
  }
  __syncthreads();
  if (KS > 1) {
    for (int i = tid; i < AG * N; i += GM_NT) {
      const int a_ = i / N, n = i - a_ * N;
      float v = bias ? bias[n] : 0.0f;
      for (int s = 0; s < KS; ++s) v += scratch[((long long)s * AG + a_) * N + n];
      if (MODE == 1) v = gm_gelu(v);
      Y[a_ * ldy + n] = v;
    }
    __syncthreads();
  }
}

// per-agent LayerNorm by ONE wave: y = (x - mean) * rstd * w + b
__device__ __forceinline__ void gm_layer_norm(float* y, const float* x, const float* __restrict__ w, const float* __restrict__ b,
                                              int C, int lane) {
  float s = 0.0f;
  for (int i = lane; i < C; i += 64) s += x[i];
  const float mean = wave_sum(s) / C;
  float q = 0.0f;
  for (int i = lane; i < C; i += 64) { const float d = x[i] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / C + 1e-5f);
  for (int i = lane; i < C; i += 64) y[i] = (x[i] - mean) * rstd * w[i] + b[i];
}

}  // namespace

// AG agents per workgroup, WPA = 16 / AG waves per agent: the agent's FIRST wave does its vector work (embedding, LayerNorm,
// residual adds, softmax, head, sampling, env step: every lane touches elements lane + 64 j only, so no barrier is needed
// inside such a chain), all WPA waves share its attention products, all 16 waves share the GEMMs.
template <int AG>
__global__ __launch_bounds__(GM_NT) void gpt_step_mfma_kernel(GptStepArgs a) {
  constexpr int WPA = GM_NT / 64 / AG, GN = WPA * 64;
  if (a.skip_flag && *a.skip_flag >= a.skip_when) {
    if (blockIdx.x == 0 && threadIdx.x == 0) a.n_done[a.step + 1] = a.skip_when;
    return;
  }
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int C = a.C, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int CP = C + 4, Q3 = 3 * C + 4, M4 = 4 * C + 4, AT = a.n_head * a.Tmax;
  float* X = sm;                    // [AG][CP]  residual stream
  float* H = X + AG * CP;           // [AG][CP]  LN output / attention output
  float* MLP = H + AG * CP;         // [AG][M4]  also the concatenated embedding parts and c_proj's split-K partials
  float* QKV = MLP + AG * M4;       // [AG][Q3]  with ATT behind it: the split-K partials of mlp.c_proj
  float* ATT = QKV + AG * Q3;       // [AG][n_head * Tmax]
  float* LG = ATT + AG * AT;        // [AG][16]  logits
  const int ks_fc2 = (4 * C <= Q3 + AT) ? 4 : 2;       // K slices of mlp.c_proj whose partials fit QKV + ATT
  const int al = wave / WPA, sub = wave - al * WPA, gt = sub * 64 + lane;   // agent of this wave, its thread within the agent
  const bool lead = sub == 0;
  const int b = blockIdx.x * AG + al;
  const bool live = b < a.B;
  const int bb = live ? b : 0;                          // padding agents of the last workgroup compute on agent 0's inputs
  float* x = X + al * CP; float* h = H + al * CP; float* qkv = QKV + al * Q3; float* mlp = MLP + al * M4;
  float* att = ATT + al * AT; float* lg = LG + al * 16;

  const int t = a.step;
  const int hs = C / a.n_head;
  const float scale = 1.0f / sqrtf((float)hs);
  int len = live ? a.cache_len[b] : 0;
  const int n_new = (a.src_mode == GPT_SRC_ENV && t == 0) ? 2 : 1;
  const bool drop = a.pdrop > 0.0f;

  GM_STAMP(0);
  for (int j = 0; j < n_new; ++j) {
    // ---------------- token embedding (per agent, by its first wave) ----------------
    const bool class_tok = (a.src_mode == GPT_SRC_CLASS) || (a.src_mode == GPT_SRC_ENV && t == 0 && j == 0);
    int p = 0;                                         // embedding parts (the same for every agent)
    if (class_tok) {
      if (lead) {
        const int cls = (live && a.classes) ? min(max((int)a.classes[b], 0), JN_N_CLASS_ROWS - 1) : 0;
        for (int i = lane; i < C; i += 64) x[i] = a.embed_class[(long long)cls * C + i];
      }
    } else if (a.src_mode == GPT_SRC_GIVEN) {
      if (lead) {
        const float* gp = a.given_emb + ((long long)bb * a.given_stride + a.given_index) * C;
        for (int i = lane; i < C; i += 64) x[i] = gp[i];
      }
    } else {
      p = 2 + (a.no_patch_emb ? 0 : 1) + (a.use_pos_emb ? 1 : 0);
      if (lead) {
        int act, row, col;
        if (a.src_mode == GPT_SRC_ENV) {
          act = (int)a.prev_action[bb];
          row = (int)a.env.positions[2 * bb]; col = (int)a.env.positions[2 * bb + 1];
        } else {
          const long long bi = (long long)bb * a.t_stride + a.t_index;
          act = (int)a.t_actions[bi];
          row = a.t_positions ? (int)a.t_positions[2 * bi] : 0;
          col = a.t_positions ? (int)a.t_positions[2 * bi + 1] : 0;
        }
        act = min(max(act, 0), a.nA - 1);
        row = min(max(row, 0), 255); col = min(max(col, 0), 255);
        float* parts = mlp;
        int q = 0;
        for (int i = lane; i < C; i += 64) parts[i] = a.wte[act * C + i];
        ++q;
        for (int i = lane; i < C; i += 64)
          parts[q * C + i] = a.dec_pos_enc ? a.pos1d[a.pos_index * C + i] : a.wpe[a.pos_index * C + i];
        ++q;
        if (!a.no_patch_emb) {
          if (a.src_mode == GPT_SRC_ENV) {
            for (int i = lane; i < C; i += 64) {
              float s = a.efpn_lin_b[i];
              for (int ks = 0; ks < a.KS; ++ks) s += a.emb_part[((long long)bb * a.KS + ks) * C + i];
              parts[q * C + i] = s;
              if (live && a.tok_emb_out) a.tok_emb_out[((long long)b * a.T + t) * C + i] = s;
            }
          } else {
            const float* pe = a.tok_emb + ((long long)bb * a.tok_emb_stride + a.tok_emb_index) * C;
            for (int i = lane; i < C; i += 64) parts[q * C + i] = pe[i];
          }
          ++q;
        }
        if (a.use_pos_emb) {
          for (int i = lane; i < C; i += 64)
            parts[q * C + i] = (i < a.pe2_ch) ? a.pe2[col * a.pe2_ch + i] : a.pe2[row * a.pe2_ch + (i - a.pe2_ch)];
          ++q;
        }
        if (!a.concat_emb)
          for (int i = lane; i < C; i += 64) {
            float s = 0.0f;
            for (int r = 0; r < q; ++r) s += parts[r * C + i];
            x[i] = s / q;
          }
      }
      if (a.concat_emb) {
        // project_concat (gpt.py:461-465): [AG][p C] -> [AG][C]; MLP holds the parts, the K slices' partials go to QKV
        // (free until the first block)
        __syncthreads();
        gm_gemm<0, AG>(X, CP, MLP, M4, a.proj_wt, a.proj_b, p * C, C, 2, QKV);
      }
    }
    if (lead) {
      if (live && a.out.final_emb)
        for (int i = lane; i < C; i += 64) a.out.final_emb[((long long)b * a.emb_stride + len) * C + i] = x[i];
      if (drop && !a.embed_only)                           // x = transformer.drop(final_emb), gpt.py:525
        for (int i = lane; i < C; i += 64) x[i] *= drop_scale(a.drop_seed, bb, len, 0, 0, i, a.pdrop);
    }
    if (a.embed_only) { ++len; __syncthreads(); continue; }

    // ---------------- transformer blocks ----------------
    for (int l = 0; l < a.n_layer; ++l) {
      const GptLayerPtrs L = a.layers[l];
      GM_STAMP0(1);
      if (lead) gm_layer_norm(h, x, L.ln1_w, L.ln1_b, C, lane);
      __syncthreads();
      GM_STAMP0(2);
      gm_gemm<0, AG>(QKV, Q3, H, CP, L.qkv_wt, L.qkv_b, C, 3 * C, 1, nullptr);
      GM_STAMP0(3);
      // ---- attention of this agent over its cache (by its WPA waves; the new token's k / v come from LDS) ----
      const int nk = len + 1;
      const float* kc = a.kcache + (((long long)l * a.B + bb) * a.Tmax) * C;
      const float* vc = a.vcache + (((long long)l * a.B + bb) * a.Tmax) * C;
      if (live) {
        float* kw = a.kcache + (((long long)l * a.B + b) * a.Tmax + len) * C;
        float* vw = a.vcache + (((long long)l * a.B + b) * a.Tmax + len) * C;
        for (int i = gt; i < C; i += GN) { kw[i] = qkv[C + i]; vw[i] = qkv[2 * C + i]; }
        for (int e = gt; e < a.n_head * nk; e += GN) {     // q . k: 16-byte loads along the head
          const int hd = e / nk, s = e - hd * nk;
          const float* qp = qkv + hd * hs;
          float d = 0.0f;
          if (s == len) {
            const float* kp = qkv + C + hd * hs;
            for (int i = 0; i < hs; ++i) d = fmaf(qp[i], kp[i], d);
          } else {
            const f32x4* kp = reinterpret_cast<const f32x4*>(kc + s * C + hd * hs);
            f32x4 d4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int i = 0; i < (hs >> 2); ++i) d4 += *reinterpret_cast<const f32x4*>(qp + 4 * i) * kp[i];
            d = (d4[0] + d4[1]) + (d4[2] + d4[3]);
          }
          att[hd * a.Tmax + s] = d * scale;
        }
      }
      __syncthreads();
      GM_STAMP0(4);
      if (live && lead) {                                  // softmax: 8 lanes per head
        const int hd = lane >> 3, j8 = lane & 7;
        float* ap = att + hd * a.Tmax;
        const bool on = hd < a.n_head;
        float m = -INFINITY;
        if (on) for (int s = j8; s < nk; s += 8) m = fmaxf(m, ap[s]);
        m = fmaxf(m, __shfl_xor(m, 1)); m = fmaxf(m, __shfl_xor(m, 2)); m = fmaxf(m, __shfl_xor(m, 4));
        float sum = 0.0f;
        if (on) for (int s = j8; s < nk; s += 8) { const float ev = expf(ap[s] - m); ap[s] = ev; sum += ev; }
        sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2); sum += __shfl_xor(sum, 4);
        const float inv = 1.0f / sum;
        if (on)
          for (int s = j8; s < nk; s += 8)                 // attn_dropout on the probabilities, gpt.py:100
            ap[s] *= drop ? inv * drop_scale(a.drop_seed, b, len, l, 1, hd * a.Tmax + s, a.pdrop) : inv;
      }
      __syncthreads();
      GM_STAMP0(5);
      if (live) {
        for (int i = gt; i < C; i += GN) {
          const float* ap = att + (i / hs) * a.Tmax;
          float acc = ap[len] * qkv[2 * C + i];
#pragma unroll 8
          for (int s = 0; s < len; ++s) acc = fmaf(ap[s], vc[s * C + i], acc);
          h[i] = acc;
        }
      } else {
        for (int i = gt; i < C; i += GN) h[i] = 0.0f;
      }
      __syncthreads();
      GM_STAMP0(6);
      // attn.c_proj: [AG][C] -> QKV[:, 0:C] (the agents' q: no longer needed); K slices through MLP
      gm_gemm<0, AG>(QKV, Q3, H, CP, L.proj_wt, L.proj_b, C, C, 4, MLP);
      GM_STAMP0(7);
      if (lead) {
        for (int i = lane; i < C; i += 64)
          x[i] += drop ? qkv[i] * drop_scale(a.drop_seed, bb, len, l, 2, i, a.pdrop) : qkv[i];
        gm_layer_norm(h, x, L.ln2_w, L.ln2_b, C, lane);
      }
      __syncthreads();
      GM_STAMP0(8);
      gm_gemm<1, AG>(MLP, M4, H, CP, L.fc_wt, L.fc_b, C, 4 * C, 1, nullptr);            // + NewGELU
      GM_STAMP0(9);
      gm_gemm<0, AG>(H, CP, MLP, M4, L.fc2_wt, L.fc2_b, 4 * C, C, ks_fc2, QKV);         // partials through QKV (+ ATT)
      GM_STAMP0(10);
      if (lead)
        for (int i = lane; i < C; i += 64)
          x[i] += drop ? h[i] * drop_scale(a.drop_seed, bb, len, l, 3, i, a.pdrop) : h[i];
    }
    ++len;
    __syncthreads();
  }
  GM_STAMP(11);

  if (!lead) return;                                       // no barrier below
  if (a.embed_only) {
    if (live && lane == 0) a.cache_len[b] = len;
    return;
  }
  // ---------------- head on the newest token (per agent, by its first wave) ----------------
  gm_layer_norm(h, x, a.lnf_w, a.lnf_b, C, lane);
  {
    // logits[n] = sum_k h[k] head_wt[k][n]: lane = (k slice of 4, column n < 16), reduced over the slices by shuffles
    const int n = lane & 15, ks = lane >> 4;
    float acc = 0.0f;
    if (n < a.nA)
      for (int k = ks; k < C; k += 4) acc = fmaf(h[k], a.head_wt[(long long)k * a.nA + n], acc);
    acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    if (lane < 16) lg[lane] = acc;
  }
  if (!live) return;
  if (a.src_mode != GPT_SRC_ENV) {
    if (lane == 0) a.cache_len[b] = len;
    if (a.logits_rows && lane < a.nA) a.logits_rows[(long long)b * a.logits_stride + lane] = lg[lane];
    GM_STAMP(12);
    return;
  }
  if (lane == 0) {
    a.cache_len[b] = len;
    const int nA = a.nA;
    float m = -INFINITY;
    int best = 0;
    for (int i = 0; i < nA; ++i)
      if (lg[i] > m) { m = lg[i]; best = i; }           // first maximum, as torch.argmax
    float sum = 0.0f;
    for (int i = 0; i < nA; ++i) sum += expf(lg[i] - m);
    const float lse = m + logf(sum);
    float ent = 0.0f;
    for (int i = 0; i < nA; ++i) { const float lp = lg[i] - lse; ent -= lp * expf(lp); }
    int act = best;
    if (a.mode == JN_MODE_FORCED) {
      act = (int)a.forced[(long long)b * a.T + t];
      act = min(max(act, 0), nA - 1);
    } else if (a.mode == JN_MODE_SAMPLE) {
      const uint4 r = philox4x32(a.seed, (uint32_t)b, (uint32_t)t, 0x53414d50u, 0u);
      const float u = u01(r.x);
      float cdf = 0.0f;
      act = nA - 1;
      for (int i = 0; i < nA; ++i) {
        cdf += expf(lg[i] - lse);
        if (u < cdf) { act = i; break; }
      }
    }
    const float logp = lg[act] - lse;
    const EnvStepResult r = env_step_one(a.env, b, act);
    a.prev_action[b] = act;
    const long long bt = (long long)b * a.T + t;
    if (a.out.rewards) a.out.rewards[bt] = r.reward;
    if (a.out.logprobs) a.out.logprobs[bt] = logp;
    if (a.out.entropies) a.out.entropies[bt] = ent;
    if (a.out.actions) a.out.actions[bt] = act;
    if (a.out.masks) a.out.masks[(long long)b * (a.T + 1) + t + 1] = r.terminated ? 0 : 1;
    if (a.out.positions) {
      a.out.positions[((long long)b * (a.T + 1) + t + 1) * 2] = r.y;
      a.out.positions[((long long)b * (a.T + 1) + t + 1) * 2 + 1] = r.x;
    }
    if (a.out.logits)
      for (int i = 0; i < nA; ++i) a.out.logits[bt * nA + i] = lg[i];
    if (r.terminated || r.truncated) atomicAdd(a.n_done + t + 1, 1);
  }
}

template <int AG>
static void gm_launch(const GptStepArgs& a, hipStream_t s) {
  const size_t smem = (size_t)AG * ((size_t)9 * a.C + 16 + (size_t)a.n_head * a.Tmax + 16) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gpt_step_mfma_kernel<AG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(gpt_step_mfma_kernel<AG>, dim3((a.B + AG - 1) / AG), dim3(GM_NT), smem, s, a);
}

// true when the kernel took the launch.  OPT-IN (JN_GPT_MFMA=1; JN_GPT_MFMA_AGENTS=4|16 picks the agents per workgroup,
// default 4 up to 64 agents): it is measured SLOWER than gpt_step_kernel at configs[4] (tools/gptstepbench.hip,
// profiles/r04_gptstepbench.txt: 446 us against 343 us per step at 4 agents per workgroup, 568 us at 16) and stays in the
// tree as the tested record of why — see the header.  Needs n_embd a multiple of 64 (macro tiles, 16-divisible K slices),
// heads of a multiple of 4 channels (16-byte key loads), at most 8 heads (8 softmax lanes each) and 16 actions.
bool launch_gpt_step_mfma(const GptStepArgs& a, hipStream_t s) {
  const char* on = std::getenv("JN_GPT_MFMA");
  if (!on || on[0] != '1') return false;
  const char* force = std::getenv("JN_GPT_MFMA_AGENTS");
  if (a.C % 64 != 0 || a.nA > 16 || a.n_head > 8 || a.C % a.n_head != 0 || (a.C / a.n_head) % 4 != 0) return false;
  int ag = a.B <= 64 ? 4 : 16;
  if (force) ag = std::atoi(force) == 16 ? 16 : 4;
  if ((size_t)ag * ((size_t)9 * a.C + 32 + (size_t)a.n_head * a.Tmax) * sizeof(float) > 160 * 1024) return false;
  if (ag == 4) gm_launch<4>(a, s); else gm_launch<16>(a, s);
  return true;
}

}  // namespace jnr
