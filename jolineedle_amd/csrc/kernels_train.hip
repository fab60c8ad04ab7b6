// Training kernels of the decision side: REINFORCE loss gradient, teacher-forced backward of
// the GPT blocks over each agent's whole trajectory, token-embedding / embed_fpn backward,
// and the fused clip + AdamW update over the flat parameter arena.
//
// The reference back-propagates through T growing prefixes (src/reinforce.py:150-153, 341);
// with a causal mask and dropout 0 that is one causal pass over the T+1 token embeddings of
// each agent, which is what gpt_backward_kernel recomputes and differentiates.
#include <hip/hip_runtime.h>

#include "jn_device.h"
#include "jn_kernels.h"
#include "jn_types.h"

namespace jnr {

constexpr int TB = 256;
// gpt_backward_kernel: one workgroup per agent, so the only parallelism inside an agent is the workgroup's width; its
// phases are short loops over <= L x 4C elements with a K-long inner product each (1024 threads: 3.4 -> see DESIGN)
constexpr int GB = 1024;

// ---- REINFORCE loss (src/reinforce.py:217-265) and d loss / d logits --------------------------
// loss = -sum(logp*adv*m)/sum(m) + w * (-sum(H*m)/sum(m)),  adv = (returns - mean)/(std + 1e-8)
// dlogits[b,t,j] = scale * m/sum(m) * ( -adv*(1[j==a] - p_j) + w * p_j*(log p_j + H) )
__global__ __launch_bounds__(TB) void reinforce_loss_kernel(LossArgs a) {
  __shared__ float red[TB];
  __shared__ float tot[4];
  const int tid = threadIdx.x;
  int S = a.T;
  if (a.stop_early)
    for (int t = 1; t <= a.T; ++t)
      if (a.n_done[t] >= a.B) { S = t; break; }
  float cnt = 0.0f;
  for (int i = tid; i < a.B * S; i += TB) {
    const int b = i / S, t = i - b * S;
    cnt += a.logit_masks[(long long)b * a.T + t] ? 1.0f : 0.0f;
  }
  red[tid] = cnt;
  __syncthreads();
  for (int o = TB / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  const float msum = red[0];
  __syncthreads();
  float la = 0.0f, le = 0.0f, lr = 0.0f;
  for (int i = tid; i < a.B * a.T; i += TB) {
    const int b = i / a.T, t = i - b * a.T;
    float* dl = a.dlogits + (long long)i * a.nA;
    const bool m = t < S && a.logit_masks[i];
    if (!m) { for (int j = 0; j < a.nA; ++j) dl[j] = 0.0f; continue; }
    const float* lg = a.logits + (long long)i * a.nA;
    float mx = -INFINITY;
    for (int j = 0; j < a.nA; ++j) mx = fmaxf(mx, lg[j]);
    float se = 0.0f;
    for (int j = 0; j < a.nA; ++j) se += expf(lg[j] - mx);
    const float lse = mx + logf(se);
    float H = 0.0f;
    for (int j = 0; j < a.nA; ++j) { const float lp = lg[j] - lse; H -= lp * expf(lp); }
    const int act = (int)a.actions[i];
    const float adv = a.reward_norm ? (a.returns[i] - a.ret_mean) / (a.ret_std + 1e-8f) : a.returns[i];
    const float k = a.scale / msum;
    for (int j = 0; j < a.nA; ++j) {
      const float lp = lg[j] - lse, p = expf(lp);
      dl[j] = k * (-adv * ((j == act ? 1.0f : 0.0f) - p) + a.entropy_weight * p * (lp + H));
    }
    la -= (lg[act] - lse) * adv;
    le -= H;
    lr += a.rewards[i];
  }
  red[tid] = la; __syncthreads();
  for (int o = TB / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) tot[0] = red[0];
  __syncthreads();
  red[tid] = le; __syncthreads();
  for (int o = TB / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) tot[1] = red[0];
  __syncthreads();
  red[tid] = lr; __syncthreads();
  for (int o = TB / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) {
    a.metrics[0] = tot[0] / msum;                                  // action_loss
    a.metrics[1] = tot[1] / msum;                                  // entropy_loss
    a.metrics[2] = a.metrics[0] + a.entropy_weight * a.metrics[1]; // loss
    a.metrics[3] = red[0] / a.B;                                   // returns (mean masked reward sum)
    a.metrics[4] = msum / a.B;                                     // episode_length
    a.metrics[5] = (float)S;
  }
}

int launch_reinforce_loss(const LossArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(reinforce_loss_kernel, dim3(1), dim3(TB), 0, s, a);
  return 0;
}

// ---- autograd bridge: d loss / d logits from the upstream gradients of the rollout's log-probs and entropies --------
// logp = log_softmax(logits), lp = logp[action], H = -sum p logp  (src/reinforce.py:73-90):
//   d lp / d logits_j = [j == action] - p_j          d H / d logits_j = -p_j (logp_j + H)
// Steps the rollout did not execute (t >= S) get zero rows.
__global__ __launch_bounds__(TB) void logits_grad_kernel(const float* __restrict__ logits, const long long* __restrict__ actions,
                                                         const float* __restrict__ dlogprobs, const float* __restrict__ dentropies,
                                                         const int* __restrict__ n_done, float* __restrict__ dlogits, int B,
                                                         int T, int nA, int stop_early) {
  int S = T;
  if (stop_early)
    for (int t = 1; t <= T; ++t)
      if (n_done[t] >= B) { S = t; break; }
  const int i = blockIdx.x * TB + threadIdx.x;
  if (i >= B * T) return;
  const int t = i % T;
  float* dl = dlogits + (long long)i * nA;
  const float glp = dlogprobs ? dlogprobs[i] : 0.0f, gh = dentropies ? dentropies[i] : 0.0f;
  if (t >= S || (glp == 0.0f && gh == 0.0f)) { for (int j = 0; j < nA; ++j) dl[j] = 0.0f; return; }
  const float* lg = logits + (long long)i * nA;
  float mx = -INFINITY;
  for (int j = 0; j < nA; ++j) mx = fmaxf(mx, lg[j]);
  float se = 0.0f;
  for (int j = 0; j < nA; ++j) se += expf(lg[j] - mx);
  const float lse = mx + logf(se);
  float H = 0.0f;
  for (int j = 0; j < nA; ++j) { const float lp = lg[j] - lse; H -= lp * expf(lp); }
  const int act = (int)actions[i];
  for (int j = 0; j < nA; ++j) {
    const float lp = lg[j] - lse, pj = expf(lp);
    dl[j] = glp * ((j == act ? 1.0f : 0.0f) - pj) - gh * pj * (lp + H);
  }
}

int launch_logits_grad(const float* logits, const int64_t* actions, const float* dlogprobs, const float* dentropies,
                       const int32_t* n_done, float* dlogits, int B, int T, int nA, int stop_early, hipStream_t s) {
  hipLaunchKernelGGL(logits_grad_kernel, dim3((B * T + TB - 1) / TB), dim3(TB), 0, s, logits, (const long long*)actions, dlogprobs,
                     dentropies, n_done, dlogits, B, T, nA, stop_early);
  return 0;
}

// ---- packed arena <-> reference (PyTorch) layout, on the device --------------------------------------------------
// One thread per element of the reference-layout buffer (same segment offsets as the arena): finds its segment by
// bisection and maps the index by the segment's packing (api.hip: PackKind / unpack_param).
__device__ __forceinline__ long long packed_index(const ArenaSeg& g, long long r) {
  switch (g.kind) {
    case 1: { const long long o = r / g.d1, i = r - o * g.d1; return i * g.d0 + o; }                       // PK_T [out][in] -> [in][out]
    case 2: {                                                                                               // PK_STEM
      const int kx = (int)(r % 3), ky = (int)((r / 3) % 3), ic = (int)((r / 9) % 12), oc = (int)(r / 108);
      const int q = ic / 3, c = ic - 3 * q, py = q & 1, px = q >> 1;
      return (long long)((c * 6 + 2 * ky + py) * 6 + 2 * kx + px) * g.d0 + oc;
    }
    case 3: { const long long c = r / 9, k = r - c * 9; return k * g.d0 + c; }                              // PK_DW [c][tap] -> [tap][c]
    case 4: { const long long tp = r % 9, k = (r / 9) % g.d1, o = r / (9LL * g.d1); return (tp * g.d0 + o) * g.d1 + k; }   // PK_CONV3
    case 5: {                                                                                               // PK_EFPN_LIN
      const long long per_o = (long long)g.d1 * g.d2, o = r / per_o, rem = r - o * per_o, ch = rem / g.d1, q = rem - ch * g.d1;
      return (q * g.d2 + ch) * g.d0 + o;
    }
    default: return r;
  }
}

__global__ __launch_bounds__(256) void arena_copy_kernel(const ArenaSeg* __restrict__ segs, int n_segs, float* __restrict__ arena,
                                                         float* __restrict__ ref, long long total, int to_ref, int accumulate) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  int lo = 0, hi = n_segs - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (segs[mid].off <= i) lo = mid; else hi = mid - 1; }
  const ArenaSeg g = segs[lo];
  const long long r = i - g.off;
  if (r >= g.numel) return;                              // padding between segments
  const long long pi = g.off + packed_index(g, r);
  if (to_ref) ref[i] = accumulate ? ref[i] + arena[pi] : arena[pi];
  else arena[pi] = ref[i];
}

int launch_arena_copy(const ArenaSeg* segs, int n_segs, float* arena, float* ref, long long total, int to_ref, int accumulate,
                      hipStream_t s) {
  hipLaunchKernelGGL(arena_copy_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, segs, n_segs, arena, ref, total,
                     to_ref, accumulate);
  return 0;
}

// ---- supervised loss (src/supervised.py:138-177): CrossEntropyLoss(weight, reduction="none") averaged over
// the non-padding tokens; dlogits = w[y] * (softmax - onehot) / n_valid --------------------------------
__global__ __launch_bounds__(TB) void ce_loss_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                     const unsigned char* __restrict__ masks, float stop_weight,
                                                     float* __restrict__ dlogits, float* __restrict__ metrics, int n,
                                                     int nA, int T) {
  __shared__ float red[TB];
  __shared__ float tot[3];
  const int tid = threadIdx.x;
  float cnt = 0.0f;
  for (int i = tid; i < n; i += TB) cnt += masks[i] ? 1.0f : 0.0f;
  red[tid] = cnt; __syncthreads();
  for (int o = TB / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  const float nvalid = red[0];
  __syncthreads();
  float ls = 0.0f, acc = 0.0f;
  for (int i = tid; i < n; i += TB) {
    float* dl = dlogits + (long long)i * nA;
    if (!masks[i]) { for (int j = 0; j < nA; ++j) dl[j] = 0.0f; continue; }
    const float* lg = logits + (long long)i * nA;
    float mx = -INFINITY; int best = 0;
    for (int j = 0; j < nA; ++j) if (lg[j] > mx) { mx = lg[j]; best = j; }
    float se = 0.0f;
    for (int j = 0; j < nA; ++j) se += expf(lg[j] - mx);
    const float lse = mx + logf(se);
    int y = (int)target[i];
    y = min(max(y, 0), nA - 1);
    const float wy = (y == 8) ? stop_weight : 1.0f;
    for (int j = 0; j < nA; ++j) dl[j] = wy * (expf(lg[j] - lse) - (j == y ? 1.0f : 0.0f)) / nvalid;
    ls += -wy * (lg[y] - lse);
    acc += (best == y) ? 1.0f : 0.0f;
  }
  red[tid] = ls; __syncthreads();
  for (int o = TB / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) tot[0] = red[0];
  __syncthreads();
  red[tid] = acc; __syncthreads();
  for (int o = TB / 2; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
  if (tid == 0) {
    metrics[0] = tot[0] / nvalid;            // action_loss (= loss without the detector term)
    metrics[1] = red[0] / nvalid;            // action_accuracy
    metrics[2] = nvalid / (n / T);           // episode_length
  }
}

int launch_ce_loss(const float* logits, const int64_t* target, const uint8_t* masks, float stop_weight, float* dlogits,
                   float* metrics, int n, int nA, int T, hipStream_t s) {
  hipLaunchKernelGGL(ce_loss_kernel, dim3(1), dim3(TB), 0, s, logits, (const long long*)target, masks, stop_weight, dlogits,
                     metrics, n, nA, T);
  return 0;
}

// ---- helpers on per-agent global scratch (all threads of the block, barrier at the end) --------
__device__ __forceinline__ void lin_fwd(float* out, const float* in, const float* __restrict__ wt,
                                        const float* __restrict__ b, int L, int K, int N) {
  for (int e = threadIdx.x; e < L * N; e += GB) {
    const int i = e / N, n = e - i * N;
    float acc = b ? b[n] : 0.0f;
    const float* ip = in + i * K;
    for (int k = 0; k < K; ++k) acc = fmaf(ip[k], wt[(long long)k * N + n], acc);
    out[e] = acc;
  }
  __syncthreads();
}
// din[i][k] = sum_n dout[i][n] * wt[k][n]
__device__ __forceinline__ void lin_bwd_data(float* din, const float* dout, const float* __restrict__ wt, int L, int K,
                                             int N, bool accumulate) {
  for (int e = threadIdx.x; e < L * K; e += GB) {
    const int i = e / K, k = e - i * K;
    float acc = 0.0f;
    const float* dp = dout + i * N;
    const float* wp = wt + (long long)k * N;
    for (int n = 0; n < N; ++n) acc = fmaf(dp[n], wp[n], acc);
    din[e] = accumulate ? din[e] + acc : acc;
  }
  __syncthreads();
}
// gwt[k][n] += sum_i in[i][k] * dout[i][n];  gb[n] += sum_i dout[i][n]
__device__ __forceinline__ void lin_bwd_weight(float* __restrict__ gwt, float* __restrict__ gb, const float* in,
                                               const float* dout, int L, int K, int N, int i0) {
  for (int e = threadIdx.x; e < K * N; e += GB) {
    const int k = e / N, n = e - k * N;
    float acc = 0.0f;
    for (int i = i0; i < L; ++i) acc = fmaf(in[i * K + k], dout[i * N + n], acc);
    atomicAdd(&gwt[e], acc);
  }
  if (gb)
    for (int n = threadIdx.x; n < N; n += GB) {
      float acc = 0.0f;
      for (int i = i0; i < L; ++i) acc += dout[i * N + n];
      atomicAdd(&gb[n], acc);
    }
  __syncthreads();
}
__device__ __forceinline__ void ln_fwd(float* out, const float* in, const float* __restrict__ w,
                                       const float* __restrict__ b, float* mu, float* rs, int L, int C) {
  for (int i = threadIdx.x; i < L; i += GB) {
    const float* x = in + i * C;
    float m = 0.0f;
    for (int c = 0; c < C; ++c) m += x[c];
    m /= C;
    float v = 0.0f;
    for (int c = 0; c < C; ++c) { const float d = x[c] - m; v += d * d; }
    mu[i] = m;
    rs[i] = 1.0f / sqrtf(v / C + 1e-5f);
  }
  __syncthreads();
  for (int e = threadIdx.x; e < L * C; e += GB) {
    const int i = e / C, c = e - i * C;
    out[e] = (in[e] - mu[i]) * rs[i] * w[c] + b[c];
  }
  __syncthreads();
}
// din (+)= LN^T(dout);  gw += sum dout*xhat, gb += sum dout
__device__ __forceinline__ void ln_bwd(float* din, const float* dout, const float* in, const float* __restrict__ w,
                                       float* __restrict__ gw, float* __restrict__ gb, const float* mu, const float* rs,
                                       int L, int C, bool accumulate, int i0) {
  for (int c = threadIdx.x; c < C; c += GB) {
    float a = 0.0f, b = 0.0f;
    for (int i = i0; i < L; ++i) { const float d = dout[i * C + c]; a += d * (in[i * C + c] - mu[i]) * rs[i]; b += d; }
    atomicAdd(&gw[c], a);
    atomicAdd(&gb[c], b);
  }
  for (int i = threadIdx.x; i < L; i += GB) {
    const float* x = in + i * C;
    const float* d = dout + i * C;
    float m1 = 0.0f, m2 = 0.0f;
    for (int c = 0; c < C; ++c) { const float dx = d[c] * w[c]; m1 += dx; m2 += dx * (x[c] - mu[i]) * rs[i]; }
    m1 /= C; m2 /= C;
    for (int c = 0; c < C; ++c) {
      const float xh = (x[c] - mu[i]) * rs[i];
      const float v = rs[i] * (d[c] * w[c] - m1 - xh * m2);
      din[i * C + c] = accumulate ? din[i * C + c] + v : v;
    }
  }
  __syncthreads();
}

__device__ __forceinline__ float gelu_f(float x) {
  return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float gelu_d(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  const float th = tanhf(u);
  return 0.5f * (1.0f + th) + 0.5f * x * (1.0f - th * th) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x * x);
}

// ---- GPT backward, one workgroup per agent ------------------------------------------------------
__global__ __launch_bounds__(GB) void gpt_backward_kernel(GptBwdArgs a) {
  __shared__ float mu[64], rs[64];
  __shared__ int s_S;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int C = a.C, nh = a.n_head, hs = C / nh, nL = a.n_layer;
  if (tid == 0) {
    int S = a.T;
    if (a.stop_early)
      for (int t = 1; t <= a.T; ++t)
        if (a.n_done[t] >= a.B) { S = t; break; }
    s_S = S;
  }
  __syncthreads();
  const int S = s_S, L = S + 1;
  const float scale = 1.0f / sqrtf((float)hs);
  float* sc = a.scratch + (long long)b * a.scratch_per_agent;
  float* X = sc;                                   // [(nL+1)][L][C]
  float* lay = X + (nL + 1) * L * C;               // per layer block
  const int lay_sz = L * 11 * C + nh * L * L;
  float* tmp = lay + nL * lay_sz;                  // backward temporaries
  float* dX = tmp;                 // [L][C]
  float* dXM = dX + L * C;         // [L][C]
  float* dH = dXM + L * C;         // [L][C]
  float* dQKV = dH + L * C;        // [L][3C]
  float* dF = dQKV + L * 3 * C;    // [L][4C]
  float* dY = dF + L * 4 * C;      // [L][C]
  float* dP = dY + L * C;          // [nh][L][L]
  float* HF = dP + nh * L * L;     // [L][C]   ln_f output
  float* PARTS = HF + L * C;       // [4C]

  // ---------------- forward recompute ----------------
  const bool drop = a.pdrop > 0.0f;
  const float pd = a.pdrop;
  const uint64_t dseed = a.drop_seed;
  for (int e = tid; e < L * C; e += GB) {
    const float v = a.final_emb[((long long)b * (a.T + 1)) * C + e];
    X[e] = drop ? v * drop_scale(dseed, b, e / C, 0, 0, e % C, pd) : v;
  }
  __syncthreads();
  for (int l = 0; l < nL; ++l) {
    const GptLayerPtrs W = a.layers[l];
    float* x = X + l * L * C;
    float* H1 = lay + l * lay_sz;
    float* QKV = H1 + L * C;
    float* ATT = QKV + L * 3 * C;
    float* Y = ATT + nh * L * L;
    float* XM = Y + L * C;
    float* H2 = XM + L * C;
    float* Fp = H2 + L * C;         // [L][4C] pre-activation
    ln_fwd(H1, x, W.ln1_w, W.ln1_b, mu, rs, L, C);
    lin_fwd(QKV, H1, W.qkv_wt, W.qkv_b, L, C, 3 * C);
    for (int e = tid; e < nh * L * L; e += GB) {
      const int h = e / (L * L), i = (e / L) % L, j = e % L;
      float d = -INFINITY;
      if (j <= i) {
        d = 0.0f;
        for (int q = 0; q < hs; ++q) d = fmaf(QKV[i * 3 * C + h * hs + q], QKV[j * 3 * C + C + h * hs + q], d);
        d *= scale;
      }
      ATT[e] = d;
    }
    __syncthreads();
    for (int e = tid; e < nh * L; e += GB) {
      float* row = ATT + e * L;
      const int i = e % L;
      float m = -INFINITY;
      for (int j = 0; j <= i; ++j) m = fmaxf(m, row[j]);
      float s = 0.0f;
      for (int j = 0; j <= i; ++j) { row[j] = expf(row[j] - m); s += row[j]; }
      for (int j = 0; j < L; ++j) row[j] = j <= i ? row[j] / s : 0.0f;
    }
    __syncthreads();
    for (int e = tid; e < L * C; e += GB) {
      const int i = e / C, c = e - i * C, h = c / hs;
      float acc = 0.0f;
      for (int j = 0; j <= i; ++j) {
        float pij = ATT[(h * L + i) * L + j];
        if (drop) pij *= drop_scale(dseed, b, i, l, 1, h * a.Tmax + j, pd);
        acc = fmaf(pij, QKV[j * 3 * C + 2 * C + c], acc);
      }
      Y[e] = acc;
    }
    __syncthreads();
    lin_fwd(XM, Y, W.proj_wt, W.proj_b, L, C, C);
    for (int e = tid; e < L * C; e += GB) XM[e] = x[e] + (drop ? XM[e] * drop_scale(dseed, b, e / C, l, 2, e % C, pd) : XM[e]);
    __syncthreads();
    ln_fwd(H2, XM, W.ln2_w, W.ln2_b, mu, rs, L, C);
    lin_fwd(Fp, H2, W.fc_wt, W.fc_b, L, C, 4 * C);
    float* xo = X + (l + 1) * L * C;
    // mlp out: needs gelu(F); use dF as a temporary activation buffer
    for (int e = tid; e < L * 4 * C; e += GB) dF[e] = gelu_f(Fp[e]);
    __syncthreads();
    lin_fwd(xo, dF, W.fc2_wt, W.fc2_b, L, 4 * C, C);
    for (int e = tid; e < L * C; e += GB) xo[e] = XM[e] + (drop ? xo[e] * drop_scale(dseed, b, e / C, l, 3, e % C, pd) : xo[e]);
    __syncthreads();
  }
  float* xl = X + nL * L * C;
  ln_fwd(HF, xl, a.lnf_w, a.lnf_b, mu, rs, L, C);

  // ---------------- head + ln_f backward ----------------
  // dHF[i][c] = sum_a dlogits[b][i-1][a] * head_wt[c][a]   (token 0 has no logits)
  for (int e = tid; e < L * C; e += GB) {
    const int i = e / C, c = e - i * C;
    float acc = 0.0f;
    if (i >= 1) {
      const float* dl = a.dlogits + ((long long)b * a.T + (i - 1)) * a.nA;
      for (int q = 0; q < a.nA; ++q) acc = fmaf(dl[q], a.head_wt[c * a.nA + q], acc);
    }
    dH[e] = acc;
  }
  for (int e = tid; e < C * a.nA; e += GB) {
    const int c = e / a.nA, q = e - c * a.nA;
    float acc = 0.0f;
    for (int i = 1; i < L; ++i) acc = fmaf(HF[i * C + c], a.dlogits[((long long)b * a.T + (i - 1)) * a.nA + q], acc);
    atomicAdd(&a.g_head_wt[e], acc);
  }
  __syncthreads();
  ln_bwd(dX, dH, xl, a.lnf_w, a.g_lnf_w, a.g_lnf_b, mu, rs, L, C, false, 0);

  // ---------------- blocks, last to first ----------------
  for (int l = nL - 1; l >= 0; --l) {
    const GptLayerPtrs W = a.layers[l];
    const GptLayerPtrs G = a.g_layers[l];
    float* x = X + l * L * C;
    float* H1 = lay + l * lay_sz;
    float* QKV = H1 + L * C;
    float* ATT = QKV + L * 3 * C;
    float* Y = ATT + nh * L * L;
    float* XM = Y + L * C;
    float* H2 = XM + L * C;
    float* Fp = H2 + L * C;
    // mlp: xo = XM + drop(fc2(gelu(fc(H2)))): dO = gradient at the fc2 output (dY is free here)
    const float* dO = dX;
    if (drop) {
      for (int e = tid; e < L * C; e += GB) dY[e] = dX[e] * drop_scale(dseed, b, e / C, l, 3, e % C, pd);
      __syncthreads();
      dO = dY;
    }
    lin_bwd_data(dF, dO, W.fc2_wt, L, 4 * C, C, false);                     // dA
    // weight grad of fc2 needs gelu(F): recompute into dQKV/dY-sized temp is too small -> use dP? no: reuse Y? keep simple:
    for (int e = tid; e < 4 * C * C; e += GB) {
      const int k = e / C, c = e - k * C;
      float acc = 0.0f;
      for (int i = 0; i < L; ++i) acc = fmaf(gelu_f(Fp[i * 4 * C + k]), dO[i * C + c], acc);
      atomicAdd(&G.fc2_wt[e], acc);
    }
    for (int c = tid; c < C; c += GB) {
      float acc = 0.0f;
      for (int i = 0; i < L; ++i) acc += dO[i * C + c];
      atomicAdd(&G.fc2_b[c], acc);
    }
    for (int e = tid; e < L * 4 * C; e += GB) dF[e] *= gelu_d(Fp[e]);
    __syncthreads();
    lin_bwd_data(dH, dF, W.fc_wt, L, C, 4 * C, false);
    lin_bwd_weight(G.fc_wt, G.fc_b, H2, dF, L, C, 4 * C, 0);
    // ln2 statistics of XM
    for (int i = tid; i < L; i += GB) {
      const float* xx = XM + i * C;
      float m = 0.0f;
      for (int c = 0; c < C; ++c) m += xx[c];
      m /= C;
      float v = 0.0f;
      for (int c = 0; c < C; ++c) { const float d = xx[c] - m; v += d * d; }
      mu[i] = m; rs[i] = 1.0f / sqrtf(v / C + 1e-5f);
    }
    __syncthreads();
    for (int e = tid; e < L * C; e += GB) dXM[e] = dX[e];
    __syncthreads();
    ln_bwd(dXM, dH, XM, W.ln2_w, G.ln2_w, G.ln2_b, mu, rs, L, C, true, 0);
    // attention output projection: XM = x + drop(proj(Y)); dPo = gradient at the proj output (dH is free here)
    const float* dPo = dXM;
    if (drop) {
      for (int e = tid; e < L * C; e += GB) dH[e] = dXM[e] * drop_scale(dseed, b, e / C, l, 2, e % C, pd);
      __syncthreads();
      dPo = dH;
    }
    lin_bwd_data(dY, dPo, W.proj_wt, L, C, C, false);
    lin_bwd_weight(G.proj_wt, G.proj_b, Y, dPo, L, C, C, 0);
    // dP[h][i][j] = sum_d dY[i][h,d] * v[j][h,d]
    for (int e = tid; e < nh * L * L; e += GB) {
      const int h = e / (L * L), i = (e / L) % L, j = e % L;
      float acc = 0.0f;
      if (j <= i) {
        for (int q = 0; q < hs; ++q) acc = fmaf(dY[i * C + h * hs + q], QKV[j * 3 * C + 2 * C + h * hs + q], acc);
        if (drop) acc *= drop_scale(dseed, b, i, l, 1, h * a.Tmax + j, pd);     // through attn_dropout
      }
      dP[e] = acc;
    }
    __syncthreads();
    // dV[j][c] = sum_{i>=j} P[h][i][j] * dY[i][c]
    for (int e = tid; e < L * C; e += GB) {
      const int j = e / C, c = e - j * C, h = c / hs;
      float acc = 0.0f;
      for (int i = j; i < L; ++i) {
        float pij = ATT[(h * L + i) * L + j];
        if (drop) pij *= drop_scale(dseed, b, i, l, 1, h * a.Tmax + j, pd);
        acc = fmaf(pij, dY[i * C + c], acc);
      }
      dQKV[j * 3 * C + 2 * C + c] = acc;
    }
    // softmax backward in place: dS = P * (dP - sum_j P*dP)
    for (int e = tid; e < nh * L; e += GB) {
      float* dp = dP + e * L;
      const float* pr = ATT + e * L;
      const int i = e % L;
      float dot = 0.0f;
      for (int j = 0; j <= i; ++j) dot = fmaf(pr[j], dp[j], dot);
      for (int j = 0; j < L; ++j) dp[j] = j <= i ? pr[j] * (dp[j] - dot) * scale : 0.0f;
    }
    __syncthreads();
    for (int e = tid; e < L * C; e += GB) {
      const int i = e / C, c = e - i * C, h = c / hs;
      float dq = 0.0f, dk = 0.0f;
      for (int j = 0; j <= i; ++j) dq = fmaf(dP[(h * L + i) * L + j], QKV[j * 3 * C + C + c], dq);
      for (int ii = i; ii < L; ++ii) dk = fmaf(dP[(h * L + ii) * L + i], QKV[ii * 3 * C + c], dk);
      dQKV[i * 3 * C + c] = dq;
      dQKV[i * 3 * C + C + c] = dk;
    }
    __syncthreads();
    lin_bwd_data(dH, dQKV, W.qkv_wt, L, C, 3 * C, false);
    lin_bwd_weight(G.qkv_wt, G.qkv_b, H1, dQKV, L, C, 3 * C, 0);
    for (int i = tid; i < L; i += GB) {
      const float* xx = x + i * C;
      float m = 0.0f;
      for (int c = 0; c < C; ++c) m += xx[c];
      m /= C;
      float v = 0.0f;
      for (int c = 0; c < C; ++c) { const float d = xx[c] - m; v += d * d; }
      mu[i] = m; rs[i] = 1.0f / sqrtf(v / C + 1e-5f);
    }
    __syncthreads();
    for (int e = tid; e < L * C; e += GB) dX[e] = dXM[e];
    __syncthreads();
    ln_bwd(dX, dH, x, W.ln1_w, G.ln1_w, G.ln1_b, mu, rs, L, C, true, 0);
  }

  // ---------------- token embeddings ----------------
  if (drop) {                                           // through transformer.drop(final_emb)
    for (int e = tid; e < L * C; e += GB) dX[e] *= drop_scale(dseed, b, e / C, 0, 0, e % C, pd);
    __syncthreads();
  }
  {                                                                                 // class token: row classes[b]
    const int cls = a.classes ? min(max((int)a.classes[b], 0), JN_N_CLASS_ROWS - 1) : 0;
    for (int c = tid; c < C; c += GB) atomicAdd(&a.g_embed_class[(long long)cls * C + c], dX[c]);
  }
  for (int i = 1; i < L; ++i) {
    const int t = i - 1;
    // rollout: token i carries the action taken BEFORE its patch (BOS = 0); teacher-forced full sequence:
    // current_actions[b][t] (src/supervised.py:863-868)
    const int act = a.tok_actions ? (int)a.tok_actions[(long long)b * a.T + t]
                                  : ((i == 1) ? 0 : (int)a.actions[(long long)b * a.T + (i - 2)]);
    const int row = (int)a.positions[((long long)b * a.pos_tokens + t) * 2];
    const int col = (int)a.positions[((long long)b * a.pos_tokens + t) * 2 + 1];
    int p = 0;
    for (int c = tid; c < C; c += GB) PARTS[c] = a.wte[act * C + c];
    ++p;
    const int p1 = a.pos1d_by_token ? t : 0;            // recurrent tokens: 1-D position 0 (gpt.py:431-449)
    for (int c = tid; c < C; c += GB) PARTS[p * C + c] = a.dec_pos_enc ? a.pos1d[p1 * C + c] : a.wpe[p1 * C + c];
    const int p_pos = p; ++p;
    int p_patch = -1, p_pos2 = -1;
    if (!a.no_patch_emb) {
      p_patch = p;
      for (int c = tid; c < C; c += GB) PARTS[p * C + c] = a.tok_emb[((long long)b * a.T + t) * C + c];
      ++p;
    }
    if (a.use_pos_emb) {
      p_pos2 = p;
      for (int c = tid; c < C; c += GB)
        PARTS[p * C + c] = (c < a.pe2_ch) ? a.pe2[col * a.pe2_ch + c] : a.pe2[row * a.pe2_ch + (c - a.pe2_ch)];
      ++p;
    }
    __syncthreads();
    const float* dx = dX + i * C;
    float* dparts = dF;   // [p*C] scratch (dF holds >= 4C floats and is free now)
    if (a.concat_emb) {
      for (int k = tid; k < p * C; k += GB) {
        float acc = 0.0f;
        for (int c = 0; c < C; ++c) acc = fmaf(dx[c], a.proj_wt[(long long)k * C + c], acc);
        dparts[k] = acc;
      }
      for (int e = tid; e < p * C * C; e += GB) atomicAdd(&a.g_proj_wt[e], PARTS[e / C] * dx[e % C]);
      for (int c = tid; c < C; c += GB) atomicAdd(&a.g_proj_b[c], dx[c]);
    } else {
      for (int k = tid; k < p * C; k += GB) dparts[k] = dx[k % C] / p;
    }
    __syncthreads();
    for (int c = tid; c < C; c += GB) atomicAdd(&a.g_wte[act * C + c], dparts[c]);
    if (!a.dec_pos_enc && a.g_wpe)
      for (int c = tid; c < C; c += GB) atomicAdd(&a.g_wpe[p1 * C + c], dparts[p_pos * C + c]);
    if (p_patch >= 0)
      for (int c = tid; c < C; c += GB) a.d_tok_emb[((long long)b * a.dte_stride_b + t * a.dte_stride_t) * C + c] = dparts[p_patch * C + c];
    (void)p_pos2;
    __syncthreads();
  }
  // steps that were never executed get a zero patch-embedding gradient
  for (int t = S; t < a.T; ++t)
    for (int c = tid; c < C; c += GB) a.d_tok_emb[((long long)b * a.dte_stride_b + t * a.dte_stride_t) * C + c] = 0.0f;
}

int launch_gpt_backward(const GptBwdArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(gpt_backward_kernel, dim3(a.B), dim3(GB), 0, s, a);
  return 0;
}

__global__ __launch_bounds__(TB) void relu_mask_kernel(float* __restrict__ de, const float* __restrict__ e, long long n4) {
  const long long i = (long long)blockIdx.x * TB + threadIdx.x;
  if (i >= n4) return;
  f32x4 d = reinterpret_cast<f32x4*>(de)[i];
  const f32x4 ev = reinterpret_cast<const f32x4*>(e)[i];
#pragma unroll
  for (int q = 0; q < 4; ++q) if (ev[q] <= 0.0f) d[q] = 0.0f;
  reinterpret_cast<f32x4*>(de)[i] = d;
}

int launch_relu_mask(float* de, const float* e, long long n, hipStream_t s) {
  const long long n4 = n / 4;       // n = rows * h*w*C with C % 4 == 0
  hipLaunchKernelGGL(relu_mask_kernel, dim3((unsigned)((n4 + TB - 1) / TB)), dim3(TB), 0, s, de, e, n4);
  return 0;
}

__global__ __launch_bounds__(TB) void colsum_add_kernel(const float* __restrict__ dpe, long long M, int Co,
                                                        float* __restrict__ gb) {
  __shared__ float part[TB];
  const int o = blockIdx.x;
  float acc = 0.0f;
  for (long long m = threadIdx.x; m < M; m += TB) acc += dpe[m * Co + o];
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int st = TB / 2; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) part[threadIdx.x] += part[threadIdx.x + st];
    __syncthreads();
  }
  if (threadIdx.x == 0) gb[o] += part[0];
}

int launch_colsum_add(const float* dpe, long long M, int Co, float* gb, hipStream_t s) {
  hipLaunchKernelGGL(colsum_add_kernel, dim3(Co), dim3(TB), 0, s, dpe, M, Co, gb);
  return 0;
}

// ---- clip_grad_value_(1) + AdamW (src/reinforce.py:344-346; torch defaults betas .9/.999, eps 1e-8,
// weight_decay .01) over the flat arena; grad_scale folds the 1/world_size of the all-reduce ----
__global__ __launch_bounds__(TB) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long long n, float lr,
                                                   float beta1, float beta2, float eps, float wd, float bc1, float bc2,
                                                   float clip, float grad_scale) {
  const long long i = (long long)blockIdx.x * TB + threadIdx.x;
  if (i >= n) return;
  float gr = g[i] * grad_scale;
  if (clip > 0.0f) gr = fminf(fmaxf(gr, -clip), clip);
  float pv = p[i];
  pv *= 1.0f - lr * wd;
  const float mi = beta1 * m[i] + (1.0f - beta1) * gr;
  const float vi = beta2 * v[i] + (1.0f - beta2) * gr * gr;
  m[i] = mi; v[i] = vi;
  const float denom = sqrtf(vi) / sqrtf(bc2) + eps;
  pv -= (lr / bc1) * (mi / denom);
  p[i] = pv;
}

int launch_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                 float eps, float wd, int step, float clip, float grad_scale, hipStream_t s) {
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)((n + TB - 1) / TB)), dim3(TB), 0, s, p, g, m, v, n, lr, beta1, beta2,
                     eps, wd, bc1, bc2, clip, grad_scale);
  return 0;
}

}  // namespace jnr
