// One glimpse step of the decision model for every agent of the batch, in ONE launch:
//   token embedding (src/models/gpt.py:419-479) -> pre-LN GPT-2 blocks with a persistent
//   per-agent KV cache (the reference recomputes the whole prefix, gpt.py:525-528; with
//   dropout 0 and a causal mask the cached form is the same function) -> ln_f ->
//   ActionHead (src/models/action_head.py:24-33) -> Categorical argmax / sample / forced
//   + log-prob + entropy (src/reinforce.py:73-90) -> env step + reward
//   (src/env/general_env.py:172-233, 321-358) -> rollout bookkeeping (reinforce.py:169-179).
// One workgroup per agent; all vectors live in LDS; Linear weights are stored transposed
// ([k][n]) so lanes read consecutive n.  Nothing here returns to the host.
#include <hip/hip_runtime.h>

#include "jn_device.h"

namespace jnr {

// Threads per agent (template parameter NT of everything below): 1024 for wide models — 16 waves, every Linear splits K
// over all of them (256 threads left a 192 -> 576 layer of gpt-mini with ONE K slice: 192 dependent loads per thread) — and
// 256 for n_embd <= 64 (gpt-nano: its largest Linear is 48 x 192, the 1024-thread form only made every barrier and block
// reduction four times as wide: 95.7 -> 119.4 us per step between rounds 1 and 2, back to the 256-thread form in round 3).

#ifndef JN_LINEAR_UNROLL
#define JN_LINEAR_UNROLL 4
#endif

template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
  constexpr int GPT_WAVES = NT / 64;
  // wave reduce by shuffles, then the waves' partials through LDS
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float s = 0.0f;
#pragma unroll
  for (int w = 0; w < GPT_WAVES; ++w) s += red[w];
  return s;
}

// y[n] = b[n] + sum_k x[k] * wt[k*N + n]   (x in LDS, wt transposed in global/L2).
// All 256 threads work whatever N is: thread = (column quad, K slice); each slice walks K with stride `slices`
// (independent dwordx4 loads), partials meet in LDS scratch `part` (>= 4 * NT floats).  With one thread per column a
// 192 -> 48 layer was 192 dependent loads on 48 threads: the whole step was a chain of L2 latencies.
template <int NT>
__device__ __forceinline__ void linear_t(float* y, const float* x, const float* __restrict__ wt,
                                         const float* __restrict__ b, int K, int N, float* part) {
  using f4 = __attribute__((ext_vector_type(4))) float;
  const int tid = threadIdx.x;
  if ((N & 3) == 0 && N <= 4 * NT) {
    const int nq = N >> 2, slices = NT / nq;
    const int q = tid % nq, sl = tid / nq;
    if (sl < slices) {
      f4 acc = {0.f, 0.f, 0.f, 0.f};
      const float* wp = wt + 4 * q;
#pragma unroll JN_LINEAR_UNROLL
      for (int k = sl; k < K; k += slices) acc += x[k] * *reinterpret_cast<const f4*>(wp + (long long)k * N);
      *reinterpret_cast<f4*>(part + sl * N + 4 * q) = acc;
    }
    __syncthreads();
    for (int n = tid; n < N; n += NT) {
      float acc = b ? b[n] : 0.0f;
      for (int i = 0; i < slices; ++i) acc += part[i * N + n];
      y[n] = acc;
    }
  } else if (N <= NT) {
    const int slices = NT / N;
    const int n = tid % N, sl = tid / N;
    if (sl < slices) {
      float acc = 0.0f;
      for (int k = sl; k < K; k += slices) acc = fmaf(x[k], wt[(long long)k * N + n], acc);
      part[sl * N + n] = acc;
    }
    __syncthreads();
    if (tid < N) {
      float acc = b ? b[tid] : 0.0f;
      for (int i = 0; i < slices; ++i) acc += part[i * N + tid];
      y[tid] = acc;
    }
  } else {
    for (int n = tid; n < N; n += NT) {
      float acc = b ? b[n] : 0.0f;
      const float* wp = wt + n;
#pragma unroll 4
      for (int k = 0; k < K; ++k) acc = fmaf(x[k], wp[(long long)k * N], acc);
      y[n] = acc;
    }
  }
}

template <int NT>
__device__ __forceinline__ void layer_norm(float* y, const float* x, const float* __restrict__ w,
                                           const float* __restrict__ b, int C, float* red) {
  float s = 0.0f;
  for (int i = threadIdx.x; i < C; i += NT) s += x[i];
  const float mean = block_sum<NT>(s, red) / C;
  float q = 0.0f;
  for (int i = threadIdx.x; i < C; i += NT) { const float d = x[i] - mean; q += d * d; }
  const float var = block_sum<NT>(q, red) / C;
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  for (int i = threadIdx.x; i < C; i += NT) y[i] = (x[i] - mean) * rstd * w[i] + b[i];
  __syncthreads();
}

__device__ __forceinline__ float gelu_tanh(float x) {
  // NewGELU, src/models/gpt.py:37-47
  return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}

template <int NT>
__global__ __launch_bounds__(NT) void gpt_step_kernel(GptStepArgs a) {
  constexpr int GPT_WAVES = NT / 64;
  if (a.skip_flag && *a.skip_flag >= a.skip_when) {
    // every env was done before this step: stay skipped for the rest of the trajectory
    if (blockIdx.x == 0 && threadIdx.x == 0) a.n_done[a.step + 1] = a.skip_when;
    return;
  }
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int C = a.C, tid = threadIdx.x, b = blockIdx.x;
  float* x = sm;                 // [C]   residual stream
  float* h = x + C;              // [C]   LN output / attention output
  float* qkv = h + C;            // [3C]
  float* mlp = qkv + 3 * C;      // [4C]  also the concatenated embedding parts
  float* att = mlp + 4 * C;      // [n_head * Tmax]
  float* red = att + a.n_head * a.Tmax;   // [GPT_WAVES]
  float* lg = red + GPT_WAVES;   // [16]  logits
  float* part = sm + ((9 * C + a.n_head * a.Tmax + GPT_WAVES + 16 + 3) & ~3);   // [4 * NT] split-K partials of linear_t (16-B aligned)

  const int t = a.step;
  const int hs = C / a.n_head;
  const float scale = 1.0f / sqrtf((float)hs);
  int len = a.cache_len[b];      // tokens already cached (0 at the start of a sequence)
  const int n_new = (a.src_mode == GPT_SRC_ENV && t == 0) ? 2 : 1;

  for (int j = 0; j < n_new; ++j) {
    // ---------------- token embedding ----------------
    const bool class_tok = (a.src_mode == GPT_SRC_CLASS) || (a.src_mode == GPT_SRC_ENV && t == 0 && j == 0);
    if (class_tok) {
      // embed_class(classes) (gpt.py:476-478); the rollout passes no ids: class 0 (reinforce.py:128-129)
      const int cls = a.classes ? min(max((int)a.classes[b], 0), JN_N_CLASS_ROWS - 1) : 0;
      for (int i = tid; i < C; i += NT) x[i] = a.embed_class[(long long)cls * C + i];
    } else if (a.src_mode == GPT_SRC_GIVEN) {
      const float* gp = a.given_emb + ((long long)b * a.given_stride + a.given_index) * C;
      for (int i = tid; i < C; i += NT) x[i] = gp[i];
    } else {
      int act, row, col;
      if (a.src_mode == GPT_SRC_ENV) {
        act = (int)a.prev_action[b];
        row = (int)a.env.positions[2 * b]; col = (int)a.env.positions[2 * b + 1];
      } else {
        const long long bi = (long long)b * a.t_stride + a.t_index;
        act = (int)a.t_actions[bi];
        row = a.t_positions ? (int)a.t_positions[2 * bi] : 0;
        col = a.t_positions ? (int)a.t_positions[2 * bi + 1] : 0;
      }
      act = min(max(act, 0), a.nA - 1);
      row = min(max(row, 0), 255); col = min(max(col, 0), 255);
      float* parts = mlp;
      int p = 0;
      for (int i = tid; i < C; i += NT) parts[i] = a.wte[act * C + i];
      ++p;
      for (int i = tid; i < C; i += NT)
        parts[p * C + i] = a.dec_pos_enc ? a.pos1d[a.pos_index * C + i] : a.wpe[a.pos_index * C + i];
      ++p;
      if (!a.no_patch_emb) {
        if (a.src_mode == GPT_SRC_ENV) {
          for (int i = tid; i < C; i += NT) {
            float s = a.efpn_lin_b[i];
            for (int ks = 0; ks < a.KS; ++ks) s += a.emb_part[((long long)b * a.KS + ks) * C + i];
            parts[p * C + i] = s;
            if (a.tok_emb_out) a.tok_emb_out[((long long)b * a.T + t) * C + i] = s;
          }
        } else {
          const float* pe = a.tok_emb + ((long long)b * a.tok_emb_stride + a.tok_emb_index) * C;
          for (int i = tid; i < C; i += NT) parts[p * C + i] = pe[i];
        }
        ++p;
      }
      if (a.use_pos_emb) {
        for (int i = tid; i < C; i += NT)
          parts[p * C + i] = (i < a.pe2_ch) ? a.pe2[col * a.pe2_ch + i] : a.pe2[row * a.pe2_ch + (i - a.pe2_ch)];
        ++p;
      }
      __syncthreads();
      if (a.concat_emb) {
        linear_t<NT>(x, parts, a.proj_wt, a.proj_b, p * C, C, part);
      } else {
        for (int i = tid; i < C; i += NT) {
          float s = 0.0f;
          for (int q = 0; q < p; ++q) s += parts[q * C + i];
          x[i] = s / p;
        }
      }
    }
    __syncthreads();
    if (a.out.final_emb)
      for (int i = tid; i < C; i += NT) a.out.final_emb[((long long)b * a.emb_stride + len) * C + i] = x[i];
    if (a.embed_only) { ++len; continue; }
    const bool drop = a.pdrop > 0.0f;
    if (drop) {                                           // x = transformer.drop(final_emb), gpt.py:525
      for (int i = tid; i < C; i += NT) x[i] *= drop_scale(a.drop_seed, b, len, 0, 0, i, a.pdrop);
      __syncthreads();
    }

    // ---------------- transformer blocks ----------------
    for (int l = 0; l < a.n_layer; ++l) {
      const GptLayerPtrs L = a.layers[l];
      float* kc = a.kcache + (((long long)l * a.B + b) * a.Tmax) * C;
      float* vc = a.vcache + (((long long)l * a.B + b) * a.Tmax) * C;
      layer_norm<NT>(h, x, L.ln1_w, L.ln1_b, C, red);
      linear_t<NT>(qkv, h, L.qkv_wt, L.qkv_b, C, 3 * C, part);
      __syncthreads();
      for (int i = tid; i < C; i += NT) { kc[len * C + i] = qkv[C + i]; vc[len * C + i] = qkv[2 * C + i]; }
      __syncthreads();   // own-block global writes are visible to the block after the barrier
      const int nk = len + 1;
      for (int e = tid; e < a.n_head * nk; e += NT) {
        const int hd = e / nk, s = e - hd * nk;
        const float* kp = (s == len) ? (qkv + C + hd * hs) : (kc + s * C + hd * hs);
        const float* qp = qkv + hd * hs;
        float d = 0.0f;
        for (int i = 0; i < hs; ++i) d = fmaf(qp[i], kp[i], d);
        att[hd * a.Tmax + s] = d * scale;
      }
      __syncthreads();
      if (tid < a.n_head) {
        float* ap = att + tid * a.Tmax;
        float m = -INFINITY;
        for (int s = 0; s < nk; ++s) m = fmaxf(m, ap[s]);
        float sum = 0.0f;
        for (int s = 0; s < nk; ++s) { const float ev = expf(ap[s] - m); ap[s] = ev; sum += ev; }
        const float inv = 1.0f / sum;
        for (int s = 0; s < nk; ++s) ap[s] *= inv;
        if (drop)                                         // attn_dropout on the probabilities, gpt.py:100
          for (int s = 0; s < nk; ++s) ap[s] *= drop_scale(a.drop_seed, b, len, l, 1, tid * a.Tmax + s, a.pdrop);
      }
      __syncthreads();
      for (int i = tid; i < C; i += NT) {
        const int hd = i / hs;
        float acc = 0.0f;
        for (int s = 0; s < nk; ++s) {
          const float vv = (s == len) ? qkv[2 * C + i] : vc[s * C + i];
          acc = fmaf(att[hd * a.Tmax + s], vv, acc);
        }
        h[i] = acc;
      }
      __syncthreads();
      linear_t<NT>(qkv, h, L.proj_wt, L.proj_b, C, C, part);      // qkv[0:C] reused as scratch
      __syncthreads();
      for (int i = tid; i < C; i += NT) x[i] += drop ? qkv[i] * drop_scale(a.drop_seed, b, len, l, 2, i, a.pdrop) : qkv[i];
      __syncthreads();
      layer_norm<NT>(h, x, L.ln2_w, L.ln2_b, C, red);
      linear_t<NT>(mlp, h, L.fc_wt, L.fc_b, C, 4 * C, part);
      __syncthreads();
      for (int i = tid; i < 4 * C; i += NT) mlp[i] = gelu_tanh(mlp[i]);
      __syncthreads();
      linear_t<NT>(qkv, mlp, L.fc2_wt, L.fc2_b, 4 * C, C, part);
      __syncthreads();
      for (int i = tid; i < C; i += NT) x[i] += drop ? qkv[i] * drop_scale(a.drop_seed, b, len, l, 3, i, a.pdrop) : qkv[i];
      __syncthreads();
    }
    ++len;
  }

  if (a.embed_only) {
    if (tid == 0) a.cache_len[b] = len;
    return;
  }
  // ---------------- head on the newest token ----------------
  layer_norm<NT>(h, x, a.lnf_w, a.lnf_b, C, red);
  linear_t<NT>(lg, h, a.head_wt, nullptr, C, a.nA, part);
  __syncthreads();

  if (a.src_mode != GPT_SRC_ENV) {
    if (tid == 0) a.cache_len[b] = len;
    if (a.logits_rows && tid < a.nA) a.logits_rows[(long long)b * a.logits_stride + tid] = lg[tid];
    return;
  }
  if (tid == 0) {
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

int launch_gpt_step(const GptStepArgs& a, hipStream_t s) {
  if (launch_gpt_step_mfma(a, s)) return 0;             // wide models: 16 agents per workgroup on the matrix pipe
  const int nt = a.C <= 64 ? 256 : 1024;
  const size_t smem = (size_t)(9 * a.C + a.n_head * a.Tmax + nt / 64 + 16 + 4 + 4 * nt) * sizeof(float);
  if (nt == 256) hipLaunchKernelGGL(gpt_step_kernel<256>, dim3(a.B), dim3(256), smem, s, a);
  else hipLaunchKernelGGL(gpt_step_kernel<1024>, dim3(a.B), dim3(1024), smem, s, a);
  return 0;
}

}  // namespace jnr
