// Backward of the decision transformer over a whole trajectory, batched over the agents (loss.backward() through
// GPT.forward of src/models/gpt.py:308-369 with the blocks of :78-127; same mathematics as gpt_backward_kernel in
// kernels_train.hip, which gives one workgroup to each agent and therefore occupies B compute units).
//
// Here every phase is one launch over ALL tokens of ALL agents: rows m = agent * Lm + token (Lm = T + 1), the forward is
// recomputed into a row-major scratch, and each Linear is one small fp32 GEMM (64 x 64 tiles through LDS, FMA):
//   forward     out[M x N]  = in[M x K] . Wt[K x N]                     (Wt: the arena's transposed Linear weight)
//   data grad   din[M x K]  = dout[M x N] . Wt^T
//   weight grad gWt[K x N] += in^T[K x M] . dout[M x N]
// The number of executed glimpse steps S is only known on the device (n_done); the first kernel publishes L = S + 1 and
// every later kernel treats token rows >= L as absent (zero operands, zero results), so launches do not depend on it.
// The workload is a few GFLOP of GEMMs with M of a few hundred rows — launch- and latency-bound, not a roofline item;
// FMA tiles keep fp32 exactness and are far from being the limit.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>

#include "jn_device.h"
#include "jn_kernels.h"
#include "jn_types.h"

namespace jnr {
namespace {

constexpr int GT = 256;

__device__ __forceinline__ float gelu_fb(float x) {
  return 0.5f * x * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}
__device__ __forceinline__ float gelu_db(float x) {
  const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
  const float th = tanhf(u);
  return 0.5f * (1.0f + th) + 0.5f * x * (1.0f - th * th) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * x * x);
}

struct GemmArgs {
  const float* A; long long a_rs, a_cs;      // A(m, k) = A[m * a_rs + k * a_cs]
  const float* B; long long b_rs, b_cs;      // B(k, n)
  float* C; long long c_rs;                  // C(m, n) = C[m * c_rs + n]
  int M, N, K;
  const float* bias;                         // + bias[n]
  const float* resid;                        // + resid[m][n] (leading dimension c_rs), after the dropout of the product
  const float* mulp;                         // * gelu'(mulp[m][n])
  int a_gelu;                                // A elements pass through gelu on load
  int accumulate;                            // C += result
  int tok_dim;                               // 0: m runs over token rows, 1: k does (weight gradients)
  int k_per_block;                           // tok_dim 1: the reduction is split over gridDim.z workgroups (atomic C +=)
  const int* L; int Lm;                      // a token row r is live iff r % Lm < *L
  float pdrop; uint64_t seed; int site, layer;   // site >= 0: dropout (jn_device.h drop_scale) of product + bias
};

template <bool A_KFAST, bool B_NFAST>
__global__ __launch_bounds__(GT) void gptb_gemm_kernel(GemmArgs g) {
  constexpr int BM = 64, BN = 64, BK = 16;
  __shared__ float As[BK][BM + 1], Bs[BK][BN + 1];
  const int tid = threadIdx.x, m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const int L = *g.L;
  const int ty = tid >> 4, tx = tid & 15;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.0f;
  const int kbeg = g.k_per_block > 0 ? blockIdx.z * g.k_per_block : 0;
  const int kend = g.k_per_block > 0 ? min(g.K, kbeg + g.k_per_block) : g.K;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + q * GT;
      int m, k;
      if (A_KFAST) { k = idx & 15; m = idx >> 4; } else { m = idx & 63; k = idx >> 6; }
      const int gm = m0 + m, gk = k0 + k;
      float v = 0.0f;
      if (gm < g.M && gk < kend) {
        const bool live = g.tok_dim == 0 ? (gm % g.Lm) < L : (gk % g.Lm) < L;
        if (live) {
          v = g.A[(long long)gm * g.a_rs + (long long)gk * g.a_cs];
          if (g.a_gelu) v = gelu_fb(v);
        }
      }
      As[k][m] = v;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + q * GT;
      int n, k;
      if (B_NFAST) { n = idx & 63; k = idx >> 6; } else { k = idx & 15; n = idx >> 4; }
      const int gn = n0 + n, gk = k0 + k;
      float v = 0.0f;
      if (gn < g.N && gk < kend && (g.tok_dim == 0 || (gk % g.Lm) < L)) v = g.B[(long long)gk * g.b_rs + (long long)gn * g.b_cs];
      Bs[k][n] = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < BK; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int gm = m0 + ty * 4 + i;
    if (gm >= g.M) continue;
    const bool live = g.tok_dim != 0 || (gm % g.Lm) < L;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gn = n0 + tx * 4 + j;
      if (gn >= g.N) continue;
      const long long o = (long long)gm * g.c_rs + gn;
      float v = acc[i][j];
      if (live) {
        if (g.bias) v += g.bias[gn];
        if (g.site >= 0 && g.pdrop > 0.0f) v *= drop_scale(g.seed, gm / g.Lm, gm % g.Lm, g.layer, g.site, gn, g.pdrop);
        if (g.resid) v += g.resid[o];
        if (g.mulp) v *= gelu_db(g.mulp[o]);
      } else {
        v = 0.0f;
      }
      if (g.k_per_block > 0) atomicAdd(&g.C[o], v);
      else g.C[o] = g.accumulate ? g.C[o] + v : v;
    }
  }
}

void gemm(GemmArgs g, hipStream_t s) {
  dim3 grid((g.N + 63) / 64, (g.M + 63) / 64);
  if (g.tok_dim == 1 && g.accumulate) {      // few output tiles, long reduction: slices of 128 tokens, combined with atomics
    g.k_per_block = 128;
    grid.z = (g.K + 127) / 128;
  }
  const bool ak = g.a_cs == 1, bn = g.b_cs == 1;
  if (ak && bn) hipLaunchKernelGGL((gptb_gemm_kernel<true, true>), grid, dim3(GT), 0, s, g);
  else if (ak) hipLaunchKernelGGL((gptb_gemm_kernel<true, false>), grid, dim3(GT), 0, s, g);
  else if (bn) hipLaunchKernelGGL((gptb_gemm_kernel<false, true>), grid, dim3(GT), 0, s, g);
  else hipLaunchKernelGGL((gptb_gemm_kernel<false, false>), grid, dim3(GT), 0, s, g);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// L = S + 1 (S = first step at which every agent is done, gpt_backward_kernel's rule) and the embedded trajectory
// X0 = drop(final_emb) for the live token rows.
__global__ __launch_bounds__(GT) void gptb_init_kernel(GptBwdArgs a, int* Lp, float* X0, int Lm) {
  __shared__ int s_L;
  if (threadIdx.x == 0) {
    int S = a.T;
    if (a.stop_early)
      for (int t = 1; t <= a.T; ++t)
        if (a.n_done[t] >= a.B) { S = t; break; }
    s_L = S + 1;
    if (blockIdx.x == 0) *Lp = S + 1;
  }
  __syncthreads();
  const int L = s_L, C = a.C;
  const long long n = (long long)a.B * Lm * C;
  for (long long e = (long long)blockIdx.x * GT + threadIdx.x; e < n; e += (long long)gridDim.x * GT) {
    const int c = (int)(e % C);
    const long long m = e / C;
    const int b = (int)(m / Lm), i = (int)(m % Lm);
    float v = 0.0f;
    if (i < L) {
      v = a.final_emb[((long long)b * (a.T + 1) + i) * C + c];
      if (a.pdrop > 0.0f) v *= drop_scale(a.drop_seed, b, i, 0, 0, c, a.pdrop);
    }
    X0[e] = v;
  }
}

// LayerNorm (eps 1e-5) of every live row, one wave per row; mean and 1 / std are kept for the backward.
__global__ __launch_bounds__(GT) void gptb_ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ b, float* __restrict__ out,
                                                         float* __restrict__ mu, float* __restrict__ rs, int M, int C,
                                                         const int* __restrict__ Lp, int Lm) {
  const int row = blockIdx.x * (GT / 64) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float* xr = x + (long long)row * C;
  float* orow = out + (long long)row * C;
  if ((row % Lm) >= *Lp) {
    for (int c = lane; c < C; c += 64) orow[c] = 0.0f;
    if (lane == 0) { mu[row] = 0.0f; rs[row] = 0.0f; }
    return;
  }
  float sm = 0.0f;
  for (int c = lane; c < C; c += 64) sm += xr[c];
  const float m = wave_sum(sm) / C;
  float q = 0.0f;
  for (int c = lane; c < C; c += 64) { const float d = xr[c] - m; q += d * d; }
  const float r = 1.0f / sqrtf(wave_sum(q) / C + 1e-5f);
  for (int c = lane; c < C; c += 64) orow[c] = (xr[c] - m) * r * w[c] + b[c];
  if (lane == 0) { mu[row] = m; rs[row] = r; }
}

// din = base + LN^T(dout) per live row (base may be null or din itself); gw += sum dout * xhat, gb += sum dout.
// One wave per row, LN_RPB rows per workgroup; the parameter sums stay in registers over the rows of a wave.
constexpr int LN_RPB = 16, LN_CQ = 16;      // C <= 64 * LN_CQ
__global__ __launch_bounds__(GT) void gptb_ln_bwd_kernel(float* __restrict__ din, const float* __restrict__ base,
                                                         const float* __restrict__ dout, const float* __restrict__ x,
                                                         const float* __restrict__ w, float* __restrict__ gw,
                                                         float* __restrict__ gb, const float* __restrict__ mu,
                                                         const float* __restrict__ rs, int M, int C,
                                                         const int* __restrict__ Lp, int Lm) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, L = *Lp;
  float aw[LN_CQ], ab[LN_CQ];
#pragma unroll
  for (int q = 0; q < LN_CQ; ++q) { aw[q] = 0.0f; ab[q] = 0.0f; }
  for (int rr = wave; rr < LN_RPB; rr += GT / 64) {
    const int row = blockIdx.x * LN_RPB + rr;
    if (row >= M) break;
    float* dr = din + (long long)row * C;
    if ((row % Lm) >= L) {
      for (int c = lane; c < C; c += 64) dr[c] = 0.0f;
      continue;
    }
    const float* xr = x + (long long)row * C;
    const float* d = dout + (long long)row * C;
    const float m = mu[row], r = rs[row];
    float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
    for (int q = 0; q < LN_CQ; ++q) {
      const int c = lane + 64 * q;
      if (c < C) {
        const float dv = d[c], xh = (xr[c] - m) * r, dx = dv * w[c];
        m1 += dx; m2 += dx * xh;
        aw[q] += dv * xh; ab[q] += dv;
      }
    }
    m1 = wave_sum(m1) / C; m2 = wave_sum(m2) / C;
    for (int c = lane; c < C; c += 64) {
      const float xh = (xr[c] - m) * r;
      const float v = r * (d[c] * w[c] - m1 - xh * m2);
      dr[c] = base ? base[(long long)row * C + c] + v : v;
    }
  }
#pragma unroll
  for (int q = 0; q < LN_CQ; ++q) {
    const int c = lane + 64 * q;
    if (c < C && (aw[q] != 0.0f || ab[q] != 0.0f)) { atomicAdd(&gw[c], aw[q]); atomicAdd(&gb[c], ab[q]); }
  }
}

// gb[n] += sum over the live rows of x[m][n]
constexpr int COLSUM_ROWS = 128;
__global__ __launch_bounds__(GT) void gptb_colsum_kernel(const float* __restrict__ x, int M, int N, float* __restrict__ gb,
                                                         const int* __restrict__ Lp, int Lm) {
  __shared__ float part[GT / 64][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = blockIdx.x * 64 + lane, L = *Lp;
  float acc = 0.0f;
  const int m0 = blockIdx.y * COLSUM_ROWS, m1 = min(M, m0 + COLSUM_ROWS);
  if (n < N)
    for (int m = m0 + wave; m < m1; m += GT / 64)
      if ((m % Lm) < L) acc += x[(long long)m * N + n];
  part[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && n < N) {
    float sum = 0.0f;
#pragma unroll
    for (int q = 0; q < GT / 64; ++q) sum += part[q][lane];
    atomicAdd(&gb[n], sum);
  }
}

// dst = src * dropout mask of (layer, site) on the live rows, 0 on the others
__global__ __launch_bounds__(GT) void gptb_drop_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n,
                                                       int C, int layer, int site, float pdrop, uint64_t seed,
                                                       const int* __restrict__ Lp, int Lm) {
  const long long e = (long long)blockIdx.x * GT + threadIdx.x;
  if (e >= n) return;
  const long long m = e / C;
  const int c = (int)(e - m * C), b = (int)(m / Lm), i = (int)(m % Lm);
  dst[e] = i < *Lp ? src[e] * drop_scale(seed, b, i, layer, site, c, pdrop) : 0.0f;
}

// Causal attention of one (agent, head): probabilities P (kept for the backward) and Y = drop(P) . V.
__global__ __launch_bounds__(GT) void gptb_attn_fwd_kernel(const float* __restrict__ QKV, float* __restrict__ ATT,
                                                           float* __restrict__ Y, int C, int nh, int Lm,
                                                           const int* __restrict__ Lp, float pdrop, uint64_t seed, int layer,
                                                           int Tmax) {
  extern __shared__ float sm[];
  const int b = blockIdx.x / nh, h = blockIdx.x % nh, hs = C / nh, tid = threadIdx.x, L = *Lp;
  float* q = sm; float* k = q + Lm * hs; float* v = k + Lm * hs; float* p = v + Lm * hs;     // p: [Lm][Lm]
  for (int e = tid; e < L * hs; e += GT) {
    const int i = e / hs, d = e - i * hs;
    const float* src = QKV + ((long long)b * Lm + i) * 3 * C + h * hs + d;
    q[e] = src[0]; k[e] = src[C]; v[e] = src[2 * C];
  }
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)hs);
  for (int e = tid; e < L * L; e += GT) {
    const int i = e / L, j = e - i * L;
    float d = -INFINITY;
    if (j <= i) {
      d = 0.0f;
      for (int t = 0; t < hs; ++t) d = fmaf(q[i * hs + t], k[j * hs + t], d);
      d *= scale;
    }
    p[i * Lm + j] = d;
  }
  __syncthreads();
  for (int i = tid; i < L; i += GT) {
    float* row = p + i * Lm;
    float m = -INFINITY;
    for (int j = 0; j <= i; ++j) m = fmaxf(m, row[j]);
    float s = 0.0f;
    for (int j = 0; j <= i; ++j) { row[j] = expf(row[j] - m); s += row[j]; }
    for (int j = 0; j < L; ++j) row[j] = j <= i ? row[j] / s : 0.0f;
  }
  __syncthreads();
  float* att = ATT + (long long)blockIdx.x * Lm * Lm;
  for (int e = tid; e < L * L; e += GT) { const int i = e / L, j = e - i * L; att[i * Lm + j] = p[i * Lm + j]; }
  for (int e = tid; e < L * hs; e += GT) {
    const int i = e / hs, d = e - i * hs;
    float acc = 0.0f;
    for (int j = 0; j <= i; ++j) {
      float pij = p[i * Lm + j];
      if (pdrop > 0.0f) pij *= drop_scale(seed, b, i, layer, 1, h * Tmax + j, pdrop);
      acc = fmaf(pij, v[j * hs + d], acc);
    }
    Y[((long long)b * Lm + i) * C + h * hs + d] = acc;
  }
}

// Backward of the same: dQKV rows of the live tokens from dY, the kept probabilities and q, k, v.
__global__ __launch_bounds__(GT) void gptb_attn_bwd_kernel(const float* __restrict__ QKV, const float* __restrict__ ATT,
                                                           const float* __restrict__ dY, float* __restrict__ dQKV, int C,
                                                           int nh, int Lm, const int* __restrict__ Lp, float pdrop,
                                                           uint64_t seed, int layer, int Tmax) {
  extern __shared__ float sm[];
  const int b = blockIdx.x / nh, h = blockIdx.x % nh, hs = C / nh, tid = threadIdx.x, L = *Lp;
  float* q = sm; float* k = q + Lm * hs; float* v = k + Lm * hs; float* dy = v + Lm * hs;
  float* p = dy + Lm * hs; float* dp = p + Lm * Lm;
  for (int e = tid; e < L * hs; e += GT) {
    const int i = e / hs, d = e - i * hs;
    const float* src = QKV + ((long long)b * Lm + i) * 3 * C + h * hs + d;
    q[e] = src[0]; k[e] = src[C]; v[e] = src[2 * C];
    dy[e] = dY[((long long)b * Lm + i) * C + h * hs + d];
  }
  const float* att = ATT + (long long)blockIdx.x * Lm * Lm;
  for (int e = tid; e < L * L; e += GT) { const int i = e / L, j = e - i * L; p[i * Lm + j] = att[i * Lm + j]; }
  __syncthreads();
  const float scale = 1.0f / sqrtf((float)hs);
  // dP[i][j] = mask(i, j) * sum_d dY[i][d] v[j][d]
  for (int e = tid; e < L * L; e += GT) {
    const int i = e / L, j = e - i * L;
    float acc = 0.0f;
    if (j <= i) {
      for (int t = 0; t < hs; ++t) acc = fmaf(dy[i * hs + t], v[j * hs + t], acc);
      if (pdrop > 0.0f) acc *= drop_scale(seed, b, i, layer, 1, h * Tmax + j, pdrop);
    }
    dp[i * Lm + j] = acc;
  }
  // dV[j][d] = sum_{i >= j} mask(i, j) P[i][j] dY[i][d]
  for (int e = tid; e < L * hs; e += GT) {
    const int j = e / hs, d = e - j * hs;
    float acc = 0.0f;
    for (int i = j; i < L; ++i) {
      float pij = p[i * Lm + j];
      if (pdrop > 0.0f) pij *= drop_scale(seed, b, i, layer, 1, h * Tmax + j, pdrop);
      acc = fmaf(pij, dy[i * hs + d], acc);
    }
    dQKV[((long long)b * Lm + j) * 3 * C + 2 * C + h * hs + d] = acc;
  }
  __syncthreads();
  // softmax backward in place: dS = P * (dP - sum_j P dP) * scale
  for (int i = tid; i < L; i += GT) {
    float* dr = dp + i * Lm;
    const float* pr = p + i * Lm;
    float dot = 0.0f;
    for (int j = 0; j <= i; ++j) dot = fmaf(pr[j], dr[j], dot);
    for (int j = 0; j < L; ++j) dr[j] = j <= i ? pr[j] * (dr[j] - dot) * scale : 0.0f;
  }
  __syncthreads();
  for (int e = tid; e < L * hs; e += GT) {
    const int i = e / hs, d = e - i * hs;
    float dq = 0.0f, dk = 0.0f;
    for (int j = 0; j <= i; ++j) dq = fmaf(dp[i * Lm + j], k[j * hs + d], dq);
    for (int ii = i; ii < L; ++ii) dk = fmaf(dp[ii * Lm + i], q[ii * hs + d], dk);
    float* dst = dQKV + ((long long)b * Lm + i) * 3 * C + h * hs + d;
    dst[0] = dq; dst[C] = dk;
  }
}

// DL[m][a] = dlogits[b][i - 1][a] for the live tokens i >= 1 (token i predicts the action of step i - 1), else 0
__global__ __launch_bounds__(GT) void gptb_dl_kernel(const float* __restrict__ dlogits, float* __restrict__ DL, int B, int T,
                                                     int nA, int Lm, const int* __restrict__ Lp) {
  const int e = blockIdx.x * GT + threadIdx.x;
  if (e >= B * Lm * nA) return;
  const int m = e / nA, q = e - m * nA, b = m / Lm, i = m % Lm;
  DL[e] = (i >= 1 && i < *Lp) ? dlogits[((long long)b * T + (i - 1)) * nA + q] : 0.0f;
}

// Token embeddings (GPT.forward's embedding stage, src/models/gpt.py:308-343), batched like everything else:
//   gptb_parts_kernel   dXe = drop0(dX) with the class-token row taken out (its gradient goes to embed_class), and the
//                       concatenated embedding parts PARTS[m][p * C] of every live patch token (action, 1-D position,
//                       patch embedding, 2-D position — the order of the forward);
//   two GEMMs           dparts = dXe . proj^T,  g_proj += PARTS^T . dXe   (concat_emb; otherwise dparts = dXe / p);
//   gptb_scatter_kernel dparts -> the action / position tables (atomics) and the patch-embedding rows d_tok_emb.
struct TokInfo { int act, p1, row, col; };
__device__ __forceinline__ TokInfo tok_info(const GptBwdArgs& a, int b, int i) {
  const int t = i - 1;
  TokInfo r;
  // rollout: token i carries the action taken BEFORE its patch (BOS = 0); teacher-forced full sequence:
  // current_actions[b][t] (src/supervised.py:863-868)
  r.act = a.tok_actions ? (int)a.tok_actions[(long long)b * a.T + t] : ((i == 1) ? 0 : (int)a.actions[(long long)b * a.T + (i - 2)]);
  r.row = (int)a.positions[((long long)b * a.pos_tokens + t) * 2];
  r.col = (int)a.positions[((long long)b * a.pos_tokens + t) * 2 + 1];
  r.p1 = a.pos1d_by_token ? t : 0;                       // recurrent tokens: 1-D position 0 (gpt.py:431-449)
  return r;
}

__global__ __launch_bounds__(GT) void gptb_parts_kernel(GptBwdArgs a, const float* __restrict__ dX, float* __restrict__ dXe,
                                                        float* __restrict__ PARTS, int np, const int* __restrict__ Lp, int Lm) {
  const int m = blockIdx.x, b = m / Lm, i = m % Lm, C = a.C, tid = threadIdx.x, L = *Lp;
  const bool live = i < L;
  for (int c = tid; c < C; c += GT) {
    float v = live ? dX[(long long)m * C + c] : 0.0f;
    if (live && a.pdrop > 0.0f) v *= drop_scale(a.drop_seed, b, i, 0, 0, c, a.pdrop);     // through transformer.drop
    if (i == 0) {                                                                         // class token: row classes[b]
      const int cls = a.classes ? min(max((int)a.classes[b], 0), JN_N_CLASS_ROWS - 1) : 0;
      if (live) atomicAdd(&a.g_embed_class[(long long)cls * C + c], v);
      v = 0.0f;
    }
    dXe[(long long)m * C + c] = v;
  }
  float* pr = PARTS + (long long)m * np * C;
  if (!live || i == 0) {
    for (int e = tid; e < np * C; e += GT) pr[e] = 0.0f;
    return;
  }
  const TokInfo ti = tok_info(a, b, i);
  int p = 0;
  for (int c = tid; c < C; c += GT) pr[c] = a.wte[ti.act * C + c];
  ++p;
  for (int c = tid; c < C; c += GT) pr[p * C + c] = a.dec_pos_enc ? a.pos1d[ti.p1 * C + c] : a.wpe[ti.p1 * C + c];
  ++p;
  if (!a.no_patch_emb) {
    for (int c = tid; c < C; c += GT) pr[p * C + c] = a.tok_emb[((long long)b * a.T + (i - 1)) * C + c];
    ++p;
  }
  if (a.use_pos_emb)
    for (int c = tid; c < C; c += GT)
      pr[p * C + c] = (c < a.pe2_ch) ? a.pe2[ti.col * a.pe2_ch + c] : a.pe2[ti.row * a.pe2_ch + (c - a.pe2_ch)];
}

// one workgroup per (agent, step t): dparts row of token t + 1 (dp_ld floats per row; without concat_emb the row is dXe
// itself and every part receives 1 / np of it)
__global__ __launch_bounds__(GT) void gptb_scatter_kernel(GptBwdArgs a, const float* __restrict__ dparts, int dp_ld, int np,
                                                          const int* __restrict__ Lp, int Lm) {
  const int b = blockIdx.x / a.T, t = blockIdx.x % a.T, i = t + 1, C = a.C, tid = threadIdx.x, L = *Lp;
  float* dte = a.d_tok_emb + ((long long)b * a.dte_stride_b + (long long)t * a.dte_stride_t) * C;
  if (i >= L) {                                          // steps that were never executed: zero patch-embedding gradient
    for (int c = tid; c < C; c += GT) dte[c] = 0.0f;
    return;
  }
  const TokInfo ti = tok_info(a, b, i);
  const float* dp = dparts + ((long long)b * Lm + i) * dp_ld;
  const float sc = a.concat_emb ? 1.0f : 1.0f / np;
  const int o_pos = a.concat_emb ? C : 0, o_patch = a.concat_emb ? 2 * C : 0;
  for (int c = tid; c < C; c += GT) {
    atomicAdd(&a.g_wte[ti.act * C + c], dp[c] * sc);
    if (!a.dec_pos_enc && a.g_wpe) atomicAdd(&a.g_wpe[ti.p1 * C + c], dp[o_pos + c] * sc);
    if (!a.no_patch_emb) dte[c] = dp[o_patch + c] * sc;
  }
}

struct Layout {
  long long M, C, X, lay, lay_sz, tmp;
  long long o_H1, o_QKV, o_ATT, o_Y, o_XM, o_H2, o_Fp, o_st;   // inside a layer block
};
Layout make_layout(int C, int nh, int nL, int B, int T) {
  Layout y{};
  const long long Lm = T + 1;
  y.M = (long long)B * Lm; y.C = C;
  y.X = 16;                                            // header (L) first
  y.lay = y.X + (long long)(nL + 1) * y.M * C;
  y.o_H1 = 0; y.o_QKV = y.o_H1 + y.M * C; y.o_ATT = y.o_QKV + y.M * 3 * C; y.o_Y = y.o_ATT + (long long)B * nh * Lm * Lm;
  y.o_XM = y.o_Y + y.M * C; y.o_H2 = y.o_XM + y.M * C; y.o_Fp = y.o_H2 + y.M * C; y.o_st = y.o_Fp + y.M * 4 * C;
  y.lay_sz = y.o_st + 4 * y.M;
  y.tmp = y.lay + (long long)nL * y.lay_sz;
  return y;
}

}  // namespace

size_t gpt_backward_batched_scratch(int C, int n_head, int n_layer, int nA, int B, int T) {
  const Layout y = make_layout(C, n_head, n_layer, B, T);
  // temporaries: dX dXM dH dY dO HF (M*C each), dQKV (3), dF (4), DL (M*nA), ln_f statistics (2M)
  return (size_t)(y.tmp + y.M * (13LL * C + nA + 2) + 64);
}

// Returns 0 when launched, 1 when the shape is outside what these kernels take (the caller falls back).
int launch_gpt_backward_batched(const GptBwdArgs& a, const GptLayerPtrs* W, const GptLayerPtrs* G, hipStream_t s) {
  const int C = a.C, nh = a.n_head, nL = a.n_layer, B = a.B, T = a.T, Lm = T + 1, hs = C / nh, nA = a.nA;
  const size_t attn_lds = ((size_t)4 * Lm * hs + (size_t)2 * Lm * Lm) * sizeof(float);
  if (C > 64 * LN_CQ || attn_lds > 64 * 1024) return 1;
  const Layout y = make_layout(C, nh, nL, B, T);
  const int M = (int)y.M;
  float* sc = a.scratch;
  int* Lp = reinterpret_cast<int*>(sc);
  float* X = sc + y.X;
  float* tmp = sc + y.tmp;
  float* dX = tmp; float* dXM = dX + (long long)M * C; float* dH = dXM + (long long)M * C; float* dY = dH + (long long)M * C;
  float* dO = dY + (long long)M * C; float* HF = dO + (long long)M * C; float* dQKV = HF + (long long)M * C;
  float* dF = dQKV + (long long)M * 3 * C; float* DL = dF + (long long)M * 4 * C; float* muF = DL + (long long)M * nA;
  float* rsF = muF + M;
  const bool drop = a.pdrop > 0.0f;
  const int row_blocks = (M + GT / 64 - 1) / (GT / 64), lnb_blocks = (M + LN_RPB - 1) / LN_RPB;
  const long long MC = (long long)M * C;
  auto G0 = [&]() {
    GemmArgs g{};
    g.L = Lp; g.Lm = Lm; g.site = -1; g.pdrop = a.pdrop; g.seed = a.drop_seed;
    return g;
  };
  // forward: out[M x N] = in[M x K] . Wt[K x N] (+ bias)
  auto fwd = [&](float* out, const float* in, const float* wt, const float* bias, int K, int N, const float* resid, int site,
                 int layer, int a_gelu) {
    GemmArgs g = G0();
    g.A = in; g.a_rs = K; g.a_cs = 1; g.B = wt; g.b_rs = N; g.b_cs = 1; g.C = out; g.c_rs = N; g.M = M; g.N = N; g.K = K;
    g.bias = bias; g.resid = resid; g.site = site; g.layer = layer; g.a_gelu = a_gelu; g.tok_dim = 0;
    gemm(g, s);
  };
  // data gradient: din[M x K] = dout[M x N] . Wt^T (optionally * gelu'(mulp))
  auto bwd_data = [&](float* din, const float* dout, const float* wt, int K, int N, const float* mulp) {
    GemmArgs g = G0();
    g.A = dout; g.a_rs = N; g.a_cs = 1; g.B = wt; g.b_rs = 1; g.b_cs = N; g.C = din; g.c_rs = K; g.M = M; g.N = K; g.K = N;
    g.mulp = mulp; g.tok_dim = 0;
    gemm(g, s);
  };
  // weight gradient: gwt[K x N] += in^T . dout; bias gradient: gb[N] += column sums of dout
  auto bwd_weight = [&](float* gwt, float* gb, const float* in, const float* dout, int K, int N, int a_gelu) {
    GemmArgs g = G0();
    g.A = in; g.a_rs = 1; g.a_cs = K; g.B = dout; g.b_rs = N; g.b_cs = 1; g.C = gwt; g.c_rs = N; g.M = K; g.N = N; g.K = M;
    g.a_gelu = a_gelu; g.accumulate = 1; g.tok_dim = 1;
    gemm(g, s);
    if (gb) hipLaunchKernelGGL(gptb_colsum_kernel, dim3((N + 63) / 64, (M + COLSUM_ROWS - 1) / COLSUM_ROWS), dim3(GT), 0, s, dout, M, N, gb, Lp, Lm);
  };
  auto ln_fwd = [&](float* out, const float* x, const float* w, const float* b, float* mu, float* rs) {
    hipLaunchKernelGGL(gptb_ln_fwd_kernel, dim3(row_blocks), dim3(GT), 0, s, x, w, b, out, mu, rs, M, C, Lp, Lm);
  };
  auto ln_bwd = [&](float* din, const float* base, const float* dout, const float* x, const float* w, float* gw, float* gb,
                    const float* mu, const float* rs) {
    hipLaunchKernelGGL(gptb_ln_bwd_kernel, dim3(lnb_blocks), dim3(GT), 0, s, din, base, dout, x, w, gw, gb, mu, rs, M, C, Lp, Lm);
  };
  auto dropmul = [&](float* dst, const float* src, int layer, int site) {
    hipLaunchKernelGGL(gptb_drop_kernel, dim3((unsigned)((MC + GT - 1) / GT)), dim3(GT), 0, s, dst, src, MC, C, layer, site,
                       a.pdrop, a.drop_seed, Lp, Lm);
  };

  hipLaunchKernelGGL(gptb_init_kernel, dim3((unsigned)std::min<long long>((MC + GT - 1) / GT, 1024)), dim3(GT), 0, s, a, Lp, X, Lm);
  // ---------------- forward recompute ----------------
  for (int l = 0; l < nL; ++l) {
    const GptLayerPtrs& w = W[l];
    float* x = X + (long long)l * MC;
    float* lay = sc + y.lay + (long long)l * y.lay_sz;
    float *H1 = lay + y.o_H1, *QKV = lay + y.o_QKV, *ATT = lay + y.o_ATT, *Yb = lay + y.o_Y, *XM = lay + y.o_XM,
          *H2 = lay + y.o_H2, *Fp = lay + y.o_Fp, *st = lay + y.o_st;
    ln_fwd(H1, x, w.ln1_w, w.ln1_b, st, st + M);
    fwd(QKV, H1, w.qkv_wt, w.qkv_b, C, 3 * C, nullptr, -1, l, 0);
    hipLaunchKernelGGL(gptb_attn_fwd_kernel, dim3(B * nh), dim3(GT), attn_lds, s, QKV, ATT, Yb, C, nh, Lm, Lp, a.pdrop,
                       a.drop_seed, l, a.Tmax);
    fwd(XM, Yb, w.proj_wt, w.proj_b, C, C, x, 2, l, 0);                       // XM = x + drop(proj(Y))
    ln_fwd(H2, XM, w.ln2_w, w.ln2_b, st + 2 * M, st + 3 * M);
    fwd(Fp, H2, w.fc_wt, w.fc_b, C, 4 * C, nullptr, -1, l, 0);                // pre-activation
    fwd(x + MC, Fp, w.fc2_wt, w.fc2_b, 4 * C, C, XM, 3, l, 1);                // x' = XM + drop(fc2(gelu(F)))
  }
  float* xl = X + (long long)nL * MC;
  ln_fwd(HF, xl, a.lnf_w, a.lnf_b, muF, rsF);
  // ---------------- head + ln_f backward ----------------
  hipLaunchKernelGGL(gptb_dl_kernel, dim3((M * nA + GT - 1) / GT), dim3(GT), 0, s, a.dlogits, DL, B, T, nA, Lm, Lp);
  bwd_data(dH, DL, a.head_wt, C, nA, nullptr);                                 // dHF = DL . head_wt^T   (head_wt: [C][nA])
  bwd_weight(a.g_head_wt, nullptr, HF, DL, C, nA, 0);
  ln_bwd(dX, nullptr, dH, xl, a.lnf_w, a.g_lnf_w, a.g_lnf_b, muF, rsF);
  // ---------------- blocks, last to first ----------------
  for (int l = nL - 1; l >= 0; --l) {
    const GptLayerPtrs& w = W[l];
    const GptLayerPtrs& gr = G[l];
    float* x = X + (long long)l * MC;
    float* lay = sc + y.lay + (long long)l * y.lay_sz;
    float *H1 = lay + y.o_H1, *QKV = lay + y.o_QKV, *ATT = lay + y.o_ATT, *Yb = lay + y.o_Y, *XM = lay + y.o_XM,
          *H2 = lay + y.o_H2, *Fp = lay + y.o_Fp, *st = lay + y.o_st;
    // mlp: x' = XM + drop(fc2(gelu(fc(H2))))
    const float* dOut = dX;
    if (drop) { dropmul(dO, dX, l, 3); dOut = dO; }
    bwd_data(dF, dOut, w.fc2_wt, 4 * C, C, Fp);                                // dF = (dO . fc2^T) * gelu'(F)
    bwd_weight(gr.fc2_wt, gr.fc2_b, Fp, dOut, 4 * C, C, 1);                    // gelu(F)^T . dO
    bwd_data(dH, dF, w.fc_wt, C, 4 * C, nullptr);
    bwd_weight(gr.fc_wt, gr.fc_b, H2, dF, C, 4 * C, 0);
    ln_bwd(dXM, dX, dH, XM, w.ln2_w, gr.ln2_w, gr.ln2_b, st + 2 * M, st + 3 * M);       // dXM = dX + LN2^T(dH)
    // attention: XM = x + drop(proj(Y))
    const float* dPo = dXM;
    if (drop) { dropmul(dO, dXM, l, 2); dPo = dO; }
    bwd_data(dY, dPo, w.proj_wt, C, C, nullptr);
    bwd_weight(gr.proj_wt, gr.proj_b, Yb, dPo, C, C, 0);
    hipLaunchKernelGGL(gptb_attn_bwd_kernel, dim3(B * nh), dim3(GT), attn_lds, s, QKV, ATT, dY, dQKV, C, nh, Lm, Lp, a.pdrop,
                       a.drop_seed, l, a.Tmax);
    bwd_data(dH, dQKV, w.qkv_wt, C, 3 * C, nullptr);
    bwd_weight(gr.qkv_wt, gr.qkv_b, H1, dQKV, C, 3 * C, 0);
    ln_bwd(dX, dXM, dH, x, w.ln1_w, gr.ln1_w, gr.ln1_b, st, st + M);                    // dX = dXM + LN1^T(dH)
  }
  // ---------------- token embeddings ----------------
  const int np = 2 + (a.no_patch_emb ? 0 : 1) + (a.use_pos_emb ? 1 : 0);
  float* dXe = dXM;                                      // free from here on
  float* PARTS = sc + y.lay + y.o_Fp;                    // layer 0's M x 4C block, free as well
  float* dparts = dF;
  hipLaunchKernelGGL(gptb_parts_kernel, dim3(M), dim3(GT), 0, s, a, dX, dXe, PARTS, np, Lp, Lm);
  if (a.concat_emb) {
    bwd_data(dparts, dXe, a.proj_wt, np * C, C, nullptr);                      // proj_wt: [np * C][C]
    bwd_weight(a.g_proj_wt, a.g_proj_b, PARTS, dXe, np * C, C, 0);
    hipLaunchKernelGGL(gptb_scatter_kernel, dim3(B * T), dim3(GT), 0, s, a, dparts, np * C, np, Lp, Lm);
  } else {
    hipLaunchKernelGGL(gptb_scatter_kernel, dim3(B * T), dim3(GT), 0, s, a, dXe, C, np, Lp, Lm);
  }
  return 0;
}

}  // namespace jnr
