// Backward kernels of the PAFPN patch encoder (train-mode BatchNorm) for gfx950.
//
// Per conv layer with output z (raw), a = silu(y), y = sc*z + sh, batch statistics (mean, invstd):
//   1. bn_bwd_reduce   sum_y = sum g_a*silu'(y),  sum_yz = sum g_a*silu'(y)*zhat      (fp64 atomics)
//   2. bn_bwd_gz       g_z = gamma*invstd*(g_y - sum_y/n - zhat*sum_yz/n), in place over g_a;
//                      dgamma += sum_yz, dbeta += sum_y
//   3. data gradient   g_a(in) (=|+=) conv^T(g_z, W)     pw: pw_mfma_kernel<.., WT>; dw: dw_bwd_data
//   4. weight gradient dW += g_z (*) a(in)                pw / dw / stem kernels below (fp32 atomics)
// g buffers mirror the activation buffers (NHWC fp32, same views).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "jn_kernels.h"
#include "jn_reduce.h"
#include "jn_types.h"

namespace jnr {

__device__ __forceinline__ float sigmoidf_(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ float silu_(float v) { return v * sigmoidf_(v); }
__device__ __forceinline__ float dsilu_(float y) {
  const float s = sigmoidf_(y);
  return s * (1.0f + y * (1.0f - s));
}
__device__ __forceinline__ f32x4 tf4_(f32x4 z, f32x4 sc, f32x4 sh, f32x4 fl) {
  f32x4 r;
  r.x = fl.x != 0.0f ? silu_(fmaf(z.x, sc.x, sh.x)) : z.x;
  r.y = fl.y != 0.0f ? silu_(fmaf(z.y, sc.y, sh.y)) : z.y;
  r.z = fl.z != 0.0f ? silu_(fmaf(z.z, sc.z, sh.z)) : z.z;
  r.w = fl.w != 0.0f ? silu_(fmaf(z.w, sc.w, sh.w)) : z.w;
  return r;
}

// ---- 1. per-channel reductions ------------------------------------------------------------
template <typename ZT>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ g, int g_ld,
                                                            const ZT* __restrict__ z, int z_ld, ChanTab t,
                                                            const float* __restrict__ save, int C, long long M,
                                                            int rows_per_block, double* __restrict__ red_out,
                                                            long long rep_stride, SlotBatch sb) {
  extern __shared__ float red[];    // [C][2]
  {
    const long long sl = blockIdx.y;
    g += sl * sb.grad; z += sl * sb.act; save += sl * sb.save; red_out += sl * sb.red;
    t.sc += sl * sb.tab; t.sh += sl * sb.tab;
  }
  for (int i = threadIdx.x; i < 2 * C; i += 256) red[i] = 0.0f;
  __syncthreads();
  const int C4 = C >> 2;
  const int c = (threadIdx.x % C4) * 4;
  const int rstep = 256 / C4;
  const long long r0 = (long long)blockIdx.x * rows_per_block;
  const long long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  const bool active = (int)(threadIdx.x / C4) < rstep;
  if (active) {
    const f32x4 sc = *reinterpret_cast<const f32x4*>(t.sc + c), sh = *reinterpret_cast<const f32x4*>(t.sh + c);
    f32x4 mean, istd;
#pragma unroll
    for (int q = 0; q < 4; ++q) { mean[q] = save[2 * (c + q)]; istd[q] = save[2 * (c + q) + 1]; }
    for (long long m = r0 + threadIdx.x / C4; m < r1; m += rstep) {
      const f32x4 gv = *reinterpret_cast<const f32x4*>(g + m * g_ld + c);
      const f32x4 zv = ld4(z + m * z_ld + c);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float gy = gv[q] * dsilu_(fmaf(zv[q], sc[q], sh[q]));
        s1[q] += gy;
        s2[q] += gy * (zv[q] - mean[q]) * istd[q];
      }
    }
  }
  if ((C4 & (C4 - 1)) == 0) {
    // power-of-two channel groups: lanes l, l + C4, ... hold the same channels -> butterfly, then C4 lanes add
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float a = s1[q], b = s2[q];
      for (int off = C4; off < 64; off <<= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
      if (lane < C4) {
        atomicAdd(&red[2 * (c + q)], a);
        atomicAdd(&red[2 * (c + q) + 1], b);
      }
    }
  } else if (active) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      atomicAdd(&red[2 * (c + q)], s1[q]);
      atomicAdd(&red[2 * (c + q) + 1], s2[q]);
    }
  }
  __syncthreads();
  double* ro = red_out + (blockIdx.x % JN_NREP) * rep_stride;
  for (int i = threadIdx.x; i < 2 * C; i += 256) atomicAdd(&ro[i], (double)red[i]);
}

int launch_bn_bwd_reduce(const float* g, int g_ld, const void* z, int z_dtype, int z_ld, ChanTab t, const float* save,
                         int C, long long M, double* red_out, long long rep_stride, hipStream_t s, const SlotBatch& sb) {
  const int rstep = 256 / (C / 4) > 0 ? 256 / (C / 4) : 1;
  const int rows_per_block = rstep * 32;
  const dim3 grid((unsigned)((M + rows_per_block - 1) / rows_per_block), sb.n);
  if (z_dtype == JN_BF16)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, grid, dim3(256), (size_t)2 * C * sizeof(float), s, g, g_ld,
                       (const bf16_t*)z, z_ld, t, save, C, M, rows_per_block, red_out, rep_stride, sb);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, grid, dim3(256), (size_t)2 * C * sizeof(float), s, g, g_ld,
                       (const float*)z, z_ld, t, save, C, M, rows_per_block, red_out, rep_stride, sb);
  return 0;
}

// ---- 2. per-channel constants, then g_a -> g_z in place --------------------------------------
__global__ void bn_bwd_consts_kernel(const double* __restrict__ red, long long rep_stride, double count,
                                     const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ save,
                                     float* __restrict__ consts, float* __restrict__ g_gamma,
                                     float* __restrict__ g_beta, int C, SlotBatch sb, int raw_moment) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int sl = blockIdx.y;
  const double* rd = red + sl * sb.red;
  const float* sv = save + sl * sb.save;
  float* cs = consts + sl * sb.consts;
  double s1 = 0.0, s2 = 0.0;
  for (int r = 0; r < JN_NREP; ++r) { s1 += rd[r * rep_stride + 2 * c]; s2 += rd[r * rep_stride + 2 * c + 1]; }
  // sums formed by a consumer's fused kernel carry sum gy * y with y = gamma * zhat + beta (the BatchNorm output the
  // kernel evaluates anyway): sum gy * zhat = (sum gy * y - beta * sum gy) / gamma.  The subtraction cancels by |beta| / |gamma|
  // (O(1) for a BatchNorm affine), not by |mean| / std of the conv output as the raw moment sum gy * z did.  gamma == 0: the
  // layer's output is the constant beta, zhat cannot be recovered from y — d gamma of such a channel is reported as 0
  // (its data gradient is exactly 0 either way: k = gamma * invstd).
  if (raw_moment) {
    const double gm = (double)gamma[c];
    s2 = gm != 0.0 ? (s2 - (double)beta[c] * s1) / gm : 0.0;
  }
  cs[3 * c] = (float)(s1 / count);
  cs[3 * c + 1] = (float)(s2 / count);
  cs[3 * c + 2] = gamma[c] * sv[2 * c + 1];
  if (sb.n == 1) { g_beta[c] += (float)s1; g_gamma[c] += (float)s2; }
  else { atomicAdd(&g_beta[c], (float)s1); atomicAdd(&g_gamma[c], (float)s2); }
}

int launch_bn_bwd_consts(const double* red, long long rep_stride, double count, const float* gamma, const float* beta,
                         const float* save, float* consts, float* g_gamma, float* g_beta, int C, hipStream_t s, const SlotBatch& sb,
                         int raw_moment) {
  hipLaunchKernelGGL(bn_bwd_consts_kernel, dim3((C + 63) / 64, sb.n), dim3(64), 0, s, red, rep_stride, count, gamma, beta, save,
                     consts, g_gamma, g_beta, C, sb, raw_moment);
  return 0;
}

template <typename ZT>
__global__ __launch_bounds__(256) void bn_bwd_gz_kernel(float* __restrict__ g, int g_ld, const ZT* __restrict__ z,
                                                        int z_ld, ChanTab t, const float* __restrict__ save,
                                                        const float* __restrict__ consts, int C, long long M,
                                                        SlotBatch sb) {
  {
    const long long sl = blockIdx.y;
    g += sl * sb.grad; z += sl * sb.act; save += sl * sb.save; consts += sl * sb.consts;
    t.sc += sl * sb.tab; t.sh += sl * sb.tab;
  }
  const int C4 = C >> 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * C4) return;
  const int c = (int)(idx % C4) * 4;
  const long long m = idx / C4;
  const f32x4 sc = *reinterpret_cast<const f32x4*>(t.sc + c), sh = *reinterpret_cast<const f32x4*>(t.sh + c);
  const f32x4 gv = *reinterpret_cast<const f32x4*>(g + m * g_ld + c);
  const f32x4 zv = ld4(z + m * z_ld + c);
  f32x4 out;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float mean = save[2 * (c + q)], istd = save[2 * (c + q) + 1];
    const float c1 = consts[3 * (c + q)], c2 = consts[3 * (c + q) + 1], k = consts[3 * (c + q) + 2];
    const float zh = (zv[q] - mean) * istd;
    const float gy = gv[q] * dsilu_(fmaf(zv[q], sc[q], sh[q]));
    out[q] = k * (gy - c1 - zh * c2);
  }
  *reinterpret_cast<f32x4*>(g + m * g_ld + c) = out;
}

int launch_bn_bwd_gz(float* g, int g_ld, const void* z, int z_dtype, int z_ld, ChanTab t, const float* save,
                     const float* consts, int C, long long M, hipStream_t s, const SlotBatch& sb) {
  const long long total = M * (C / 4);
  const dim3 grid((unsigned)((total + 255) / 256), sb.n);
  if (z_dtype == JN_BF16)
    hipLaunchKernelGGL(bn_bwd_gz_kernel<bf16_t>, grid, dim3(256), 0, s, g, g_ld, (const bf16_t*)z, z_ld, t, save, consts, C, M, sb);
  else
    hipLaunchKernelGGL(bn_bwd_gz_kernel<float>, grid, dim3(256), 0, s, g, g_ld, (const float*)z, z_ld, t, save, consts, C, M, sb);
  return 0;
}

// gw[i] += sum_rep wpart[rep][i]; wpart is left zeroed for the next user
__global__ __launch_bounds__(256) void wpart_reduce_kernel(float* __restrict__ gw, float* __restrict__ wpart, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0.0f;
#pragma unroll 8
  for (int r = 0; r < JN_NREP; ++r) { s += wpart[r * JN_WPART_MAX + i]; wpart[r * JN_WPART_MAX + i] = 0.0f; }
  gw[i] += s;
}

int launch_wpart_reduce(float* gw, float* wpart, int n, hipStream_t s) {
  hipLaunchKernelGGL(wpart_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, s, gw, wpart, n);
  return 0;
}

// ---- 4a. pointwise weight gradient: dW[n][k] += sum_m gz[m][n] * T(x[m][k]) ---------------------
// Workgroup = (row chunk, 16*CTN channels of n, 16*CTK channels of k); 64 rows staged per
// iteration in LDS, each wave contracts 16 of them per iteration on v_mfma_f32_16x16x4_f32.
constexpr int WG_RB = 64;

template <int CTN, int CTK, typename XT>
__global__ __launch_bounds__(256) void pw_bwd_weight_kernel(const float* __restrict__ gz, int g_ld,
                                                            const XT* __restrict__ x, int x_ld, ChanTab it,
                                                            float* __restrict__ gw, int rep, long long M, int N,
                                                            int K, int rows_per_block, int chunks_per_slot,
                                                            long long gz_slot, long long x_slot, long long tab_slot) {
  constexpr int LDN = 16 * CTN + 4, LDK = 16 * CTK + 4;
  const int slot = blockIdx.x / chunks_per_slot, chunk = blockIdx.x - slot * chunks_per_slot;
  gz += slot * gz_slot; x += slot * x_slot;
  it.sc += slot * tab_slot; it.sh += slot * tab_slot; it.fl += slot * tab_slot;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Gs = sm;                      // [WG_RB][LDN]
  float* As = sm + WG_RB * LDN;        // [WG_RB][LDK]
  float* Ts = As + WG_RB * LDK;        // [16*CTN][16*CTK] block partial
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.y * 16 * CTN, k0 = blockIdx.z * 16 * CTK;
  const long long r0 = (long long)chunk * rows_per_block;
  const long long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  f32x4 acc[CTN][CTK];
#pragma unroll
  for (int a = 0; a < CTN; ++a)
#pragma unroll
    for (int b = 0; b < CTK; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (long long rb = r0; rb < r1; rb += WG_RB) {
    __syncthreads();
    for (int i = tid; i < WG_RB * 4 * CTN; i += 256) {
      const int r = i / (4 * CTN), q = i % (4 * CTN);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (rb + r < r1 && n0 + 4 * q < N) v = *reinterpret_cast<const f32x4*>(gz + (rb + r) * g_ld + n0 + 4 * q);
      *reinterpret_cast<f32x4*>(Gs + r * LDN + 4 * q) = v;
    }
    for (int i = tid; i < WG_RB * 4 * CTK; i += 256) {
      const int r = i / (4 * CTK), q = i % (4 * CTK);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int kk = k0 + 4 * q;
      if (rb + r < r1 && kk < K)
        v = tf4_(*reinterpret_cast<const f32x4*>(x + (rb + r) * x_ld + kk), *reinterpret_cast<const f32x4*>(it.sc + kk),
                 *reinterpret_cast<const f32x4*>(it.sh + kk), *reinterpret_cast<const f32x4*>(it.fl + kk));
      *reinterpret_cast<f32x4*>(As + r * LDK + 4 * q) = v;
    }
    __syncthreads();
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int row = wave * 16 + 4 * st + g;
      float av[CTN], bv[CTK];
#pragma unroll
      for (int a = 0; a < CTN; ++a) av[a] = Gs[row * LDN + 16 * a + lm];
#pragma unroll
      for (int b = 0; b < CTK; ++b) bv[b] = As[row * LDK + 16 * b + lm];
#pragma unroll
      for (int a = 0; a < CTN; ++a)
#pragma unroll
        for (int b = 0; b < CTK; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
  }
  // cross-wave sum of the four partial tiles: one wave at a time, plain LDS read-modify-write
  // (LDS float atomics cost ~3x the whole kernel here: ds_add_f32 with 4-way bank conflicts)
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int a = 0; a < CTN; ++a)
#pragma unroll
        for (int b = 0; b < CTK; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* tp = &Ts[(16 * a + 4 * g + r) * (16 * CTK) + 16 * b + lm];
            *tp = (w == 0 ? 0.0f : *tp) + acc[a][b][r];
          }
    }
  }
  __syncthreads();
  float* dst = rep ? gw + (blockIdx.x % JN_NREP) * JN_WPART_MAX : gw;
  for (int i = tid; i < 256 * CTN * CTK; i += 256) {
    const int n = n0 + i / (16 * CTK), k = k0 + i % (16 * CTK);
    if (n < N && k < K) atomicAdd(&dst[(long long)n * K + k], Ts[i]);
  }
}

template <int CTN, int CTK>
static void launch_pw_bw_t(const float* gz, int g_ld, const void* x, int x_dtype, int x_ld, ChanTab it, float* gw, int rep,
                           long long M, int N, int K, hipStream_t s, const SlotBatch& sb, long long gz_slot) {
  // aim at ~768 workgroups: enough to fill 256 CUs x 3, few enough that the final atomics stay cheap
  const long long tiles = (long long)((N + 16 * CTN - 1) / (16 * CTN)) * ((K + 16 * CTK - 1) / (16 * CTK));
  long long rpb = (M * sb.n * tiles / 768 + 63) / 64 * 64;
  if (rpb < 64) rpb = 64;
  if (rpb > 4096) rpb = 4096;
  const int rows_per_block = (int)rpb;
  const int chunks_per_slot = (int)((M + rows_per_block - 1) / rows_per_block);
  const long long x_slot = sb.act, tab_slot = sb.tab;
  dim3 grid((unsigned)(chunks_per_slot * sb.n), (N + 16 * CTN - 1) / (16 * CTN), (K + 16 * CTK - 1) / (16 * CTK));
  const size_t smem = ((size_t)WG_RB * (16 * CTN + 4 + 16 * CTK + 4) + 256 * CTN * CTK) * sizeof(float);
  if (x_dtype == JN_BF16)
    hipLaunchKernelGGL((pw_bwd_weight_kernel<CTN, CTK, bf16_t>), grid, dim3(256), smem, s, gz, g_ld, (const bf16_t*)x, x_ld,
                       it, gw, rep, M, N, K, rows_per_block, chunks_per_slot, gz_slot, x_slot, tab_slot);
  else
    hipLaunchKernelGGL((pw_bwd_weight_kernel<CTN, CTK, float>), grid, dim3(256), smem, s, gz, g_ld, (const float*)x, x_ld,
                       it, gw, rep, M, N, K, rows_per_block, chunks_per_slot, gz_slot, x_slot, tab_slot);
}

// Wide layers (N, K >= 128): workgroup tile = 128 x 128 outputs, wave w owns the 64 x 64 quadrant (w & 1, w >> 1) and
// contracts ALL 64 rows of a stage itself (no cross-wave sum).  Against the 64 x 64 tiles above every input row is
// re-read half as often: (N/128 + K/128) instead of (N/64 + K/64) passes over g_z / x.
// SP (the default; JN_WW_EXACT=1 keeps fp32 MFMA): both operands are split into two bf16 terms when they leave LDS
// (x = hi + lo to 2^-17) and each 16 x 16 x 32 block takes three v_mfma_f32_16x16x32_bf16 (hi*hi + hi*lo + lo*hi; lo*lo
// < 2^-16 of the product is dropped) — 48 matrix-pipe cycles instead of the 256 of eight fp32 MFMAs, which turns the
// kernel from matrix-bound (~40 % of the fp32 peak) into a pass at memory speed.  A weight gradient is a leaf: its ~1e-5
// relative error goes to the optimiser and nowhere else (the forward / data-gradient GEMMs stay exact).
// Lane group g of a k-step takes rows 4g + (e & 3) + 16 (e >> 2): any row order is a valid contraction order, and this
// one puts the four groups 16 banks apart (row stride 132 floats).
template <typename XT, bool SP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void pw_bwd_weight_wide_kernel(const float* __restrict__ gz, int g_ld,
                                                                 const XT* __restrict__ x, int x_ld, ChanTab it,
                                                                 float* __restrict__ gw, long long M, int N, int K,
                                                                 int rows_per_block, int chunks_per_slot,
                                                                 long long gz_slot, long long x_slot, long long tab_slot) {
  constexpr int LD = 128 + 4;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Gs = sm;                 // [64][LD]
  float* As = sm + 64 * LD;       // [64][LD]
  const int slot = blockIdx.x / chunks_per_slot, chunk = blockIdx.x - slot * chunks_per_slot;
  gz += slot * gz_slot; x += slot * x_slot;
  it.sc += slot * tab_slot; it.sh += slot * tab_slot; it.fl += slot * tab_slot;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.y * 128, k0 = blockIdx.z * 128;
  const int wn = (wave & 1) * 64, wk = (wave >> 1) * 64;
  const long long r0 = (long long)chunk * rows_per_block;
  const long long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the channel quad of a thread is fixed (256 % 32 == 0)
  const int q = tid & 31;
  f32x4 i_sc = {1.f, 1.f, 1.f, 1.f}, i_sh = {0.f, 0.f, 0.f, 0.f}, i_fl = {0.f, 0.f, 0.f, 0.f};
  if (k0 + 4 * q < K) {
    i_sc = *reinterpret_cast<const f32x4*>(it.sc + k0 + 4 * q); i_sh = *reinterpret_cast<const f32x4*>(it.sh + k0 + 4 * q);
    i_fl = *reinterpret_cast<const f32x4*>(it.fl + k0 + 4 * q);
  }
  // the next 64-row block is fetched into registers while the MFMAs of the current one run (two workgroups per CU
  // alone left the matrix pipe 50 % busy)
  f32x4 pg[8], px[8];
  auto fetch = [&](long long rb) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = (tid >> 5) + 8 * j;
      pg[j] = f32x4{0.f, 0.f, 0.f, 0.f}; px[j] = pg[j];
      if (rb + r < r1) {
        if (n0 + 4 * q < N) pg[j] = *reinterpret_cast<const f32x4*>(gz + (rb + r) * g_ld + n0 + 4 * q);
        if (k0 + 4 * q < K) px[j] = ld4(x + (rb + r) * x_ld + k0 + 4 * q);
      }
    }
  };
  if (r0 < r1) fetch(r0);
  for (long long rb = r0; rb < r1; rb += 64) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int r = (tid >> 5) + 8 * j;
      f32x4 u = {0.f, 0.f, 0.f, 0.f};
      if (rb + r < r1 && k0 + 4 * q < K) u = tf4_(px[j], i_sc, i_sh, i_fl);
      *reinterpret_cast<f32x4*>(Gs + r * LD + 4 * q) = pg[j];
      *reinterpret_cast<f32x4*>(As + r * LD + 4 * q) = u;
    }
    __syncthreads();
    if (rb + 64 < r1) fetch(rb + 64);
    if constexpr (SP) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const float* gs = Gs + (ks * 32 + 4 * g) * LD + wn + lm;
        const float* as = As + (ks * 32 + 4 * g) * LD + wk + lm;
        bf16x8 bh[4], bl[4];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float v = as[((e & 3) + 16 * (e >> 2)) * LD + 16 * b];
            const bf16_t h = (bf16_t)v;
            bh[b][e] = h; bl[b][e] = (bf16_t)(v - (float)h);
          }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          bf16x8 ah, al;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float v = gs[((e & 3) + 16 * (e >> 2)) * LD + 16 * a];
            const bf16_t h = (bf16_t)v;
            ah[e] = h; al[e] = (bf16_t)(v - (float)h);
          }
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[b], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[b], acc[a][b], 0, 0, 0);
          }
        }
      }
    } else {
#pragma unroll 4
      for (int st = 0; st < 16; ++st) {
        const int row = 4 * st + g;
        float av[4], bv[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) av[a] = Gs[row * LD + wn + 16 * a + lm];
#pragma unroll
        for (int b = 0; b < 4; ++b) bv[b] = As[row * LD + wk + 16 * b + lm];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], acc[a][b], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn + 16 * a + 4 * g + r, k = k0 + wk + 16 * b + lm;
        if (n < N && k < K) atomicAdd(&gw[(long long)n * K + k], acc[a][b][r]);
      }
}

static void launch_pw_bwd_weight_wide(const float* gz, int g_ld, const void* x, int x_dtype, int x_ld, ChanTab it, float* gw,
                                      long long M, int N, int K, hipStream_t s, const SlotBatch& sb, long long gz_slot) {
  const long long tiles = (long long)((N + 127) / 128) * ((K + 127) / 128);
  long long rpb = (M * sb.n * tiles / 1024 + 63) / 64 * 64;     // ~1024 workgroups (2 fit per CU)
  if (rpb < 256) rpb = 256;
  if (rpb > 8192) rpb = 8192;
  const int chunks_per_slot = (int)((M + rpb - 1) / rpb);
  dim3 grid((unsigned)(chunks_per_slot * sb.n), (N + 127) / 128, (K + 127) / 128);
  const size_t smem = (size_t)2 * 64 * (128 + 4) * sizeof(float);
  static const bool exact = std::getenv("JN_WW_EXACT") != nullptr;
#define JN_WW(T_, SP_)                                                                                                \
  {                                                                                                                   \
    static bool raised = false;                                                                                       \
    if (!raised) {                                                                                                    \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_bwd_weight_wide_kernel<T_, SP_>),                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                               \
      raised = true;                                                                                                  \
    }                                                                                                                 \
    hipLaunchKernelGGL((pw_bwd_weight_wide_kernel<T_, SP_>), grid, dim3(256), smem, s, gz, g_ld, (const T_*)x, x_ld, it, gw, M, \
                       N, K, (int)rpb, chunks_per_slot, gz_slot, sb.act, sb.tab);                                     \
  }
  if (x_dtype == JN_BF16) { if (exact) JN_WW(bf16_t, false) else JN_WW(bf16_t, true) }
  else { if (exact) JN_WW(float, false) else JN_WW(float, true) }
#undef JN_WW
}

int launch_pw_bwd_weight(const float* gz, int g_ld, const void* x, int x_dtype, int x_ld, ChanTab it, float* gw_final,
                         float* wpart, long long M, int N, int K, hipStream_t s, const SlotBatch& sb,
                         long long gz_slot_stride) {
  const long long gz_slot = gz_slot_stride >= 0 ? gz_slot_stride : sb.grad;
  if (N >= 128 && K >= 128 && N % 4 == 0 && K % 4 == 0) {
    launch_pw_bwd_weight_wide(gz, g_ld, x, x_dtype, x_ld, it, gw_final, M, N, K, s, sb, gz_slot);
    return 0;
  }
  const int rep = (wpart && N * K <= JN_WPART_MAX) ? 1 : 0;
  float* gw = rep ? wpart : gw_final;
  const int tn = (N + 15) / 16, tk = (K + 15) / 16;
  const int cn = tn >= 4 ? 4 : (tn == 3 ? 3 : tn), ck = tk >= 4 ? 4 : (tk == 3 ? 3 : tk);
#define JN_BW(A, B) if (cn == A && ck == B) { launch_pw_bw_t<A, B>(gz, g_ld, x, x_dtype, x_ld, it, gw, rep, M, N, K, s, sb, gz_slot); \
    if (rep) launch_wpart_reduce(gw_final, wpart, N * K, s); return 0; }
  JN_BW(1, 1) JN_BW(1, 2) JN_BW(1, 3) JN_BW(1, 4) JN_BW(2, 1) JN_BW(2, 2) JN_BW(2, 3) JN_BW(2, 4)
  JN_BW(3, 1) JN_BW(3, 2) JN_BW(3, 3) JN_BW(3, 4) JN_BW(4, 1) JN_BW(4, 2) JN_BW(4, 3) JN_BW(4, 4)
#undef JN_BW
  return -1;
}

// ---- 2+3+4 fused for pointwise layers with few channels ------------------------------------------
// One pass over (g_a, z) of the layer output and z of the layer input does the whole layer:
//   g_z = k * (g_a * silu'(y) - c1 - zhat * c2)      (never written to HBM)
//   g_in (=|+=) g_z . W                              (phase 2: the forward kernel's MFMA loop with X = g_z, W^T)
//   dW  += g_z^T . a_in                              (phase 3: contraction over the 64 pixels of the tile)
// HBM traffic 8 B/output + 8 B/input element instead of 20 + 8 for the bn_bwd_gz / data / weight kernels.
// Workgroup = 4 waves, 64 pixels per iteration, persistent over the tiles of ONE slot (blockIdx.y) so the
// per-channel constants sit in registers and dW in accumulators; Cout = 16*CTN, Cin = 16*CTK, CTN*CTK <= 16.
// RED: the layer input is the output of ONE BatchNorm conv with no other consumer: the per-channel sums of ITS backward
// (bn_bwd_reduce: sum g_a*silu'(y), sum g_a*silu'(y)*zhat) are accumulated here from the data gradient still in the
// MFMA accumulators (z of the input tile is re-read, L2-hot), so that layer's reduce pass over (g, z) never runs.
// RED2 (round 4; round 3 had it inside every RED instantiation, where its operands cost all of them a wave of occupancy:
// now its own instantiation): the first run of the input is a materialised shortcut sum silu(bn(z2)) + res; the kernel
// writes the sum's FINAL gradient, which is also the gradient of the activation of z2 — the bottleneck's last pointwise
// conv — so the sums of that run are formed with z2 / its table and go to THAT conv's BatchNorm.
template <int CTN, int CTK, bool RED, bool RED2 = false>
__global__ __launch_bounds__(256) void pw_bwd_fused_kernel(
    const float* __restrict__ g, int g_ld, const float* __restrict__ z, int z_ld, ChanTab ot,
    const float* __restrict__ save, const float* __restrict__ consts, const float* __restrict__ x, int x_ld,
    ChanTab it, const float* __restrict__ w, float* __restrict__ gx, int gx_ld, int accumulate,
    float* __restrict__ gw, int rep, long long M, SlotBatch sb, double* __restrict__ red_in, long long red_rep_stride,
    const float* __restrict__ gadd, int gadd_ld, double* __restrict__ red_in2, int red_split,
    const float* __restrict__ red2_z, int red2_ld, const float* __restrict__ red2_sc, const float* __restrict__ red2_sh) {
  constexpr int N = 16 * CTN, K = 16 * CTK;          // output / input channels
  constexpr int LDG = N + 4, LDA = K + 4, LDW = N + 4;
  constexpr int NG = 64 * (N / 4) / 256, NA = 64 * (K / 4) / 256;     // f32x4 per thread per tile (may be 0 -> 1)
  constexpr int NGq = NG > 0 ? NG : 1, NAq = NA > 0 ? NA : 1;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Gs = sm;                       // [64][LDG]  g_z tile
  float* As = Gs + 64 * LDG;            // [64][LDA]  activated input tile
  float* Wt = As + 64 * LDA;            // [K][LDW]   W^T (cin-major, cout contiguous)
  float* Cs = Wt + K * LDW;             // [7][N] output-side constants, [3][K] input table
  float* Ts = sm;                       // [N][K]     cross-wave dW sum; reuses the tile space after the loop
  {
    const long long sl = blockIdx.y;
    g += sl * sb.grad; z += sl * sb.act; x += sl * sb.act; gx += sl * sb.grad;
    save += sl * sb.save; consts += sl * sb.consts;
    if (RED) { if (red_in) red_in += sl * sb.red; if (red_in2) red_in2 += sl * sb.red; }
    if (gadd) gadd += sl * sb.grad;
    if constexpr (RED2) { red2_z += sl * sb.act; red2_sc += sl * sb.tab; red2_sh += sl * sb.tab; }
    ot.sc += sl * sb.tab; ot.sh += sl * sb.tab;
    it.sc += sl * sb.tab; it.sh += sl * sb.tab; it.fl += sl * sb.tab;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, gq = lane >> 4;
  for (int i = tid; i < N * K; i += 256) {           // w is [N][K] row-major
    const int n = i / K, k = i - n * K;
    Wt[k * LDW + n] = w[i];
  }
  // per-thread channel quads of the staging loops never change (256 % (N/4) == 0, 256 % (K/4) == 0); the per-channel
  // constants live in LDS (7 x N + 3 x K floats) rather than in 40 registers: with them in registers the 64x64 layers
  // sat at one wave per SIMD
  const int nq = tid % (N / 4), kq = tid % (K / 4);
  for (int c = tid; c < N; c += 256) {
    Cs[c] = ot.sc[c]; Cs[N + c] = ot.sh[c]; Cs[2 * N + c] = save[2 * c]; Cs[3 * N + c] = save[2 * c + 1];
    Cs[4 * N + c] = consts[3 * c]; Cs[5 * N + c] = consts[3 * c + 1]; Cs[6 * N + c] = consts[3 * c + 2];
  }
  for (int c = tid; c < K; c += 256) { Cs[7 * N + c] = it.sc[c]; Cs[7 * N + K + c] = it.sh[c]; Cs[7 * N + 2 * K + c] = it.fl[c]; }
  if (RED) {
    for (int c = tid; c < 8 * K; c += 256) Cs[7 * N + 3 * K + c] = 0.0f;       // [4 waves][K][2] input-layer BN sums
    if constexpr (RED2)                                                         // table of the BatchNorm behind the shortcut sum
      for (int c = tid; c < K; c += 256) {
        const bool in = c < red_split;
        Cs[7 * N + 11 * K + c] = in ? red2_sc[c] : 1.0f; Cs[7 * N + 12 * K + c] = in ? red2_sh[c] : 0.0f;
      }
  }
  // weight-gradient tiles of this wave: all of them (fp32 form) or one half, split along cin when CTK is even, else cout
#ifdef JN_FUSED_DW_EXACT
  constexpr bool DW_SPLIT = false;
#else
  constexpr bool DW_SPLIT = (CTK % 2 == 0) || (CTN % 2 == 0);
#endif
  constexpr bool DW_HALF_K = DW_SPLIT && CTK % 2 == 0;
  constexpr int DW_AN = !DW_SPLIT ? CTN : (DW_HALF_K ? CTN : CTN / 2), DW_BN = !DW_SPLIT ? CTK : (DW_HALF_K ? CTK / 2 : CTK);
  const int dw_a0 = (DW_SPLIT && !DW_HALF_K) ? (wave & 1) * DW_AN : 0, dw_b0 = DW_HALF_K ? (wave & 1) * DW_BN : 0;
  f32x4 dw[DW_AN][DW_BN];
#pragma unroll
  for (int a = 0; a < DW_AN; ++a)
#pragma unroll
    for (int b = 0; b < DW_BN; ++b) dw[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const long long n_tiles = (M + 63) / 64;
  f32x4 rg[NGq], rz[NGq], rx[NAq];
  auto fetch = [&](long long m0) {
#pragma unroll
    for (int j = 0; j < NGq; ++j) {
      const int i = tid + 256 * j, r = i / (N / 4);
      rg[j] = f32x4{0.f, 0.f, 0.f, 0.f}; rz[j] = rg[j];
      if (i < 64 * (N / 4) && m0 + r < M) {
        rg[j] = *reinterpret_cast<const f32x4*>(g + (m0 + r) * g_ld + 4 * nq);
        rz[j] = *reinterpret_cast<const f32x4*>(z + (m0 + r) * z_ld + 4 * nq);
      }
    }
#pragma unroll
    for (int j = 0; j < NAq; ++j) {
      const int i = tid + 256 * j, r = i / (K / 4);
      rx[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < 64 * (K / 4) && m0 + r < M) rx[j] = *reinterpret_cast<const f32x4*>(x + (m0 + r) * x_ld + 4 * kq);
    }
  };
  auto stage = [&](long long m0) {
#pragma unroll
    for (int j = 0; j < NGq; ++j) {
      const int i = tid + 256 * j, r = i / (N / 4);
      if (i < 64 * (N / 4)) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m0 + r < M) {
          const f32x4 o_sc = *reinterpret_cast<const f32x4*>(Cs + 4 * nq), o_sh = *reinterpret_cast<const f32x4*>(Cs + N + 4 * nq),
                      o_mean = *reinterpret_cast<const f32x4*>(Cs + 2 * N + 4 * nq), o_istd = *reinterpret_cast<const f32x4*>(Cs + 3 * N + 4 * nq),
                      o_c1 = *reinterpret_cast<const f32x4*>(Cs + 4 * N + 4 * nq), o_c2 = *reinterpret_cast<const f32x4*>(Cs + 5 * N + 4 * nq),
                      o_k = *reinterpret_cast<const f32x4*>(Cs + 6 * N + 4 * nq);
#ifdef JN_DBG_NO_STAGE_MATH            // (JN_DBG_*: attribution builds of tools/bwd_attrib.sh, never in the product library)
          v = rg[j] + rz[j] * o_k + o_sc + o_sh + o_mean + o_istd + o_c1 + o_c2;
#else
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float zh = (rz[j][q] - o_mean[q]) * o_istd[q];
            const float gy = rg[j][q] * dsilu_(fmaf(rz[j][q], o_sc[q], o_sh[q]));
            v[q] = o_k[q] * (gy - o_c1[q] - zh * o_c2[q]);
          }
#endif
        }
        *reinterpret_cast<f32x4*>(Gs + r * LDG + 4 * nq) = v;
      }
    }
#pragma unroll
    for (int j = 0; j < NAq; ++j) {
      const int i = tid + 256 * j, r = i / (K / 4);
      if (i < 64 * (K / 4)) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (m0 + r < M) {
          if constexpr (RED) v = rx[j];     // raw z: activated where phase 3 reads it (each element once), z itself feeds the RED sums
          else v = tf4_(rx[j], *reinterpret_cast<const f32x4*>(Cs + 7 * N + 4 * kq), *reinterpret_cast<const f32x4*>(Cs + 7 * N + K + 4 * kq),
                        *reinterpret_cast<const f32x4*>(Cs + 7 * N + 2 * K + 4 * kq));
        }
        *reinterpret_cast<f32x4*>(As + r * LDA + 4 * kq) = v;
      }
    }
  };

  float p_sc[RED ? CTK : 1], p_sh[RED ? CTK : 1], p_fl[RED ? CTK : 1];      // input table of channels 16 b + lm (phase 3)
  if constexpr (RED) {
#pragma unroll
    for (int b = 0; b < CTK; ++b) { p_sc[b] = it.sc[16 * b + lm]; p_sh[b] = it.sh[16 * b + lm]; p_fl[b] = it.fl[16 * b + lm]; }
  }
  long long tile = blockIdx.x;
  if (tile < n_tiles) fetch(tile * 64);
  for (; tile < n_tiles; tile += gridDim.x) {
    const long long m0 = tile * 64;
    __syncthreads();                                  // previous tile's readers are done (and Wt is in place)
    stage(m0);
    __syncthreads();
    if (tile + gridDim.x < n_tiles) fetch((tile + gridDim.x) * 64);
    // ---- phase 2: g_in[pixel][cin] = sum_cout g_z[pixel][cout] * W[cout][cin]; D = W^T . G^T, wave = 16 pixels
    {
      f32x4 acc[CTK];
#pragma unroll
      for (int b = 0; b < CTK; ++b) acc[b] = f32x4{0.f, 0.f, 0.f, 0.f};
      // RED2: the raw output of the conv behind the shortcut sum, in the layout of the accumulators (this lane's pixel,
      // channels 16 b + 4 gq), requested BEFORE the matrix loop that hides its latency
      f32x4 z2v[RED2 ? CTK : 1];
      if constexpr (RED2) {
        const long long mz = m0 + wave * 16 + lm;
#pragma unroll
        for (int b = 0; b < CTK; ++b) {
          z2v[b] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (16 * b < red_split && mz < M) z2v[b] = *reinterpret_cast<const f32x4*>(red2_z + mz * red2_ld + 16 * b + 4 * gq);
        }
      }
      const float* grow = Gs + (wave * 16 + lm) * LDG + 4 * gq;
      const float* wrow = Wt + lm * LDW + 4 * gq;
#ifndef JN_DBG_NO_P2
#pragma unroll
      for (int kk = 0; kk < N; kk += 16) {
        const f32x4 gb = *reinterpret_cast<const f32x4*>(grow + kk);
#pragma unroll
        for (int b = 0; b < CTK; ++b) {
          const f32x4 wa = *reinterpret_cast<const f32x4*>(wrow + b * 16 * LDW + kk);
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j], gb[j], acc[b], 0, 0, 0);
        }
      }
#else
      acc[0][0] = grow[0] + wrow[0];
#endif
      const long long m = m0 + wave * 16 + lm;
      if (m < M) {
#pragma unroll
        for (int b = 0; b < CTK; ++b) {
          float* op = gx + m * gx_ld + 16 * b + 4 * gq;
          f32x4 v = acc[b];
          // shortcut of a bottleneck: the gradient of the sum joins here (the shortcut add's own backward pass is gone)
          if (gadd) v += *reinterpret_cast<const f32x4*>(gadd + m * gadd_ld + 16 * b + 4 * gq);
          if (accumulate) v += *reinterpret_cast<const f32x4*>(op);
          *reinterpret_cast<f32x4*>(op) = v;
          acc[b] = v;                                  // RED below sums the FINAL gradient of the input
        }
      }
#ifdef JN_DBG_NO_RED
      if constexpr (false) {
#else
      if constexpr (RED) {
#endif
        // this wave's 16 pixels: row sums by DPP, accumulated in the wave's own LDS slots (plain read-modify-write by
        // the row leaders) -- per-lane sums held in registers across the tiles cost a wave of occupancy
        float* rslot = Cs + 7 * N + 3 * K + wave * 2 * K;
#pragma unroll
        for (int b = 0; b < CTK; ++b) {
          const int ch = 16 * b + 4 * gq;
          f32x4 zv = *reinterpret_cast<const f32x4*>(As + (wave * 16 + lm) * LDA + ch);      // raw z of this pixel
          f32x4 i_sc = *reinterpret_cast<const f32x4*>(Cs + 7 * N + ch), i_sh = *reinterpret_cast<const f32x4*>(Cs + 7 * N + K + ch);
          if constexpr (RED2) if (16 * b < red_split) {  // (uniform) the sums of this run go to the conv behind the shortcut sum
            zv = z2v[b];
            i_sc = *reinterpret_cast<const f32x4*>(Cs + 7 * N + 11 * K + ch); i_sh = *reinterpret_cast<const f32x4*>(Cs + 7 * N + 12 * K + ch);
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float yv = fmaf(zv[q], i_sc[q], i_sh[q]);
            const float gy = m < M ? acc[b][q] * dsilu_(yv) : 0.0f;
            const float a1 = row16_sum(gy);
            // second moment against the NORMALISED value y = gamma * zhat + beta (bn_bwd_consts turns it into sum gy * zhat):
            // the raw moment sum gy * z cancels against mean * sum gy there, and its fp32 rounding comes back amplified by
            // |mean| / std of the producer's output — 1e2 .. 1e3 on the near-constant maps of a fresh network
            const float a2 = row16_sum(gy * yv);
            if (lm == 0) { rslot[2 * (ch + q)] += a1; rslot[2 * (ch + q) + 1] += a2; }
          }
        }
      }
    }
    // ---- phase 3: dW[cout][cin] += sum_pixel g_z[pixel][cout] * a[pixel][cin]
#ifdef JN_DBG_NO_P3
    if constexpr (true) {
    } else if constexpr (DW_SPLIT) {
#else
    if constexpr (DW_SPLIT) {
#endif
      // split-bf16 products (a weight gradient is a leaf, see pw_bwd_weight_wide_kernel): one 32-pixel k-step per wave —
      // waves 0, 1 take pixels 0..31, waves 2, 3 pixels 32..63 — and half of the output tiles each, so a wave holds
      // HALF the accumulators of the fp32 form.  Lane group gq, element e -> pixel 4 gq + (e & 3) + 16 (e >> 2).
      const int prow = 32 * (wave >> 1) + 4 * gq;
      bf16x8 bh[DW_BN], bl[DW_BN];
#pragma unroll
      for (int b = 0; b < DW_BN; ++b)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = As[(prow + (e & 3) + 16 * (e >> 2)) * LDA + 16 * (dw_b0 + b) + lm];
          if constexpr (RED) v = p_fl[dw_b0 + b] != 0.0f ? silu_(fmaf(v, p_sc[dw_b0 + b], p_sh[dw_b0 + b])) : v;
          const bf16_t h = (bf16_t)v;
          bh[b][e] = h; bl[b][e] = (bf16_t)(v - (float)h);
        }
#pragma unroll
      for (int a = 0; a < DW_AN; ++a) {
        bf16x8 ah, al;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float v = Gs[(prow + (e & 3) + 16 * (e >> 2)) * LDG + 16 * (dw_a0 + a) + lm];
          const bf16_t h = (bf16_t)v;
          ah[e] = h; al[e] = (bf16_t)(v - (float)h);
        }
#pragma unroll
        for (int b = 0; b < DW_BN; ++b) {
          f32x4 d = dw[a][b];
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[b], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[b], d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[b], d, 0, 0, 0);
          dw[a][b] = d;
        }
      }
    } else {
      // fp32 MFMA: wave = 16 pixels = 4 k-steps, every output tile in every wave
#pragma unroll
      for (int st = 0; st < 4; ++st) {
        const int row = wave * 16 + 4 * st + gq;
        float av[CTN], bv[CTK];
#pragma unroll
        for (int a = 0; a < CTN; ++a) av[a] = Gs[row * LDG + 16 * a + lm];
#pragma unroll
        for (int b = 0; b < CTK; ++b) {
          bv[b] = As[row * LDA + 16 * b + lm];
          if constexpr (RED) bv[b] = p_fl[b] != 0.0f ? silu_(fmaf(bv[b], p_sc[b], p_sh[b])) : bv[b];
        }
#pragma unroll
        for (int a = 0; a < CTN; ++a)
#pragma unroll
          for (int b = 0; b < CTK; ++b) dw[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[b], dw[a][b], 0, 0, 0);
      }
    }
  }
  if constexpr (RED) {
    // per-channel sums of the input layer's BN backward: the four waves' LDS slots -> fp64 atomics
    __syncthreads();
    // (two channel runs: [0, red_split) -> red_in, [red_split, K) -> red_in2 — the input of a CSP's conv3 is
    //  [bottleneck output | conv2 half of the pair]; either base may be null: that run's producer has no BatchNorm or
    //  another reader)
    const float* rs = Cs + 7 * N + 3 * K;
    if (tid < 2 * K) {
      const bool first = tid < 2 * red_split;
      double* base = first ? red_in : red_in2;
      if (base)
        atomicAdd(&base[(blockIdx.x % JN_NREP) * red_rep_stride + (first ? tid : tid - 2 * red_split)],
                  (double)(rs[tid] + rs[2 * K + tid] + rs[4 * K + tid] + rs[6 * K + tid]));
    }
  }
  // cross-wave sum of dW (plain LDS read-modify-write, one round per wave that shares tiles), then one set of atomics
  if constexpr (DW_SPLIT) {
    for (int ph = 0; ph < 2; ++ph) {                 // waves (0, 1) hold disjoint tiles, (2, 3) the same two sets
      __syncthreads();
      if ((wave >> 1) == ph) {
#pragma unroll
        for (int a = 0; a < DW_AN; ++a)
#pragma unroll
          for (int b = 0; b < DW_BN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float* tp = &Ts[(16 * (dw_a0 + a) + 4 * gq + r) * K + 16 * (dw_b0 + b) + lm];
              *tp = (ph == 0 ? 0.0f : *tp) + dw[a][b][r];
            }
      }
    }
  } else {
    for (int wv = 0; wv < 4; ++wv) {
      __syncthreads();
      if (wave == wv) {
#pragma unroll
        for (int a = 0; a < DW_AN; ++a)
#pragma unroll
          for (int b = 0; b < DW_BN; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float* tp = &Ts[(16 * a + 4 * gq + r) * K + 16 * b + lm];
              *tp = (wv == 0 ? 0.0f : *tp) + dw[a][b][r];
            }
      }
    }
  }
  __syncthreads();
  float* dst = rep ? gw + ((blockIdx.x + 5 * blockIdx.y) % JN_NREP) * JN_WPART_MAX : gw;
  for (int i = tid; i < N * K; i += 256) atomicAdd(&dst[i], Ts[i]);
}

template <int CTN, int CTK, bool RED, bool RED2 = false>
static void launch_pw_bwd_fused_r(const PwBwdFusedArgs& a, hipStream_t s) {
  constexpr int N = 16 * CTN, K = 16 * CTK;
  size_t smem = ((size_t)64 * (N + 4) + 64 * (K + 4) + (size_t)K * (N + 4) + 7 * N + 3 * K + (RED ? (RED2 ? 10 : 8) * K : 0)) * sizeof(float);
  if (smem < (size_t)N * K * sizeof(float)) smem = (size_t)N * K * sizeof(float);
  if (smem > 64 * 1024) {
    static bool raised = false;       // per instantiation
    if (!raised) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_bwd_fused_kernel<CTN, CTK, RED, RED2>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      raised = true;
    }
  }
  const long long n_tiles = (a.M + 63) / 64;
  // ~1024 persistent workgroups over all slots, in WHOLE resident rounds: an instantiation that fits 3 workgroups per CU
  // (768 places) ran 1020 of them, i.e. a second round at a third of the occupancy
  static int places = 0;
  if (!places) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&pw_bwd_fused_kernel<CTN, CTK, RED, RED2>), 256, smem) != hipSuccess || per_cu < 1) per_cu = 2;
    places = per_cu * 256;
  }
  const int rounds = std::max(1, (1024 + places / 2) / places);
  long long bx = (long long)rounds * places / a.sb.n;
  if (bx < 1) bx = 1;
  if (bx > n_tiles) bx = n_tiles;
  const int rep = (a.wpart && N * K <= JN_WPART_MAX) ? 1 : 0;
  hipLaunchKernelGGL((pw_bwd_fused_kernel<CTN, CTK, RED, RED2>), dim3((unsigned)bx, a.sb.n), dim3(256), smem, s, a.g, a.g_ld, a.z,
                     a.z_ld, a.ot, a.save, a.consts, a.x, a.x_ld, a.it, a.w, a.gx, a.gx_ld, a.accumulate,
                     rep ? a.wpart : a.gw, rep, a.M, a.sb, a.red_in, a.red_rep_stride, a.gadd, a.gadd_ld, a.red_in2,
                     a.red_split > 0 ? a.red_split : K, a.red2_z, a.red2_ld, a.red2_sc, a.red2_sh);
  if (rep) launch_wpart_reduce(a.gw, a.wpart, N * K, s);
}

template <int CTN, int CTK>
static void launch_pw_bwd_fused_t(const PwBwdFusedArgs& a, hipStream_t s) {
  if constexpr (CTK <= 4) {
    if (a.red2_z) { launch_pw_bwd_fused_r<CTN, CTK, true, true>(a, s); return; }
    if (a.red_in || a.red_in2) { launch_pw_bwd_fused_r<CTN, CTK, true>(a, s); return; }
  }
  launch_pw_bwd_fused_r<CTN, CTK, false>(a, s);
}

bool pw_bwd_fused_reduces_input(int cout, int cin) { return pw_bwd_fused_supported(cout, cin) && cin <= 64; }   // (cin = 128 measured: the epilogue of the one-workgroup-per-CU kernels costs more than the 28x28 passes it saves, profiles/r04_ab_red128.txt)

bool pw_bwd_fused_supported(int cout, int cin) {
  if (cout % 16 || cin % 16) return false;
  const int tn = cout / 16, tk = cin / 16;
  auto pow2 = [](int v) { return v == 1 || v == 2 || v == 4 || v == 8; };
  static const bool no_wide = std::getenv("JN_NO_FUSED_BWD_WIDE") != nullptr;     // 128 x 128: round 3 (one workgroup per CU)
  return pow2(tn) && pow2(tk) && (tn * tk <= 32 || (!no_wide && tn == 8 && tk == 8));
}

int launch_pw_bwd_fused(const PwBwdFusedArgs& a, hipStream_t s) {
  const int tn = a.cout / 16, tk = a.cin / 16;
#define JN_PF(A, B) if (tn == A && tk == B) { launch_pw_bwd_fused_t<A, B>(a, s); return 0; }
  JN_PF(1, 1) JN_PF(1, 2) JN_PF(1, 4) JN_PF(1, 8) JN_PF(2, 1) JN_PF(2, 2) JN_PF(2, 4) JN_PF(2, 8)
  JN_PF(4, 1) JN_PF(4, 2) JN_PF(4, 4) JN_PF(4, 8) JN_PF(8, 1) JN_PF(8, 2) JN_PF(8, 4) JN_PF(8, 8)
#undef JN_PF
  return -1;
}

// ---- 3b. depthwise data gradient ----------------------------------------------------------------
// g_in[iy][ix][c] (=|+=) sum_{ky,kx} gz[(iy+1-ky)/S][(ix+1-kx)/S][c] * w[ky][kx][c]   (where divisible)
template <int S>
__global__ __launch_bounds__(256) void dw_bwd_data_kernel(const float* __restrict__ gz, int g_ld,
                                                          const float* __restrict__ w, float* __restrict__ gin,
                                                          int gin_ld, int C, int H, int W, int OH, int OW, int N,
                                                          int accumulate, long long g_slot) {
  gz += blockIdx.y * g_slot; gin += blockIdx.y * g_slot;
  const int C4 = C >> 2;
  const long long total = (long long)N * H * W * C4;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C4) * 4;
  const int ix = (int)((idx / C4) % W);
  const int iy = (int)((idx / ((long long)C4 * W)) % H);
  const long long n = idx / ((long long)C4 * W * H);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int ty = iy + 1 - ky;
    if (ty < 0 || (S == 2 && (ty & 1))) continue;
    const int oy = ty / S;
    if (oy >= OH) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int tx = ix + 1 - kx;
      if (tx < 0 || (S == 2 && (tx & 1))) continue;
      const int ox = tx / S;
      if (ox >= OW) continue;
      acc += *reinterpret_cast<const f32x4*>(gz + ((n * OH + oy) * OW + ox) * g_ld + c) *
             *reinterpret_cast<const f32x4*>(w + (ky * 3 + kx) * C + c);
    }
  }
  float* op = gin + ((n * H + iy) * W + ix) * gin_ld + c;
  if (accumulate) acc += *reinterpret_cast<const f32x4*>(op);
  *reinterpret_cast<f32x4*>(op) = acc;
}

int launch_dw_bwd_data(const float* gz, int g_ld, const float* w, float* gin, int gin_ld, int C, int H, int W, int OH,
                       int OW, int N, int stride, int accumulate, hipStream_t s, const SlotBatch& sb) {
  const long long total = (long long)N * H * W * (C / 4);
  const dim3 blocks((unsigned)((total + 255) / 256), sb.n);
  if (stride == 1)
    hipLaunchKernelGGL(dw_bwd_data_kernel<1>, blocks, dim3(256), 0, s, gz, g_ld, w, gin, gin_ld, C, H, W, OH, OW, N,
                       accumulate, sb.grad);
  else
    hipLaunchKernelGGL(dw_bwd_data_kernel<2>, blocks, dim3(256), 0, s, gz, g_ld, w, gin, gin_ld, C, H, W, OH, OW, N,
                       accumulate, sb.grad);
  return 0;
}

// ---- 4b. depthwise weight gradient: dW[tap][c] += sum gz[oy][ox][c] * T(x)[oy*S-1+ky][ox*S-1+kx][c]
template <int S, typename XT>
__global__ __launch_bounds__(256) void dw_bwd_weight_kernel(const float* __restrict__ gz, int g_ld,
                                                            const XT* __restrict__ x, int x_ld, ChanTab it,
                                                            float* __restrict__ gw, int rep, int C, int H, int W,
                                                            int OH, int OW, int N, SlotBatch sb) {
  extern __shared__ float red[];   // [9][C]
  {
    const long long sl = blockIdx.y;
    gz += sl * sb.grad; x += sl * sb.act;
    it.sc += sl * sb.tab; it.sh += sl * sb.tab; it.fl += sl * sb.tab;
  }
  for (int i = threadIdx.x; i < 9 * C; i += 256) red[i] = 0.0f;
  __syncthreads();
  const int C4 = C >> 2;
  const int YS = (OH + 3) >> 2;
  const long long total = (long long)N * YS * OW * C4;
  const long long idx0 = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long gstride = (long long)gridDim.x * 256;          // multiple of C4: the channel group stays fixed
  {
    const int c = (int)(idx0 % C4) * 4;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(it.sc + c), sh = *reinterpret_cast<const f32x4*>(it.sh + c),
                fl = *reinterpret_cast<const f32x4*>(it.fl + c);
    f32x4 dw[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) dw[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (long long idx = idx0; idx < total; idx += gstride) {
    const int ox = (int)((idx / C4) % OW);
    const int ys = (int)((idx / ((long long)C4 * OW)) % YS);
    const long long n = idx / ((long long)C4 * OW * YS);
    f32x4 gv[4];
    const int oy0 = ys * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      gv[j] = (oy0 + j < OH) ? *reinterpret_cast<const f32x4*>(gz + ((n * OH + oy0 + j) * OW + ox) * g_ld + c)
                             : f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int R = 3 * S + 3;
    const XT* xb = x + n * H * W * (long long)x_ld + c;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int iy = oy0 * S - 1 + r;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * S - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        const f32x4 v = tf4_(ld4(xb + ((long long)iy * W + ix) * x_ld), sc, sh, fl);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ky = r - j * S;
          if (ky >= 0 && ky < 3) dw[ky * 3 + kx] += v * gv[j];
        }
      }
    }
    }
    // lanes l, l + C4, l + 2*C4, ... of a wave hold the same channel group: butterfly-sum them first so that
    // only C4 lanes per wave touch the (slow) LDS float atomics
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float v = dw[t][q];
        for (int off = C4; off < 64; off <<= 1) v += __shfl_xor(v, off);
        if (lane < C4) atomicAdd(&red[t * C + c + q], v);
      }
  }
  __syncthreads();
  float* dst = rep ? gw + ((blockIdx.x + 5 * blockIdx.y) % JN_NREP) * JN_WPART_MAX : gw;
  for (int i = threadIdx.x; i < 9 * C; i += 256) atomicAdd(&dst[i], red[i]);
}

int launch_dw_bwd_weight(const float* gz, int g_ld, const void* x, int x_dtype, int x_ld, ChanTab it, float* gw_final,
                         float* wpart, int C, int H, int W, int OH, int OW, int N, int stride, hipStream_t s,
                         const SlotBatch& sb) {
  const int rep = (wpart && 9 * C <= JN_WPART_MAX) ? 1 : 0;
  float* gw = rep ? wpart : gw_final;
  const int YS = (OH + 3) / 4;
  const long long total = (long long)N * YS * OW * (C / 4);
  long long nb = (total + 255) / 256;
  const long long cap = sb.n >= 8 ? 256 : 2048 / sb.n;
  if (nb > cap) nb = cap;                         // grid-stride: each thread folds many strips before its atomics
  const dim3 blocks((unsigned)nb, sb.n);
  const size_t smem = (size_t)9 * C * sizeof(float);
#define JN_DWW(S_, T_)                                                                                            \
  hipLaunchKernelGGL((dw_bwd_weight_kernel<S_, T_>), blocks, dim3(256), smem, s, gz, g_ld, (const T_*)x, x_ld, it, \
                     gw, rep, C, H, W, OH, OW, N, sb)
  if (x_dtype == JN_BF16) { if (stride == 1) JN_DWW(1, bf16_t); else JN_DWW(2, bf16_t); }
  else { if (stride == 1) JN_DWW(1, float); else JN_DWW(2, float); }
#undef JN_DWW
  if (rep) launch_wpart_reduce(gw_final, wpart, 9 * C, s);
  return 0;
}

// ---- 2+3b+4b fused for depthwise layers ------------------------------------------------------------
// One LDS-tiled pass: g_z (from g_a, z) over the output tile + halo and the activated input tile + halo are
// staged once; every thread then forms the data gradient of its input pixel(s) and its share of the 9-tap weight
// gradient from LDS.  HBM traffic ~ (8 B/output + 4 B/input) x halo factor + 4 B/input written, instead of
// bn_bwd_gz (12) + dw_bwd_data (4 + 4) + dw_bwd_weight (4 + 4).  Tile = 8 x 16 output pixels, CB channels;
// stride 2: thread (a, b) owns output pixel (a, b) and the 2 x 2 input block under it.
constexpr int DF_TW = 16;

// RED: as in pw_bwd_fused_kernel — the BN-backward sums of the layer that produced the input are formed from the data
// gradient of each input pixel (its raw z re-read, L2-hot) and added to that layer's fp64 accumulators.
template <int S, int CB, int DF_TH, bool RED>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void dw_bwd_fused_kernel(
    const float* __restrict__ g, int g_ld, const float* __restrict__ z, int z_ld, ChanTab ot,
    const float* __restrict__ save, const float* __restrict__ consts, const float* __restrict__ x, int x_ld,
    ChanTab it, const float* __restrict__ w, float* __restrict__ gin, int gin_ld, int accumulate,
    float* __restrict__ gw, int rep, int C, int H, int W, int OH, int OW, int tiles_x, int tiles_y, int n_tiles,
    SlotBatch sb, double* __restrict__ red_in, long long red_rep_stride) {
  constexpr int Q = CB / 4, PS = S == 1 ? CB : CB + JN_DW_S2_PAD;     // stride-1 taps: a wave reads 1 KB contiguous, no padding
  constexpr int GH = S == 1 ? DF_TH + 2 : DF_TH + 1, GW = S == 1 ? DF_TW + 2 : DF_TW + 1;
  constexpr int AH = S == 1 ? DF_TH + 2 : 2 * DF_TH + 1, AW = S == 1 ? DF_TW + 2 : 2 * DF_TW + 1;
  constexpr int GROUPS = 256 / (DF_TW * Q), RPG = DF_TH / GROUPS;
  static_assert(RPG >= 1 && RPG * GROUPS == DF_TH, "tile / thread mapping");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Gs = sm;                        // [GH*GW][PS] g_z
  float* As = Gs + GH * GW * PS;         // [AH*AW][PS] activated input
  float* red = As + AH * AW * PS;        // [9][CB]
  float* Cs = red + 11 * CB;             // [7][CB] output-side constants, [3][CB] input table, [9][CB] weights (registers
                                         // are the occupancy limit of this kernel)
  {
    const long long sl = blockIdx.y;
    g += sl * sb.grad; z += sl * sb.act; x += sl * sb.act; gin += sl * sb.grad;
    save += sl * sb.save; consts += sl * sb.consts;
    if (RED) red_in += sl * sb.red;
    ot.sc += sl * sb.tab; ot.sh += sl * sb.tab;
    it.sc += sl * sb.tab; it.sh += sl * sb.tab; it.fl += sl * sb.tab;
  }
  const int tid = threadIdx.x;
  const int ncb = C / CB, cb = blockIdx.x % ncb;
  const int q = tid % Q, c = cb * CB + 4 * q;
  for (int i = tid; i < 11 * CB; i += 256) red[i] = 0.0f;
  for (int i = tid; i < CB; i += 256) {
    const int cc = cb * CB + i;
    Cs[i] = ot.sc[cc]; Cs[CB + i] = ot.sh[cc]; Cs[2 * CB + i] = save[2 * cc]; Cs[3 * CB + i] = save[2 * cc + 1];
    Cs[4 * CB + i] = consts[3 * cc]; Cs[5 * CB + i] = consts[3 * cc + 1]; Cs[6 * CB + i] = consts[3 * cc + 2];
    Cs[7 * CB + i] = it.sc[cc]; Cs[8 * CB + i] = it.sh[cc]; Cs[9 * CB + i] = it.fl[cc];
#pragma unroll
    for (int t = 0; t < 9; ++t) Cs[(10 + t) * CB + i] = w[t * C + cc];
  }
#define JN_C4(ROW) (*reinterpret_cast<const f32x4*>(Cs + (ROW) * CB + 4 * q))
  f32x4 dw[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) dw[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 r1 = {0.f, 0.f, 0.f, 0.f}, r2 = {0.f, 0.f, 0.f, 0.f};
  auto reduce_in = [&](const f32x4& zv, const f32x4& dx) {      // dx = data gradient of the input pixel with raw value zv
    const f32x4 i_sc = JN_C4(7), i_sh = JN_C4(8);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float yv = fmaf(zv[k], i_sc[k], i_sh[k]);
      const float gy = dx[k] * dsilu_(yv);
      r1[k] += gy;
      r2[k] += gy * yv;             // moment against y = gamma * zhat + beta: bn_bwd_consts turns it into sum gy * zhat (fp64)
    }
  };
  const int xx = (tid / Q) % DF_TW, grp = tid / (Q * DF_TW), j0 = grp * RPG;
  const int wg = blockIdx.x / ncb, n_wg = gridDim.x / ncb;
  // stride 2 with RED (round 4): every input pixel of the tile outside its first row / column belongs to the 2 x 2 block
  // of exactly ONE thread — the thread that forms its data gradient.  That thread stages it (raw value kept in registers,
  // activated value to LDS), so the reduction epilogue needs no second read of the input; the first version of this variant
  // re-read the raw input from global memory inside the gradient loop and lost more than the separate pass costs.
  constexpr bool OWNER = S == 2 && RED;
  static_assert(!OWNER || RPG == 1, "owner staging: one output pixel per thread and tile");
  f32x4 zraw[OWNER ? 4 : 1];
  for (int tile = wg; tile < n_tiles; tile += n_wg) {
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / (tiles_x * tiles_y);
    const int oy0 = ty * DF_TH, ox0 = tx * DF_TW;
    const int gy0 = S == 1 ? oy0 - 1 : oy0, gx0 = S == 1 ? ox0 - 1 : ox0;         // first output row / col in Gs
    const int ay0 = oy0 * S - 1, ax0 = ox0 * S - 1;                              // first input row / col in As
    __syncthreads();
    for (int i = tid; i < GH * GW * Q; i += 256) {
      const int p = i / Q, r = p / GW, cx = p - r * GW;
      const int oy = gy0 + r, ox = gx0 + cx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (oy >= 0 && oy < OH && ox >= 0 && ox < OW) {
        const long long pix = ((long long)n * OH + oy) * OW + ox;
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + pix * g_ld + c);
        const f32x4 zv = *reinterpret_cast<const f32x4*>(z + pix * z_ld + c);
        const f32x4 o_sc = JN_C4(0), o_sh = JN_C4(1), o_mean = JN_C4(2), o_istd = JN_C4(3), o_c1 = JN_C4(4), o_c2 = JN_C4(5),
                    o_k = JN_C4(6);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float zh = (zv[k] - o_mean[k]) * o_istd[k];
          const float gy = gv[k] * dsilu_(fmaf(zv[k], o_sc[k], o_sh[k]));
          v[k] = o_k[k] * (gy - o_c1[k] - zh * o_c2[k]);
        }
      }
      *reinterpret_cast<f32x4*>(Gs + p * PS + 4 * q) = v;
    }
    if constexpr (OWNER) {
#pragma unroll
      for (int d = 0; d < 4; ++d) {                 // the thread's own 2 x 2 block: tile rows 2 j0 + 1 (+ 1), columns 2 xx + 1 (+ 1)
        const int r = 2 * j0 + 1 + (d >> 1), cx = 2 * xx + 1 + (d & 1);
        const int iy = ay0 + r, ix = ax0 + cx;      // >= 0 by construction
        zraw[d] = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy < H && ix < W) {
          zraw[d] = *reinterpret_cast<const f32x4*>(x + (((long long)n * H + iy) * W + ix) * x_ld + c);
          v = tf4_(zraw[d], JN_C4(7), JN_C4(8), JN_C4(9));
        }
        *reinterpret_cast<f32x4*>(As + (r * AW + cx) * PS + 4 * q) = v;
      }
      for (int i = tid; i < (AW + AH - 1) * Q; i += 256) {     // the halo: row 0, then column 0 of the rows below
        const int p = i / Q, r = p < AW ? 0 : p - AW + 1, cx = p < AW ? p : 0;
        const int iy = ay0 + r, ix = ax0 + cx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W)
          v = tf4_(*reinterpret_cast<const f32x4*>(x + (((long long)n * H + iy) * W + ix) * x_ld + c), JN_C4(7), JN_C4(8), JN_C4(9));
        *reinterpret_cast<f32x4*>(As + (r * AW + cx) * PS + 4 * q) = v;
      }
    } else {
      for (int i = tid; i < AH * AW * Q; i += 256) {
        const int p = i / Q, r = p / AW, cx = p - r * AW;
        const int iy = ay0 + r, ix = ax0 + cx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W)
          v = tf4_(*reinterpret_cast<const f32x4*>(x + (((long long)n * H + iy) * W + ix) * x_ld + c), JN_C4(7), JN_C4(8), JN_C4(9));
        *reinterpret_cast<f32x4*>(As + p * PS + 4 * q) = v;
      }
    }
    __syncthreads();
#pragma unroll 1
    for (int j = 0; j < RPG; ++j) {
      const int y = j0 + j;                       // tile-relative output row of this thread
      if (S == 1) {
        // input pixel (y, xx) == output pixel (y, xx); Gs / As carry a 1-pixel halo
        const f32x4 gc = *reinterpret_cast<const f32x4*>(Gs + ((y + 1) * GW + xx + 1) * PS + 4 * q);
        f32x4 zin = {0.f, 0.f, 0.f, 0.f};              // RED: raw z of this input pixel, requested before the tap loop
        if (RED && oy0 + y < H && ox0 + xx < W)
          zin = *reinterpret_cast<const f32x4*>(x + (((long long)n * H + oy0 + y) * W + ox0 + xx) * x_ld + c);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            acc += *reinterpret_cast<const f32x4*>(Gs + ((y + 2 - ky) * GW + xx + 2 - kx) * PS + 4 * q) * JN_C4(10 + ky * 3 + kx);
            dw[ky * 3 + kx] += gc * *reinterpret_cast<const f32x4*>(As + ((y + ky) * AW + xx + kx) * PS + 4 * q);
          }
        const int iy = oy0 + y, ix = ox0 + xx;
        if (iy < H && ix < W) {
          float* op = gin + (((long long)n * H + iy) * W + ix) * gin_ld + c;
          if (RED) reduce_in(zin, acc);
          if (accumulate) acc += *reinterpret_cast<const f32x4*>(op);
          *reinterpret_cast<f32x4*>(op) = acc;
        }
      } else {
        const f32x4 g00 = *reinterpret_cast<const f32x4*>(Gs + (y * GW + xx) * PS + 4 * q);
        const f32x4 g01 = *reinterpret_cast<const f32x4*>(Gs + (y * GW + xx + 1) * PS + 4 * q);
        const f32x4 g10 = *reinterpret_cast<const f32x4*>(Gs + ((y + 1) * GW + xx) * PS + 4 * q);
        const f32x4 g11 = *reinterpret_cast<const f32x4*>(Gs + ((y + 1) * GW + xx + 1) * PS + 4 * q);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
            dw[ky * 3 + kx] += g00 * *reinterpret_cast<const f32x4*>(As + ((2 * y + ky) * AW + 2 * xx + kx) * PS + 4 * q);
        // the 2 x 2 input block (2y + dy, 2xx + dx): taps with (dy + 1 - ky), (dx + 1 - kx) even; one block pixel at a
        // time (formed, reduced, stored) so that only one of the four gradients is live
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const int iy = 2 * (oy0 + y) + (d >> 1), ix = 2 * (ox0 + xx) + (d & 1);
          if (iy < H && ix < W) {
            float* op = gin + (((long long)n * H + iy) * W + ix) * gin_ld + c;
            f32x4 v;
            if (d == 0) v = g00 * JN_C4(14);                                                              // (even, even): ky = kx = 1
            else if (d == 1) v = g01 * JN_C4(13) + g00 * JN_C4(15);                                       // (even, odd): ky = 1, kx = 0 | 2
            else if (d == 2) v = g10 * JN_C4(11) + g00 * JN_C4(17);                                       // (odd, even): kx = 1, ky = 0 | 2
            else v = g11 * JN_C4(10) + g10 * JN_C4(12) + g01 * JN_C4(16) + g00 * JN_C4(18);               // (odd, odd)
            if constexpr (OWNER) reduce_in(zraw[d], v);
            if (accumulate) v += *reinterpret_cast<const f32x4*>(op);
            *reinterpret_cast<f32x4*>(op) = v;
          }
        }
      }
    }
  }
  // lanes l, l + Q, ... hold the same channel quad: butterfly, then Q lanes per wave add into LDS
  const int lane = tid & 63;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v = dw[t][k];
      for (int off = Q; off < 64; off <<= 1) v += __shfl_xor(v, off);
      if (lane < Q) atomicAdd(&red[t * CB + 4 * q + k], v);
    }
  if (RED) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float a = r1[k], b = r2[k];
      for (int off = Q; off < 64; off <<= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
      if (lane < Q) { atomicAdd(&red[9 * CB + 2 * (4 * q + k)], a); atomicAdd(&red[9 * CB + 2 * (4 * q + k) + 1], b); }
    }
  }
  __syncthreads();
  float* dst = (rep ? gw + ((wg + 5 * blockIdx.y) % JN_NREP) * JN_WPART_MAX : gw);
  for (int i = tid; i < 9 * CB; i += 256) atomicAdd(&dst[(i / CB) * C + cb * CB + (i % CB)], red[i]);
  if (RED)
    if (tid < 2 * CB) atomicAdd(&red_in[(wg % JN_NREP) * red_rep_stride + 2 * cb * CB + tid], (double)red[9 * CB + tid]);
#undef JN_C4
}

template <int S, int CB, int DF_TH, bool RED>
static void launch_dw_bwd_fused_r(const DwBwdFusedArgs& a, hipStream_t s) {
  constexpr int PS = S == 1 ? CB : CB + JN_DW_S2_PAD;
  constexpr int GH = S == 1 ? DF_TH + 2 : DF_TH + 1, GW = S == 1 ? DF_TW + 2 : DF_TW + 1;
  constexpr int AH = S == 1 ? DF_TH + 2 : 2 * DF_TH + 1, AW = S == 1 ? DF_TW + 2 : 2 * DF_TW + 1;
  const size_t smem = ((size_t)(GH * GW + AH * AW) * PS + 11 * CB + 19 * CB) * sizeof(float);
  const int tiles_x = (a.OW + DF_TW - 1) / DF_TW, tiles_y = (a.OH + DF_TH - 1) / DF_TH;
  const int n_tiles = tiles_x * tiles_y * a.N, ncb = a.C / CB;
  int n_wg = 1536 / (ncb * a.sb.n);                 // persistent: one set of weight-gradient atomics per workgroup
  if (n_wg < 8) n_wg = 8;
  if (n_wg > n_tiles) n_wg = n_tiles;
  const int rep = (a.wpart && 9 * a.C <= JN_WPART_MAX) ? 1 : 0;
  hipLaunchKernelGGL((dw_bwd_fused_kernel<S, CB, DF_TH, RED>), dim3((unsigned)(n_wg * ncb), a.sb.n), dim3(256), smem, s, a.g, a.g_ld, a.z,
                     a.z_ld, a.ot, a.save, a.consts, a.x, a.x_ld, a.it, a.w, a.gin, a.gin_ld, a.accumulate,
                     rep ? a.wpart : a.gw, rep, a.C, a.H, a.W, a.OH, a.OW, tiles_x, tiles_y, n_tiles, a.sb, a.red_in,
                     a.red_rep_stride);
  if (rep) launch_wpart_reduce(a.gw, a.wpart, 9 * a.C, s);
}

template <int S, int CB, int DF_TH>
static void launch_dw_bwd_fused_t(const DwBwdFusedArgs& a, hipStream_t s) {
  if (a.red_in) launch_dw_bwd_fused_r<S, CB, DF_TH, true>(a, s);
  else launch_dw_bwd_fused_r<S, CB, DF_TH, false>(a, s);
}

bool dw_bwd_fused_supported(int C, int H, int W, int OH, int OW, int stride) {
  if (C % 16) return false;
  return stride == 1 ? (OH == H && OW == W) : (stride == 2 && OH == (H + 1) / 2 && OW == (W + 1) / 2);
}

int launch_dw_bwd_fused(const DwBwdFusedArgs& a, hipStream_t s) {
  if (a.stride == 1) { if (a.C % 32 == 0) launch_dw_bwd_fused_t<1, 32, 8>(a, s); else launch_dw_bwd_fused_t<1, 16, 8>(a, s); }
  else launch_dw_bwd_fused_t<2, 16, 4>(a, s);       // stride 2: 4-row tiles halve the LDS tiles (more workgroups per CU)
  return 0;
}

// ---- 4c. stem weight gradient: dW[(c,dy,dx)][oc] += sum_pixels gz[p][oc] * img[c][2oy-2+dy][2ox-2+dx]
constexpr int SB_TY = 8, SB_TX = 32, SB_IH = 2 * SB_TY + 4, SB_IW = 2 * SB_TX + 4;
// LDS layouts chosen for conflict-free MFMA operand reads (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE was 0.43):
// image rows 70 floats apart -> tap (dy, dx) of 16 consecutive k lands on bank 6 dy + dx = k % 36, consecutive; the four
// k-slots of an MFMA take pixels 8 apart (16 banks apart); g_z rows of 8 pixels are padded by 16 floats per slot group
constexpr int SB_IWP = 70, SB_GZG = 8 * 16 + 16, SB_GZROW = 4 * SB_GZG;

constexpr int SB_W4 = SB_IW / 4 + 1;     // float4 groups per row of the aligned window (see stem_mfma_kernel, VEC)
template <bool VEC>
__global__ __launch_bounds__(256) void stem_bwd_weight_kernel(
    const float* __restrict__ src, const long long* __restrict__ pos, int pos_stride, long long sample_stride,
    long long chan_stride, int row_stride, int P, const float* __restrict__ gz, int g_ld, int cout, int ocg,
    int tiles_x, int tiles_y, int n_tiles, float* __restrict__ gw, long long pos_slot, long long g_slot,
    const float* __restrict__ z, int z_ld, ChanTab ot, const float* __restrict__ save,
    const float* __restrict__ consts, SlotBatch sb) {
  __shared__ __attribute__((aligned(16))) float tile[3 * SB_IH * SB_IWP];
  if (pos) pos += blockIdx.z * pos_slot;
  gz += blockIdx.z * g_slot;
  // z != null: `gz` holds d loss / d activation and g_z is formed while staging (no bn_bwd_gz pass over the
  // largest map of the network)
  // the seven per-channel constants of the 16 output channels sit in LDS (a thread reads its quad while staging): in 28
  // registers they kept the kernel at two workgroups per CU
  __shared__ __attribute__((aligned(16))) float Cq[7 * 16];
  if (z) {
    const long long sl = blockIdx.z;
    z += sl * sb.act; save += sl * sb.save; consts += sl * sb.consts; ot.sc += sl * sb.tab; ot.sh += sl * sb.tab;
    if (threadIdx.x < 16) {
      const int c = blockIdx.y * 16 + threadIdx.x;
      Cq[threadIdx.x] = ot.sc[c]; Cq[16 + threadIdx.x] = ot.sh[c]; Cq[32 + threadIdx.x] = save[2 * c]; Cq[48 + threadIdx.x] = save[2 * c + 1];
      Cq[64 + threadIdx.x] = consts[3 * c]; Cq[80 + threadIdx.x] = consts[3 * c + 1]; Cq[96 + threadIdx.x] = consts[3 * c + 2];
    }
  }
  __shared__ __attribute__((aligned(16))) float Gz[SB_TY * SB_GZROW];
  __shared__ float Ts[16 * 112];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, g = lane >> 4;
  const int og = blockIdx.y;
  const int OH = P / 2;
  for (int i = tid; i < 16 * 112; i += 256) Ts[i] = 0.0f;
  int koff[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) {
    const int k = 16 * t + lm;
    const int c = k / 36, dy = (k % 36) / 6, dx = k % 6;
    koff[t] = k < 108 ? (c * SB_IH + dy) * SB_IWP + dx : 0;
  }
  f32x4 acc[7];
#pragma unroll
  for (int t = 0; t < 7; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  // persistent over (image, tile): the partial dW stays in registers, ONE set of atomics per workgroup; the next
  // tile's image values and gradients are fetched into registers while the MFMAs of the current one run
  constexpr int NI = (3 * SB_IH * SB_IW + 255) / 256, NG = SB_TY * SB_TX * 4 / 256;
  constexpr int NI4 = (3 * SB_IH * SB_W4 + 255) / 256;
  float pim[VEC ? 1 : NI];
  f32x4 pim4[VEC ? NI4 : 1];
  f32x4 pg[NG], pz[NG];
  auto fetch = [&](int tl) {
    const int n = tl / (tiles_x * tiles_y), tr = tl % (tiles_x * tiles_y);
    const int oy0 = (tr / tiles_x) * SB_TY, ox0 = (tr % tiles_x) * SB_TX;
    const float* base = src + (long long)n * sample_stride;
    if (pos) base += pos[(long long)pos_stride * n] * (long long)P * row_stride + pos[(long long)pos_stride * n + 1] * (long long)P;
    if constexpr (VEC) {
#pragma unroll
      for (int j = 0; j < NI4; ++j) {
        const int i = tid + 256 * j;
        const int c = i / (SB_IH * SB_W4), r = (i / SB_W4) % SB_IH, q4 = i % SB_W4;
        const int iy = 2 * oy0 - 2 + r, ix = 2 * ox0 - 4 + 4 * q4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (i < 3 * SB_IH * SB_W4 && iy >= 0 && iy < P && ix >= 0 && ix < P)
          v = *reinterpret_cast<const f32x4*>(base + c * chan_stride + (long long)iy * row_stride + ix);
        pim4[j] = v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int i = tid + 256 * j;
        const int c = i / (SB_IH * SB_IW), r = (i / SB_IW) % SB_IH, q = i % SB_IW;
        const int iy = 2 * oy0 - 2 + r, ix = 2 * ox0 - 2 + q;
        float v = 0.0f;
        if (i < 3 * SB_IH * SB_IW && iy >= 0 && iy < P && ix >= 0 && ix < P) v = base[c * chan_stride + (long long)iy * row_stride + ix];
        pim[j] = v;
      }
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int i = tid + 256 * j;
      const int p = i >> 2, q = i & 3, ty = p / SB_TX, tx = p % SB_TX;
      pg[j] = f32x4{0.f, 0.f, 0.f, 0.f}; pz[j] = pg[j];
      if (oy0 + ty < OH && ox0 + tx < OH) {
        const long long pix = ((long long)n * OH + oy0 + ty) * OH + ox0 + tx;
        pg[j] = *reinterpret_cast<const f32x4*>(gz + pix * g_ld + og * 16 + 4 * q);
        if (z) pz[j] = *reinterpret_cast<const f32x4*>(z + pix * z_ld + og * 16 + 4 * q);
      }
    }
  };
  int tl = blockIdx.x;
  if (tl < n_tiles) fetch(tl);
  for (; tl < n_tiles; tl += gridDim.x) {
    __syncthreads();
    if constexpr (VEC) {
#pragma unroll
      for (int j = 0; j < NI4; ++j) {
        const int i = tid + 256 * j;
        if (i < 3 * SB_IH * SB_W4) {
          const int q4 = i % SB_W4;
          float* d = tile + (i / SB_W4) * SB_IWP + 4 * q4 - 2;               // 8-byte aligned (row = 280 bytes)
          if (q4 > 0) *reinterpret_cast<float2*>(d) = float2{pim4[j].x, pim4[j].y};
          if (q4 < SB_W4 - 1) *reinterpret_cast<float2*>(d + 2) = float2{pim4[j].z, pim4[j].w};
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int i = tid + 256 * j;
        if (i < 3 * SB_IH * SB_IW) tile[(i / SB_IW) * SB_IWP + i % SB_IW] = pim[j];
      }
    }
#pragma unroll
    for (int j = 0; j < NG; ++j) {
      const int i = tid + 256 * j;
      f32x4 v = pg[j];
      if (z) {                                   // pixels outside the map carry g = 0 -> gy = 0, but c1 must not leak in
        const int cq = 4 * (i & 3);
        const f32x4 o_sc = *reinterpret_cast<const f32x4*>(Cq + cq), o_sh = *reinterpret_cast<const f32x4*>(Cq + 16 + cq),
                    o_mean = *reinterpret_cast<const f32x4*>(Cq + 32 + cq), o_istd = *reinterpret_cast<const f32x4*>(Cq + 48 + cq),
                    o_c1 = *reinterpret_cast<const f32x4*>(Cq + 64 + cq), o_c2 = *reinterpret_cast<const f32x4*>(Cq + 80 + cq),
                    o_k = *reinterpret_cast<const f32x4*>(Cq + 96 + cq);
        const int p = i >> 2, ty = p / SB_TX, tx = p % SB_TX;
        const int tr = tl % (tiles_x * tiles_y);
        const bool inside = (tr / tiles_x) * SB_TY + ty < OH && (tr % tiles_x) * SB_TX + tx < OH;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float zh = (pz[j][k] - o_mean[k]) * o_istd[k];
          const float gy = v[k] * dsilu_(fmaf(pz[j][k], o_sc[k], o_sh[k]));
          v[k] = inside ? o_k[k] * (gy - o_c1[k] - zh * o_c2[k]) : 0.0f;
        }
      }
      {
        const int p = i >> 2, ty = p / SB_TX, tx = p % SB_TX;
        *reinterpret_cast<f32x4*>(&Gz[ty * SB_GZROW + (tx >> 3) * SB_GZG + (tx & 7) * 16 + 4 * (i & 3)]) = v;
      }
    }
    __syncthreads();
    if (tl + (int)gridDim.x < n_tiles) fetch(tl + gridDim.x);
    // Software-pipelined over the 16 k-steps (round 3): the operands of step st + 1 are read from LDS BEFORE the seven
    // MFMAs of step st issue.  Left to the compiler (unroll 2) every step's reads were issued right before its MFMAs and
    // the first MFMA of each step waited out the LDS latency; the matrix pipe was busy 36 % of the kernel.
    {
      // the wave owns tile rows 2 wave, 2 wave + 1; k-slot g of step st = pixel (ty, 8 g + st % 8)
      auto lds_step = [&](int st, float& av, float (&bv)[7]) {
        const int ty = 2 * wave + (st >> 3), tx = 8 * g + (st & 7);
        av = Gz[ty * SB_GZROW + g * SB_GZG + (st & 7) * 16 + lm];               // A[i = oc][kk = pixel]
        const int pbase = (2 * ty) * SB_IWP + 2 * tx;
#pragma unroll
        for (int t = 0; t < 7; ++t) bv[t] = tile[pbase + koff[t]];
      };
      float av_c, bv_c[7];
      lds_step(0, av_c, bv_c);
#pragma unroll
      for (int st = 0; st < SB_TY * SB_TX / 16; ++st) {
        float av_n = 0.0f, bv_n[7];
        if (st + 1 < SB_TY * SB_TX / 16) lds_step(st + 1, av_n, bv_n);
#pragma unroll
        for (int t = 0; t < 7; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_c, bv_c[t], acc[t], 0, 0, 0);
        av_c = av_n;
#pragma unroll
        for (int t = 0; t < 7; ++t) bv_c[t] = bv_n[t];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 7; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) atomicAdd(&Ts[(4 * g + r) * 112 + 16 * t + lm], acc[t][r]);
  __syncthreads();
  float* dst = gw + ((blockIdx.x + 5 * blockIdx.z) % JN_NREP) * JN_WPART_MAX;
  for (int i = tid; i < 16 * 108; i += 256) {
    const int oc = i / 108, k = i % 108;
    atomicAdd(&dst[(long long)k * cout + og * 16 + oc], Ts[oc * 112 + k]);
  }
}

int launch_stem_bwd_weight(const StemArgs& a, const float* gz, int g_ld, float* gw, float* wpart, hipStream_t s,
                           const SlotBatch& sb, const float* z, int z_ld, ChanTab ot, const float* save,
                           const float* consts) {
  const int OH = a.P / 2, ocg = a.cout / 16;
  const int tiles_x = (OH + SB_TX - 1) / SB_TX, tiles_y = (OH + SB_TY - 1) / SB_TY;
  const int n_tiles = tiles_x * tiles_y * a.N;
  const int cap = sb.n >= 4 ? 256 : 1024 / sb.n;
  dim3 grid(n_tiles < cap ? n_tiles : cap, ocg, sb.n);
#define JN_STEMB(V_)                                                                                                      \
  hipLaunchKernelGGL(stem_bwd_weight_kernel<V_>, grid, dim3(256), 0, s, a.src, (const long long*)a.positions, a.pos_stride, \
                     a.sample_stride, a.chan_stride, a.row_stride, a.P, gz, g_ld, a.cout, ocg, tiles_x, tiles_y, n_tiles,  \
                     wpart, sb.pos, sb.grad, z, z_ld, ot, save, consts, sb)
  if (stem_rows_aligned(a)) JN_STEMB(true); else JN_STEMB(false);
#undef JN_STEMB
  launch_wpart_reduce(gw, wpart, 108 * a.cout, s);
  return 0;
}

// ---- SPP backward: g0 += route(g1, argmax5) + route(g2, argmax9) + route(g3, argmax13) ----------
template <typename AT>
__global__ __launch_bounds__(256) void spp_bwd_kernel(const AT* __restrict__ cat, float* __restrict__ gcat, int ld,
                                                      int h, int H, int W, int cb, ChanTab it, SlotBatch sb) {
  extern __shared__ float sp[];
  {
    const long long sl = blockIdx.z;
    cat += sl * sb.act; gcat += sl * sb.grad;
    it.sc += sl * sb.tab; it.sh += sl * sb.tab; it.fl += sl * sb.tab;
  }
  const int HW = H * W;
  float* A = sp;                 // activation of slice 0
  float* G = sp + HW * cb;       // gradient accumulator for slice 0
  float* Rv = G + HW * cb;       // row pass: max of the horizontal window ...
  int* Rx = reinterpret_cast<int*>(Rv + HW * cb);   // ... and the column it sits in (leftmost on ties)
  const int n = blockIdx.y, c0 = blockIdx.x * cb;
  const AT* base = cat + (long long)n * HW * ld + c0;
  float* gbase = gcat + (long long)n * HW * ld + c0;
  const int tid = threadIdx.x;
  for (int e = tid; e < HW * cb; e += 256) {
    const int c = e % cb;
    const float z = ld1(base + (long long)(e / cb) * ld + c);
    A[e] = it.fl[c0 + c] != 0.0f ? silu_(fmaf(z, it.sc[c0 + c], it.sh[c0 + c])) : z;
    G[e] = 0.0f;
  }
  __syncthreads();
  // The gradient of each pooled value goes to the FIRST maximum of its window in row-major order (MaxPool2d).  The 2-D
  // arg-max is separable with that tie rule: leftmost maximum of every window row, then the topmost row holding the
  // overall maximum — 2 (2 r + 1) reads per element and radius instead of (2 r + 1)^2 (275 reads for r = 2, 4, 6).
  // compile-time radii: the window loads are independent and issued back to back (a run-time trip count serialised
  // one LDS round trip per tap)
  auto route = [&](auto rad_c, int stage) {
    constexpr int RAD = decltype(rad_c)::value;
    for (int e = tid; e < HW * cb; e += 256) {
      const int p = e / cb, c = e % cb, y = p / W, x = p - y * W;
      float v[2 * RAD + 1];
#pragma unroll
      for (int d = -RAD; d <= RAD; ++d) {
        const int xx = x + d;
        v[d + RAD] = (xx >= 0 && xx < W) ? A[(y * W + xx) * cb + c] : -INFINITY;
      }
      float best = -INFINITY;
      int bx = x;
#pragma unroll
      for (int d = -RAD; d <= RAD; ++d)
        if (v[d + RAD] > best) { best = v[d + RAD]; bx = x + d; }
      Rv[e] = best; Rx[e] = bx;
    }
    __syncthreads();
    for (int e = tid; e < HW * cb; e += 256) {
      const int p = e / cb, c = e % cb, y = p / W, x = p - y * W;
      float v[2 * RAD + 1];
#pragma unroll
      for (int d = -RAD; d <= RAD; ++d) {
        const int yy = y + d;
        v[d + RAD] = (yy >= 0 && yy < H) ? Rv[(yy * W + x) * cb + c] : -INFINITY;
      }
      float best = -INFINITY;
      int by = y;
#pragma unroll
      for (int d = -RAD; d <= RAD; ++d)
        if (v[d + RAD] > best) { best = v[d + RAD]; by = y + d; }
      const int bi = by * W + Rx[(by * W + x) * cb + c];
      atomicAdd(&G[bi * cb + c], gbase[(long long)p * ld + stage * h + c]);
    }
    __syncthreads();
  };
  route(std::integral_constant<int, 2>{}, 1);
  route(std::integral_constant<int, 4>{}, 2);
  route(std::integral_constant<int, 6>{}, 3);
  for (int e = tid; e < HW * cb; e += 256) gbase[(long long)(e / cb) * ld + (e % cb)] += G[e];
}

// fp32, 16-byte form (round 4): a thread owns channel quads (16-byte loads of the activation and of the three pooled
// gradients, one 16-byte read-modify-write of g0), the four channels of a quad are routed independently; the tie rule and
// the separable arg-max are those of spp_bwd_kernel.  16 channels per workgroup.
template <int CBQ>
__global__ __launch_bounds__(256) void spp4_bwd_kernel(const float* __restrict__ cat, float* __restrict__ gcat, int ld, int h, int H,
                                                       int W, ChanTab it, SlotBatch sb) {
  using i32x4 = __attribute__((ext_vector_type(4))) int;
  extern __shared__ __attribute__((aligned(16))) float sp[];
  {
    const long long sl = blockIdx.z;
    cat += sl * sb.act; gcat += sl * sb.grad;
    it.sc += sl * sb.tab; it.sh += sl * sb.tab; it.fl += sl * sb.tab;
  }
  constexpr int NI = 8;
  const int HW = H * W, NEL = HW * CBQ;
  f32x4* A = reinterpret_cast<f32x4*>(sp);             // activation of slice 0
  float* G = sp + 4 * NEL;                             // gradient accumulator for slice 0 (LDS atomics per channel)
  f32x4* Rv = reinterpret_cast<f32x4*>(sp + 8 * NEL);  // row pass: max of the horizontal window ...
  i32x4* Rx = reinterpret_cast<i32x4*>(sp + 12 * NEL); // ... and the column it sits in (leftmost on ties)
  const int n = blockIdx.y, c0 = blockIdx.x * 4 * CBQ;
  const float* base = cat + (long long)n * HW * ld + c0;
  float* gbase = gcat + (long long)n * HW * ld + c0;
  const int tid = threadIdx.x, q = tid % CBQ;
  f32x4 sc, sh, fl;
#pragma unroll
  for (int k = 0; k < 4; ++k) { sc[k] = it.sc[c0 + 4 * q + k]; sh[k] = it.sh[c0 + 4 * q + k]; fl[k] = it.fl[c0 + 4 * q + k]; }
  int py[NI], px[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int e = tid + 256 * i, p = e / CBQ;
    py[i] = p / W; px[i] = p - py[i] * W;
    if (e < NEL) {
      const f32x4 z = *reinterpret_cast<const f32x4*>(base + (long long)p * ld + 4 * q);
      f32x4 a;
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] = fl[k] != 0.0f ? silu_(fmaf(z[k], sc[k], sh[k])) : z[k];
      A[e] = a;
      *reinterpret_cast<f32x4*>(G + 4 * e) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  __syncthreads();
  auto route = [&](auto rad_c, int stage) {
    constexpr int RAD = decltype(rad_c)::value;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int e = tid + 256 * i;
      if (e < NEL) {
        const int x = px[i];
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        i32x4 bx = {x, x, x, x};
#pragma unroll
        for (int d = -RAD; d <= RAD; ++d) {
          const int xx = x + d;
          if (xx >= 0 && xx < W) {
            const f32x4 v = A[e + d * CBQ];
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (v[k] > best[k]) { best[k] = v[k]; bx[k] = xx; }
          }
        }
        Rv[e] = best; Rx[e] = bx;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int e = tid + 256 * i;
      if (e < NEL) {
        const int y = py[i], x = px[i];
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        i32x4 by = {y, y, y, y};
#pragma unroll
        for (int d = -RAD; d <= RAD; ++d) {
          const int yy = y + d;
          if (yy >= 0 && yy < H) {
            const f32x4 v = Rv[e + d * W * CBQ];
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (v[k] > best[k]) { best[k] = v[k]; by[k] = yy; }
          }
        }
        const f32x4 g = *reinterpret_cast<const f32x4*>(gbase + (long long)(e / CBQ) * ld + stage * h + 4 * q);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int src = (by[k] * W + x) * CBQ + q;
          const int bi = by[k] * W + reinterpret_cast<const int*>(Rx + src)[k];
          atomicAdd(&G[(bi * CBQ + q) * 4 + k], g[k]);
        }
      }
    }
    __syncthreads();
  };
  route(std::integral_constant<int, 2>{}, 1);
  route(std::integral_constant<int, 4>{}, 2);
  route(std::integral_constant<int, 6>{}, 3);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int e = tid + 256 * i;
    if (e < NEL) {
      f32x4* gp = reinterpret_cast<f32x4*>(gbase + (long long)(e / CBQ) * ld + 4 * q);
      *gp += *reinterpret_cast<const f32x4*>(G + 4 * e);
    }
  }
}

int launch_spp_bwd(const void* cat, int dtype, float* gcat, int ld, int h, int H, int W, int N, ChanTab it,
                   hipStream_t s, const SlotBatch& sb) {
  static const bool no_v4 = std::getenv("JN_NO_SPP_V4") != nullptr;
  if (dtype == JN_F32 && !no_v4 && h % 16 == 0 && ld % 4 == 0 && H * W * 4 <= 2048) {
    const size_t smem4 = (size_t)H * W * 4 * 16 * sizeof(float);        // at most 128 KB (H * W <= 512)
    static bool raised = false;
    if (!raised && smem4 > 64 * 1024) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&spp4_bwd_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      raised = true;
    }
    hipLaunchKernelGGL(spp4_bwd_kernel<4>, dim3(h / 16, N, sb.n), dim3(256), smem4, s, (const float*)cat, gcat, ld, h, H, W, it, sb);
    return 0;
  }
  const int cb0 = 8;   // channels per workgroup, measured at B = 64, 20 steps: 16: 1068 us, 8: 847 us, 4: 888 us
  int cb = cb0;
  while (cb > 4 && (size_t)H * W * cb * 4 * sizeof(float) > 60 * 1024) cb >>= 1;
  dim3 grid(h / cb, N, sb.n);
  const size_t smem = (size_t)H * W * cb * 4 * sizeof(float);
  if (dtype == JN_BF16)
    hipLaunchKernelGGL(spp_bwd_kernel<bf16_t>, grid, dim3(256), smem, s, (const bf16_t*)cat, gcat, ld, h, H, W, cb, it, sb);
  else
    hipLaunchKernelGGL(spp_bwd_kernel<float>, grid, dim3(256), smem, s, (const float*)cat, gcat, ld, h, H, W, cb, it, sb);
  return 0;
}

// upsample backward: g_src (=|+=) sum of the 2x2 children
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float* __restrict__ gdst, int dst_ld,
                                                           float* __restrict__ gsrc, int src_ld, int C, int H, int W,
                                                           long long total, int accumulate, long long g_slot) {
  gdst += blockIdx.y * g_slot; gsrc += blockIdx.y * g_slot;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int C4 = C >> 2;
  const int c = (int)(idx % C4) * 4;
  const int x = (int)((idx / C4) % W);
  const int y = (int)((idx / ((long long)C4 * W)) % H);
  const long long n = idx / ((long long)C4 * W * H);
  const int OW = 2 * W, OH = 2 * H;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int dy = 0; dy < 2; ++dy)
#pragma unroll
    for (int dx = 0; dx < 2; ++dx)
      acc += *reinterpret_cast<const f32x4*>(gdst + ((n * OH + 2 * y + dy) * OW + 2 * x + dx) * dst_ld + c);
  float* op = gsrc + ((n * H + y) * W + x) * src_ld + c;
  if (accumulate) acc += *reinterpret_cast<const f32x4*>(op);
  *reinterpret_cast<f32x4*>(op) = acc;
}

int launch_upsample_bwd(const float* gdst, int dst_ld, float* gsrc, int src_ld, int C, int H, int W, int N,
                        int accumulate, hipStream_t s, const SlotBatch& sb) {
  const long long total = (long long)N * H * W * (C / 4);
  hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)((total + 255) / 256), sb.n), dim3(256), 0, s, gdst, dst_ld, gsrc,
                     src_ld, C, H, W, total, accumulate, sb.grad);
  return 0;
}

// dst (=|+=) src over a [M][C] view (shortcut backward, gradient seeding)
__global__ __launch_bounds__(256) void grad_copy_kernel(const float* __restrict__ src, int src_ld,
                                                        float* __restrict__ dst, int dst_ld, int C, long long M,
                                                        int accumulate, long long g_slot) {
  src += blockIdx.y * g_slot; dst += blockIdx.y * g_slot;
  const int C4 = C >> 2;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * C4) return;
  const int c = (int)(idx % C4) * 4;
  const long long m = idx / C4;
  f32x4 v = *reinterpret_cast<const f32x4*>(src + m * src_ld + c);
  if (accumulate) v += *reinterpret_cast<const f32x4*>(dst + m * dst_ld + c);
  *reinterpret_cast<f32x4*>(dst + m * dst_ld + c) = v;
}

int launch_grad_copy(const float* src, int src_ld, float* dst, int dst_ld, int C, long long M, int accumulate,
                     hipStream_t s, const SlotBatch& sb) {
  const long long total = M * (C / 4);
  hipLaunchKernelGGL(grad_copy_kernel, dim3((unsigned)((total + 255) / 256), sb.n), dim3(256), 0, s, src, src_ld, dst, dst_ld,
                     C, M, accumulate, sb.grad);
  return 0;
}

// NCHW gradient (boundary, parity API) -> NHWC view (=|+=)
__global__ __launch_bounds__(256) void nchw_to_nhwc_grad_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                                int out_ld, int C, int HW, long long total,
                                                                int accumulate) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const int p = (int)((idx / C) % HW);
  const long long n = idx / ((long long)HW * C);
  float v = in[(n * C + c) * HW + p];
  float* op = out + (n * HW + p) * out_ld + c;
  if (accumulate) v += *op;
  *op = v;
}

int launch_nchw_to_nhwc_grad(const float* in, float* out, int out_ld, int C, int HW, int N, int accumulate,
                             hipStream_t s) {
  const long long total = (long long)N * C * HW;
  hipLaunchKernelGGL(nchw_to_nhwc_grad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, out_ld, C,
                     HW, total, accumulate);
  return 0;
}

}  // namespace jnr
