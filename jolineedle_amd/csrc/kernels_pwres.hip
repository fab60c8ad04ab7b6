// Wide 1x1 convolutions (GEMM with K, N >= 64) for gfx950: weight-resident, persistent, software-pipelined.
//
// z[m][n] = sum_k T(x[m][k]) * w[n][k]   (forward; T = "normalize on read", see kernels_conv.hip)
// gx[m][n] = sum_k g[m][k] * w[k][n]     (WT: data gradient of a layer whose forward weight is w[k][n])
//
// The generic pw_mfma_kernel re-stages the weight tile in every workgroup and every K chunk, single-buffered, with two
// barriers per 32-wide chunk: on the 28x28 / 14x14 maps (196 - 784 workgroups of one tile each) it ran at a third of
// either roof (profiles/r01_g_*).  Here
//   * a workgroup keeps its [16*CT][K] weight slice in LDS for its whole life (one read of W per workgroup, not per
//     tile and chunk) and walks over pixel tiles (persistent grid: <= 256 * workgroups-per-CU, a multiple of 8 so that
//     the N slices of one pixel tile sit on one XCD and share its L2);
//   * the pixel operand streams in [BM][KC] chunks through TWO LDS buffers with ONE barrier per chunk, and the global
//     loads run PD chunks ahead in registers (the load -> transform -> LDS -> MFMA chain of a chunk overlaps the MFMAs
//     of the previous PD chunks);
//   * LDS rows are K + 8 / KC + 8 floats: with K % 32 == 0 the row stride is 8 or 40 mod 64 banks, which makes the
//     k-permuted ds_read_b128 fragment reads of all four 16-lane groups conflict-free (MI355X_MICROARCH.md, LDS);
//   * BatchNorm sums stay in registers across the tiles: one set of fp64 atomics per workgroup.
// Exact fp32 (v_mfma_f32_16x16x4_f32), same fragment layout and k order as pw_mfma_kernel.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "jn_kernels.h"
#include "jn_reduce.h"
#include "jn_tab.h"
#include "jn_types.h"

namespace jnr {

#ifdef JN_PWRES_STAMPS
__device__ long long* g_pwres_dbg = nullptr;      // [workgroups][32] wall-clock stamps (10 ns ticks), tools/pwbench.hip
#define JN_STAMP(i) do { if (g_pwres_dbg && threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (i) < 32) g_pwres_dbg[blockIdx.x * 32 + (i)] = wall_clock64(); } while (0)
#else
#define JN_STAMP(i) do { } while (0)
#endif

template <int CT, int KC, int BM, int PD, bool WT>
__global__ __launch_bounds__(256) void pw_res_kernel(
    const float* __restrict__ x, int x_ld, ChanTab it, const float* __restrict__ w, int w_ld, float* __restrict__ out,
    int out_ld, long long M, int K, int Nc, int accumulate, double* __restrict__ stats, long long rep_stride, int nrep,
    const int* __restrict__ skip_flag, int skip_when, long long x_slot, long long out_slot, long long tab_slot) {
  if (skip_flag && *skip_flag >= skip_when) return;
  JN_STAMP(0);
  x += blockIdx.z * x_slot; out += blockIdx.z * out_slot;        // step-batched launches (gradients)
  it.sc += blockIdx.z * tab_slot; it.sh += blockIdx.z * tab_slot; it.fl += blockIdx.z * tab_slot;
  constexpr int WN = (CT >= 2) ? 2 : 1, WM = 4 / WN;             // waves along the channels / the pixels
  constexpr int PT = BM / (16 * WM), CTW = CT / WN;              // 16x16 tiles per wave: pixels x channels
  static_assert(PT >= 1 && CTW >= 1 && PT * 16 * WM == BM && CTW * WN == CT, "pw_res tile mapping");
  constexpr int LDX = KC + 8, Q4 = KC / 4, RPP = 256 / Q4, NX = BM * Q4 / 256;
  static_assert(NX >= 1 && 256 % Q4 == 0, "pw_res staging mapping");
  const int LDW = K + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* Ws = reinterpret_cast<float*>(smem_raw);                 // [16*CT][LDW]   resident weight slice
  float* Xs = Ws + 16 * CT * LDW;                                 // [2][BM][LDX]   pixel chunks
  float* Tb = Xs + 2 * BM * LDX;                                  // [3][K]         input table
  float* red = Tb + 3 * K;                                        // [WM][16*CT][2] statistics slots
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int lm = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.y * (16 * CT);
  const int nchunks = K / KC;
  const long long n_tiles = (M + BM - 1) / BM;
  const long long my_tiles = blockIdx.x < n_tiles ? (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
  const long long n_it = my_tiles * nchunks;
  const int q = tid % Q4, r0 = tid / Q4;

  // ---- pipeline registers: PD chunks of raw pixels in flight ----
  f32x4 xr[PD][NX];
  long long pf_tile = blockIdx.x; int pf_chunk = 0;               // next (tile, chunk) to fetch
  auto fetch = [&](f32x4 (&dst)[NX]) {
    const long long m0 = pf_tile * BM;
    const int k0 = pf_chunk * KC;
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const long long m = m0 + r0 + RPP * j;
      dst[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (m < M) dst[j] = *reinterpret_cast<const f32x4*>(x + m * x_ld + k0 + 4 * q);
    }
    if (++pf_chunk == nchunks) { pf_chunk = 0; pf_tile += gridDim.x; }
  };
#pragma unroll
  for (int u = 0; u < PD; ++u)
    if (u < n_it) fetch(xr[u]);

  // ---- prologue: table and weight slice -> LDS ----
  // The slice is fetched in batches of WB float4 per thread, all loads of a batch in flight before the first LDS store (a
  // plain load -> store loop with a run-time trip count waits one global round trip per iteration: 16 of them for a
  // 128 x 128 slice); the table (deferred entries: batch sums) is derived while the first batch is in flight.
  {
    constexpr int WB = 8;
    const int NQ = 4 * CT, KQ = K / 4;
    const int total = WT ? K * NQ : 16 * CT * KQ;
    f32x4 wr[WB];
    auto wload = [&](int base) {
#pragma unroll
      for (int j = 0; j < WB; ++j) {
        const int i = base + tid + 256 * j;
        wr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < total) {
          if (WT) {           // w is [K][w_ld] (the forward weight of the differentiated layer): Ws[n][k] = w[k][n0 + n]
            const int k = i / NQ, nq = i - k * NQ;
            if (n0 + 4 * nq < Nc) wr[j] = *reinterpret_cast<const f32x4*>(w + (long long)k * w_ld + n0 + 4 * nq);
          } else {
            const int r = i / KQ, kq = i - r * KQ;
            if (n0 + r < Nc) wr[j] = *reinterpret_cast<const f32x4*>(w + (long long)(n0 + r) * w_ld + 4 * kq);
          }
        }
      }
    };
    auto wstore = [&](int base) {
#pragma unroll
      for (int j = 0; j < WB; ++j) {
        const int i = base + tid + 256 * j;
        if (i < total) {
          if (WT) {
            const int k = i / NQ, nq = i - k * NQ;
#pragma unroll
            for (int e = 0; e < 4; ++e) Ws[(4 * nq + e) * LDW + k] = wr[j][e];
          } else {
            const int r = i / KQ, kq = i - r * KQ;
            *reinterpret_cast<f32x4*>(Ws + r * LDW + 4 * kq) = wr[j];
          }
        }
      }
    };
    wload(0);
    tab_to_lds(Tb, K, K, it, tid, 256);
    wstore(0);
    for (int base = 256 * WB; base < total; base += 256 * WB) { wload(base); wstore(base); }
  }
  JN_STAMP(1);
  __syncthreads();
  JN_STAMP(2);
  int stamp_i = 3;

  f32x4 acc[PT][CTW];
#pragma unroll
  for (int p = 0; p < PT; ++p)
#pragma unroll
    for (int c = 0; c < CTW; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 s1[CTW], s2[CTW];
#pragma unroll
  for (int c = 0; c < CTW; ++c) { s1[c] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[c] = s1[c]; }

  long long cur_tile = blockIdx.x; int cur_chunk = 0;             // (tile, chunk) being computed
  for (long long f = 0; f < n_it; f += PD) {
#pragma unroll
    for (int u = 0; u < PD; ++u) {
      if (f + u >= n_it) break;
      const long long m0 = cur_tile * BM;
      const int k0 = cur_chunk * KC;
      float* Xb = Xs + ((PD % 2 == 0) ? u % 2 : (int)((f + u) & 1)) * (BM * LDX);
      {   // stage: raw chunk -> activated operand tile
        const f32x4 t_sc = *reinterpret_cast<const f32x4*>(Tb + k0 + 4 * q), t_sh = *reinterpret_cast<const f32x4*>(Tb + K + k0 + 4 * q),
                    t_fl = *reinterpret_cast<const f32x4*>(Tb + 2 * K + k0 + 4 * q);
#pragma unroll
        for (int j = 0; j < NX; ++j) {
          const int r = r0 + RPP * j;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (m0 + r < M) v = tf4_tab(xr[u][j], t_sc, t_sh, t_fl);
          *reinterpret_cast<f32x4*>(Xb + r * LDX + 4 * q) = v;
        }
      }
      JN_STAMP(stamp_i); ++stamp_i;
      __syncthreads();
      JN_STAMP(stamp_i); ++stamp_i;
      if (f + u + PD < n_it) fetch(xr[u]);
      {
        const float* xrow = Xb + (wm * PT * 16 + lm) * LDX + 4 * g;
        const float* wrow = Ws + (wn * CTW * 16 + lm) * LDW + k0 + 4 * g;
#pragma unroll
        for (int kk = 0; kk < KC; kk += 16) {
          f32x4 xb[PT], wa[CTW];
#pragma unroll
          for (int p = 0; p < PT; ++p) xb[p] = *reinterpret_cast<const f32x4*>(xrow + p * 16 * LDX + kk);
#pragma unroll
          for (int c = 0; c < CTW; ++c) wa[c] = *reinterpret_cast<const f32x4*>(wrow + c * 16 * LDW + kk);
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int c = 0; c < CTW; ++c)
#pragma unroll
              for (int p = 0; p < PT; ++p)
                acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][j], xb[p][j], acc[p][c], 0, 0, 0);
        }
      }
      JN_STAMP(stamp_i); ++stamp_i;
      if (++cur_chunk == nchunks) {
        cur_chunk = 0;
#pragma unroll
        for (int p = 0; p < PT; ++p) {
          const long long m = m0 + (wm * PT + p) * 16 + lm;
#pragma unroll
          for (int c = 0; c < CTW; ++c) {
            const int n = n0 + (wn * CTW + c) * 16 + 4 * g;
            f32x4 v = acc[p][c];
            acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (m >= M || n >= Nc) continue;
            float* op = out + m * out_ld + n;
            if (accumulate) v += *reinterpret_cast<const f32x4*>(op);
            *reinterpret_cast<f32x4*>(op) = v;
            s1[c] += v;
            s2[c] += v * v;
          }
        }
        cur_tile += gridDim.x;
        JN_STAMP(stamp_i); ++stamp_i;
      }
    }
  }
  JN_STAMP(30);
  (void)stamp_i;
  if (stats) {
    __syncthreads();
    wave_stats_to_lds<CTW>(s1, s2, red + wm * 32 * CT + 2 * (wn * CTW * 16), lane, Nc - n0 - wn * CTW * 16);
    __syncthreads();
    if (tid < 32 * CT && n0 + (tid >> 1) < Nc) {
      float v = 0.0f;
#pragma unroll
      for (int qq = 0; qq < WM; ++qq) v += red[qq * 32 * CT + tid];
      atomicAdd(&stats[(blockIdx.x % nrep) * rep_stride + 2 * n0 + tid], (double)v);
    }
  }
  JN_STAMP(31);
}

const char* g_pw_res_force = nullptr;   // tools/pwbench.hip: forced "ct,kc,bm,pd"

template <int CT, int KC, int BM, int PD, bool WT>
static void launch_pw_res_t(const ConvArgs& a, long long M, int max_wg_per_cu, hipStream_t s) {
  const int K = a.cin;
  const size_t smem = ((size_t)16 * CT * (K + 8) + 2 * BM * (KC + 8) + 3 * K + 32 * CT * 4) * sizeof(float);
  auto kern = pw_res_kernel<CT, KC, BM, PD, WT>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const long long n_tiles = (M + BM - 1) / BM;
  const int ny = (a.cout + 16 * CT - 1) / (16 * CT);
  int per_cu = (int)std::min<size_t>((size_t)max_wg_per_cu, (160 * 1024) / smem);
  if (per_cu < 1) per_cu = 1;
  const int nz = a.n_slots > 1 ? a.n_slots : 1;
  long long gx = (256LL * per_cu) / ((long long)ny * nz);
  gx = std::max<long long>(8, gx / 8 * 8);
  if (gx > n_tiles) gx = n_tiles;
  dim3 grid((unsigned)gx, (unsigned)ny, (unsigned)nz);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, (const float*)a.in, a.in_ld, a.itab, a.w,
                     WT ? a.cout : a.cin, (float*)a.out, a.out_ld, M, K, a.cout, a.accumulate, a.stats, a.stats_rep_stride,
                     a.stats_nrep > 0 ? a.stats_nrep : JN_NREP, a.skip_flag, a.skip_when, a.in_slot_stride, a.out_slot_stride,
                     a.tab_slot_stride);
}

// Shapes the resident-weight kernel takes (everything else stays with pw_mfma_kernel / pw_narrow_kernel): fp32, no
// bias / activation epilogue, K a multiple of 64 (32 for K = 32 * odd), at least 64 input and output channels.
bool pw_res_supported(const ConvArgs& a) {
  static const bool off = std::getenv("JN_NO_PW_RES") != nullptr;
  if (off || a.bf16_mfma || a.in_dtype != JN_F32 || a.out_dtype != JN_F32 || a.bias || a.act != ACT_NONE) return false;
  if (a.cin < 64 || a.cout < 64 || a.cin % 32 != 0 || a.cout % 4 != 0 || a.cin > 1024) return false;
  return true;
}

// Picks the weight slice (CT channel tiles per workgroup) so that the slice fits LDS with the two pixel buffers, and
// enough slices / tiles exist to occupy the chip.
int launch_pw_res(const ConvArgs& a, hipStream_t s) {
  const long long M = (long long)a.N * a.H * a.W;
  const int K = a.cin, N = a.cout;
  const int nz = a.n_slots > 1 ? a.n_slots : 1;
  // largest CT in {8, 4, 2} whose slice + buffers fit 160 KB (leaving room for two workgroups when the slice is small)
  auto lds = [&](int ct, int kc, int bm) { return ((size_t)16 * ct * (K + 8) + 2 * bm * (kc + 8) + 3 * K + 128 * ct) * sizeof(float); };
  const int kc = (K % 64 == 0) ? 64 : 32;
  int ct = 8;
  while (ct > 2 && (16 * ct > N || lds(ct, kc, 64) > 150 * 1024)) ct >>= 1;
  if (lds(ct, kc, 64) > 150 * 1024) return -1;
  // small problems: more, narrower slices so that >= ~256 workgroups exist
  const long long tiles64 = (M + 63) / 64;
  while (ct > 2 && tiles64 * nz * ((N + 16 * ct - 1) / (16 * ct)) < 320) ct >>= 1;
  static const char* force_env = std::getenv("JN_PW_RES_CFG");       // "ct,kc,bm,pd": tuning aid
  const char* force = g_pw_res_force ? g_pw_res_force : force_env;
  int bm = 64, pd = 2, kcc = kc;
  if (force) { int f_ct, f_kc, f_bm, f_pd; if (sscanf(force, "%d,%d,%d,%d", &f_ct, &f_kc, &f_bm, &f_pd) == 4) { if (f_ct) ct = f_ct; if (f_kc && K % f_kc == 0) kcc = f_kc; if (f_bm) bm = f_bm; if (f_pd) pd = f_pd; } }
  const bool wt = a.w_transposed != 0;
#define JN_PR(CT_, KC_, BM_, PD_)                                                                          \
  if (ct == CT_ && kcc == KC_ && bm == BM_ && pd == PD_) {                                                 \
    if (wt) launch_pw_res_t<CT_, KC_, BM_, PD_, true>(a, M, 2, s); else launch_pw_res_t<CT_, KC_, BM_, PD_, false>(a, M, 2, s); \
    return 0;                                                                                              \
  }
  JN_PR(8, 64, 64, 2) JN_PR(4, 64, 64, 2) JN_PR(2, 64, 64, 2)
  JN_PR(8, 32, 64, 2) JN_PR(4, 32, 64, 2) JN_PR(2, 32, 64, 2)
  JN_PR(8, 64, 64, 1) JN_PR(4, 64, 64, 1) JN_PR(2, 64, 64, 1)
  JN_PR(8, 64, 32, 2) JN_PR(4, 64, 32, 2) JN_PR(2, 64, 32, 2)
  JN_PR(8, 64, 128, 2) JN_PR(4, 64, 128, 2)
#undef JN_PR
  return -1;
}

}  // namespace jnr
