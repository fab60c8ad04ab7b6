// Wide 1x1 convolutions (GEMM with K, N >= 64) for gfx950: weight-stationary, barrier-free (pw_dir_kernel).  The
// LDS-pipelined predecessor of round 2 (pw_res_kernel: two pixel buffers, one barrier per chunk — measured at parity with
// pw_mfma_kernel and superseded) lives in tools/pwres_legacy.hip for the comparison benchmark tools/pwbench.hip only;
// the notes below on the LDS row stride and the persistent grid apply to both.
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

// fp32 value -> three bf16 terms whose sum carries its 24 significant bits (round-to-nearest at every step)
__device__ __forceinline__ void split3(const f32x4& v, bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bf16_t a = (bf16_t)v[e];
    const float r = v[e] - (float)a;
    const bf16_t b = (bf16_t)r;
    const float r2 = r - (float)b;
    h[e] = a; m[e] = b; l[e] = (bf16_t)r2;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight-stationary, barrier-free form for slices of at most 128 output channels: the pixel operand never goes through
// LDS.  A wave owns 16-pixel tiles; it fetches its tile's values straight into registers IN MFMA FRAGMENT LAYOUT (lane
// (pixel lm, group g) reads the 4 — split path: 8 — consecutive input channels its k-steps need), applies the
// "normalize on read" transform there, and multiplies against weight fragments read from the resident LDS copy.  After
// the one barrier behind the weight load the waves never synchronise again: loads of the next step, transform VALU,
// MFMAs and stores of different waves overlap freely (two workgroups per CU).  The in-kernel stamps of pw_res_kernel
// showed why this matters on these shapes: with one 4-wave workgroup per CU its stage -> barrier -> MFMA -> store
// phases ran back to back and the matrix pipe was busy a third of the time.
// A step = KH = 64 input channels of one tile; steps are prefetched PF ahead in registers.
template <int CTW, bool WT, bool SP, int PF = 2, bool ST = true, bool TF = true>
__global__ __launch_bounds__(256, 2) void pw_dir_kernel(
    const float* __restrict__ x, int x_ld, ChanTab it, const float* __restrict__ w, int w_ld, float* __restrict__ out,
    int out_ld, long long M, int K, int Nc, int accumulate, double* __restrict__ stats, long long rep_stride, int nrep,
    const int* __restrict__ skip_flag, int skip_when, long long x_slot, long long out_slot, long long tab_slot) {
  if (skip_flag && *skip_flag >= skip_when) return;
  x += blockIdx.z * x_slot; out += blockIdx.z * out_slot;
  it.sc += blockIdx.z * tab_slot; it.sh += blockIdx.z * tab_slot; it.fl += blockIdx.z * tab_slot;
  constexpr int KH = 64, NF = KH / 16;                           // frags (float4 per lane) per step; PF = prefetch depth (steps)
  constexpr int NCH = 16 * CTW;                                   // output channels of the slice
  const int LDW = K + 8, LDWh = K + 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* Ws = reinterpret_cast<float*>(smem_raw);                 // [NCH][LDW]
  bf16_t* Wh = reinterpret_cast<bf16_t*>(smem_raw);               // split: [3][NCH][LDWh]
  float* Tb = SP ? reinterpret_cast<float*>(Wh + 3 * NCH * LDWh) : Ws + NCH * LDW;    // [3][K]
  float* red = Tb + 3 * K;                                        // [4 waves][NCH][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.y * NCH;
  const int nsteps = K / KH;
  const long long n_tiles = (M + 15) / 16;
  const long long wid = (long long)blockIdx.x * 4 + wave, wstride = (long long)gridDim.x * 4;
  const long long my_tiles = wid < n_tiles ? (n_tiles - wid + wstride - 1) / wstride : 0;
  const long long n_it = my_tiles * nsteps;

  f32x4 xr[PF][NF];
  long long pf_tile = wid; int pf_step = 0;
  auto fetch = [&](f32x4 (&dst)[NF]) {
#ifdef JN_PWDIR_HOT                         // tools/pwdirbench.hip: every fetch from the same few (cache-resident) tiles
    const long long m = (pf_tile & 7) * 16 + lm;
#else
    const long long m = pf_tile * 16 + lm;
#endif
    const float* xp = x + m * x_ld + pf_step * KH;
#pragma unroll
    for (int j = 0; j < NF; ++j) {
      dst[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      // fp32 MFMA: k-steps 16 j .. 16 j + 15, this lane the quad 4 g;  split path: k 32 (j / 2) + 8 g + 4 (j & 1)
      const int k = SP ? 32 * (j >> 1) + 8 * g + 4 * (j & 1) : 16 * j + 4 * g;
      if (m < M) dst[j] = *reinterpret_cast<const f32x4*>(xp + k);
    }
    if (++pf_step == nsteps) { pf_step = 0; pf_tile += wstride; }
  };
#pragma unroll
  for (int u = 0; u < PF; ++u)
    if (u < n_it) fetch(xr[u]);

  {   // weight slice -> LDS (all loads of a batch in flight before the first store), table meanwhile
    constexpr int WB = 8;                                         // a 64 x 128 slice in one batch
    const int NQ = NCH / 4, KQ = K / 4;
    const int total = WT ? K * NQ : NCH * KQ;
    f32x4 wr[WB];
    auto wload = [&](int base) {
#pragma unroll
      for (int j = 0; j < WB; ++j) {
        const int i = base + tid + 256 * j;
        wr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (i < total) {
          if (WT) {
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
        if (i >= total) continue;
        if constexpr (SP) {
          bf16x4 h, m, l;
          split3(wr[j], h, m, l);
          const int plane = NCH * LDWh;
          if (WT) {
            const int k = i / NQ, nq = i - k * NQ;
#pragma unroll
            for (int e = 0; e < 4; ++e) { bf16_t* d = Wh + (4 * nq + e) * LDWh + k; d[0] = h[e]; d[plane] = m[e]; d[2 * plane] = l[e]; }
          } else {
            const int r = i / KQ, kq = i - r * KQ;
            bf16_t* d = Wh + r * LDWh + 4 * kq;
            *reinterpret_cast<bf16x4*>(d) = h; *reinterpret_cast<bf16x4*>(d + plane) = m; *reinterpret_cast<bf16x4*>(d + 2 * plane) = l;
          }
        } else if (WT) {
          const int k = i / NQ, nq = i - k * NQ;
#pragma unroll
          for (int e = 0; e < 4; ++e) Ws[(4 * nq + e) * LDW + k] = wr[j][e];
        } else {
          const int r = i / KQ, kq = i - r * KQ;
          *reinterpret_cast<f32x4*>(Ws + r * LDW + 4 * kq) = wr[j];
        }
      }
    };
    wload(0);
    if constexpr (TF) tab_to_lds(Tb, K, K, it, tid, 256);
    wstore(0);
    for (int base = 256 * WB; base < total; base += 256 * WB) { wload(base); wstore(base); }
  }
  __syncthreads();                                                // the only barrier before the statistics epilogue

  f32x4 acc[CTW], s1[CTW], s2[CTW];
#pragma unroll
  for (int c = 0; c < CTW; ++c) { acc[c] = f32x4{0.f, 0.f, 0.f, 0.f}; s1[c] = acc[c]; s2[c] = acc[c]; }
  long long cur_tile = wid; int cur_step = 0;
  for (long long f = 0; f < n_it; f += PF) {
#pragma unroll
    for (int u = 0; u < PF; ++u) {
      if (f + u >= n_it) break;
      const int k0 = cur_step * KH;
      // ---- transform in registers ----
      f32x4 xa[NF];
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const int k = k0 + (SP ? 32 * (j >> 1) + 8 * g + 4 * (j & 1) : 16 * j + 4 * g);
        if constexpr (TF)
          xa[j] = tf4_tab(xr[u][j], *reinterpret_cast<const f32x4*>(Tb + k), *reinterpret_cast<const f32x4*>(Tb + K + k),
                          *reinterpret_cast<const f32x4*>(Tb + 2 * K + k));
        else
          xa[j] = xr[u][j];
      }
      if (cur_tile * 16 + lm >= M) {                              // rows past the end contribute nothing (and store nothing)
#pragma unroll
        for (int j = 0; j < NF; ++j) xa[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (f + u + PF < n_it) fetch(xr[u]);
      // ---- MFMAs against the resident weights ----
      if constexpr (SP) {
        const int plane = NCH * LDWh;
#pragma unroll
        for (int kk = 0; kk < KH / 32; ++kk) {
          bf16x4 h0, m0, l0, h1, m1, l1;
          split3(xa[2 * kk], h0, m0, l0);
          split3(xa[2 * kk + 1], h1, m1, l1);
          const bf16x8 xh = __builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7);
          const bf16x8 xm = __builtin_shufflevector(m0, m1, 0, 1, 2, 3, 4, 5, 6, 7);
          const bf16x8 xl = __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7);
          const bf16_t* wrow = Wh + lm * LDWh + k0 + 32 * kk + 8 * g;
#pragma unroll
          for (int c = 0; c < CTW; ++c) {
            const bf16x8 wh = *reinterpret_cast<const bf16x8*>(wrow + c * 16 * LDWh);
            const bf16x8 wm = *reinterpret_cast<const bf16x8*>(wrow + c * 16 * LDWh + plane);
            const bf16x8 wl = *reinterpret_cast<const bf16x8*>(wrow + c * 16 * LDWh + 2 * plane);
            f32x4 d = acc[c];
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, xh, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xl, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xm, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, xh, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xm, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, xh, d, 0, 0, 0);
            acc[c] = d;
          }
        }
      } else {
        // weight fragments of k-substep j + 1 are read from LDS while the MFMAs of substep j run (left to itself the
        // compiler issues every ds_read right before its four MFMAs and waits out the LDS latency each time: the matrix
        // pipe sat idle for about half of every step)
        const float* wrow = Ws + lm * LDW + k0 + 4 * g;
        f32x4 wa[2][CTW];
#pragma unroll
        for (int c = 0; c < CTW; ++c) wa[0][c] = *reinterpret_cast<const f32x4*>(wrow + c * 16 * LDW);
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          if (j + 1 < NF) {
#pragma unroll
            for (int c = 0; c < CTW; ++c) wa[(j + 1) & 1][c] = *reinterpret_cast<const f32x4*>(wrow + c * 16 * LDW + 16 * (j + 1));
          }
#ifdef JN_PWDIR_NOMFMA                     // tools/pwdirbench.hip: the same loads and transform, no matrix work
#pragma unroll
          for (int c = 0; c < CTW; ++c) acc[c] += wa[j & 1][c] * xa[j];
#else
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < CTW; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j & 1][c][e], xa[j][e], acc[c], 0, 0, 0);
#endif
        }
      }
      if (++cur_step == nsteps) {
        cur_step = 0;
        const long long m = cur_tile * 16 + lm;
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
          const int n = n0 + c * 16 + 4 * g;
          f32x4 v = acc[c];
          acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (m >= M || n >= Nc) continue;
          float* op = out + m * out_ld + n;
          if (accumulate) v += *reinterpret_cast<const f32x4*>(op);
          *reinterpret_cast<f32x4*>(op) = v;
          if constexpr (ST) { s1[c] += v; s2[c] += v * v; }
        }
        cur_tile += wstride;
      }
    }
  }
  if (ST && stats) {
    wave_stats_to_lds<CTW>(s1, s2, red + wave * 2 * NCH, lane, Nc - n0);
    __syncthreads();
    if (tid < 2 * NCH && n0 + (tid >> 1) < Nc)
      atomicAdd(&stats[(blockIdx.x % nrep) * rep_stride + 2 * n0 + tid],
                (double)(red[tid] + red[2 * NCH + tid] + red[4 * NCH + tid] + red[6 * NCH + tid]));
  }
}

static size_t pw_dir_lds(int ctw, int K, bool split) {
  const size_t tail = ((size_t)3 * K + 8 * 16 * ctw) * sizeof(float);
  return (split ? (size_t)6 * 16 * ctw * (K + 16) : (size_t)4 * 16 * ctw * (K + 8)) + tail;
}

template <int CTW, bool WT, bool SP, int PF = 2, bool ST = true, bool TF = true>
static void launch_pw_dir_t(const ConvArgs& a, long long M, hipStream_t s) {
  const int K = a.cin;
  const size_t smem = pw_dir_lds(CTW, K, SP);
  auto kern = pw_dir_kernel<CTW, WT, SP, PF, ST, TF>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const long long n_tiles = (M + 15) / 16;
  const int ny = (a.cout + 16 * CTW - 1) / (16 * CTW), nz = a.n_slots > 1 ? a.n_slots : 1;
  const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, (160 * 1024) / smem));
  // one resident round: as many workgroups as the chip holds at once, never a few more (20 steps x 4 slices with the
  // old round-up-to-8 rule gave 640 workgroups for 512 places: a second round at a quarter of the occupancy)
  long long gx = std::max<long long>(1, 256LL * per_cu / ((long long)ny * nz));
  if (nz == 1 && ny > 1) gx = std::max<long long>(8, gx / 8 * 8);
  gx = std::min<long long>(gx, (n_tiles + 3) / 4);
  dim3 grid((unsigned)gx, (unsigned)ny, (unsigned)nz);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, (const float*)a.in, a.in_ld, a.itab, a.w, WT ? a.cout : a.cin,
                     (float*)a.out, a.out_ld, M, K, a.cout, a.accumulate, a.stats, a.stats_rep_stride,
                     a.stats_nrep > 0 ? a.stats_nrep : JN_NREP, a.skip_flag, a.skip_when, a.in_slot_stride, a.out_slot_stride,
                     a.tab_slot_stride);
}

// ctw = channel tiles per workgroup slice (0: automatic — the widest slice <= 128 channels whose weights leave room for two
// workgroups per CU); returns -1 when the shape does not fit (K % 64, slice too large).
int g_pw_dir_pf = 0;      // tools/pwbench.hip: prefetch depth override (3, 4) for the fp32 64-channel-slice kernel
int launch_pw_dir(const ConvArgs& a, int ctw, int split, hipStream_t s) {
  const long long M = (long long)a.N * a.H * a.W;
  const int K = a.cin, N = a.cout;
  if (K % 64 != 0) return -1;
  if (!ctw) {
    // 64-channel slices measured best on every shape of the nano PAFPN (tools/pwbench.hip: 28x28 128 -> 128 28.4 us
    // against 29.3 with 128-channel and 32.1 with 32-channel slices); narrower only where the weights would not leave
    // room for two workgroups per CU (K = 512)
    for (int c : {4, 2}) {
      if (16 * c > N && c > 2) continue;
      if (pw_dir_lds(c, K, split != 0) <= 78 * 1024) { ctw = c; break; }
    }
    if (!ctw) return -1;
  }
  if (pw_dir_lds(ctw, K, split != 0) > 160 * 1024) return -1;
  const bool wt = a.w_transposed != 0;
  // data gradients (transposed weights) collect no statistics: without the two sum registers per tile a third step of
  // prefetch fits the 256-VGPR budget (depth 4: measured best of 2 / 3 / 4 in round 2)
  if (wt && !split && !a.stats && (ctw == 4 || ctw == 2)) {
    // (gradient views carry the identity table: the transform — two transcendentals and half a dozen VALU ops per value,
    //  as long as the MFMAs of the step, tools/pwdirbench.hip — is compiled out)
    if (a.in_identity) {
      if (ctw == 4) launch_pw_dir_t<4, true, false, 4, false, false>(a, M, s); else launch_pw_dir_t<2, true, false, 4, false, false>(a, M, s);
    } else {
      if (ctw == 4) launch_pw_dir_t<4, true, false, 4, false>(a, M, s); else launch_pw_dir_t<2, true, false, 4, false>(a, M, s);
    }
    return 0;
  }
  if (wt && split && !a.stats && a.in_identity && (ctw == 4 || ctw == 2)) {
    if (ctw == 4) launch_pw_dir_t<4, true, true, 2, false, false>(a, M, s); else launch_pw_dir_t<2, true, true, 2, false, false>(a, M, s);
    return 0;
  }
  if (ctw == 4 && !split && !wt && g_pw_dir_pf == 3) { launch_pw_dir_t<4, false, false, 3>(a, M, s); return 0; }
  if (ctw == 4 && !split && !wt && g_pw_dir_pf == 4) { launch_pw_dir_t<4, false, false, 4>(a, M, s); return 0; }
#define JN_PD(C_)                                                                                   \
  if (ctw == C_) {                                                                                  \
    if (split) { if (wt) launch_pw_dir_t<C_, true, true>(a, M, s); else launch_pw_dir_t<C_, false, true>(a, M, s); }     \
    else { if (wt) launch_pw_dir_t<C_, true, false>(a, M, s); else launch_pw_dir_t<C_, false, false>(a, M, s); }         \
    return 0;                                                                                       \
  }
  JN_PD(8) JN_PD(4) JN_PD(2)
#undef JN_PD
  return -1;
}

// Shapes the resident-weight kernel takes (everything else stays with pw_mfma_kernel / pw_narrow_kernel): fp32, no
// bias / activation epilogue, K a multiple of 64 (32 for K = 32 * odd), at least 64 input and output channels.
bool pw_res_supported(const ConvArgs& a) {
  if (a.bf16_mfma || a.in_dtype != JN_F32 || a.out_dtype != JN_F32 || a.bias || a.act != ACT_NONE) return false;
  if (a.cin < 64 || a.cout < 64 || a.cin % 32 != 0 || a.cout % 4 != 0 || a.cin > 1024) return false;
  return true;
}

// (name kept from round 2: the shape test of the resident-weight kernels)
// The production route for wide 1x1 layers (forward and data gradient): the barrier-free weight-stationary kernel on
// exact-fp32 MFMA.  JN_PW_SPLIT=1 switches its products to the split-bf16 form: 2.07 against 2.15 ms per forward pass and
// 117.0 against 118.1 ms per iteration (these kernels are latency-, not matrix-bound), but a rounding noise ~10x that of
// an fp32 fma chain per layer, which the 77 train-mode BatchNorms amplify to 1.0 - 1.5e-3 on the first layers'
// gradients — not worth 1 %.  Shapes the kernel measured no better on stay with their old kernels: 64 -> 64 on fewer
// than 65536 pixels (pw_mfma_kernel, 14.3 against 14.6 us at 28x28).
int launch_pw_wide(const ConvArgs& a, hipStream_t s) {
  static const bool off = std::getenv("JN_NO_PW_DIR") != nullptr;
  static const bool exact = std::getenv("JN_PW_SPLIT") == nullptr;
  if (off || !pw_res_supported(a) || a.cin % 64 != 0) return -1;
  const long long M = (long long)a.N * a.H * a.W * (a.n_slots > 1 ? a.n_slots : 1);
  if (a.cin == 64 && a.cout == 64 && M < 65536) return -1;
  // (split products for the data gradients only — the backward is linear in g, so a GEMM error of ~5e-6 travels up the
  //  chain without the amplification a forward perturbation gets — were measured in round 2: 3.30 -> 3.27 ms per step,
  //  not taken)
  return launch_pw_dir(a, 0, exact ? 0 : 1, s);
}

}  // namespace jnr
