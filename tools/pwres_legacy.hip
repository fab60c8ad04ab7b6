// Round 2's LDS-pipelined resident-weight 1x1 kernel (pw_res_kernel), superseded by pw_dir_kernel (kernels_pwres.hip) and
// kept OUT of libjnroll.so: a comparison kernel for tools/pwbench.hip (include after kernels_pwres.hip).
#pragma once
namespace jnr {

// SP: operands split into three bf16 terms each, six v_mfma_f32_16x16x32_bf16 per product block (all cross terms down
// to 2^-24 of the product: the accuracy of an fp32 multiply-add chain) — 96 matrix-pipe cycles per 16 x 16 x 32 block
// instead of the 256 of eight v_mfma_f32_16x16x4_f32.  SP = false: exact fp32 (the default; JN_PW_SPLIT=1 selects SP).
template <int CT, int KC, int BM, int PD, bool WT, bool SP>
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
  static_assert(!SP || KC % 32 == 0, "split path consumes 32 k per MFMA");
  const int LDW = K + 8;
  constexpr int LDXh = KC + 16;                                   // split path: bf16 rows (40 dwords mod 64 for KC = 64)
  const int LDWh = K + 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* Ws = reinterpret_cast<float*>(smem_raw);                 // [16*CT][LDW]   resident weight slice
  float* Xs = Ws + 16 * CT * LDW;                                 // [2][BM][LDX]   pixel chunks
  bf16_t* Wh = reinterpret_cast<bf16_t*>(smem_raw);               // split path: [3][16*CT][LDWh]
  bf16_t* Xh = Wh + 3 * 16 * CT * LDWh;                           //             [2][3][BM][LDXh]
  float* Tb = SP ? reinterpret_cast<float*>(Xh + 6 * BM * LDXh) : Xs + 2 * BM * LDX;   // [3][K] input table
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
    constexpr int WB = 16;        // 16 float4 per thread cover a 128 x 128 slice in ONE round trip
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
          if constexpr (SP) {
            bf16x4 h, m, l;
            split3(wr[j], h, m, l);
            const int plane = 16 * CT * LDWh;
            if (WT) {
              const int k = i / NQ, nq = i - k * NQ;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                bf16_t* d = Wh + (4 * nq + e) * LDWh + k;
                d[0] = h[e]; d[plane] = m[e]; d[2 * plane] = l[e];
              }
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
      const int bsel = (PD % 2 == 0) ? u % 2 : (int)((f + u) & 1);
      float* Xb = Xs + bsel * (BM * LDX);
      bf16_t* Xbh = Xh + bsel * (3 * BM * LDXh);
      {   // stage: raw chunk -> activated operand tile
        const f32x4 t_sc = *reinterpret_cast<const f32x4*>(Tb + k0 + 4 * q), t_sh = *reinterpret_cast<const f32x4*>(Tb + K + k0 + 4 * q),
                    t_fl = *reinterpret_cast<const f32x4*>(Tb + 2 * K + k0 + 4 * q);
#pragma unroll
        for (int j = 0; j < NX; ++j) {
          const int r = r0 + RPP * j;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (m0 + r < M) v = tf4_tab(xr[u][j], t_sc, t_sh, t_fl);
          if constexpr (SP) {
            bf16x4 h, m, l;
            split3(v, h, m, l);
            bf16_t* d = Xbh + r * LDXh + 4 * q;
            *reinterpret_cast<bf16x4*>(d) = h; *reinterpret_cast<bf16x4*>(d + BM * LDXh) = m;
            *reinterpret_cast<bf16x4*>(d + 2 * BM * LDXh) = l;
          } else {
            *reinterpret_cast<f32x4*>(Xb + r * LDX + 4 * q) = v;
          }
        }
      }
      JN_STAMP(stamp_i); ++stamp_i;
      __syncthreads();
      JN_STAMP(stamp_i); ++stamp_i;
      if (f + u + PD < n_it) fetch(xr[u]);
      if constexpr (SP) {
        const bf16_t* xrow = Xbh + (wm * PT * 16 + lm) * LDXh + 8 * g;
        const bf16_t* wrow = Wh + (wn * CTW * 16 + lm) * LDWh + k0 + 8 * g;
        const int wplane = 16 * CT * LDWh;
#pragma unroll
        for (int kk = 0; kk < KC; kk += 32) {
          bf16x8 xb[PT][3], wa[CTW][3];
#pragma unroll
          for (int p = 0; p < PT; ++p)
#pragma unroll
            for (int t = 0; t < 3; ++t) xb[p][t] = *reinterpret_cast<const bf16x8*>(xrow + t * (BM * LDXh) + p * 16 * LDXh + kk);
#pragma unroll
          for (int c = 0; c < CTW; ++c)
#pragma unroll
            for (int t = 0; t < 3; ++t) wa[c][t] = *reinterpret_cast<const bf16x8*>(wrow + t * wplane + c * 16 * LDWh + kk);
          // smallest terms first: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)
#pragma unroll
          for (int c = 0; c < CTW; ++c)
#pragma unroll
            for (int p = 0; p < PT; ++p) {
              f32x4 d = acc[p][c];
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[c][2], xb[p][0], d, 0, 0, 0);
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[c][0], xb[p][2], d, 0, 0, 0);
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[c][1], xb[p][1], d, 0, 0, 0);
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[c][1], xb[p][0], d, 0, 0, 0);
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[c][0], xb[p][1], d, 0, 0, 0);
              d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[c][0], xb[p][0], d, 0, 0, 0);
              acc[p][c] = d;
            }
        }
      } else {
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

static size_t pw_res_lds(int ct, int kc, int bm, int K, bool split) {
  const size_t tail = ((size_t)3 * K + 32 * ct * 4) * sizeof(float);
  if (split) return (size_t)6 * 16 * ct * (K + 16) + (size_t)12 * bm * (kc + 16) + tail;
  return ((size_t)16 * ct * (K + 8) + 2 * bm * (kc + 8)) * sizeof(float) + tail;
}

template <int CT, int KC, int BM, int PD, bool WT, bool SP>
static void launch_pw_res_t(const ConvArgs& a, long long M, int max_wg_per_cu, hipStream_t s) {
  const int K = a.cin;
  const size_t smem = pw_res_lds(CT, KC, BM, K, SP);
  auto kern = pw_res_kernel<CT, KC, BM, PD, WT, SP>;
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
  long long gx = (256LL * per_cu + (long long)ny * nz - 1) / ((long long)ny * nz);
  if (ny > 1) gx = std::max<long long>(8, (gx + 7) / 8 * 8);     // the N slices of a pixel tile on one XCD (shared L2)
  if (gx > n_tiles) gx = n_tiles;
  dim3 grid((unsigned)gx, (unsigned)ny, (unsigned)nz);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, s, (const float*)a.in, a.in_ld, a.itab, a.w,
                     WT ? a.cout : a.cin, (float*)a.out, a.out_ld, M, K, a.cout, a.accumulate, a.stats, a.stats_rep_stride,
                     a.stats_nrep > 0 ? a.stats_nrep : JN_NREP, a.skip_flag, a.skip_when, a.in_slot_stride, a.out_slot_stride,
                     a.tab_slot_stride);
}

// Picks the configuration: the weight slice (CT channel tiles per workgroup) must fit LDS with the two pixel buffers;
// two workgroups per CU whenever a slice allows it (one workgroup's staging / stores run under the other's MFMAs); more,
// narrower slices when the problem has too few pixel tiles to occupy the chip.
int launch_pw_res(const ConvArgs& a, hipStream_t s) {
  const long long M = (long long)a.N * a.H * a.W;
  const int K = a.cin, N = a.cout;
  const int nz = a.n_slots > 1 ? a.n_slots : 1;
  static const bool exact = std::getenv("JN_PW_SPLIT") == nullptr;     // fp32 MFMA unless the split-bf16 products are asked for
  static const char* force_env = std::getenv("JN_PW_RES_CFG");         // "ct,kc,bm,pd,split": tuning aid (0 = automatic)
  const char* force = g_pw_res_force ? g_pw_res_force : force_env;
  int sp = exact ? 0 : 1;
  int kc = (K % 64 == 0) ? 64 : 32, bm = 32, pd = 2, ct = 0;
  if (force) {
    int f_ct = 0, f_kc = 0, f_bm = 0, f_pd = 0, f_sp = -1;
    if (sscanf(force, "%d,%d,%d,%d,%d", &f_ct, &f_kc, &f_bm, &f_pd, &f_sp) >= 4) {
      if (f_ct) ct = f_ct;
      if (f_kc && K % f_kc == 0) kc = f_kc;
      if (f_bm) bm = f_bm;
      if (f_pd) pd = f_pd;
      if (f_sp >= 0) sp = f_sp;
    }
  }
  if (!ct) {
    const long long tiles = (M + bm - 1) / bm;
    for (int c : {8, 4, 2}) {                  // widest slice that leaves room for two workgroups per CU
      if (16 * c > N && c > 2) continue;
      if (pw_res_lds(c, kc, bm, K, sp) <= 78 * 1024) { ct = c; break; }
    }
    if (!ct)
      for (int c : {8, 4, 2}) {
        if (16 * c > N && c > 2) continue;
        if (pw_res_lds(c, kc, bm, K, sp) <= 156 * 1024) { ct = c; break; }
      }
    if (!ct) return -1;
    while (ct > 2 && tiles * nz * ((N + 16 * ct - 1) / (16 * ct)) < 400) ct >>= 1;
  }
  if (pw_res_lds(ct, kc, bm, K, sp) > 160 * 1024) return -1;
  const bool wt = a.w_transposed != 0;
#define JN_PR2(CT_, KC_, BM_, SP_)                                                                                            \
  if (ct == CT_ && kc == KC_ && bm == BM_ && sp == SP_ && pd == 2) {                                                          \
    if (wt) launch_pw_res_t<CT_, KC_, BM_, 2, true, SP_ != 0>(a, M, 2, s); else launch_pw_res_t<CT_, KC_, BM_, 2, false, SP_ != 0>(a, M, 2, s); \
    return 0;                                                                                                                 \
  }
#define JN_PR(CT_, KC_, BM_) JN_PR2(CT_, KC_, BM_, 0) JN_PR2(CT_, KC_, BM_, 1)
  JN_PR(8, 64, 64) JN_PR(4, 64, 64) JN_PR(2, 64, 64)
  JN_PR(8, 64, 32) JN_PR(4, 64, 32) JN_PR(2, 64, 32)
  JN_PR(8, 32, 64) JN_PR(4, 32, 64) JN_PR(2, 32, 64)
  JN_PR(8, 32, 32) JN_PR(4, 32, 32) JN_PR(2, 32, 32)
#undef JN_PR
#undef JN_PR2
  if (ct == 8 && kc == 64 && bm == 64 && pd == 1) {
    if (wt) launch_pw_res_t<8, 64, 64, 1, true, false>(a, M, 2, s); else launch_pw_res_t<8, 64, 64, 1, false, false>(a, M, 2, s);
    return 0;
  }
  return -1;
}

}  // namespace jnr
