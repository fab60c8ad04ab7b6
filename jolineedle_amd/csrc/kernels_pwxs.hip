// Pixel-stationary 1x1 convolution for the small maps of a forward pass (28x28 / 14x14 at the headline batch: 12 544 -
// 50 176 pixels per launch, K, N in {64 .. 512}) — round 3.
//
// z[m][n] = sum_k T(x[m][k]) * w[n][k]      (T = "normalize on read", jn_tab.h)
//
// Why another kernel: on these shapes the weight-stationary pw_dir_kernel (kernels_pwres.hip) spends a third of its time
// outside the matrix pipe (profiles/r02_pwdirbench_limiters.txt) — every 64-channel output slice is its own workgroup
// that loads a 64 KB weight slice behind a barrier, re-reads and re-TRANSFORMS the whole pixel operand (4x for N = 256:
// two transcendentals per value each time), and a wave sees only one or two tiles, so nothing reaches a steady state.
// Here the roles are swapped:
//   * a workgroup owns BM = 16 * PT pixels and ALL N output channels: its [BM][K] operand is read from HBM once,
//     transformed once (thread t always handles channel quad t % (K / 4), so its (scale, shift, flag) quad sits in
//     registers; the table is formed once per workgroup while the operand loads are in flight) and staged in LDS once:
//     two barriers in the kernel, none in the matrix loop;
//   * each of the four waves owns N / 4 output channels and streams their weight rows straight from L2 into registers in
//     MFMA fragment layout (lane (row, g) reads w[n0 + row][16 j + 4 g .. + 3]), D steps ahead of the MFMAs that use
//     them: the weights never touch LDS and are not shared between waves, so no second barrier;
//   * per 16-wide k step a wave issues PT ds_read_b128 + CTW global loads for 4 * PT * CTW MFMAs (64 for 64 pixels x 64
//     channels): the matrix pipe is the only busy unit inside the loop;
//   * the BatchNorm sums of a wave's channels are complete inside the wave (DPP row sums): no cross-wave reduction, one
//     coalesced set of fp64 atomics per workgroup.
// Exact fp32 (v_mfma_f32_16x16x4_f32), the same fragment layout and k order as pw_mfma_kernel / pw_dir_kernel: results
// are bit-identical to theirs.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "jn_kernels.h"
#include "jn_reduce.h"
#include "jn_tab.h"
#include "jn_types.h"

namespace jnr {

// UPS (round 4): the layer's output is also the source of a nearest x2 upsample into a concat slice of the next stage
// (lateral_conv0, reduce_conv1): the kernel stores every output quad to its four places in that slice as well, and the
// upsample launch — a read of the map it has just written and a 4x write — disappears.  Own instantiation, two shapes.
struct UpsDst { float* out; int ld; int W; int HW; };     // destination slice (pixel stride ld floats), source map W x (HW / W)
template <bool UPS>
__device__ __forceinline__ void ups_store(const UpsDst& u, long long m, int n, const f32x4& v) {
  if constexpr (UPS) {
    const long long img = m / u.HW;
    const int rem = (int)(m - img * u.HW), y = rem / u.W, x = rem - y * u.W;
    float* d = u.out + ((img * 2 * (u.HW / u.W) + 2 * y) * (2LL * u.W) + 2 * x) * u.ld + n;
    *reinterpret_cast<f32x4*>(d) = v;
    *reinterpret_cast<f32x4*>(d + u.ld) = v;
    d += 2LL * u.W * u.ld;
    *reinterpret_cast<f32x4*>(d) = v;
    *reinterpret_cast<f32x4*>(d + u.ld) = v;
  }
}

template <int K, int CTW, int PT, int D, bool UPS = false>
__global__ __launch_bounds__(256, (K <= 64 ? 3 : (K <= 128 || (K == 256 && CTW <= 2)) ? 2 : 1)) void pw_xs_kernel(
    const float* __restrict__ x, int x_ld, ChanTab it, const float* __restrict__ w, float* __restrict__ out, int out_ld,
    long long M, double* __restrict__ stats, long long rep_stride, int nrep, const int* __restrict__ skip_flag,
    int skip_when, UpsDst ups) {
  if (skip_flag && *skip_flag >= skip_when) return;
  constexpr int BM = 16 * PT, LDX = K + 8, KQ = K / 4, NJ = K / 16, N = 64 * CTW;
  constexpr int NX = BM * KQ / 256;                    // float4 of the operand tile per thread
  constexpr int NXB = NX < 16 ? NX : 16;               // ... fetched in batches of at most 16 (64 VGPRs in flight)
  static_assert(256 % KQ == 0 && NX >= 1 && NX % NXB == 0 && NJ % D == 0, "pw_xs tile mapping");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* Xs = reinterpret_cast<float*>(smem_raw);      // [BM][LDX]
  float* Tb = Xs + BM * LDX;                           // [3][K], later [N][2] statistics
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, g = lane >> 4;
  const long long n_tiles = (M + BM - 1) / BM;

  // ---- weight fragments: the first D k-steps (D == K / 16: all of them, once per workgroup) ----
  const float* wrow[CTW];
#pragma unroll
  for (int c = 0; c < CTW; ++c) wrow[c] = w + (long long)((wave * CTW + c) * 16 + lm) * K + 4 * g;
  f32x4 wr[D][CTW];
  auto wload_head = [&]() {
#pragma unroll
    for (int u = 0; u < D; ++u)
#pragma unroll
      for (int c = 0; c < CTW; ++c) wr[u][c] = *reinterpret_cast<const f32x4*>(wrow[c] + 16 * u);
  };
  wload_head();

  // ---- operand tile loads (every load unconditional: rows past the end re-read the last row and are zeroed after the
  //      transform — a per-load bounds branch made the compiler wait for each load before issuing the next) ----
  const int q = tid % KQ, r0 = tid / KQ;               // channel quad of this thread, first row; rows advance by 256 / KQ
  constexpr int RS = 256 / KQ;
  f32x4 xr[NXB];
  auto fetch = [&](long long m0, int b) {
#pragma unroll
    for (int u = 0; u < NXB; ++u) {
      long long m = m0 + r0 + (long long)(b * NXB + u) * RS;
      m = m < M ? m : M - 1;
      xr[u] = *reinterpret_cast<const f32x4*>(x + m * x_ld + 4 * q);
    }
  };
  long long tile = blockIdx.x;
  fetch(tile * BM, 0);
  // the (scale, shift, flag) entries of the K input channels, once per workgroup (deferred entries: fp64 arithmetic on the
  // producer's batch sums, jn_tab.h), while the first operand loads are in flight
  tab_to_lds(Tb, K, K, it, tid, 256);
  __syncthreads();
  const f32x4 sc = *reinterpret_cast<const f32x4*>(Tb + 4 * q), sh = *reinterpret_cast<const f32x4*>(Tb + K + 4 * q),
              fl = *reinterpret_cast<const f32x4*>(Tb + 2 * K + 4 * q);
  f32x4 s1[CTW], s2[CTW];
#pragma unroll
  for (int c = 0; c < CTW; ++c) { s1[c] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[c] = s1[c]; }

  // ---- persistent over pixel tiles: the next tile's operand is fetched into registers under the MFMAs of this one ----
#pragma unroll 1
  for (; tile < n_tiles; tile += gridDim.x) {
    const long long m0 = tile * BM;
#pragma unroll
    for (int b = 0; b < NX / NXB; ++b) {
      if (b > 0) fetch(m0, b);
#pragma unroll
      for (int u = 0; u < NXB; ++u) {
        const int r = r0 + (b * NXB + u) * RS;
        f32x4 v = tf4_tab(xr[u], sc, sh, fl);
        if (m0 + r >= M) v = f32x4{0.f, 0.f, 0.f, 0.f};  // rows past the end contribute nothing (T(0) is not 0)
        *reinterpret_cast<f32x4*>(Xs + r * LDX + 4 * q) = v;
      }
    }
    __syncthreads();
    if (tile + gridDim.x < n_tiles) fetch((tile + gridDim.x) * BM, 0);

    // ---- MFMAs: D-step ring of weight fragments, operand fragments from LDS ----
    f32x4 acc[CTW][PT];
#pragma unroll
    for (int c = 0; c < CTW; ++c)
#pragma unroll
      for (int p = 0; p < PT; ++p) acc[c][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xrow = Xs + lm * LDX + 4 * g;
#pragma unroll 1
    for (int j0 = 0; j0 < NJ; j0 += D) {
#pragma unroll
      for (int u = 0; u < D; ++u) {
        const int j = j0 + u;
        f32x4 xa[PT], wa[CTW];
#pragma unroll
        for (int p = 0; p < PT; ++p) xa[p] = *reinterpret_cast<const f32x4*>(xrow + p * 16 * LDX + 16 * j);
#pragma unroll
        for (int c = 0; c < CTW; ++c) wa[c] = wr[u][c];
        if constexpr (D < NJ) {                           // ring (D == NJ: every fragment stays in registers)
          // step j + D; past the end: the head of the NEXT tile's ring (same fragments for every tile)
          const int jn = j + D < NJ ? j + D : j + D - NJ;
#pragma unroll
          for (int c = 0; c < CTW; ++c) wr[u][c] = *reinterpret_cast<const f32x4*>(wrow[c] + 16 * jn);
          // keep the loads HERE, ahead of this step's MFMAs (left alone, the scheduler sank all of a body's loads to its
          // end and the next body waited out an L2 round trip on its first step)
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int c = 0; c < CTW; ++c)
#pragma unroll
            for (int p = 0; p < PT; ++p) acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][e], xa[p][e], acc[c][p], 0, 0, 0);
      }
    }

    // ---- store (lane = 4 consecutive channels of one pixel), BatchNorm sums in registers across the tiles ----
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
      const int n = (wave * CTW + c) * 16 + 4 * g;
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        const long long m = m0 + 16 * p + lm;
        const f32x4 v = acc[c][p];
        if (m < M) { *reinterpret_cast<f32x4*>(out + m * out_ld + n) = v; ups_store<UPS>(ups, m, n, v); }
        s1[c] += v; s2[c] += v * v;                       // rows past the end are exact zeros
      }
    }
    __syncthreads();                                      // every wave is done reading the tile
  }

  // ---- BatchNorm sums: through LDS to ONE coalesced set of fp64 atomics per workgroup (thread t -> value t of [N][2]):
  // issued straight from the four row-leader lanes of every wave they were 8 * CTW instructions of 4 scattered lanes each —
  // ten times the L2 atomic transactions, and the kernel's time was proportional to their number (36 - 140 us).
  if (stats) {
    float* red = Tb;                                    // [N][2] (3 K >= 2 N for every instantiated shape but K = 64, N = 128)
    if constexpr (3 * K < 2 * N) red = Xs;
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
      const int n = (wave * CTW + c) * 16 + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = row16_sum(s1[c][r]);
        const float b = row16_sum(s2[c][r]);
        if (lm == 0) { red[2 * (n + r)] = a; red[2 * (n + r) + 1] = b; }
      }
    }
    __syncthreads();
    double* st = stats + (blockIdx.x % nrep) * rep_stride;
    for (int i = tid; i < 2 * N; i += 256) atomicAdd(st + i, (double)red[i]);
  }
}

template <int K, int CTW, int PT, int D, bool UPS = false>
static void launch_pw_xs_t(const ConvArgs& a, long long M, int wg_per_cu, hipStream_t s) {
  constexpr int BM = 16 * PT;
  const size_t smem = ((size_t)BM * (K + 8) + 3 * K) * sizeof(float);
  auto kern = pw_xs_kernel<K, CTW, PT, D, UPS>;
  static int places = 0;
  if (!places) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256, smem) != hipSuccess || per_cu < 1) per_cu = 1;
    places = per_cu;
  }
  // persistent grid: at most `wg_per_cu` (<= what fits) workgroups per CU, so that every workgroup is resident from the
  // start and walks over its tiles with the next operand in flight
  const long long n_tiles = (M + BM - 1) / BM;
  const int per_cu = std::max(1, std::min(places, wg_per_cu > 0 ? wg_per_cu : 2));
  const long long gx = std::min<long long>(n_tiles, 256LL * per_cu);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(256), smem, s, (const float*)a.in, a.in_ld, a.itab, a.w, (float*)a.out,
                     a.out_ld, M, a.stats, a.stats_rep_stride, a.stats_nrep > 0 ? a.stats_nrep : JN_NREP, a.skip_flag,
                     a.skip_when, UpsDst{(float*)a.up_out, a.up_ld, a.W, a.H * a.W});
}

// Shapes the kernel takes: fp32 forward (no bias / activation epilogue, weight [N][K] not transposed, one slot),
// K in {64, 128, 256, 512}, N in {64, 128, 256}.  pt = pixel tiles per workgroup (2 or 4; 0 = the measured best for the
// shape, tools/pwxsbench.hip).  Returns -1 when the shape is not covered.
bool pw_xs_supported(const ConvArgs& a) {
  if (a.bf16_mfma || a.in_dtype != JN_F32 || a.out_dtype != JN_F32 || a.bias || a.act != ACT_NONE || a.w_transposed ||
      a.accumulate || a.n_slots > 1)
    return false;
  const int K = a.cin, N = a.cout;
  if (!(K == 64 || K == 128 || K == 256 || K == 512) || !(N == 64 || N == 128 || N == 256)) return false;
  if (K == 64 && N == 256) return false;
  if (K == 512 && N != 256) return false;
  if (K == 256 && N == 64) return false;
  return a.in_ld % 4 == 0 && a.out_ld % 4 == 0;
}

int launch_pw_xs(const ConvArgs& a, int pt, hipStream_t s, int wg_per_cu) {
  if (!pw_xs_supported(a)) return -1;
  const long long M = (long long)a.N * a.H * a.W;
  const int K = a.cin, ctw = a.cout / 64;
  // measured best per shape (tools/pwxsbench.hip, profiles/r03_pwxsbench.txt): 32-pixel tiles and two persistent
  // workgroups per CU almost everywhere; 64-pixel tiles, one workgroup per CU for K = 512 (LDS) and for 64 -> 128
  if (pt == 0) {
    const bool big = K == 512 || (K == 64 && ctw == 2);
    pt = big ? 4 : 2;
    if (wg_per_cu == 0) wg_per_cu = big ? 1 : 2;
  }
  if (a.up_out) {                       // fused x2 upsample of the output: the one shape that needs it on this route
    if (!(K == 256 && ctw == 2 && pt == 2)) return -1;
    launch_pw_xs_t<256, 2, 2, 4, true>(a, M, wg_per_cu, s);
    return 0;
  }
#define JN_XS(K_, C_, D_)                                                          \
  if (K == K_ && ctw == C_) {                                                      \
    if (pt == 4) launch_pw_xs_t<K_, C_, 4, D_>(a, M, wg_per_cu, s); else launch_pw_xs_t<K_, C_, 2, D_>(a, M, wg_per_cu, s); \
    return 0;                                                                      \
  }
  // D: k-steps of weight fragments in flight per wave — a step is 4 * PT * CTW MFMAs (32 cycles each), an L2 round trip
  // about 1500 cycles: the narrower the wave's channel slice, the deeper the ring; D = K / 16 (K <= 128 with at most two
  // channel tiles per wave): every fragment is fetched up front, no loads inside the matrix loop
  JN_XS(64, 1, 4) JN_XS(64, 2, 4)
  JN_XS(128, 1, 8) JN_XS(128, 2, 8) JN_XS(128, 4, 2)
  JN_XS(256, 2, 4) JN_XS(256, 4, 2)
  JN_XS(512, 4, 2)
#undef JN_XS
  return -1;
}


// ---- the same kernel on the bf16 matrix pipe at fp32 accuracy: "x3" -------------------------------------------------
// gfx950 has no reduced-precision fast path for fp32 operands: v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate, and
// on these shapes its 32-cycle instructions are HALF of the kernel's time (tools/pwxsbench.hip prints the floor).  An
// fp32 value is exactly the sum of three bf16 values (8 + 8 + 8 significand bits: h = bf16(v), m = bf16(v - h),
// l = bf16(v - h - m)), so  x * w = (xh + xm + xl)(wh + wm + wl); the three products with |.| <= 2^-25 |x w| (m*l, l*m,
// l*l) are dropped and the other six run on v_mfma_f32_16x16x32_bf16 into the same fp32 accumulator: 6 x 16 cycles per
// 32 k instead of 8 x 32.  The error against an fp64 sum is that of the fp32 fmaf chain (measured, pwxsbench).
//   * the operand tile is split where it is transformed (once per element) and staged as three bf16 planes;
//   * the weights are split once per pass for the whole parameter arena (w_split3_kernel): groups of 8 consecutive
//     elements as [h x 8 | m x 8 | l x 8], i.e. the 48 bytes lane (row, g) needs for one 32-wide k step.
__global__ __launch_bounds__(256) void w_split3_kernel(const float* __restrict__ w, bf16_t* __restrict__ w3, long long n8) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n8) return;
  const f32x4 a = *reinterpret_cast<const f32x4*>(w + 8 * i), b = *reinterpret_cast<const f32x4*>(w + 8 * i + 4);
  bf16x8 h, m, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = e < 4 ? a[e] : b[e - 4];
    const bf16_t vh = (bf16_t)v;
    const float r1 = v - (float)vh;
    const bf16_t vm = (bf16_t)r1;
    const float r2 = r1 - (float)vm;
    h[e] = vh; m[e] = vm; l[e] = (bf16_t)r2;
  }
  bf16x8* o = reinterpret_cast<bf16x8*>(w3 + 24 * i);
  o[0] = h; o[1] = m; o[2] = l;
}

void launch_w_split3(const float* w, void* w3, long long n_floats, hipStream_t s) {
  const long long n8 = n_floats / 8;
  if (n8 <= 0) return;
  hipLaunchKernelGGL(w_split3_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, s, w, (bf16_t*)w3, n8);
}

// NP = 3, XT = float: the fp32 route.  NP = 1, XT = bf16: the bf16 inference mode on the same kernel — bf16 activations in
// and out, ONE plane (the operand rounded to bf16, as that mode defines its products), the weights' h plane.
// SLOTS (round 4, the data gradient of the wide layers in the step-batched backward): the operand is n_slots maps of M pixels
// each, `sl.in` / `sl.out` elements apart (the workspace slots of the glimpse steps); a tile never straddles two slots.
struct SlotSpan { int n_slots; long long in, out; };
// With SLOTS the split weights come in FRAGMENT ORDER: block ((c16 * K / 32 + j) * 3 + plane) of 64 lanes x 8 bf16, lane
// (g, lm) holding w[16 c16 + lm][32 j + 8 g .. + 7] — a wave's load of one fragment is 1 KB contiguous (8 cache lines).  The
// [row][k / 8][h | m | l] order of the forward planes makes the same load 64 pieces of 16 bytes in 32 lines, three times over
// for the three planes: with K = 256 the weights are streamed per tile and the L1 tag rate, not the matrix pipe, set the pace
// (383 us per 250 880-pixel launch against 79 us of matrix time).
template <int K, int CTW, int PT, int D, int NP, typename XT, bool UPS = false, bool SLOTS = false>
__global__ __launch_bounds__(256, (K <= 64 ? 3 : CTW <= 2 ? 2 : 1)) void pw_x3_kernel(
    const XT* __restrict__ x, int x_ld, ChanTab it, const bf16_t* __restrict__ w3, XT* __restrict__ out, int out_ld,
    long long M, double* __restrict__ stats, long long rep_stride, int nrep, const int* __restrict__ skip_flag,
    int skip_when, UpsDst ups, SlotSpan sl) {
  if (skip_flag && *skip_flag >= skip_when) return;
  // LDK: row stride 2 K + 32 bytes = 32 x odd: the 16-lane groups of a ds_read_b128 ({0-3, 12-15, 20-27}, ...: rows lm at
  // 16 g bytes) then cover the 64 banks exactly once (with K + 8 the planes read at 37 - 39 % conflict cycles, PMC)
  constexpr int BM = 16 * PT, LDK = K + 16, KQ = K / 4, NJ = K / 32, N = 64 * CTW;
  constexpr int NX = BM * KQ / 256;
  constexpr int NXB = NX < 16 ? NX : 16;
  static_assert(256 % KQ == 0 && NX >= 1 && NX % NXB == 0 && NJ % D == 0, "pw_x3 tile mapping");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Xp = reinterpret_cast<bf16_t*>(smem_raw);    // [NP][BM][LDK]: h, m, l planes of the transformed operand
  float* Tb = reinterpret_cast<float*>(Xp + NP * BM * LDK);      // [3][K], later [N][2] statistics
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, g = lane >> 4;
  const long long tiles_per_slot = (M + BM - 1) / BM;
  const long long n_tiles = SLOTS ? tiles_per_slot * sl.n_slots : tiles_per_slot;

  // weight fragments of k step j, channel tile c: 3 x bf16x8 at wrow[c] + 96 j (elements)
  const bf16_t* wrow[CTW];
#pragma unroll
  for (int c = 0; c < CTW; ++c)
    wrow[c] = SLOTS ? w3 + (long long)(wave * CTW + c) * (K / 32) * 3 * 512 + lane * 8
                    : w3 + ((long long)((wave * CTW + c) * 16 + lm) * (K / 8) + g) * 24;
  // fragment (k step j, plane t) of channel tile c
  auto wfrag = [&](int c, int j, int t) -> bf16x8 {
    if constexpr (SLOTS) return *reinterpret_cast<const bf16x8*>(wrow[c] + (j * 3 + t) * 512);
    else return *reinterpret_cast<const bf16x8*>(wrow[c] + 96 * j + 8 * t);
  };
  bf16x8 wr[D][CTW][NP];
#pragma unroll
  for (int u = 0; u < D; ++u)
#pragma unroll
    for (int c = 0; c < CTW; ++c)
#pragma unroll
      for (int t = 0; t < NP; ++t) wr[u][c][t] = wfrag(c, u, t);

  const int q = tid % KQ, r0 = tid / KQ;
  constexpr int RS = 256 / KQ;
  f32x4 xr[NXB];
  const XT* xs = x;                                    // SLOTS: the slot of the tile being fetched
  auto fetch = [&](long long m0, int b) {
#pragma unroll
    for (int u = 0; u < NXB; ++u) {
      long long m = m0 + r0 + (long long)(b * NXB + u) * RS;
      m = m < M ? m : M - 1;
      xr[u] = ld4(xs + m * x_ld + 4 * q);
    }
  };
  // tile index -> first pixel inside its slot (and the slot's operand base)
  auto locate = [&](long long t) -> long long {
    if constexpr (SLOTS) {
      const long long slot = t / tiles_per_slot;
      xs = x + slot * sl.in;
      return (t - slot * tiles_per_slot) * BM;
    } else {
      return t * BM;
    }
  };
  long long tile = blockIdx.x;
  long long m0 = locate(tile);
  fetch(m0, 0);
  tab_to_lds(Tb, K, K, it, tid, 256);
  __syncthreads();
  const f32x4 sc = *reinterpret_cast<const f32x4*>(Tb + 4 * q), sh = *reinterpret_cast<const f32x4*>(Tb + K + 4 * q),
              fl = *reinterpret_cast<const f32x4*>(Tb + 2 * K + 4 * q);
  f32x4 s1[CTW], s2[CTW];
#pragma unroll
  for (int c = 0; c < CTW; ++c) { s1[c] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[c] = s1[c]; }

#pragma unroll 1
  for (; tile < n_tiles; tile += gridDim.x) {
    XT* outs = out;
    if constexpr (SLOTS) outs = out + (tile / tiles_per_slot) * sl.out;
#pragma unroll
    for (int b = 0; b < NX / NXB; ++b) {
      if (b > 0) fetch(m0, b);
#pragma unroll
      for (int u = 0; u < NXB; ++u) {
        const int r = r0 + (b * NXB + u) * RS;
        f32x4 v = tf4_tab(xr[u], sc, sh, fl);
        if (m0 + r >= M) v = f32x4{0.f, 0.f, 0.f, 0.f};
        const bf16x4 vh = __builtin_convertvector(v, bf16x4);
        bf16_t* dst = Xp + r * LDK + 4 * q;
        *reinterpret_cast<bf16x4*>(dst) = vh;
        if constexpr (NP == 3) {
          const f32x4 r1 = v - __builtin_convertvector(vh, f32x4);
          const bf16x4 vm = __builtin_convertvector(r1, bf16x4);
          const f32x4 r2 = r1 - __builtin_convertvector(vm, f32x4);
          const bf16x4 vl = __builtin_convertvector(r2, bf16x4);
          *reinterpret_cast<bf16x4*>(dst + BM * LDK) = vm;
          *reinterpret_cast<bf16x4*>(dst + 2 * BM * LDK) = vl;
        }
      }
    }
    __syncthreads();
    const long long m0_this = m0;
    if (tile + gridDim.x < n_tiles) { m0 = locate(tile + gridDim.x); fetch(m0, 0); }

    f32x4 acc[CTW][PT];
#pragma unroll
    for (int c = 0; c < CTW; ++c)
#pragma unroll
      for (int p = 0; p < PT; ++p) acc[c][p] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16_t* xrow = Xp + lm * LDK + 8 * g;
#pragma unroll 1
    for (int j0 = 0; j0 < NJ; j0 += D) {
#pragma unroll
      for (int u = 0; u < D; ++u) {
        const int j = j0 + u;
        bf16x8 xa[PT][NP], wa[CTW][NP];
#pragma unroll
        for (int p = 0; p < PT; ++p)
#pragma unroll
          for (int t = 0; t < NP; ++t) xa[p][t] = *reinterpret_cast<const bf16x8*>(xrow + t * BM * LDK + p * 16 * LDK + 32 * j);
#pragma unroll
        for (int c = 0; c < CTW; ++c)
#pragma unroll
          for (int t = 0; t < NP; ++t) wa[c][t] = wr[u][c][t];
        if constexpr (D < NJ) {
          const int jn = j + D < NJ ? j + D : j + D - NJ;
#pragma unroll
          for (int c = 0; c < CTW; ++c)
#pragma unroll
            for (int t = 0; t < NP; ++t) wr[u][c][t] = wfrag(c, jn, t);
          __builtin_amdgcn_sched_barrier(0);
        }
        // six products, the small ones first: (w, x) = (l, h) (h, l) (m, m) (m, h) (h, m) (h, h)
        constexpr int TW[6] = {2, 0, 1, 1, 0, 0}, TX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
        for (int e = (NP == 3 ? 0 : 5); e < 6; ++e)
#pragma unroll
          for (int c = 0; c < CTW; ++c)
#pragma unroll
            for (int p = 0; p < PT; ++p)
              acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[c][TW[e]], xa[p][TX[e]], acc[c][p], 0, 0, 0);
      }
    }

#pragma unroll
    for (int c = 0; c < CTW; ++c) {
      const int n = (wave * CTW + c) * 16 + 4 * g;
#pragma unroll
      for (int p = 0; p < PT; ++p) {
        const long long m = m0_this + 16 * p + lm;
        const f32x4 v = acc[c][p];
        if (m < M) { st4(outs + m * out_ld + n, v); ups_store<UPS>(ups, m, n, v); }
        s1[c] += v; s2[c] += v * v;
      }
    }
    __syncthreads();
  }

  if (stats) {
    float* red = Tb;
    if constexpr (3 * K < 2 * N) red = reinterpret_cast<float*>(Xp);
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
      const int n = (wave * CTW + c) * 16 + 4 * g;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = row16_sum(s1[c][r]);
        const float b = row16_sum(s2[c][r]);
        if (lm == 0) { red[2 * (n + r)] = a; red[2 * (n + r) + 1] = b; }
      }
    }
    __syncthreads();
    double* st = stats + (blockIdx.x % nrep) * rep_stride;
    for (int i = tid; i < 2 * N; i += 256) atomicAdd(st + i, (double)red[i]);
  }
}

template <int K, int CTW, int PT, int D, int NP, typename XT, bool UPS = false, bool SLOTS = false>
static void launch_pw_x3_t(const ConvArgs& a, long long M, int wg_per_cu, hipStream_t s) {
  constexpr int BM = 16 * PT;
  const size_t smem = (size_t)NP * BM * (K + 16) * sizeof(bf16_t) + 3 * K * sizeof(float);
  auto kern = pw_x3_kernel<K, CTW, PT, D, NP, XT, UPS, SLOTS>;
  static int places = 0;
  if (!places) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), 256, smem) != hipSuccess || per_cu < 1) per_cu = 1;
    places = per_cu;
  }
  const int n_slots = SLOTS ? std::max(1, a.n_slots) : 1;
  const long long n_tiles = (M + BM - 1) / BM * n_slots;
  const int per_cu = std::max(1, std::min(places, wg_per_cu > 0 ? wg_per_cu : 2));
  const long long gx = std::min<long long>(n_tiles, 256LL * per_cu);
  hipLaunchKernelGGL(kern, dim3((unsigned)gx), dim3(256), smem, s, (const XT*)a.in, a.in_ld, a.itab, (const bf16_t*)a.w_x3,
                     (XT*)a.out, a.out_ld, M, a.stats, a.stats_rep_stride, a.stats_nrep > 0 ? a.stats_nrep : JN_NREP,
                     a.skip_flag, a.skip_when, UpsDst{(float*)a.up_out, a.up_ld, a.W, a.H * a.W},
                     SlotSpan{n_slots, a.in_slot_stride, a.out_slot_stride});
}

// Shapes the x3 kernel is built for — the ones where it beats pw_xs_kernel (tools/pwxsbench.hip, profiles/r03_x3bench.txt:
// 8 - 21 %); with K >= 256 the weight fragments are streamed per tile and their 1.5x bytes cost more than the matrix
// cycles saved, except for 256 -> 256 on 64-pixel tiles.
static bool pw_xs_shape(int K, int N) {
  if (!(K == 64 || K == 128 || K == 256 || K == 512) || !(N == 64 || N == 128 || N == 256)) return false;
  return !(K == 64 && N == 256) && !(K == 512 && N != 256) && !(K == 256 && N == 64);
}

bool pw_x3_preferred(const ConvArgs& a) {
  if (!a.w_x3 || !pw_xs_supported(a)) return false;
  const int K = a.cin, N = a.cout;
  return (K == 64 && (N == 64 || N == 128)) || (K == 128 && (N == 64 || N == 128)) || (K == 256 && N == 256);
}

// bf16 inference mode (bf16 activations in and out, bf16 MFMA): the single-plane form of the kernel, every xs shape
bool pw_x1_supported(const ConvArgs& a) {
  return a.w_x3 && a.bf16_mfma && a.in_dtype == JN_BF16 && a.out_dtype == JN_BF16 && !a.bias && a.act == ACT_NONE && !a.w_transposed &&
         !a.accumulate && a.n_slots <= 1 && pw_xs_shape(a.cin, a.cout) && a.in_ld % 4 == 0 && a.out_ld % 4 == 0;
}

int launch_pw_x1(const ConvArgs& a, hipStream_t s) {
  if (!pw_x1_supported(a)) return -1;
  const long long M = (long long)a.N * a.H * a.W;
  const int K = a.cin, ctw = a.cout / 64;
  const int wg = (K == 512 || ctw == 4) ? 1 : 2;
#define JN_X1(K_, C_, P_, D_) if (K == K_ && ctw == C_) { launch_pw_x3_t<K_, C_, P_, D_, 1, bf16_t>(a, M, wg, s); return 0; }
  JN_X1(64, 1, 2, 2) JN_X1(64, 2, 2, 2) JN_X1(128, 1, 2, 4) JN_X1(128, 2, 2, 4) JN_X1(128, 4, 2, 4)
  JN_X1(256, 2, 4, 2) JN_X1(256, 4, 4, 2) JN_X1(512, 4, 4, 2)
#undef JN_X1
  return -1;
}

int launch_pw_x3(const ConvArgs& a, int pt, hipStream_t s, int wg_per_cu) {
  if (!pw_xs_supported(a) || !a.w_x3) return -1;
  const long long M = (long long)a.N * a.H * a.W;
  const int K = a.cin, ctw = a.cout / 64;
  if (pt == 0) {
    // measured best per shape: 32-pixel tiles, two persistent workgroups per CU (one when there is at most a tile or two
    // per workgroup anyway, or for 64 -> 128); 256 -> 256: 64-pixel tiles halve the weight stream
    pt = K == 256 ? 4 : 2;
    if (wg_per_cu == 0) wg_per_cu = (K == 256 || (K == 64 && ctw == 2) || (K == 128 && ctw == 2 && M <= 16384)) ? 1 : 2;
  }
  if (a.up_out) {                       // fused x2 upsample of the output: the one shape that needs it on this route
    if (!(K == 128 && ctw == 1 && pt == 2)) return -1;
    launch_pw_x3_t<128, 1, 2, 4, 3, float, true>(a, M, wg_per_cu, s);
    return 0;
  }
#define JN_X3(K_, C_, P_, D_) if (K == K_ && ctw == C_ && pt == P_) { launch_pw_x3_t<K_, C_, P_, D_, 3, float>(a, M, wg_per_cu, s); return 0; }
  JN_X3(64, 1, 2, 2) JN_X3(64, 2, 2, 2) JN_X3(128, 1, 2, 4) JN_X3(128, 2, 2, 4) JN_X3(256, 4, 4, 2)
#ifdef JN_X3_ALL_SHAPES        // tools/pwxsbench.hip: every shape and tile size, to show where the kernel loses
  JN_X3(64, 1, 4, 2) JN_X3(64, 2, 4, 2) JN_X3(128, 1, 4, 4) JN_X3(128, 2, 4, 4) JN_X3(128, 4, 2, 2) JN_X3(128, 4, 4, 2)
  JN_X3(256, 2, 2, 2) JN_X3(256, 2, 4, 2) JN_X3(256, 4, 2, 2) JN_X3(512, 4, 2, 2)
#endif
#undef JN_X3
  return -1;
}

// ---- the data gradient of the WIDE 1x1 layers on the same kernel (round 4) ---------------------------------------------
// g_x[m][k] = sum_n g_z[m][n] w[n][k] is the 1x1 conv of g_z with the transposed weight.  In the step-batched backward its
// M is 20 x the forward's and pw_dir_kernel<.., WT> runs it at 0.74 matrix-pipe busy on the fp32 pipe
// (profiles/r04_z_pmc_lds_mfma_train_iteration.txt): matrix-bound, the one regime where the three-way bf16 split pays in
// full (3 cycles per k instead of 8).  The transposed weight is split per layer at the start of its backward
// (w_split3_t_kernel, in the fragment order pw_x3_kernel<.., SLOTS> streams); 512 input channels: two
// launches of 256 output columns each.
__global__ __launch_bounds__(256) void w_split3_t_kernel(const float* __restrict__ w, bf16_t* __restrict__ w3t, int cout, int cin) {
  const int i = blockIdx.x * 256 + threadIdx.x;        // (row n' = input channel, group of 8 output channels)
  const int groups = cout / 8;
  if (i >= cin * groups) return;
  const int n = i % cin, gq = i / cin;                 // consecutive threads: consecutive columns of w (coalesced reads)
  const int NJ = cout / 32, j = gq >> 2, g = gq & 3;   // fragment order (pw_x3_kernel, SLOTS): k step j, lane group g
  bf16x8 h, m, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float v = w[(long long)(8 * gq + e) * cin + n];
    const bf16_t vh = (bf16_t)v;
    const float r1 = v - (float)vh;
    const bf16_t vm = (bf16_t)r1;
    const float r2 = r1 - (float)vm;
    h[e] = vh; m[e] = vm; l[e] = (bf16_t)r2;
  }
  bf16_t* o = w3t + ((long long)((n >> 4) * NJ + j) * 3 * 64 + g * 16 + (n & 15)) * 8;
  *reinterpret_cast<bf16x8*>(o) = h;
  *reinterpret_cast<bf16x8*>(o + 512) = m;
  *reinterpret_cast<bf16x8*>(o + 1024) = l;
}

// shapes (cout = the gradient's K, cin = its N) this route takes
bool pw_x3_bwd_data_supported(int cout, int cin) {
  return (cout == 256 && (cin == 128 || cin == 256 || cin == 512)) || (cout == 128 && cin == 256);
}

// a: the ConvArgs of the gradient conv as api.hip builds them for launch_pw (in = g_z, out = g_x, cin = layer's cout,
// cout = layer's cin, slots); w = the layer's [cout][cin] weight, w3t = 3 * cout * cin bf16 of scratch owned by the layer
int launch_pw_x3_bwd_data(const ConvArgs& a, const float* w, void* w3t, hipStream_t s) {
  const int K = a.cin, N = a.cout;                      // of the gradient conv
  if (!pw_x3_bwd_data_supported(K, N) || a.accumulate || a.in_dtype != JN_F32 || a.out_dtype != JN_F32 || a.bias ||
      a.act != ACT_NONE || a.in_ld % 4 != 0 || a.out_ld % 4 != 0 || a.stats)
    return -1;
  hipLaunchKernelGGL(w_split3_t_kernel, dim3((unsigned)((N * (K / 8) + 255) / 256)), dim3(256), 0, s, w, (bf16_t*)w3t, K, N);
  const long long M = (long long)a.N * a.H * a.W;
  ConvArgs b = a;
  b.w_x3 = w3t; b.up_out = nullptr;
  // tilings: the best of tools/x3bwdbench.hip (profiles/r04_x3bwdbench.txt) — 32-pixel tiles everywhere (more tiles to
  // balance over the persistent grid), K = 128: all weight fragments in registers (D = K / 32)
  if (K == 256 && N == 128) { launch_pw_x3_t<256, 2, 2, 2, 3, float, false, true>(b, M, 2, s); return 0; }
  if (K == 128 && N == 256) { launch_pw_x3_t<128, 4, 2, 4, 3, float, false, true>(b, M, 2, s); return 0; }
  for (int n0 = 0; n0 < N; n0 += 256) {                 // 256 -> 256, and 256 -> 512 as two column halves
    b.w_x3 = (const bf16_t*)w3t + (long long)n0 * K * 3;
    b.out = (float*)a.out + n0;
    launch_pw_x3_t<256, 4, 2, 2, 3, float, false, true>(b, M, 2, s);
  }
  return 0;
}

}  // namespace jnr
