// Detector kernels (yolox-s style, non-depthwise): dense 3x3 conv on fp32 MFMA, the YOLOX head's
// prediction convs fused with the box decode, and class-agnostic postprocess (threshold, sort, NMS).
// Restated from the published YOLOX head / postprocess (reference call sites
// src/models/yolox.py:55, 77-86, 93-113).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "jn_kernels.h"
#include "jn_reduce.h"
#include "jn_types.h"

namespace jnr {

__device__ __forceinline__ float silu_d(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ f32x4 tf4_d(f32x4 z, f32x4 sc, f32x4 sh, f32x4 fl) {
  f32x4 r;
  r.x = fl.x != 0.0f ? silu_d(fmaf(z.x, sc.x, sh.x)) : z.x;
  r.y = fl.y != 0.0f ? silu_d(fmaf(z.y, sc.y, sh.y)) : z.y;
  r.z = fl.z != 0.0f ? silu_d(fmaf(z.z, sc.z, sh.z)) : z.z;
  r.w = fl.w != 0.0f ? silu_d(fmaf(z.w, sc.w, sh.w)) : z.w;
  return r;
}

// ------------------------------------------------------------------------------------
// dense 3x3 conv (pad 1, stride S): z[m][n] = sum_tap sum_k T(x[m (+) tap][k]) * w[tap][n][k]
// im2col-free: nine shifted 1x1 GEMMs over ONE LDS halo tile per K chunk.
// Workgroup = 8 x 16 output pixels x 64 channels; wave = 2 rows x 16 pixels; K chunk = 16.
// ------------------------------------------------------------------------------------
constexpr int C3_TH = 8, C3_TW = 16, C3_KC = 16, C3_LD = C3_KC + 4, C3_BN = 64;

// WT (stride 1 only): the data gradient of a stride-1 layer is the same conv over g_z with the taps mirrored and the
// weight read transposed: w'[tap][n][k] = w[8 - tap][k][n] (w is the forward [tap][Nc_fwd = K here][K_fwd = Nc here]).
// accumulate: out += (gradient views with several writers); stats: BN sum / sumsq accumulators (train mode).
template <int S, typename AT, bool WT, int PR>      // PR = output rows per wave: tile = 4 PR x 16 pixels
__global__ __launch_bounds__(256) void conv3_mfma_kernel(const AT* __restrict__ x, int x_ld, ChanTab it,
                                                         const float* __restrict__ w, AT* __restrict__ out,
                                                         int out_ld, int H, int W, int OH, int OW, int K, int Nc,
                                                         int tiles_x, int tiles_y, int accumulate,
                                                         double* __restrict__ stats, long long rep_stride,
                                                         const int* __restrict__ skip_flag, int skip_when,
                                                         long long x_slot, long long out_slot) {
  if (skip_flag && *skip_flag >= skip_when) return;
  x += blockIdx.z * x_slot; out += blockIdx.z * out_slot;          // step-batched gradient launches
  constexpr int TH = 4 * PR;
  constexpr int IH = TH * S + 2, IW = C3_TW * S + 2;
  // row strides (floats): 96 B = 32 x odd makes the ds_read_b128 of 16 consecutive rows conflict-free (80 B: half of the LDS
  // cycles were bank conflicts, PMC); at stride 2 the pixel rows are 2 x 80 B = 32 x 5 apart already, and 96-byte weight rows
  // would cost the second workgroup per CU
  constexpr int LDX = S == 1 ? 24 : C3_LD, LDW = S == 1 ? 24 : C3_LD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                              // [IH*IW][LDX]
  float* Ws = smem + IH * IW * LDX;              // [9][C3_BN][LDW]
  float* red = Ws + 9 * C3_BN * LDW;             // [4 waves][C3_BN][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x % (tiles_x * tiles_y);
  const int n_img = blockIdx.x / (tiles_x * tiles_y);
  const int oy0 = (tile / tiles_x) * TH, ox0 = (tile % tiles_x) * C3_TW;
  const int n0 = blockIdx.y * C3_BN;
  const AT* xb = x + (long long)n_img * H * W * x_ld;
  f32x4 acc[PR][4];
#pragma unroll
  for (int p = 0; p < PR; ++p)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  // K chunks are software-pipelined: chunk i + 1 is fetched into registers while the 288 MFMAs of chunk i run
  constexpr int NXR = (IH * IW * 4 + 255) / 256, NWR = 9 * C3_BN * 4 / 256;
  f32x4 xr[NXR], wr[NWR];
  // the thread's channel quad (tid & 3) is the same for all its staged pixels: its (scale, shift, flag) entries of the
  // next chunk travel with the prefetch instead of being read from global memory inside the staging
  f32x4 t_sc = {1.f, 1.f, 1.f, 1.f}, t_sh = {0.f, 0.f, 0.f, 0.f}, t_fl = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int k0) {
    if (k0 + 4 * (tid & 3) < K) {
      t_sc = *reinterpret_cast<const f32x4*>(it.sc + k0 + 4 * (tid & 3));
      t_sh = *reinterpret_cast<const f32x4*>(it.sh + k0 + 4 * (tid & 3));
      t_fl = *reinterpret_cast<const f32x4*>(it.fl + k0 + 4 * (tid & 3));
    }
#pragma unroll
    for (int j = 0; j < NXR; ++j) {
      const int i = tid + 256 * j, pix = i >> 2, q = i & 3;
      const int iy = oy0 * S - 1 + pix / IW, ix = ox0 * S - 1 + pix % IW;
      xr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < IH * IW * 4 && iy >= 0 && iy < H && ix >= 0 && ix < W && k0 + 4 * q < K)
        xr[j] = ld4(xb + ((long long)iy * W + ix) * x_ld + k0 + 4 * q);
    }
#pragma unroll
    for (int j = 0; j < NWR; ++j) {
      const int i = tid + 256 * j, q = i & 3, r = (i >> 2) % C3_BN, tp = i / (4 * C3_BN);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n0 + r < Nc && k0 + 4 * q < K) {
        if (WT) {
          const float* wp = w + ((long long)(8 - tp) * K + k0 + 4 * q) * Nc + n0 + r;
          v = f32x4{wp[0], wp[Nc], wp[2 * Nc], wp[3 * Nc]};
        } else {
          v = *reinterpret_cast<const f32x4*>(w + ((long long)tp * Nc + n0 + r) * K + k0 + 4 * q);
        }
      }
      wr[j] = v;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += C3_KC) {
    if (k0) __syncthreads();
#pragma unroll
    for (int j = 0; j < NXR; ++j) {
      const int i = tid + 256 * j, pix = i >> 2, q = i & 3;
      if (i < IH * IW * 4) {
        const int iy = oy0 * S - 1 + pix / IW, ix = ox0 * S - 1 + pix % IW;
        const int kk = k0 + 4 * q;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W && kk < K)
          v = tf4_d(xr[j], t_sc, t_sh, t_fl);
        *reinterpret_cast<f32x4*>(Xs + pix * LDX + 4 * q) = v;
      }
    }
#pragma unroll
    for (int j = 0; j < NWR; ++j) {
      const int i = tid + 256 * j, q = i & 3, r = (i >> 2) % C3_BN, tp = i / (4 * C3_BN);
      *reinterpret_cast<f32x4*>(Ws + (tp * C3_BN + r) * LDW + 4 * q) = wr[j];
    }
    __syncthreads();
    if (k0 + C3_KC < K) fetch(k0 + C3_KC);
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      const int ky = tp / 3, kx = tp % 3;
      f32x4 xbv[PR];
#pragma unroll
      for (int p = 0; p < PR; ++p)
        xbv[p] = *reinterpret_cast<const f32x4*>(Xs + (((PR * wave + p) * S + ky) * IW + lm * S + kx) * LDX + 4 * g);
      f32x4 wa[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) wa[c] = *reinterpret_cast<const f32x4*>(Ws + (tp * C3_BN + 16 * c + lm) * LDW + 4 * g);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int p = 0; p < PR; ++p) acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][j], xbv[p][j], acc[p][c], 0, 0, 0);
    }
  }
  f32x4 s1[4], s2[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { s1[c] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[c] = s1[c]; }
#pragma unroll
  for (int p = 0; p < PR; ++p) {
    const int oy = oy0 + PR * wave + p, ox = ox0 + lm;
    if (oy >= OH || ox >= OW) continue;
    AT* op = out + (((long long)n_img * OH + oy) * OW + ox) * out_ld;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int n = n0 + 16 * c + 4 * g;
      if (n < Nc) {
        f32x4 v = acc[p][c];
        if (accumulate) v += ld4(op + n);
        st4(op + n, v);
        s1[c] += v;
        s2[c] += v * v;
      }
    }
  }
  if (stats) {
    __syncthreads();                               // the MFMA loop's LDS readers are done; red aliases nothing but be safe
    wave_stats_to_lds<4>(s1, s2, red + wave * 2 * C3_BN, lane, Nc - n0);
    __syncthreads();
    if (tid < 2 * C3_BN && n0 + (tid >> 1) < Nc)
      atomicAdd(&stats[(blockIdx.x % JN_NREP) * rep_stride + 2 * n0 + tid],
                (double)(red[tid] + red[2 * C3_BN + tid] + red[4 * C3_BN + tid] + red[6 * C3_BN + tid]));
  }
}

// bf16 inference mode: the same nine-shifted-GEMMs scheme on v_mfma_f32_16x16x32_bf16 (16x the fp32 MFMA rate; the
// detector is MFMA-bound: 13 GFLOP per 448-px patch).  bf16 activations in and out, fp32 master weights converted
// while staging, fp32 accumulation.  K chunk = 32 = one MFMA k-step per tap; rows of 8 consecutive k per lane.
constexpr int C3B_KC = 32, C3B_LD = C3B_KC + 8;
constexpr int C3B_TH = 16;          // 16 x 16 pixels per workgroup (wave = 4 rows): the 9 x 64 x 32 weight tile is staged once per 256 pixels

template <int S>
__global__ __launch_bounds__(256) void conv3_bf16_kernel(const bf16_t* __restrict__ x, int x_ld, ChanTab it,
                                                         const bf16_t* __restrict__ w, bf16_t* __restrict__ out,
                                                         int out_ld, int H, int W, int OH, int OW, int K, int Nc,
                                                         int tiles_x, int tiles_y, const int* __restrict__ skip_flag,
                                                         int skip_when) {
  if (skip_flag && *skip_flag >= skip_when) return;
  constexpr int IH = C3B_TH * S + 2, IW = C3_TW * S + 2;
  constexpr int NXR = (IH * IW * (C3B_KC / 4) + 255) / 256;       // 4-channel groups of the halo tile per thread
  constexpr int NWR = 9 * C3_BN * (C3B_KC / 8) / 256;              // 8-channel (16-B) groups of the weight chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
  bf16_t* Xs = reinterpret_cast<bf16_t*>(smem_b);      // [IH*IW][C3B_LD]
  bf16_t* Ws = Xs + IH * IW * C3B_LD;                  // [9][C3_BN][C3B_LD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x % (tiles_x * tiles_y);
  const int n_img = blockIdx.x / (tiles_x * tiles_y);
  const int oy0 = (tile / tiles_x) * C3B_TH, ox0 = (tile % tiles_x) * C3_TW;
  const int n0 = blockIdx.y * C3_BN;
  const bf16_t* xb = x + (long long)n_img * H * W * x_ld;
  f32x4 acc[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  // K chunks are software-pipelined through registers (chunk i + 1 is in flight while the MFMAs of chunk i run)
  bf16x4 xr[NXR];
  bf16x8 wr[NWR];
  static_assert(256 % (C3B_KC / 4) == 0, "the thread's channel quad is fixed");
  f32x4 t_sc = {1.f, 1.f, 1.f, 1.f}, t_sh = {0.f, 0.f, 0.f, 0.f}, t_fl = {0.f, 0.f, 0.f, 0.f};   // table quad of the prefetched chunk
  auto fetch = [&](int k0) {
    {
      const int kq = k0 + 4 * (tid % (C3B_KC / 4));
      if (kq < K) {
        t_sc = *reinterpret_cast<const f32x4*>(it.sc + kq); t_sh = *reinterpret_cast<const f32x4*>(it.sh + kq);
        t_fl = *reinterpret_cast<const f32x4*>(it.fl + kq);
      }
    }
#pragma unroll
    for (int j = 0; j < NXR; ++j) {
      const int i = tid + 256 * j, pix = i / (C3B_KC / 4), q = i % (C3B_KC / 4);
      const int iy = oy0 * S - 1 + pix / IW, ix = ox0 * S - 1 + pix % IW;
      bf16x4 v = {};
      if (i < IH * IW * (C3B_KC / 4) && iy >= 0 && iy < H && ix >= 0 && ix < W && k0 + 4 * q < K)
        v = *reinterpret_cast<const bf16x4*>(xb + ((long long)iy * W + ix) * x_ld + k0 + 4 * q);
      xr[j] = v;
    }
#pragma unroll
    for (int j = 0; j < NWR; ++j) {
      const int i = tid + 256 * j, q = i % (C3B_KC / 8), r = (i / (C3B_KC / 8)) % C3_BN, tp = i / ((C3B_KC / 8) * C3_BN);
      bf16x8 v = {};
      if (n0 + r < Nc && k0 + 8 * q < K) v = *reinterpret_cast<const bf16x8*>(w + ((long long)tp * Nc + n0 + r) * K + k0 + 8 * q);
      wr[j] = v;
    }
  };
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += C3B_KC) {
    if (k0) __syncthreads();
#pragma unroll
    for (int j = 0; j < NXR; ++j) {
      const int i = tid + 256 * j, pix = i / (C3B_KC / 4), q = i % (C3B_KC / 4);
      if (i < IH * IW * (C3B_KC / 4)) {
        const int iy = oy0 * S - 1 + pix / IW, ix = ox0 * S - 1 + pix % IW;
        const int kk = k0 + 4 * q;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W && kk < K)
          v = tf4_d(__builtin_convertvector(xr[j], f32x4), t_sc, t_sh, t_fl);
        st4(Xs + pix * C3B_LD + 4 * q, v);
      }
    }
#pragma unroll
    for (int j = 0; j < NWR; ++j) {
      const int i = tid + 256 * j, q = i % (C3B_KC / 8), r = (i / (C3B_KC / 8)) % C3_BN, tp = i / ((C3B_KC / 8) * C3_BN);
      *reinterpret_cast<bf16x8*>(Ws + (tp * C3_BN + r) * C3B_LD + 8 * q) = wr[j];
    }
    __syncthreads();
    if (k0 + C3B_KC < K) fetch(k0 + C3B_KC);
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      const int ky = tp / 3, kx = tp % 3;
      bf16x8 xbv[4];
#pragma unroll
      for (int p = 0; p < 4; ++p)
        xbv[p] = *reinterpret_cast<const bf16x8*>(Xs + (((4 * wave + p) * S + ky) * IW + lm * S + kx) * C3B_LD + 8 * g);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const bf16x8 wa = *reinterpret_cast<const bf16x8*>(Ws + (tp * C3_BN + 16 * c + lm) * C3B_LD + 8 * g);
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[p][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xbv[p], acc[p][c], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int oy = oy0 + 4 * wave + p, ox = ox0 + lm;
    if (oy >= OH || ox >= OW) continue;
    bf16_t* op = out + (((long long)n_img * OH + oy) * OW + ox) * out_ld;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int n = n0 + 16 * c + 4 * g;
      if (n < Nc) st4(op + n, acc[p][c]);
    }
  }
}

template <int S>
static void launch_conv3_bf16(const ConvArgs& a, hipStream_t s) {
  const int tiles_x = (a.OW + C3_TW - 1) / C3_TW, tiles_y = (a.OH + C3B_TH - 1) / C3B_TH;
  dim3 grid(tiles_x * tiles_y * a.N, (a.cout + C3_BN - 1) / C3_BN);
  const size_t smem = ((size_t)(S * C3B_TH + 2) * (S * C3_TW + 2) + 9 * C3_BN) * C3B_LD * sizeof(bf16_t);
  if (smem > 64 * 1024) {
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_bf16_kernel<S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr_set = true;
    }
  }
  hipLaunchKernelGGL((conv3_bf16_kernel<S>), grid, dim3(256), smem, s, (const bf16_t*)a.in, a.in_ld, a.itab,
                     (const bf16_t*)a.w_bf16, (bf16_t*)a.out,
                     a.out_ld, a.H, a.W, a.OH, a.OW, a.cin, a.cout, tiles_x, tiles_y, a.skip_flag, a.skip_when);
}

// ---- dense 3x3 on the bf16 matrix pipe at fp32 accuracy ("x3", round 3) --------------------------------------------
// The fp32 kernel above runs AT the fp32 MFMA rate (configs[4]: 161 ms of a 367 ms iteration are its forward and
// data-gradient launches) — 1/16 of the bf16 rate.  As in pw_x3_kernel (kernels_pwxs.hip): every fp32 operand is the exact
// sum of three bf16 values, the six products above 2^-25 |x w| go to v_mfma_f32_16x16x32_bf16, small terms first.
//   * K chunk = 32 (one MFMA k-step per tap and product); the halo tile is transformed and split ONCE per chunk into three
//     bf16 planes [pixel][32 + 8];
//   * the weight chunk goes through LDS one kernel ROW at a time (3 taps x 64 channels x 32 k x three planes = 46 KB; all
//     nine taps would not fit beside the halo tile): three phases per chunk, the next phase's weights (and, during the
//     last phase, the next chunk's halo tile) are in flight in registers while the matrix loop of the current one runs;
//   * wave = PR rows x 16 pixels x 64 channels: per tap PR x 3 + 4 x 3 ds_read_b128 feed PR x 4 x 6 MFMAs.
// WT: data gradient of a stride-1 layer (mirrored taps, transposed weight), as in conv3_mfma_kernel.
// Eight waves (two per SIMD: one wave's operand reads run under the other's MFMAs — four waves on a CU of their own were
// no faster than the fp32 kernel), two rows per wave: 16 x 16 pixels x 64 channels per workgroup.
constexpr int C3X_KC = 32, C3X_LD = C3X_KC + 16;     // 96-byte rows = 32 x 3: conflict-free ds_read_b128 (80-byte rows: half of the LDS cycles were bank conflicts, PMC)

template <int S, bool WT, int PR, int NW>      // NW waves per workgroup, PR rows per wave: tile = NW PR x 16 pixels
__global__ __launch_bounds__(64 * NW) void conv3_x3_kernel(const float* __restrict__ x, int x_ld, ChanTab it,
                                                       const float* __restrict__ w, float* __restrict__ out, int out_ld,
                                                       int H, int W, int OH, int OW, int K, int Nc, int tiles_x, int tiles_y,
                                                       int accumulate, double* __restrict__ stats, long long rep_stride,
                                                       const int* __restrict__ skip_flag, int skip_when, long long x_slot,
                                                       long long out_slot) {
  if (skip_flag && *skip_flag >= skip_when) return;
  x += blockIdx.z * x_slot; out += blockIdx.z * out_slot;
  constexpr int NT = 64 * NW, TH = NW * PR, IH = TH * S + 2, IW = C3_TW * S + 2, NP = IH * IW;
  constexpr int XPL = NP * C3X_LD, WPL = 3 * C3_BN * C3X_LD;       // elements per plane
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_x3[];
  bf16_t* Xp = reinterpret_cast<bf16_t*>(smem_x3);                  // [3 planes][NP][C3X_LD]
  bf16_t* Wp = Xp + 3 * XPL;                                        // [3 planes][3 taps][C3_BN][C3X_LD]
  float* red = reinterpret_cast<float*>(Wp + 3 * WPL);              // [NW waves][C3_BN][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x % (tiles_x * tiles_y);
  const int n_img = blockIdx.x / (tiles_x * tiles_y);
  const int oy0 = (tile / tiles_x) * TH, ox0 = (tile % tiles_x) * C3_TW;
  const int n0 = blockIdx.y * C3_BN;
  const float* xb = x + (long long)n_img * H * W * x_ld;
  f32x4 acc[PR][4];
#pragma unroll
  for (int p = 0; p < PR; ++p)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NXR = (NP * 8 + NT - 1) / NT;            // f32x4 of the halo tile per thread (8 quads per pixel)
  constexpr int NWR = 3 * C3_BN * 8 / NT;             // f32x4 of one kernel row's weight chunk per thread (= 6)
  f32x4 xr[NXR], wr[NWR];
  f32x4 t_sc = {1.f, 1.f, 1.f, 1.f}, t_sh = {0.f, 0.f, 0.f, 0.f}, t_fl = {0.f, 0.f, 0.f, 0.f};
  auto fetch_x = [&](int k0) {
    const int kq = k0 + 4 * (tid & 7);                 // the thread's channel quad is the same for all its pixels
    t_sc = *reinterpret_cast<const f32x4*>(it.sc + kq); t_sh = *reinterpret_cast<const f32x4*>(it.sh + kq);
    t_fl = *reinterpret_cast<const f32x4*>(it.fl + kq);
#pragma unroll
    for (int j = 0; j < NXR; ++j) {
      const int i = tid + NT * j, pix = i >> 3;
      const int iy = oy0 * S - 1 + pix / IW, ix = ox0 * S - 1 + pix % IW;
      xr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < NP * 8 && iy >= 0 && iy < H && ix >= 0 && ix < W)
        xr[j] = *reinterpret_cast<const f32x4*>(xb + ((long long)iy * W + ix) * x_ld + kq);
    }
  };
  auto fetch_w = [&](int k0, int ky) {
#pragma unroll
    for (int j = 0; j < NWR; ++j) {
      const int i = tid + NT * j, q = i & 7, r = (i >> 3) % C3_BN, kx = i / (8 * C3_BN), tp = 3 * ky + kx;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n0 + r < Nc) {
        if (WT) {
          const float* wp = w + ((long long)(8 - tp) * K + k0 + 4 * q) * Nc + n0 + r;
          v = f32x4{wp[0], wp[Nc], wp[2 * Nc], wp[3 * Nc]};
        } else {
          v = *reinterpret_cast<const f32x4*>(w + ((long long)tp * Nc + n0 + r) * K + k0 + 4 * q);
        }
      }
      wr[j] = v;
    }
  };
  auto split3 = [](bf16_t* dst, int plane_stride, f32x4 v) {
    const bf16x4 h = __builtin_convertvector(v, bf16x4);
    const f32x4 r1 = v - __builtin_convertvector(h, f32x4);
    const bf16x4 m = __builtin_convertvector(r1, bf16x4);
    const bf16x4 l = __builtin_convertvector(r1 - __builtin_convertvector(m, f32x4), bf16x4);
    *reinterpret_cast<bf16x4*>(dst) = h;
    *reinterpret_cast<bf16x4*>(dst + plane_stride) = m;
    *reinterpret_cast<bf16x4*>(dst + 2 * plane_stride) = l;
  };
  fetch_x(0);
  fetch_w(0, 0);
  for (int k0 = 0; k0 < K; k0 += C3X_KC) {
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {
      __syncthreads();                                 // the previous phase's readers are done
      if (ky == 0) {
#pragma unroll
        for (int j = 0; j < NXR; ++j) {
          const int i = tid + NT * j, pix = i >> 3, q = i & 7;
          if (i < NP * 8) {
            const int iy = oy0 * S - 1 + pix / IW, ix = ox0 * S - 1 + pix % IW;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};              // padding stays 0 (not silu(shift))
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = tf4_d(xr[j], t_sc, t_sh, t_fl);
            split3(Xp + pix * C3X_LD + 4 * q, XPL, v);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NWR; ++j) {
        const int i = tid + NT * j, q = i & 7, r = (i >> 3) % C3_BN, kx = i / (8 * C3_BN);
        split3(Wp + (kx * C3_BN + r) * C3X_LD + 4 * q, WPL, wr[j]);
      }
      __syncthreads();
      // next phase's operands into registers under this phase's matrix loop
      if (ky < 2) fetch_w(k0, ky + 1);
      else if (k0 + C3X_KC < K) { fetch_w(k0 + C3X_KC, 0); fetch_x(k0 + C3X_KC); }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        bf16x8 xv[PR][3];
#pragma unroll
        for (int p = 0; p < PR; ++p)
#pragma unroll
          for (int t = 0; t < 3; ++t)
            xv[p][t] = *reinterpret_cast<const bf16x8*>(Xp + t * XPL + (((PR * wave + p) * S + ky) * IW + lm * S + kx) * C3X_LD + 8 * g);
#pragma unroll
        for (int c2 = 0; c2 < 4; c2 += 2) {              // two channel tiles at a time: 2 PR independent accumulators per product
          bf16x8 wa[2][3];
#pragma unroll
          for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int t = 0; t < 3; ++t)
              wa[cc][t] = *reinterpret_cast<const bf16x8*>(Wp + t * WPL + (kx * C3_BN + 16 * (c2 + cc) + lm) * C3X_LD + 8 * g);
          constexpr int TW[6] = {2, 0, 1, 1, 0, 0}, TX[6] = {0, 2, 1, 0, 1, 0};     // (w, x): l h, h l, m m, m h, h m, h h
#pragma unroll
          for (int e = 0; e < 6; ++e)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
              for (int p = 0; p < PR; ++p)
                acc[p][c2 + cc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cc][TW[e]], xv[p][TX[e]], acc[p][c2 + cc], 0, 0, 0);
        }
      }
    }
  }
  f32x4 s1[4], s2[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { s1[c] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[c] = s1[c]; }
#pragma unroll
  for (int p = 0; p < PR; ++p) {
    const int oy = oy0 + PR * wave + p, ox = ox0 + lm;
    if (oy >= OH || ox >= OW) continue;
    float* op = out + (((long long)n_img * OH + oy) * OW + ox) * out_ld;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int n = n0 + 16 * c + 4 * g;
      if (n < Nc) {
        f32x4 v = acc[p][c];
        if (accumulate) v += *reinterpret_cast<const f32x4*>(op + n);
        *reinterpret_cast<f32x4*>(op + n) = v;
        s1[c] += v;
        s2[c] += v * v;
      }
    }
  }
  if (stats) {
    wave_stats_to_lds<4>(s1, s2, red + wave * 2 * C3_BN, lane, Nc - n0);
    __syncthreads();
    if (tid < 2 * C3_BN && n0 + (tid >> 1) < Nc) {
      float sum = 0.0f;
#pragma unroll
      for (int wv = 0; wv < NW; ++wv) sum += red[wv * 2 * C3_BN + tid];
      atomicAdd(&stats[(blockIdx.x % JN_NREP) * rep_stride + 2 * n0 + tid], (double)sum);
    }
  }
}

template <int S, bool WT, int PR, int NW>
static void launch_conv3_x3(const ConvArgs& a, hipStream_t s) {
  constexpr int TH = NW * PR;
  const int tiles_x = (a.OW + C3_TW - 1) / C3_TW, tiles_y = (a.OH + TH - 1) / TH;
  dim3 grid(tiles_x * tiles_y * a.N, (a.cout + C3_BN - 1) / C3_BN, a.n_slots > 1 ? a.n_slots : 1);
  const size_t smem = ((size_t)3 * (S * TH + 2) * (S * C3_TW + 2) * C3X_LD + (size_t)3 * 3 * C3_BN * C3X_LD) * sizeof(bf16_t) +
                      (size_t)NW * 2 * C3_BN * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_x3_kernel<S, WT, PR, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3_x3_kernel<S, WT, PR, NW>), grid, dim3(64 * NW), smem, s, (const float*)a.in, a.in_ld, a.itab, a.w,
                     (float*)a.out, a.out_ld, a.H, a.W, a.OH, a.OW, a.cin, a.cout, tiles_x, tiles_y, a.accumulate, a.stats,
                     a.stats_rep_stride, a.skip_flag, a.skip_when, a.in_slot_stride, a.out_slot_stride);
}

// ---- stride 2 on the same scheme: parity classes -----------------------------------------------------------------
// A stride-2 3x3 conv reads, per output pixel, input rows 2 oy - 1 .. 2 oy + 1: split the input into its four parity
// classes (row parity py, column parity px) and every class is a STRIDE-1 problem on a (TH + 1) x 17 block grid with 4 / 2 /
// 2 / 1 of the nine taps: class (1, 1): taps (0|2, 0|2); (1, 0): (0|2, 1); (0, 1): (1, 0|2); (0, 0): (1, 1) — tap k = 0 sits
// one block back, k = 1, 2 in the output pixel's own block.  The class tile (17 x 17 pixels x 32 k x three planes = 83 KB) and
// at most two taps of the weight chunk (37 KB) share the LDS: five phases per K chunk (class (1, 1) takes two), every input
// value staged once, the next phase's tile / weights in flight under the matrix loop.  The stride-1 kernel's LDS halo tile at
// stride 2 would be 34 x 34 pixels (333 KB); with one row per wave (the 4 x 16 tile that fits) the operand reads exceeded
// the LDS rate (measured 5 % slower than the fp32 kernel).
template <int PR, int NW>
__global__ __launch_bounds__(64 * NW) void conv3_x3s2_kernel(const float* __restrict__ x, int x_ld, ChanTab it,
                                                         const float* __restrict__ w, float* __restrict__ out, int out_ld,
                                                         int H, int W, int OH, int OW, int K, int Nc, int tiles_x, int tiles_y,
                                                         double* __restrict__ stats, long long rep_stride,
                                                         const int* __restrict__ skip_flag, int skip_when) {
  if (skip_flag && *skip_flag >= skip_when) return;
  constexpr int NT = 64 * NW, TH = NW * PR, BH = TH + 1, BW = C3_TW + 1, NP = BH * BW;
  constexpr int XPL = NP * C3X_LD, WPL = 2 * C3_BN * C3X_LD;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_x3[];
  bf16_t* Xp = reinterpret_cast<bf16_t*>(smem_x3);                  // [3 planes][NP][C3X_LD]   one parity class
  bf16_t* Wp = Xp + 3 * XPL;                                        // [3 planes][2 taps][C3_BN][C3X_LD]
  float* red = reinterpret_cast<float*>(Wp + 3 * WPL);              // [NW waves][C3_BN][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x % (tiles_x * tiles_y);
  const int n_img = blockIdx.x / (tiles_x * tiles_y);
  const int oy0 = (tile / tiles_x) * TH, ox0 = (tile % tiles_x) * C3_TW;
  const int n0 = blockIdx.y * C3_BN;
  const float* xb = x + (long long)n_img * H * W * x_ld;
  f32x4 acc[PR][4];
#pragma unroll
  for (int p = 0; p < PR; ++p)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NXR = (NP * 8 + NT - 1) / NT;            // f32x4 of a class tile per thread
  constexpr int NWR = (2 * C3_BN * 8 + NT - 1) / NT;     // f32x4 of two taps of the weight chunk per thread
  f32x4 xr[NXR], wr[NWR];
  f32x4 t_sc = {1.f, 1.f, 1.f, 1.f}, t_sh = {0.f, 0.f, 0.f, 0.f}, t_fl = {0.f, 0.f, 0.f, 0.f};
  // phases of a K chunk: parity class (2 py + px), its taps (ky * 3 + kx; the second one repeated when there is one), whether
  // the class tile is new
  constexpr int PH_CLS[5] = {3, 3, 2, 1, 0}, PH_NTAP[5] = {2, 2, 2, 2, 1}, PH_T0[5] = {0, 6, 1, 3, 4}, PH_T1[5] = {2, 8, 7, 5, 4};
  constexpr int PH_NEWX[5] = {1, 0, 1, 1, 1};
  auto fetch_x = [&](int k0, int cls) {
    const int py = cls >> 1, px = cls & 1;
    const int kq = k0 + 4 * (tid & 7);
    t_sc = *reinterpret_cast<const f32x4*>(it.sc + kq); t_sh = *reinterpret_cast<const f32x4*>(it.sh + kq);
    t_fl = *reinterpret_cast<const f32x4*>(it.fl + kq);
#pragma unroll
    for (int j = 0; j < NXR; ++j) {
      const int i = tid + NT * j, pix = i >> 3;
      const int iy = 2 * (oy0 - 1 + pix / BW) + py, ix = 2 * (ox0 - 1 + pix % BW) + px;
      xr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < NP * 8 && iy >= 0 && iy < H && ix >= 0 && ix < W)
        xr[j] = *reinterpret_cast<const f32x4*>(xb + ((long long)iy * W + ix) * x_ld + kq);
    }
  };
  auto fetch_w = [&](int k0, int t0, int t1) {
#pragma unroll
    for (int j = 0; j < NWR; ++j) {
      const int i = tid + NT * j, q = i & 7, r = (i >> 3) % C3_BN, ti = i / (8 * C3_BN);
      const int tp = ti == 0 ? t0 : t1;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < 2 * C3_BN * 8 && n0 + r < Nc) v = *reinterpret_cast<const f32x4*>(w + ((long long)tp * Nc + n0 + r) * K + k0 + 4 * q);
      wr[j] = v;
    }
  };
  auto split3 = [](bf16_t* dst, int plane_stride, f32x4 v) {
    const bf16x4 h = __builtin_convertvector(v, bf16x4);
    const f32x4 r1 = v - __builtin_convertvector(h, f32x4);
    const bf16x4 m = __builtin_convertvector(r1, bf16x4);
    const bf16x4 l = __builtin_convertvector(r1 - __builtin_convertvector(m, f32x4), bf16x4);
    *reinterpret_cast<bf16x4*>(dst) = h;
    *reinterpret_cast<bf16x4*>(dst + plane_stride) = m;
    *reinterpret_cast<bf16x4*>(dst + 2 * plane_stride) = l;
  };
  fetch_x(0, PH_CLS[0]);
  fetch_w(0, PH_T0[0], PH_T1[0]);
  for (int k0 = 0; k0 < K; k0 += C3X_KC) {
#pragma unroll
    for (int ph = 0; ph < 5; ++ph) {
      __syncthreads();                                 // the previous phase's readers are done
      if (PH_NEWX[ph]) {
        const int py = PH_CLS[ph] >> 1, px = PH_CLS[ph] & 1;
#pragma unroll
        for (int j = 0; j < NXR; ++j) {
          const int i = tid + NT * j, pix = i >> 3, q = i & 7;
          if (i < NP * 8) {
            const int iy = 2 * (oy0 - 1 + pix / BW) + py, ix = 2 * (ox0 - 1 + pix % BW) + px;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};              // padding stays 0 (not silu(shift))
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = tf4_d(xr[j], t_sc, t_sh, t_fl);
            split3(Xp + pix * C3X_LD + 4 * q, XPL, v);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NWR; ++j) {
        const int i = tid + NT * j, q = i & 7, r = (i >> 3) % C3_BN, ti = i / (8 * C3_BN);
        if (i < 2 * C3_BN * 8) split3(Wp + (ti * C3_BN + r) * C3X_LD + 4 * q, WPL, wr[j]);
      }
      __syncthreads();
      // next phase's operands into registers under this phase's matrix loop
      if (ph < 4) {
        fetch_w(k0, PH_T0[ph + 1], PH_T1[ph + 1]);
        if (PH_NEWX[ph + 1]) fetch_x(k0, PH_CLS[ph + 1]);
      } else if (k0 + C3X_KC < K) {
        fetch_w(k0 + C3X_KC, PH_T0[0], PH_T1[0]);
        fetch_x(k0 + C3X_KC, PH_CLS[0]);
      }
#pragma unroll
      for (int ti = 0; ti < PH_NTAP[ph]; ++ti) {
        const int tp = ti == 0 ? PH_T0[ph] : PH_T1[ph];
        const int dr = (tp / 3) == 0 ? 0 : 1, dc = (tp % 3) == 0 ? 0 : 1;      // tap 0: one block back; taps 1, 2: this block
        bf16x8 xv[PR][3];
#pragma unroll
        for (int p = 0; p < PR; ++p)
#pragma unroll
          for (int t = 0; t < 3; ++t)
            xv[p][t] = *reinterpret_cast<const bf16x8*>(Xp + t * XPL + ((PR * wave + p + dr) * BW + lm + dc) * C3X_LD + 8 * g);
#pragma unroll
        for (int c2 = 0; c2 < 4; c2 += 2) {
          bf16x8 wa[2][3];
#pragma unroll
          for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int t = 0; t < 3; ++t)
              wa[cc][t] = *reinterpret_cast<const bf16x8*>(Wp + t * WPL + (ti * C3_BN + 16 * (c2 + cc) + lm) * C3X_LD + 8 * g);
          constexpr int TW[6] = {2, 0, 1, 1, 0, 0}, TX[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
          for (int e = 0; e < 6; ++e)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
              for (int p = 0; p < PR; ++p)
                acc[p][c2 + cc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cc][TW[e]], xv[p][TX[e]], acc[p][c2 + cc], 0, 0, 0);
        }
      }
    }
  }
  f32x4 s1[4], s2[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { s1[c] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[c] = s1[c]; }
#pragma unroll
  for (int p = 0; p < PR; ++p) {
    const int oy = oy0 + PR * wave + p, ox = ox0 + lm;
    if (oy >= OH || ox >= OW) continue;
    float* op = out + (((long long)n_img * OH + oy) * OW + ox) * out_ld;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int n = n0 + 16 * c + 4 * g;
      if (n < Nc) {
        const f32x4 v = acc[p][c];
        *reinterpret_cast<f32x4*>(op + n) = v;
        s1[c] += v;
        s2[c] += v * v;
      }
    }
  }
  if (stats) {
    wave_stats_to_lds<4>(s1, s2, red + wave * 2 * C3_BN, lane, Nc - n0);
    __syncthreads();
    if (tid < 2 * C3_BN && n0 + (tid >> 1) < Nc) {
      float sum = 0.0f;
#pragma unroll
      for (int wv = 0; wv < NW; ++wv) sum += red[wv * 2 * C3_BN + tid];
      atomicAdd(&stats[(blockIdx.x % JN_NREP) * rep_stride + 2 * n0 + tid], (double)sum);
    }
  }
}

static void launch_conv3_x3s2(const ConvArgs& a, hipStream_t s) {
  constexpr int PR = 2, NW = 8, TH = NW * PR;
  const int tiles_x = (a.OW + C3_TW - 1) / C3_TW, tiles_y = (a.OH + TH - 1) / TH;
  dim3 grid(tiles_x * tiles_y * a.N, (a.cout + C3_BN - 1) / C3_BN);
  const size_t smem = ((size_t)3 * (TH + 1) * (C3_TW + 1) * C3X_LD + (size_t)3 * 2 * C3_BN * C3X_LD) * sizeof(bf16_t) +
                      (size_t)NW * 2 * C3_BN * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_x3s2_kernel<PR, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3_x3s2_kernel<PR, NW>), grid, dim3(64 * NW), smem, s, (const float*)a.in, a.in_ld, a.itab, a.w,
                     (float*)a.out, a.out_ld, a.H, a.W, a.OH, a.OW, a.cin, a.cout, tiles_x, tiles_y, a.stats, a.stats_rep_stride,
                     a.skip_flag, a.skip_when);
}

// fp32 activations, K a multiple of 32 (every dense 3x3 layer of yolox-s / -m / -l), stride 1; JN_NO_CONV3_X3=1 (read per
// launch): fp32 matrix pipe.  Stride 2 stays on the fp32 kernel: its halo tile leaves room for one row per wave only, the
// operand reads of the three planes then need more than the LDS delivers (measured: 5 % slower).
// The x3 workgroup is 16 x 16 pixels on a whole CU (124 KB of LDS), the fp32 one 8 x 16 with two per CU; per 16 rows of
// pixels the x3 kernel takes ~0.65 of the fp32 kernel's time (tools/c5_layers.sh: 80x80 64 -> 64: 91 -> 62 us; 20x20
// 256 -> 256: 159 -> 107), but a launch is whole rounds of workgroups and a 40 x 40 map is 2.5 tiles high — pick per launch:
// rounds x time per round (forward of 16 patches at 40 x 40: 480 fp32 workgroups = one round, 288 x3 workgroups = two).
static bool conv3_x3_ok(const ConvArgs& a) {
  if (std::getenv("JN_NO_CONV3_X3") || a.stride != 1 || a.in_dtype != JN_F32 || a.out_dtype != JN_F32 || a.cin % 32 || a.cout % 4 ||
      a.in_ld % 4 || a.out_ld % 4 || a.bias || a.act != ACT_NONE)
    return false;
  const long long slots = a.n_slots > 1 ? a.n_slots : 1, nbo = (a.cout + C3_BN - 1) / C3_BN, tx = (a.OW + C3_TW - 1) / C3_TW;
  const long long wg_f32 = tx * ((a.OH + 7) / 8) * a.N * nbo * slots, wg_x3 = tx * ((a.OH + 15) / 16) * a.N * nbo * slots;
  const double cost_f32 = (double)((wg_f32 + 511) / 512), cost_x3 = 0.65 * (double)((wg_x3 + 255) / 256);
  return cost_x3 < cost_f32;
}

// stride 2 forward on the parity-class kernel: whole rounds again (16 x 16 output tiles on a CU of their own against 4 x 16 fp32
// tiles, two workgroups per CU); JN_NO_CONV3_X3S2=1 / JN_NO_CONV3_X3=1: fp32 matrix pipe
static bool conv3_x3s2_ok(const ConvArgs& a) {
  if (std::getenv("JN_NO_CONV3_X3") || std::getenv("JN_NO_CONV3_X3S2") || a.stride != 2 || a.in_dtype != JN_F32 || a.out_dtype != JN_F32 ||
      a.cin % 32 || a.cout % 4 || a.in_ld % 4 || a.out_ld % 4 || a.bias || a.act != ACT_NONE || a.accumulate || a.n_slots > 1 ||
      a.w_transposed)
    return false;
  const long long nbo = (a.cout + C3_BN - 1) / C3_BN, tx = (a.OW + C3_TW - 1) / C3_TW;
  const long long wg_f32 = tx * ((a.OH + 3) / 4) * a.N * nbo, wg_x3 = tx * ((a.OH + 15) / 16) * a.N * nbo;
  // an fp32 workgroup is 4 rows, two per CU: a round of 512 of them covers what 128 x3 workgroups cover
  const double cost_f32 = (double)((wg_f32 + 511) / 512) * 0.5, cost_x3 = 0.65 * (double)((wg_x3 + 255) / 256);
  return cost_x3 < cost_f32;
}

template <int S, typename AT, bool WT>
static void launch_conv3_t(const ConvArgs& a, hipStream_t s) {
  // 8 x 16 pixels per workgroup (2 rows per wave).  16 x 16 (PR = 4) halves the weight-tile staging per MFMA but
  // measured 6 % slower in fp32 (72 KB of LDS per workgroup: fewer workgroups per CU)
  // stride 2: 4 x 16 pixels (PR = 1) — the 18 x 34 halo tile of 8 rows left room for one workgroup per CU
  constexpr int PR = S == 2 ? 1 : 2, TH = 4 * PR;
  const int tiles_x = (a.OW + C3_TW - 1) / C3_TW, tiles_y = (a.OH + TH - 1) / TH;
  dim3 grid(tiles_x * tiles_y * a.N, (a.cout + C3_BN - 1) / C3_BN, a.n_slots > 1 ? a.n_slots : 1);
  constexpr int LD = S == 1 ? 24 : C3_LD;
  const size_t smem = (((size_t)(S * TH + 2) * (S * C3_TW + 2) + 9 * C3_BN) * LD + 4 * 2 * C3_BN) * sizeof(float);
  if (smem > 64 * 1024) {             // stride 2: 95 KB of the CU's 160 KB LDS, above the 64 KB default cap
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_mfma_kernel<S, AT, WT, PR>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr_set = true;
    }
  }
  hipLaunchKernelGGL((conv3_mfma_kernel<S, AT, WT, PR>), grid, dim3(256), smem, s, (const AT*)a.in, a.in_ld, a.itab, a.w,
                     (AT*)a.out, a.out_ld, a.H, a.W, a.OH, a.OW, a.cin, a.cout, tiles_x, tiles_y, a.accumulate, a.stats,
                     a.stats_rep_stride, a.skip_flag, a.skip_when, a.in_slot_stride, a.out_slot_stride);
}

int launch_conv3(const ConvArgs& a, hipStream_t s) {
  if (a.w_transposed) {               // data gradient of a stride-1 layer (fp32 gradient buffers)
    if (a.stride != 1 || a.in_dtype != JN_F32 || a.cin % 4) return -1;
    if (conv3_x3_ok(a)) launch_conv3_x3<1, true, 2, 8>(a, s); else launch_conv3_t<1, float, true>(a, s);
    return 0;
  }
  static const bool no_bf16_conv3 = std::getenv("JN_NO_BF16_CONV3") != nullptr;
  if (!no_bf16_conv3 && a.w_bf16 && a.in_dtype == JN_BF16 && a.out_dtype == JN_BF16 && !a.stats && !a.accumulate && a.cin % 8 == 0) {
    if (a.stride == 1) launch_conv3_bf16<1>(a, s); else launch_conv3_bf16<2>(a, s);
    return 0;
  }
  if (a.in_dtype == JN_BF16) { if (a.stride == 1) launch_conv3_t<1, bf16_t, false>(a, s); else launch_conv3_t<2, bf16_t, false>(a, s); }
  else if (conv3_x3_ok(a)) launch_conv3_x3<1, false, 2, 8>(a, s);
  else if (conv3_x3s2_ok(a)) launch_conv3_x3s2(a, s);
  else { if (a.stride == 1) launch_conv3_t<1, float, false>(a, s); else launch_conv3_t<2, float, false>(a, s); }
  return 0;
}

// ---- data gradient of a stride-2 dense 3x3 layer -------------------------------------------------------
// g_in[iy][ix][k] = sum over taps with (iy + 1 - ky), (ix + 1 - kx) even of g_z[(iy+1-ky)/2][(ix+1-kx)/2][o] * w[tap][o][k].
// The input pixels of one parity class (iy & 1, ix & 1) over a 16 x 32 input tile form an 8 x 16 grid that maps 1:1
// onto output pixels (a, b) = (iy >> 1, ix >> 1) (+1 for the odd taps): the same MFMA tile loop as the forward
// kernel with 1 / 2 / 2 / 4 taps.  blockIdx.z = parity class (x slot); g_z tile (9 x 17 pixels) per K chunk in LDS.
constexpr int C3S2_LD = 24;      // 96-byte rows: conflict-free ds_read_b128 (with 80-byte rows half of the LDS cycles were conflicts)
__global__ __launch_bounds__(256) void conv3_bwd_data_s2_kernel(const float* __restrict__ gz, int g_ld,
                                                                const float* __restrict__ w, float* __restrict__ gin,
                                                                int gin_ld, int H, int W, int OH, int OW, int Co, int Ci,
                                                                int tiles_x, int tiles_y, int accumulate,
                                                                long long g_slot) {
  constexpr int GH = C3_TH + 1, GW = C3_TW + 1;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Gs = smem;                              // [GH*GW][C3S2_LD]   g_z, K chunk of 16 output channels
  float* Ws = smem + GH * GW * C3S2_LD;            // [4][C3_BN][C3S2_LD] the class's taps: [cin row][cout chunk]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, g = lane >> 4;
  const int cls = blockIdx.z & 3, py = cls >> 1, px = cls & 1;
  const long long sl = blockIdx.z >> 2;
  gz += sl * g_slot; gin += sl * g_slot;
  const int tile = blockIdx.x % (tiles_x * tiles_y), n_img = blockIdx.x / (tiles_x * tiles_y);
  const int a0 = (tile / tiles_x) * C3_TH, b0 = (tile % tiles_x) * C3_TW;     // output-space origin of the tile
  const int n0 = blockIdx.y * C3_BN;                                          // input-channel block
  // taps of this class: even coordinate -> k = 1 (offset 0); odd -> k = 0 (offset +1) and k = 2 (offset 0)
  const int nky = py ? 2 : 1, nkx = px ? 2 : 1;
  f32x4 acc[2][4];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = 0; k0 < Co; k0 += C3_KC) {
    if (k0) __syncthreads();
    for (int i = tid; i < GH * GW * 4; i += 256) {
      const int pix = i >> 2, q = i & 3;
      const int oy = a0 + pix / GW, ox = b0 + pix % GW;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (oy < OH && ox < OW && k0 + 4 * q < Co)
        v = *reinterpret_cast<const f32x4*>(gz + (((long long)n_img * OH + oy) * OW + ox) * g_ld + k0 + 4 * q);
      *reinterpret_cast<f32x4*>(Gs + pix * C3S2_LD + 4 * q) = v;
    }
    for (int i = tid; i < nky * nkx * C3_BN * 4; i += 256) {
      const int q = i & 3, r = (i >> 2) % C3_BN, tq = i / (4 * C3_BN);
      const int ky = py ? (tq / nkx) * 2 : 1, kx = px ? (tq % nkx) * 2 : 1;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n0 + r < Ci && k0 + 4 * q < Co) {
        const float* wp = w + ((long long)(ky * 3 + kx) * Co + k0 + 4 * q) * Ci + n0 + r;      // w[tap][o][k]
        v = f32x4{wp[0], wp[Ci], wp[2 * Ci], wp[3 * Ci]};
      }
      *reinterpret_cast<f32x4*>(Ws + (tq * C3_BN + r) * C3S2_LD + 4 * q) = v;
    }
    __syncthreads();
    for (int tq = 0; tq < nky * nkx; ++tq) {
      const int ky = py ? (tq / nkx) * 2 : 1, kx = px ? (tq % nkx) * 2 : 1;
      const int dy = ky == 0 ? 1 : 0, dx = kx == 0 ? 1 : 0;       // (iy + 1 - ky) / 2 - a
      f32x4 xb0 = *reinterpret_cast<const f32x4*>(Gs + ((2 * wave + dy) * GW + lm + dx) * C3S2_LD + 4 * g);
      f32x4 xb1 = *reinterpret_cast<const f32x4*>(Gs + ((2 * wave + 1 + dy) * GW + lm + dx) * C3S2_LD + 4 * g);
      f32x4 wa[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) wa[c] = *reinterpret_cast<const f32x4*>(Ws + (tq * C3_BN + 16 * c + lm) * C3S2_LD + 4 * g);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][j], xb0[j], acc[0][c], 0, 0, 0);
          acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][j], xb1[j], acc[1][c], 0, 0, 0);
        }
    }
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int iy = 2 * (a0 + 2 * wave + p) + py, ix = 2 * (b0 + lm) + px;
    if (iy >= H || ix >= W) continue;
    float* op = gin + (((long long)n_img * H + iy) * W + ix) * gin_ld;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int n = n0 + 16 * c + 4 * g;
      if (n < Ci) {
        f32x4 v = acc[p][c];
        if (accumulate) v += *reinterpret_cast<const f32x4*>(op + n);
        *reinterpret_cast<f32x4*>(op + n) = v;
      }
    }
  }
}

// The same data gradient on the bf16 matrix pipe with three-way split operands (round 4) — the scheme of conv3_x3_kernel<WT>
// applied per parity class: the input pixels of class (py, px) over a 32 x 32 input tile form a 16 x 16 grid that maps 1:1
// onto g_z pixels (a, b) (+1 for the odd taps), i.e. a STRIDE-1 problem over the g_z tile with 1 / 2 / 2 / 4 of the nine
// taps.  g_z chunk (17 x 17 pixels x 32 output channels) split once into three bf16 planes; the class's weight taps go
// through LDS one tap ROW at a time (at most two taps), transposed on the way in; next phase's operands in registers under
// the matrix loop; eight waves, two rows per wave, 64 input channels per workgroup.  blockIdx.z = class + 4 * slot.
template <int PR, int NW>
__global__ __launch_bounds__(64 * NW) void conv3_x3_bwd_data_s2_kernel(const float* __restrict__ gz, int g_ld,
                                                                    const float* __restrict__ w, float* __restrict__ gin,
                                                                    int gin_ld, int H, int W, int OH, int OW, int Co, int Ci,
                                                                    int tiles_x, int tiles_y, int accumulate, long long g_slot) {
  constexpr int NT = 64 * NW, TH = NW * PR, IH = TH + 1, IW = C3_TW + 1, NP = IH * IW;
  constexpr int XPL = NP * C3X_LD, WPL = 2 * C3_BN * C3X_LD;      // elements per plane: g_z tile, two taps of weights
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_x3b[];
  bf16_t* Xp = reinterpret_cast<bf16_t*>(smem_x3b);                // [3 planes][NP][C3X_LD]
  bf16_t* Wp = Xp + 3 * XPL;                                       // [3 planes][2 taps][C3_BN][C3X_LD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, g = lane >> 4;
  const int cls = blockIdx.z & 3, py = cls >> 1, px = cls & 1;
  const long long sl = blockIdx.z >> 2;
  gz += sl * g_slot; gin += sl * g_slot;
  const int tile = blockIdx.x % (tiles_x * tiles_y), n_img = blockIdx.x / (tiles_x * tiles_y);
  const int a0 = (tile / tiles_x) * TH, b0 = (tile % tiles_x) * C3_TW;     // g_z-space origin of the tile
  const int n0 = blockIdx.y * C3_BN;                                       // input-channel block
  const int nky = py ? 2 : 1, nkx = px ? 2 : 1;
  // tap slot -> kernel index and g_z offset: even coordinate: k = 1 (offset 0); odd: slot 0 = k 0 (offset + 1), slot 1 = k 2 (offset 0)
  auto tap_k = [](int odd, int slot) { return odd ? 2 * slot : 1; };
  const float* gb = gz + (long long)n_img * OH * OW * g_ld;
  f32x4 acc[PR][4];
#pragma unroll
  for (int p = 0; p < PR; ++p)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int NXR = (NP * 8 + NT - 1) / NT;            // f32x4 of the g_z tile per thread (8 quads per pixel)
  constexpr int NWR = 2 * C3_BN * 8 / NT;                // f32x4 of one tap row's weight chunk per thread (two taps at most)
  f32x4 xr[NXR], wr[NWR];
  auto fetch_x = [&](int k0) {
    const int kq = k0 + 4 * (tid & 7);
#pragma unroll
    for (int j = 0; j < NXR; ++j) {
      const int i = tid + NT * j, pix = i >> 3;
      const int oy = a0 + pix / IW, ox = b0 + pix % IW;
      xr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < NP * 8 && oy < OH && ox < OW && kq < Co) xr[j] = *reinterpret_cast<const f32x4*>(gb + ((long long)oy * OW + ox) * g_ld + kq);
    }
  };
  auto fetch_w = [&](int k0, int sy) {
    const int ky = tap_k(py, sy);
#pragma unroll
    for (int j = 0; j < NWR; ++j) {
      const int i = tid + NT * j, q = i & 7, r = (i >> 3) % C3_BN, sx = i / (8 * C3_BN);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (sx < nkx && n0 + r < Ci && k0 + 4 * q < Co) {
        const float* wp = w + ((long long)(ky * 3 + tap_k(px, sx)) * Co + k0 + 4 * q) * Ci + n0 + r;      // w[tap][o][k]
        v = f32x4{wp[0], wp[Ci], wp[2 * Ci], wp[3 * Ci]};
      }
      wr[j] = v;
    }
  };
  auto split3 = [](bf16_t* dst, int plane_stride, f32x4 v) {
    const bf16x4 h = __builtin_convertvector(v, bf16x4);
    const f32x4 r1 = v - __builtin_convertvector(h, f32x4);
    const bf16x4 m = __builtin_convertvector(r1, bf16x4);
    const bf16x4 l = __builtin_convertvector(r1 - __builtin_convertvector(m, f32x4), bf16x4);
    *reinterpret_cast<bf16x4*>(dst) = h;
    *reinterpret_cast<bf16x4*>(dst + plane_stride) = m;
    *reinterpret_cast<bf16x4*>(dst + 2 * plane_stride) = l;
  };
  fetch_x(0);
  fetch_w(0, 0);
  for (int k0 = 0; k0 < Co; k0 += C3X_KC) {
#pragma unroll 1
    for (int sy = 0; sy < nky; ++sy) {
      __syncthreads();                                 // the previous phase's readers are done
      if (sy == 0) {
#pragma unroll
        for (int j = 0; j < NXR; ++j) {
          const int i = tid + NT * j, pix = i >> 3, q = i & 7;
          if (i < NP * 8) split3(Xp + pix * C3X_LD + 4 * q, XPL, xr[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < NWR; ++j) {
        const int i = tid + NT * j, q = i & 7, r = (i >> 3) % C3_BN, sx = i / (8 * C3_BN);
        split3(Wp + (sx * C3_BN + r) * C3X_LD + 4 * q, WPL, wr[j]);
      }
      __syncthreads();
      if (sy + 1 < nky) fetch_w(k0, sy + 1);
      else if (k0 + C3X_KC < Co) { fetch_w(k0 + C3X_KC, 0); fetch_x(k0 + C3X_KC); }
      const int dy = (py && sy == 0) ? 1 : 0;          // (iy + 1 - ky) / 2 - a
      for (int sx = 0; sx < nkx; ++sx) {
        const int dx = (px && sx == 0) ? 1 : 0;
        bf16x8 xv[PR][3];
#pragma unroll
        for (int p = 0; p < PR; ++p)
#pragma unroll
          for (int t = 0; t < 3; ++t)
            xv[p][t] = *reinterpret_cast<const bf16x8*>(Xp + t * XPL + ((PR * wave + p + dy) * IW + lm + dx) * C3X_LD + 8 * g);
#pragma unroll
        for (int c2 = 0; c2 < 4; c2 += 2) {
          bf16x8 wa[2][3];
#pragma unroll
          for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int t = 0; t < 3; ++t)
              wa[cc][t] = *reinterpret_cast<const bf16x8*>(Wp + t * WPL + (sx * C3_BN + 16 * (c2 + cc) + lm) * C3X_LD + 8 * g);
          constexpr int TW[6] = {2, 0, 1, 1, 0, 0}, TX[6] = {0, 2, 1, 0, 1, 0};     // (w, x): l h, h l, m m, m h, h m, h h
#pragma unroll
          for (int e = 0; e < 6; ++e)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
              for (int p = 0; p < PR; ++p)
                acc[p][c2 + cc] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa[cc][TW[e]], xv[p][TX[e]], acc[p][c2 + cc], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int p = 0; p < PR; ++p) {
    const int iy = 2 * (a0 + PR * wave + p) + py, ix = 2 * (b0 + lm) + px;
    if (iy >= H || ix >= W) continue;
    float* op = gin + (((long long)n_img * H + iy) * W + ix) * gin_ld;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int n = n0 + 16 * c + 4 * g;
      if (n < Ci) {
        f32x4 v = acc[p][c];
        if (accumulate) v += *reinterpret_cast<const f32x4*>(op + n);
        *reinterpret_cast<f32x4*>(op + n) = v;
      }
    }
  }
}

int launch_conv3_bwd_data_s2(const float* gz, int g_ld, const float* w, float* gin, int gin_ld, int H, int W, int OH,
                             int OW, int Co, int Ci, int N, int accumulate, hipStream_t s, const SlotBatch& sb) {
  if (Co % 4 || Ci % 4) return -1;
  // bf16 matrix pipe (three-way split operands): whole 16 x 16 class tiles on a CU of their own against the fp32 kernel's
  // 8 x 16 tiles, two per CU — chosen like the forward routes, by rounds of workgroups x time per round
  if (!std::getenv("JN_NO_CONV3_X3") && !std::getenv("JN_NO_CONV3_X3S2") && !std::getenv("JN_NO_CONV3_X3S2_BWD") && Co % 32 == 0 && g_ld % 4 == 0 &&
      gin_ld % 4 == 0) {
    constexpr int PR = 2, NW = 8, TH = PR * NW;
    const int txs = (OW + C3_TW - 1) / C3_TW, tys = (OH + TH - 1) / TH;
    const long long nbi = (Ci + C3_BN - 1) / C3_BN;
    const long long wg_x3 = (long long)txs * tys * N * nbi * 4 * sb.n, wg_f32 = (long long)txs * ((OH + C3_TH - 1) / C3_TH) * N * nbi * 4 * sb.n;
    const double cost_f32 = (double)((wg_f32 + 511) / 512), cost_x3 = 0.65 * (double)((wg_x3 + 255) / 256);
    if (cost_x3 < cost_f32) {
      const size_t smem = ((size_t)3 * (TH + 1) * (C3_TW + 1) * C3X_LD + (size_t)3 * 2 * C3_BN * C3X_LD) * sizeof(bf16_t);
      static bool attr_set = false;
      if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_x3_bwd_data_s2_kernel<PR, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        attr_set = true;
      }
      dim3 grid((unsigned)(txs * tys * N), (unsigned)nbi, 4 * sb.n);
      hipLaunchKernelGGL((conv3_x3_bwd_data_s2_kernel<PR, NW>), grid, dim3(64 * NW), smem, s, gz, g_ld, w, gin, gin_ld, H, W, OH, OW,
                         Co, Ci, txs, tys, accumulate, sb.grad);
      return 0;
    }
  }
  const int tiles_x = (OW + C3_TW - 1) / C3_TW, tiles_y = (OH + C3_TH - 1) / C3_TH;
  dim3 grid(tiles_x * tiles_y * N, (Ci + C3_BN - 1) / C3_BN, 4 * sb.n);
  const size_t smem = ((size_t)(C3_TH + 1) * (C3_TW + 1) + 4 * C3_BN) * C3S2_LD * sizeof(float);
  hipLaunchKernelGGL(conv3_bwd_data_s2_kernel, grid, dim3(256), smem, s, gz, g_ld, w, gin, gin_ld, H, W, OH, OW, Co, Ci,
                     tiles_x, tiles_y, accumulate, sb.grad);
  return 0;
}

// ---- weight gradient of a dense 3x3 layer: dW[tap][o][k] += sum_pixels g_z[p][o] * a[p (+) tap][k] ------------
// Workgroup = (pixel-tile group, 64 output channels, 32 input channels): per 8 x 16 pixel tile the g_z tile [128][64]
// and the activated input halo tile [(8S+2)(16S+2)][32] sit in LDS; wave w contracts the 128 pixels for output-channel
// tile w, 2 input tiles and the 9 taps (18 accumulators), persistent over its tiles; one set of atomics at the end.
constexpr int CW_BO = 64, CW_BK = 32, CW_LDG = CW_BO + 4, CW_LDX = CW_BK + 4;

// TH = tile rows: 8 for stride 1; 4 for stride 2 (the 18 x 34 input halo tile of an 8-row tile left room for ONE
// workgroup per CU and nothing to overlap its loads with: matrix pipe 28 % busy)
// SP (default; JN_WW_EXACT=1 keeps fp32 MFMA): as in pw_bwd_weight_wide_kernel, the operands are split into bf16 hi + lo
// when they leave LDS and every 16 x 16 x 32 block takes three v_mfma_f32_16x16x32_bf16 (hi*hi + hi*lo + lo*hi) — a weight
// gradient is a leaf, its ~1e-5 relative error reaches the optimiser only.  Lane group g of a 32-pixel k-step takes
// pixels (row e >> 2, column 4 g + (e & 3)) of the step's two tile rows: conflict-free for both LDS tiles at stride 1.
template <int S, typename XT, int TH, bool SP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(S == 2 ? 2 : 1))) void conv3_bwd_weight_kernel(const float* __restrict__ gz, int g_ld,
                                                               const XT* __restrict__ x, int x_ld, ChanTab it,
                                                               float* __restrict__ gw, int H, int W, int OH, int OW,
                                                               int Co, int Ci, int tiles_x, int tiles_y, int n_tiles,
                                                               SlotBatch sb) {
  constexpr int IH = TH * S + 2, IW = C3_TW * S + 2, NPIX = TH * C3_TW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Gs = smem;                              // [NPIX][CW_LDG]
  float* Xs = smem + NPIX * CW_LDG;               // [IH*IW][CW_LDX]
  {
    const long long sl = blockIdx.z / ((Ci + CW_BK - 1) / CW_BK);
    gz += sl * sb.grad; x += sl * sb.act;
    it.sc += sl * sb.tab; it.sh += sl * sb.tab; it.fl += sl * sb.tab;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, g = lane >> 4;
  const int o0 = blockIdx.y * CW_BO, k0 = (blockIdx.z % ((Ci + CW_BK - 1) / CW_BK)) * CW_BK;
  f32x4 acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = acc[t][0]; }
  // the next tile's values are fetched into registers while the MFMAs of the current one run
  constexpr int NGF = (NPIX * (CW_BO / 4) + 255) / 256, NXF = (IH * IW * (CW_BK / 4) + 255) / 256;
  f32x4 pg[NGF], px[NXF];
  static_assert(256 % (CW_BK / 4) == 0, "the thread's channel quad is fixed");
  f32x4 w_sc = {1.f, 1.f, 1.f, 1.f}, w_sh = {0.f, 0.f, 0.f, 0.f}, w_fl = {0.f, 0.f, 0.f, 0.f};   // k0 and the quad never change
  if (k0 + 4 * (tid % (CW_BK / 4)) < Ci) {
    const int kq = k0 + 4 * (tid % (CW_BK / 4));
    w_sc = *reinterpret_cast<const f32x4*>(it.sc + kq); w_sh = *reinterpret_cast<const f32x4*>(it.sh + kq);
    w_fl = *reinterpret_cast<const f32x4*>(it.fl + kq);
  }
  auto fetch = [&](int tile) {
    const int tr = tile % (tiles_x * tiles_y), n_img = tile / (tiles_x * tiles_y);
    const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * C3_TW;
#pragma unroll
    for (int j = 0; j < NGF; ++j) {
      const int i = tid + 256 * j, pix = i / (CW_BO / 4), q = i % (CW_BO / 4);
      const int oy = oy0 + pix / C3_TW, ox = ox0 + pix % C3_TW;
      pg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < NPIX * (CW_BO / 4) && oy < OH && ox < OW && o0 + 4 * q < Co)
        pg[j] = *reinterpret_cast<const f32x4*>(gz + (((long long)n_img * OH + oy) * OW + ox) * g_ld + o0 + 4 * q);
    }
#pragma unroll
    for (int j = 0; j < NXF; ++j) {
      const int i = tid + 256 * j, pix = i / (CW_BK / 4), q = i % (CW_BK / 4);
      const int iy = oy0 * S - 1 + pix / IW, ix = ox0 * S - 1 + pix % IW;
      px[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < IH * IW * (CW_BK / 4) && iy >= 0 && iy < H && ix >= 0 && ix < W && k0 + 4 * q < Ci)
        px[j] = ld4(x + (((long long)n_img * H + iy) * W + ix) * x_ld + k0 + 4 * q);
    }
  };
  if ((int)blockIdx.x < n_tiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int tr = tile % (tiles_x * tiles_y);
    const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * C3_TW;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NGF; ++j) {
      const int i = tid + 256 * j, pix = i / (CW_BO / 4), q = i % (CW_BO / 4);
      if (i < NPIX * (CW_BO / 4)) *reinterpret_cast<f32x4*>(Gs + pix * CW_LDG + 4 * q) = pg[j];
    }
#pragma unroll
    for (int j = 0; j < NXF; ++j) {
      const int i = tid + 256 * j, pix = i / (CW_BK / 4), q = i % (CW_BK / 4);
      if (i < IH * IW * (CW_BK / 4)) {
        const int iy = oy0 * S - 1 + pix / IW, ix = ox0 * S - 1 + pix % IW;
        const int kk = k0 + 4 * q;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};                     // padding stays 0 (not silu(shift))
        if (iy >= 0 && iy < H && ix >= 0 && ix < W && kk < Ci)
          v = tf4_d(px[j], w_sc, w_sh, w_fl);
        *reinterpret_cast<f32x4*>(Xs + pix * CW_LDX + 4 * q) = v;
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < n_tiles) fetch(tile + gridDim.x);
    if constexpr (SP) {
      static_assert(C3_TW == 16 && NPIX % 32 == 0, "a 32-pixel k-step is two rows of the tile");
#pragma unroll 1
      for (int ks = 0; ks < NPIX / 32; ++ks) {
        bf16x8 ah, al;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int pix = 32 * ks + 16 * (e >> 2) + 4 * g + (e & 3);
          const float v = Gs[pix * CW_LDG + 16 * wave + lm];          // A[i = cout][k = pixel]
          const bf16_t h = (bf16_t)v;
          ah[e] = h; al[e] = (bf16_t)(v - (float)h);
        }
        const float* xp = Xs + ((2 * ks * S) * IW + 4 * g * S) * CW_LDX + lm;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float* xt = xp + ((t / 3) * IW + (t % 3)) * CW_LDX;
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            bf16x8 bh, bl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float v = xt[(((e >> 2) * S) * IW + (e & 3) * S) * CW_LDX + 16 * hh];
              const bf16_t h = (bf16_t)v;
              bh[e] = h; bl[e] = (bf16_t)(v - (float)h);
            }
            f32x4 d = acc[t][hh];
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, d, 0, 0, 0);
            acc[t][hh] = d;
          }
        }
      }
    } else {
#pragma unroll 2
      for (int st = 0; st < NPIX / 4; ++st) {              // 4 pixels per k-step: lane group g takes pixel 4 st + g
        const int pix = 4 * st + g, ty = pix / C3_TW, tx = pix % C3_TW;
        const float av = Gs[pix * CW_LDG + 16 * wave + lm];            // A[i = cout][k = pixel]
        const float* xp = Xs + ((ty * S) * IW + tx * S) * CW_LDX + lm;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const float* xt = xp + ((t / 3) * IW + (t % 3)) * CW_LDX;
          acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xt[0], acc[t][0], 0, 0, 0);
          acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, xt[16], acc[t][1], 0, 0, 0);
        }
      }
    }
  }
  // D[i = cout 4g + r][j = cin lm]: dW[tap][o][k], k contiguous
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = o0 + 16 * wave + 4 * g + r, k = k0 + 16 * h + lm;
        if (o < Co && k < Ci) atomicAdd(&gw[((long long)t * Co + o) * Ci + k], acc[t][h][r]);
      }
}

// ---- the same contraction with the operands split ONCE and read transposed (round 3) -----------------------------
// conv3_bwd_weight_kernel<.., SP> splits every value into bf16 hi + lo on its way OUT of LDS: 8 strided scalar reads and
// ~40 VALU operations per MFMA fragment, nine taps x two halves per k-step — the matrix pipe sat at 11 - 28 % busy under
// the VALU (stride 2: 14 ms per launch at configs[4]).  Here the tiles are split when they are STAGED (once per value)
// and kept as bf16 hi / lo planes in the layout the loads deliver, [pixel][channel]; a fragment — 8 pixels of one
// channel per lane — is two ds_read_b64_tr_b16 (the hardware transposes a 4-pixel x 16-channel block per 16 lanes), with
// per-lane row addresses, so tap shifts and stride 2 cost nothing.  Inside the loop: 4 transposed reads per 3 MFMAs,
// no VALU.  Row strides (G 160 B, X 96 B at stride 1 / 80 B at stride 2) put the 8 pixel rows a 32-lane half reads at
// once on 8 disjoint bank octets.  k of the 32-pixel step <-> pixel: k = 16 h + 8 b + 4 r + q  ->  tile row 2 ks + h,
// column 8 r + 4 b + q (lane group g = 2 h + b, read r, row q of the block) — any bijection works as long as both
// operands use it; this one makes the rows of a half consecutive pixels.
template <int S>
__global__ __launch_bounds__(256, 2) void conv3_bwd_weight_tr_kernel(const float* __restrict__ gz, int g_ld,
                                                                     const float* __restrict__ x, int x_ld, ChanTab it,
                                                                     float* __restrict__ gw, int H, int W, int OH, int OW,
                                                                     int Co, int Ci, int tiles_x, int tiles_y, int n_tiles,
                                                                     SlotBatch sb) {
  constexpr int TH = 4, IH = TH * S + 2, IW = C3_TW * S + 2, NPIX = TH * C3_TW;
  constexpr int GLD = 80, XLD = S == 1 ? 48 : 40;          // row strides in bf16 elements (see above)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_tr[];
  bf16_t* Gh = reinterpret_cast<bf16_t*>(smem_tr);          // [NPIX][GLD]
  bf16_t* Gl = Gh + NPIX * GLD;
  bf16_t* Xh = Gl + NPIX * GLD;                             // [IH * IW][XLD]
  bf16_t* Xl = Xh + IH * IW * XLD;
  {
    const long long sl = blockIdx.z / ((Ci + CW_BK - 1) / CW_BK);
    gz += sl * sb.grad; x += sl * sb.act;
    it.sc += sl * sb.tab; it.sh += sl * sb.tab; it.fl += sl * sb.tab;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, g = lane >> 4;
  const int o0 = blockIdx.y * CW_BO, k0 = (blockIdx.z % ((Ci + CW_BK - 1) / CW_BK)) * CW_BK;
  f32x4 acc[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = acc[t][0]; }
  constexpr int NGF = (NPIX * (CW_BO / 4) + 255) / 256, NXF = (IH * IW * (CW_BK / 4) + 255) / 256;
  f32x4 pg[NGF], px[NXF];
  f32x4 w_sc = {1.f, 1.f, 1.f, 1.f}, w_sh = {0.f, 0.f, 0.f, 0.f}, w_fl = {0.f, 0.f, 0.f, 0.f};
  if (k0 + 4 * (tid % (CW_BK / 4)) < Ci) {
    const int kq = k0 + 4 * (tid % (CW_BK / 4));
    w_sc = *reinterpret_cast<const f32x4*>(it.sc + kq); w_sh = *reinterpret_cast<const f32x4*>(it.sh + kq);
    w_fl = *reinterpret_cast<const f32x4*>(it.fl + kq);
  }
  auto fetch = [&](int tile) {
    const int tr = tile % (tiles_x * tiles_y), n_img = tile / (tiles_x * tiles_y);
    const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * C3_TW;
#pragma unroll
    for (int j = 0; j < NGF; ++j) {
      const int i = tid + 256 * j, pix = i / (CW_BO / 4), q = i % (CW_BO / 4);
      const int oy = oy0 + pix / C3_TW, ox = ox0 + pix % C3_TW;
      pg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < NPIX * (CW_BO / 4) && oy < OH && ox < OW && o0 + 4 * q < Co)
        pg[j] = *reinterpret_cast<const f32x4*>(gz + (((long long)n_img * OH + oy) * OW + ox) * g_ld + o0 + 4 * q);
    }
#pragma unroll
    for (int j = 0; j < NXF; ++j) {
      const int i = tid + 256 * j, pix = i / (CW_BK / 4), q = i % (CW_BK / 4);
      const int iy = oy0 * S - 1 + pix / IW, ix = ox0 * S - 1 + pix % IW;
      px[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < IH * IW * (CW_BK / 4) && iy >= 0 && iy < H && ix >= 0 && ix < W && k0 + 4 * q < Ci)
        px[j] = *reinterpret_cast<const f32x4*>(x + (((long long)n_img * H + iy) * W + ix) * x_ld + k0 + 4 * q);
    }
  };
  auto split_store = [](bf16_t* hi, bf16_t* lo, f32x4 v) {
    const bf16x4 h = __builtin_convertvector(v, bf16x4);
    const bf16x4 l = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), bf16x4);
    *reinterpret_cast<bf16x4*>(hi) = h; *reinterpret_cast<bf16x4*>(lo) = l;
  };
  // transposed-read addresses of this lane: row q = (lane & 15) >> 2 of the block, channels 4 p .. 4 p + 3
  const int tq = (lane & 15) >> 2, tp = lane & 3, th = g >> 1, tb = g & 1;
  // operand A (g_z): pixel (2 ks + th, 8 r + 4 tb + tq), channels 16 wave + 4 tp
  const int a_off = (th * C3_TW + 4 * tb + tq) * GLD + 16 * wave + 4 * tp;
  // operand B (input): halo pixel (S (2 ks + th) + ky, S (8 r + 4 tb + tq) + kx), channels 16 hh + 4 tp
  const int b_off = ((S * th) * IW + S * (4 * tb + tq)) * XLD + 4 * tp;
  if ((int)blockIdx.x < n_tiles) fetch(blockIdx.x);
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int tr = tile % (tiles_x * tiles_y);
    const int oy0 = (tr / tiles_x) * TH, ox0 = (tr % tiles_x) * C3_TW;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NGF; ++j) {
      const int i = tid + 256 * j, pix = i / (CW_BO / 4), q = i % (CW_BO / 4);
      if (i < NPIX * (CW_BO / 4)) split_store(Gh + pix * GLD + 4 * q, Gl + pix * GLD + 4 * q, pg[j]);
    }
#pragma unroll
    for (int j = 0; j < NXF; ++j) {
      const int i = tid + 256 * j, pix = i / (CW_BK / 4), q = i % (CW_BK / 4);
      if (i < IH * IW * (CW_BK / 4)) {
        const int iy = oy0 * S - 1 + pix / IW, ix = ox0 * S - 1 + pix % IW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};                     // padding stays 0 (not silu(shift))
        if (iy >= 0 && iy < H && ix >= 0 && ix < W && k0 + 4 * q < Ci) v = tf4_d(px[j], w_sc, w_sh, w_fl);
        split_store(Xh + pix * XLD + 4 * q, Xl + pix * XLD + 4 * q, v);
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < n_tiles) fetch(tile + gridDim.x);
#pragma unroll
    for (int ks = 0; ks < NPIX / 32; ++ks) {
      const int ga = a_off + (2 * ks) * C3_TW * GLD;
      const bf16x8 ah = tr_frag(Gh + ga, Gh + ga + 8 * GLD), al = tr_frag(Gl + ga, Gl + ga + 8 * GLD);
      const int xb = b_off + (S * 2 * ks) * IW * XLD;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int xt = xb + ((t / 3) * IW + (t % 3)) * XLD;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const bf16x8 bh = tr_frag(Xh + xt + 16 * hh, Xh + xt + 16 * hh + 8 * S * XLD);
          const bf16x8 bl = tr_frag(Xl + xt + 16 * hh, Xl + xt + 16 * hh + 8 * S * XLD);
          f32x4 d = acc[t][hh];
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, d, 0, 0, 0);
          d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, d, 0, 0, 0);
          acc[t][hh] = d;
        }
      }
    }
  }
  // D[i = cout 4g + r][j = cin lm]: dW[tap][o][k], k contiguous
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = o0 + 16 * wave + 4 * g + r, k = k0 + 16 * h + lm;
        if (o < Co && k < Ci) atomicAdd(&gw[((long long)t * Co + o) * Ci + k], acc[t][h][r]);
      }
}

int launch_conv3_bwd_weight(const float* gz, int g_ld, const void* x, int x_dtype, int x_ld, ChanTab it, float* gw, int H,
                            int W, int OH, int OW, int Co, int Ci, int N, int stride, hipStream_t s, const SlotBatch& sb) {
  const int TH = 4;
  const int tiles_x = (OW + C3_TW - 1) / C3_TW, tiles_y = (OH + TH - 1) / TH, n_tiles = tiles_x * tiles_y * N;
  const int nbo = (Co + CW_BO - 1) / CW_BO, nbk = (Ci + CW_BK - 1) / CW_BK;
  long long gx = 2048 / ((long long)nbo * nbk * sb.n);      // persistent workgroups over the pixel tiles
  if (gx < 1) gx = 1;
  if (gx > n_tiles) gx = n_tiles;
  dim3 grid((unsigned)gx, nbo, nbk * sb.n);
  static const bool exact = std::getenv("JN_WW_EXACT") != nullptr;
  static const bool no_tr = std::getenv("JN_NO_CONV3_WTR") != nullptr;
  if (!exact && !no_tr && x_dtype == JN_F32 && Co % 16 == 0 && Ci % 4 == 0 && g_ld % 4 == 0 && x_ld % 4 == 0) {
    // split planes + transposed LDS reads (conv3_bwd_weight_tr_kernel)
#define JN_CWT(S_)                                                                                                      \
    {                                                                                                                   \
      constexpr int IHW = (4 * S_ + 2) * (C3_TW * S_ + 2);                                                              \
      const size_t smem = ((size_t)2 * 4 * C3_TW * 80 + (size_t)2 * IHW * (S_ == 1 ? 48 : 40)) * sizeof(bf16_t);        \
      static bool raised = false;                                                                                       \
      if (!raised) {                                                                                                    \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_bwd_weight_tr_kernel<S_>),                      \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                               \
        raised = true;                                                                                                  \
      }                                                                                                                 \
      hipLaunchKernelGGL((conv3_bwd_weight_tr_kernel<S_>), grid, dim3(256), smem, s, gz, g_ld, (const float*)x, x_ld, it, gw, \
                         H, W, OH, OW, Co, Ci, tiles_x, tiles_y, n_tiles, sb);                                          \
    }
    if (stride == 1) JN_CWT(1) else JN_CWT(2)
#undef JN_CWT
    return 0;
  }
#define JN_CW(S_, T_, TH_, SP_)                                                                                       \
  {                                                                                                                   \
    const size_t smem = ((size_t)TH_ * C3_TW * CW_LDG + (size_t)(TH_ * S_ + 2) * (C3_TW * S_ + 2) * CW_LDX) * sizeof(float); \
    if (smem > 64 * 1024) {                                                                                           \
      static bool raised = false;                                                                                     \
      if (!raised) {                                                                                                  \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3_bwd_weight_kernel<S_, T_, TH_, SP_>),          \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                             \
        raised = true;                                                                                                \
      }                                                                                                               \
    }                                                                                                                 \
    hipLaunchKernelGGL((conv3_bwd_weight_kernel<S_, T_, TH_, SP_>), grid, dim3(256), smem, s, gz, g_ld, (const T_*)x, x_ld, \
                       it, gw, H, W, OH, OW, Co, Ci, tiles_x, tiles_y, n_tiles, sb);                                  \
  }
#define JN_CW2(S_, T_) { if (exact) JN_CW(S_, T_, 4, false) else JN_CW(S_, T_, 4, true) }
  if (x_dtype == JN_BF16) { if (stride == 1) JN_CW2(1, bf16_t) else JN_CW2(2, bf16_t) }
  else { if (stride == 1) JN_CW2(1, float) else JN_CW2(2, float) }
#undef JN_CW2
#undef JN_CW
  return 0;
}

// ------------------------------------------------------------------------------------
// YOLOXHead predictors of one level + decode: raw[n][a0 + p][0..5] =
//   ((reg_xy + grid) * stride, exp(reg_wh) * stride, sigmoid(obj), sigmoid(cls))
// wp = [6][hid]: rows 0-3 reg_pred, 4 obj_pred (both read reg_feat), 5 cls_pred (reads cls_feat); bp[6].
// ------------------------------------------------------------------------------------
template <typename AT>
__global__ __launch_bounds__(256) void head_pred_kernel(const AT* __restrict__ reg, int reg_ld, ChanTab rt,
                                                        const AT* __restrict__ cls, int cls_ld, ChanTab ct,
                                                        const float* __restrict__ wp, const float* __restrict__ bp,
                                                        float* __restrict__ raw, int hid, int Hl, int Wl, int stride,
                                                        int A, int a0, int N, int logits_only) {
  extern __shared__ float sw[];     // [6][hid] + tables
  for (int i = threadIdx.x; i < 6 * hid; i += 256) sw[i] = wp[i];
  __syncthreads();
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)N * Hl * Wl) return;
  const int p = (int)(idx % (Hl * Wl));
  const long long n = idx / (Hl * Wl);
  const AT* rp = reg + idx * reg_ld;
  const AT* cp = cls + idx * cls_ld;
  float o[6] = {bp[0], bp[1], bp[2], bp[3], bp[4], bp[8]};     // bias layout: reg 0..3 | obj 4 (+pad) | cls 8 (+pad)
  for (int k = 0; k < hid; k += 4) {
    const f32x4 rv = tf4_d(ld4(rp + k), *reinterpret_cast<const f32x4*>(rt.sc + k),
                           *reinterpret_cast<const f32x4*>(rt.sh + k), *reinterpret_cast<const f32x4*>(rt.fl + k));
    const f32x4 cv = tf4_d(ld4(cp + k), *reinterpret_cast<const f32x4*>(ct.sc + k),
                           *reinterpret_cast<const f32x4*>(ct.sh + k), *reinterpret_cast<const f32x4*>(ct.fl + k));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int j = 0; j < 5; ++j) o[j] = fmaf(rv[q], sw[j * hid + k + q], o[j]);
      o[5] = fmaf(cv[q], sw[5 * hid + k + q], o[5]);
    }
  }
  const int gy = p / Wl, gx = p % Wl;
  float* dst = raw + (n * A + a0 + p) * 6;
  if (logits_only) {                 // training: raw predictor outputs, the loss kernel decodes
#pragma unroll
    for (int j = 0; j < 6; ++j) dst[j] = o[j];
    return;
  }
  dst[0] = (o[0] + (float)gx) * (float)stride;
  dst[1] = (o[1] + (float)gy) * (float)stride;
  dst[2] = expf(o[2]) * (float)stride;
  dst[3] = expf(o[3]) * (float)stride;
  dst[4] = 1.0f / (1.0f + expf(-o[4]));
  dst[5] = 1.0f / (1.0f + expf(-o[5]));
}

int launch_head_pred(const void* reg, int reg_ld, ChanTab rt, const void* cls, int cls_ld, ChanTab ct, int dtype,
                     const float* wp, const float* bp, float* raw, int hid, int Hl, int Wl, int stride, int A, int a0,
                     int N, hipStream_t s, int logits_only) {
  const long long total = (long long)N * Hl * Wl;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == JN_BF16)
    hipLaunchKernelGGL(head_pred_kernel<bf16_t>, grid, dim3(256), (size_t)6 * hid * sizeof(float), s, (const bf16_t*)reg,
                       reg_ld, rt, (const bf16_t*)cls, cls_ld, ct, wp, bp, raw, hid, Hl, Wl, stride, A, a0, N, logits_only);
  else
    hipLaunchKernelGGL(head_pred_kernel<float>, grid, dim3(256), (size_t)6 * hid * sizeof(float), s, (const float*)reg,
                       reg_ld, rt, (const float*)cls, cls_ld, ct, wp, bp, raw, hid, Hl, Wl, stride, A, a0, N, logits_only);
  return 0;
}

// ------------------------------------------------------------------------------------
// postprocess (class-agnostic, one class): cxcywh -> xyxy, keep obj*cls >= conf, sort by score
// (descending, ties by anchor index), greedy NMS (IoU > thr suppressed), clamp to [0, P-1].
// One workgroup per patch; candidates live in LDS (cap DET_CAP, lowest anchor indices kept).
// ------------------------------------------------------------------------------------
constexpr int DET_CAP = 2048;

__global__ __launch_bounds__(256) void postprocess_kernel(const float* __restrict__ raw, int A, float conf, float nms_thr,
                                                          float clamp_max, float* __restrict__ boxes,
                                                          int* __restrict__ counts, int max_out) {
  __shared__ float sc[DET_CAP];
  __shared__ int id[DET_CAP];
  __shared__ unsigned char dead[DET_CAP];
  __shared__ int s_n, s_keep;
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* r = raw + (long long)n * A * 6;
  if (tid == 0) { s_n = 0; s_keep = 0; }
  __syncthreads();
  // ordered compaction, chunk by chunk so that lower anchor indices win the cap
  for (int base = 0; base < A; base += 256) {
    const int a = base + tid;
    float score = -1.0f;
    if (a < A) score = r[a * 6 + 4] * r[a * 6 + 5];
    const bool ok = a < A && score >= conf;
    const unsigned long long m = __ballot(ok);
    __shared__ int wave_cnt[4];
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) wave_cnt[wv] = __popcll(m);
    __syncthreads();
    int off = s_n;
    for (int w2 = 0; w2 < wv; ++w2) off += wave_cnt[w2];
    off += __popcll(m & ((1ull << lane) - 1ull));
    if (ok && off < DET_CAP) { sc[off] = score; id[off] = a; }
    __syncthreads();
    if (tid == 0) s_n = min(DET_CAP, s_n + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3]);
    __syncthreads();
  }
  const int cnt = s_n;
  // bitonic sort on (score desc, index asc)
  int np2 = 1;
  while (np2 < cnt) np2 <<= 1;
  for (int i = cnt + tid; i < np2; i += 256) { sc[i] = -INFINITY; id[i] = 0x7fffffff; }
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < np2; i += 256) {
        const int l = i ^ j;
        if (l > i) {
          const bool up = (i & k) == 0;
          const bool before = sc[i] > sc[l] || (sc[i] == sc[l] && id[i] < id[l]);   // i should precede l
          if (up ? !before : before) {
            const float ts = sc[i]; sc[i] = sc[l]; sc[l] = ts;
            const int ti = id[i]; id[i] = id[l]; id[l] = ti;
          }
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < cnt; i += 256) dead[i] = 0;
  __syncthreads();
  for (int i = 0; i < cnt; ++i) {
    if (dead[i]) continue;                          // uniform: dead[] is shared and synced
    const float* bi = r + id[i] * 6;
    const float ax1 = bi[0] - bi[2] * 0.5f, ay1 = bi[1] - bi[3] * 0.5f, ax2 = bi[0] + bi[2] * 0.5f, ay2 = bi[1] + bi[3] * 0.5f;
    const float aa = (ax2 - ax1) * (ay2 - ay1);
    if (tid == 0) {
      const int kidx = s_keep;
      if (kidx < max_out) {
        float* o = boxes + ((long long)n * max_out + kidx) * 7;
        o[0] = fminf(fmaxf(ax1, 0.0f), clamp_max); o[1] = fminf(fmaxf(ay1, 0.0f), clamp_max);
        o[2] = fminf(fmaxf(ax2, 0.0f), clamp_max); o[3] = fminf(fmaxf(ay2, 0.0f), clamp_max);
        o[4] = bi[4]; o[5] = bi[5]; o[6] = 0.0f;
      }
      s_keep = kidx + 1;
    }
    for (int j = i + 1 + tid; j < cnt; j += 256) {
      if (dead[j]) continue;
      const float* bj = r + id[j] * 6;
      const float bx1 = bj[0] - bj[2] * 0.5f, by1 = bj[1] - bj[3] * 0.5f, bx2 = bj[0] + bj[2] * 0.5f, by2 = bj[1] + bj[3] * 0.5f;
      const float iw = fmaxf(fminf(ax2, bx2) - fmaxf(ax1, bx1), 0.0f), ih = fmaxf(fminf(ay2, by2) - fmaxf(ay1, by1), 0.0f);
      const float inter = iw * ih;
      const float iou = inter / (aa + (bx2 - bx1) * (by2 - by1) - inter);
      if (iou > nms_thr) dead[j] = 1;
    }
    __syncthreads();
  }
  if (tid == 0) counts[n] = min(s_keep, max_out);
}

int launch_postprocess(const float* raw, int A, int N, float conf, float nms_thr, float clamp_max, float* boxes,
                       int* counts, int max_out, hipStream_t s) {
  hipLaunchKernelGGL(postprocess_kernel, dim3(N), dim3(256), 0, s, raw, A, conf, nms_thr, clamp_max, boxes, counts,
                     max_out);
  return 0;
}

// boxes [B][K][7] / counts [B] of one glimpse step -> column `col` of [B][cols][K][7] / [B][cols]
__global__ void det_scatter_kernel(const float* __restrict__ boxes, const int* __restrict__ counts,
                                   float* __restrict__ out_boxes, int* __restrict__ out_counts, int B, int cols, int col,
                                   int K, const int* __restrict__ skip_flag, int skip_when) {
  if (skip_flag && *skip_flag >= skip_when) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * K * 7) return;
  const int b = i / (K * 7), r = i % (K * 7);
  out_boxes[((long long)b * cols + col) * K * 7 + r] = (r / 7 < counts[b]) ? boxes[i] : 0.0f;
  if (r == 0) out_counts[b * cols + col] = counts[b];
}

int launch_det_scatter(const float* boxes, const int* counts, float* out_boxes, int* out_counts, int B, int cols, int col,
                       int K, const int* skip_flag, int skip_when, hipStream_t s) {
  hipLaunchKernelGGL(det_scatter_kernel, dim3((B * K * 7 + 255) / 256), dim3(256), 0, s, boxes, counts, out_boxes,
                     out_counts, B, cols, col, K, skip_flag, skip_when);
  return 0;
}

}  // namespace jnr
