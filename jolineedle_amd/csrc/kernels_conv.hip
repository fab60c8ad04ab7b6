// Convolution kernels of the YOLOX PAFPN patch encoder for gfx950 (MI355X).
//
// Data model ("normalize on read").  Every conv writes its RAW output z (no bias, no BN, no
// activation) exactly once, NHWC fp32.  Each buffer channel carries an affine (scale, shift)
// and a flag in a small per-workspace table; every consumer applies
//        a = flag ? silu(z * scale + shift) : z
// while loading.  In eval mode the table holds the BatchNorm running statistics (set once at
// weight load); in train mode each conv also accumulates per-channel sum / sum-of-squares of
// z (fp64 atomics, one pair per channel per workgroup) and a tiny finalize kernel turns them
// into the batch-statistics (scale, shift) of that layer before its consumers run — the
// reference trains with batch statistics per glimpse step (src/reinforce.py:304,
// src/models/yolox.py:54-55).  One code path serves both modes and the backward pass
// (which needs z, not a).
//
//   stem_mfma_kernel  Focus + dense 3x3 == 6x6 stride-2 conv on the image (K = 108 = 27 x 4)
//                     on v_mfma_f32_16x16x4_f32; reads the patch straight out of the big image
//                     at the agent's position (gather fused), LDS halo tile.
//   dw3x3_kernel      depthwise 3x3, stride 1/2, float4 over channels, 4-row strips.
//   pw_mfma_kernel    1x1 conv as GEMM D = W * X^T on v_mfma_f32_16x16x4_f32 (exact fp32).
//   spp_kernel        SPP max-pools 5/9/13 as cascaded separable 5-pools in LDS.
//   upsample_kernel   nearest x2 copy of raw z into a concat slice.
//   addact_kernel     bottleneck shortcut: out = T(res) + silu(bn(z)) materialised.
//   bn_finalize_kernel  batch statistics -> (scale, shift) table, running-stat update.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "jn_kernels.h"
#include "jn_reduce.h"
#include "jn_tab.h"
#include "jn_types.h"

namespace jnr {

__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == ACT_SILU) return silu(v);
  if (act == ACT_RELU) return fmaxf(v, 0.0f);
  if (act == ACT_SIGMOID) return 1.0f / (1.0f + __expf(-v));
  return v;
}

// a = flag ? silu(z * sc + sh) : z, 4 channels at once
__device__ __forceinline__ f32x4 tf4(f32x4 z, f32x4 sc, f32x4 sh, f32x4 fl) {
  f32x4 r;
  r.x = fl.x != 0.0f ? silu(fmaf(z.x, sc.x, sh.x)) : z.x;
  r.y = fl.y != 0.0f ? silu(fmaf(z.y, sc.y, sh.y)) : z.y;
  r.z = fl.z != 0.0f ? silu(fmaf(z.z, sc.z, sh.z)) : z.z;
  r.w = fl.w != 0.0f ? silu(fmaf(z.w, sc.w, sh.w)) : z.w;
  return r;
}

// ------------------------------------------------------------------------------------
// stem: z[n][oy][ox][oc] = sum_{c,dy,dx} w[(c,dy,dx)][oc] * img[c][2oy-2+dy][2ox-2+dx]
// Workgroup = 16 x 32 output pixels; 4 waves x 4 rows; per row two 16-pixel MFMA tiles.
// ------------------------------------------------------------------------------------
constexpr int ST_TY = 16, ST_TX = 32;
constexpr int ST_IH = 2 * ST_TY + 4, ST_IW = 2 * ST_TX + 4;   // 36 x 68
constexpr int ST_KS = 27;                                     // 108 / 4 k-steps

// Persistent: a workgroup keeps its weight fragments in registers and walks over (patch, tile) pairs; the image
// values of the NEXT tile are fetched into registers while the MFMAs of the current one run (the per-tile
// load -> LDS -> MFMA sequence left the matrix pipe idle two thirds of the time).
constexpr int ST_NL = (3 * ST_IH * ST_IW + 255) / 256;        // image values per thread per tile
// VEC: the image tile is fetched with 16-byte loads.  The tile's row window [2 ox0 - 2, 2 ox0 + 66) starts 8 bytes off
// a 16-byte boundary, so the fetch takes the ALIGNED window [2 ox0 - 4, 2 ox0 + 68): 18 float4 per row, of which the
// outer halves of the first and last are dropped when the row is written to LDS.  With 4-byte loads the image read was
// bound by the number of load instructions, not by bytes: 2.3 - 2.5 TB/s whatever the segment length, against 4.1 - 6.2
// TB/s for the same tiles with 16-byte loads (tools/imgreadbench.hip, profiles/r04_imgreadbench.txt).  Needs 16-byte
// aligned rows (base pointer, strides and the patch size multiples of four floats: checked by the launcher).
constexpr int ST_W4 = ST_IW / 4 + 1;                            // float4 groups per row of the aligned window
constexpr int ST_NL4 = (3 * ST_IH * ST_W4 + 255) / 256;

template <typename OT, bool VEC>
__global__ __launch_bounds__(256) void stem_mfma_kernel(
    const float* __restrict__ src, const long long* __restrict__ pos, int pos_stride, long long sample_stride,
    long long chan_stride, int row_stride, int P, const float* __restrict__ w, OT* __restrict__ out,
    int out_ld, int cout, int tiles_x, int tiles_y, int n_tiles, double* __restrict__ stats, long long rep_stride,
    const int* __restrict__ skip_flag, int skip_when, int nrep) {
  if (skip_flag && *skip_flag >= skip_when) return;
  __shared__ __attribute__((aligned(16))) float tile[3 * ST_IH * ST_IW];
  __shared__ float red[4 * 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, g = lane >> 4;
  const int og = blockIdx.y;
  const int OH = P / 2;
  // A operand (weights): lane holds W[oc = lm][k = 4s + g] for every k-step s; B operand offset table
  float wreg[ST_KS];
  int koff[ST_KS];
#pragma unroll
  for (int s = 0; s < ST_KS; ++s) {
    const int k = 4 * s + g;
    wreg[s] = w[k * cout + og * 16 + lm];
    const int c = k / 36, dy = (k % 36) / 6, dx = k % 6;
    koff[s] = (c * ST_IH + dy) * ST_IW + dx;
  }
  float pre[VEC ? 1 : ST_NL];
  f32x4 pre4[VEC ? ST_NL4 : 1];
  auto fetch = [&](int tl) {
    const int n = tl / (tiles_x * tiles_y), tr = tl - n * (tiles_x * tiles_y);
    const int oy0 = (tr / tiles_x) * ST_TY, ox0 = (tr % tiles_x) * ST_TX;
    const float* base = src + (long long)n * sample_stride;
    if (pos) base += pos[(long long)pos_stride * n] * (long long)P * row_stride + pos[(long long)pos_stride * n + 1] * (long long)P;
    if constexpr (VEC) {
#pragma unroll
      for (int j = 0; j < ST_NL4; ++j) {
        const int i = tid + 256 * j;
        const int c = i / (ST_IH * ST_W4), r = (i / ST_W4) % ST_IH, q4 = i % ST_W4;
        const int iy = 2 * oy0 - 2 + r, ix = 2 * ox0 - 4 + 4 * q4;          // P % 4 == 0: a group is inside the patch or outside
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (i < 3 * ST_IH * ST_W4 && iy >= 0 && iy < P && ix >= 0 && ix < P)
          v = *reinterpret_cast<const f32x4*>(base + c * chan_stride + (long long)iy * row_stride + ix);
        pre4[j] = v;
      }
    } else {
#pragma unroll
      for (int j = 0; j < ST_NL; ++j) {
        const int i = tid + 256 * j;
        const int c = i / (ST_IH * ST_IW), r = (i / ST_IW) % ST_IH, q = i % ST_IW;
        const int iy = 2 * oy0 - 2 + r, ix = 2 * ox0 - 2 + q;
        float v = 0.0f;
        if (i < 3 * ST_IH * ST_IW && iy >= 0 && iy < P && ix >= 0 && ix < P) v = base[c * chan_stride + (long long)iy * row_stride + ix];
        pre[j] = v;
      }
    }
  };
  f32x4 s1[1] = {f32x4{0.f, 0.f, 0.f, 0.f}}, s2[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
  int tl = blockIdx.x;
  if (tl < n_tiles) fetch(tl);
  for (; tl < n_tiles; tl += gridDim.x) {
    const int n = tl / (tiles_x * tiles_y), tr = tl - n * (tiles_x * tiles_y);
    const int oy0 = (tr / tiles_x) * ST_TY, ox0 = (tr % tiles_x) * ST_TX;
    __syncthreads();                                   // the previous tile's MFMAs have read their operands
    if constexpr (VEC) {
#pragma unroll
      for (int j = 0; j < ST_NL4; ++j) {
        const int i = tid + 256 * j;
        if (i < 3 * ST_IH * ST_W4) {
          const int q4 = i % ST_W4;
          float* d = tile + (i / ST_W4) * ST_IW + 4 * q4 - 2;                // 8-byte aligned (row = 272 bytes)
          if (q4 > 0) *reinterpret_cast<float2*>(d) = float2{pre4[j].x, pre4[j].y};
          if (q4 < ST_W4 - 1) *reinterpret_cast<float2*>(d + 2) = float2{pre4[j].z, pre4[j].w};
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < ST_NL; ++j) {
        const int i = tid + 256 * j;
        if (i < 3 * ST_IH * ST_IW) tile[i] = pre[j];
      }
    }
    __syncthreads();
    if (tl + (int)gridDim.x < n_tiles) fetch(tl + gridDim.x);
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
      const int ty = wave * 4 + r;
      const int oy = oy0 + ty;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int pbase = (2 * ty) * ST_IW + 2 * (16 * h + lm);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < ST_KS; ++s)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[s], tile[pbase + koff[s]], acc, 0, 0, 0);
        const int ox = ox0 + 16 * h + lm;
        if (oy < OH && ox < OH) {
          st4(out + (((long long)n * OH + oy) * OH + ox) * out_ld + og * 16 + 4 * g, acc);
          s1[0] += acc;
          s2[0] += acc * acc;
        }
      }
    }
  }
  if (stats) {
    wave_stats_to_lds<1>(s1, s2, red + 32 * wave, lane, 16);
    __syncthreads();
    const int rep = (blockIdx.x + blockIdx.y) % nrep;
    if (tid < 32)
      atomicAdd(&stats[rep * rep_stride + 2 * (og * 16) + tid], (double)(red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid]));
  }
}

// 16-byte loads of image rows: base pointer, every stride and the patch size must be multiples of four floats
bool stem_rows_aligned(const StemArgs& a) {
  return (reinterpret_cast<uintptr_t>(a.src) & 15) == 0 && a.sample_stride % 4 == 0 && a.chan_stride % 4 == 0 &&
         a.row_stride % 4 == 0 && a.P % 4 == 0;
}

int launch_stem(const StemArgs& a, hipStream_t s) {
  const int OH = a.P / 2;
  const int ocg = a.cout / 16;
  const int tiles_x = (OH + ST_TX - 1) / ST_TX, tiles_y = (OH + ST_TY - 1) / ST_TY, n_tiles = tiles_x * tiles_y * a.N;
  int nwg = 512 / ocg;                                 // 2 persistent workgroups per CU (222 VGPRs)
  if (nwg < 1) nwg = 1;
  if (nwg > n_tiles) nwg = n_tiles;
  dim3 grid(nwg, ocg);
  const int nrep = a.stats_nrep > 0 ? a.stats_nrep : JN_NREP;
  const bool vec = stem_rows_aligned(a);
#define JN_STEM(OT_, V_)                                                                                                       \
  hipLaunchKernelGGL((stem_mfma_kernel<OT_, V_>), grid, dim3(256), 0, s, a.src, (const long long*)a.positions, a.pos_stride, \
                     a.sample_stride, a.chan_stride, a.row_stride, a.P, a.w, (OT_*)a.out, a.out_ld, a.cout, tiles_x,         \
                     tiles_y, n_tiles, a.stats, a.stats_rep_stride, a.skip_flag, a.skip_when, nrep)
  if (a.out_dtype == JN_BF16) { if (vec) JN_STEM(bf16_t, true); else JN_STEM(bf16_t, false); }
  else { if (vec) JN_STEM(float, true); else JN_STEM(float, false); }
#undef JN_STEM
  return 0;
}

// ------------------------------------------------------------------------------------
// depthwise 3x3 (pad 1), stride S; thread = 4 channels x 4 output rows of one column.
// ------------------------------------------------------------------------------------
template <int S, typename AT>
__global__ __launch_bounds__(256) void dw3x3_kernel(
    const AT* __restrict__ in, int in_ld, ChanTab it, const float* __restrict__ w, AT* __restrict__ out,
    int out_ld, int C, int H, int W, int OH, int OW, int N, double* __restrict__ stats, long long rep_stride,
    const int* __restrict__ skip_flag, int skip_when) {
  if (skip_flag && *skip_flag >= skip_when) return;
  extern __shared__ float red[];   // [C][2] when stats
  if (stats) {
    for (int i = threadIdx.x; i < 2 * C; i += 256) red[i] = 0.0f;
    __syncthreads();
  }
  const int C4 = C >> 2;
  const int YS = (OH + 3) >> 2;
  const long long total = (long long)N * YS * OW * C4;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  const int cstat = (int)(idx % C4) * 4;
  if (idx < total) {
    const int c4 = (int)(idx % C4);
    const int ox = (int)((idx / C4) % OW);
    const int ys = (int)((idx / ((long long)C4 * OW)) % YS);
    const int n = (int)(idx / ((long long)C4 * OW * YS));
    const int c = c4 * 4;
    f32x4 wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4*>(w + t * C + c);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(it.sc + c);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(it.sh + c);
    const f32x4 fl = *reinterpret_cast<const f32x4*>(it.fl + c);
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int oy0 = ys * 4;
    constexpr int R = 3 * S + 3;
    const AT* inb = in + (long long)n * H * W * in_ld + c;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int iy = oy0 * S - 1 + r;
      if (iy < 0 || iy >= H) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * S - 1 + kx;
        if (ix < 0 || ix >= W) continue;
        const f32x4 v = tf4(ld4(inb + ((long long)iy * W + ix) * in_ld), sc, sh, fl);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ky = r - j * S;
          if (ky >= 0 && ky < 3) acc[j] += v * wv[ky * 3 + kx];
        }
      }
    }
    AT* ob = out + (long long)n * OH * OW * out_ld + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int oy = oy0 + j;
      if (oy < OH) {
        st4(ob + ((long long)oy * OW + ox) * out_ld, acc[j]);
        s1 += acc[j];
        s2 += acc[j] * acc[j];
      }
    }
  }
  if (stats) {
    const int lane = threadIdx.x & 63;         // lanes l, l + C4, ... share the channel group: butterfly first
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float a = s1[q], b = s2[q];
      for (int off = C4; off < 64; off <<= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
      if (lane < C4) {
        atomicAdd(&red[2 * (cstat + q)], a);
        atomicAdd(&red[2 * (cstat + q) + 1], b);
      }
    }
    __syncthreads();
    double* st = stats + (blockIdx.x % JN_NREP) * rep_stride;
    for (int i = threadIdx.x; i < 2 * C; i += 256) atomicAdd(&st[i], (double)red[i]);
  }
}

// LDS-tiled variant (the one used whenever C % 16 == 0): the workgroup stages the ACTIVATED input tile
// (halo included) of CB channels once, so silu(bn(z)) is evaluated ~1.3x per input element instead of once per
// tap (4.5x - 6.75x in the strip kernel above, which made the layer ALU-bound); taps then come from LDS.
// Tile = 8 x 16 output pixels; thread = (channel quad, column, row group), sliding down its rows.
constexpr int DW_TW = 16;

template <int S, int CB, int DW_TH, typename AT>
__global__ __launch_bounds__(256) void dw3x3_lds_kernel(
    const AT* __restrict__ in, int in_ld, ChanTab it, const float* __restrict__ w, AT* __restrict__ out,
    int out_ld, int C, int H, int W, int OH, int OW, int tiles_x, int tiles_y, double* __restrict__ stats,
    long long rep_stride, const int* __restrict__ skip_flag, int skip_when, int nrep) {
  if (skip_flag && *skip_flag >= skip_when) return;
  constexpr int Q = CB / 4;                          // channel quads per workgroup
  constexpr int IH = S * (DW_TH - 1) + 3, IW = S * (DW_TW - 1) + 3;
  constexpr int PS = S == 1 ? CB : CB + JN_DW_S2_PAD;           // pixel stride in LDS: stride-1 taps read 1 KB contiguous per wave
  constexpr int GROUPS = 256 / (DW_TW * Q), RPG = DW_TH / GROUPS;   // row groups, output rows per thread
  static_assert(GROUPS >= 1 && RPG >= 1 && RPG * GROUPS == DW_TH, "tile / thread mapping");
  extern __shared__ __attribute__((aligned(16))) float sm[];        // [IH*IW][PS] then red[2*CB]
  float* red = sm + IH * IW * PS;
  const int tid = threadIdx.x;
  const int ncb = C / CB;
  const int cb = blockIdx.x % ncb;
  const int tx = (blockIdx.x / ncb) % tiles_x, ty = blockIdx.x / (ncb * tiles_x);
  const int n = blockIdx.y;
  const int q = tid % Q;
  const int c = cb * CB + 4 * q;
  const int oy0 = ty * DW_TH, ox0 = tx * DW_TW;
  const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
  // the taps are requested first, with the tile: fetched after the staging they were one more global round trip on every
  // workgroup's critical path
  f32x4 wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4*>(w + t * C + c);
  {
    const AT* inb = in + (long long)n * H * W * in_ld + c;
    // all loads of the tile are issued back to back into registers, then transformed and stored (a load -> transform ->
    // store loop with a run-time trip count waited for one global round trip per iteration); 256 % Q == 0: the quad
    // of a thread never changes
    constexpr int NT = (IH * IW * Q + 255) / 256;
    f32x4 rv[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int i = tid + 256 * j, p = i / Q, r = p / IW, cx = p - r * IW;
      const int iy = iy0 + r, ix = ix0 + cx;
      rv[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < IH * IW * Q && iy >= 0 && iy < H && ix >= 0 && ix < W) rv[j] = ld4(inb + ((long long)iy * W + ix) * in_ld);
    }
    f32x4 sc, sh, fl;
    if (it.dsrc) {
      // deferred entries: CB threads derive (scale, shift) of the workgroup's channels from the batch sums while the tile
      // loads are in flight; the statistics slots are free until the epilogue
      if (tid < CB) {
        float a, b, f;
        tab_entry(it, cb * CB + tid, a, b, f);
        red[tid] = a; red[CB + tid] = b; red[2 * CB + tid] = f;
      }
      __syncthreads();
      sc = *reinterpret_cast<const f32x4*>(red + 4 * q); sh = *reinterpret_cast<const f32x4*>(red + CB + 4 * q);
      fl = *reinterpret_cast<const f32x4*>(red + 2 * CB + 4 * q);
    } else {
      sc = *reinterpret_cast<const f32x4*>(it.sc + c); sh = *reinterpret_cast<const f32x4*>(it.sh + c);
      fl = *reinterpret_cast<const f32x4*>(it.fl + c);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int i = tid + 256 * j, p = i / Q, r = p / IW, cx = p - r * IW;
      const int iy = iy0 + r, ix = ix0 + cx;
      if (i < IH * IW * Q) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = tf4(rv[j], sc, sh, fl);
        *reinterpret_cast<f32x4*>(sm + p * PS + 4 * q) = v;
      }
    }
  }
  __syncthreads();
  const int x = (tid / Q) % DW_TW, grp = tid / (Q * DW_TW);
  const int j0 = grp * RPG;                           // first output row (within the tile) of this thread
  f32x4 acc[RPG];
#pragma unroll
  for (int j = 0; j < RPG; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int R = S * (RPG - 1) + 3;
  const float* sp = sm + ((j0 * S) * IW + x * S) * PS + 4 * q;
#pragma unroll
  for (int r = 0; r < R; ++r) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(sp + (r * IW + kx) * PS);
#pragma unroll
      for (int j = 0; j < RPG; ++j) {
        const int ky = r - j * S;
        if (ky >= 0 && ky < 3) acc[j] += v * wv[ky * 3 + kx];
      }
    }
  }
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  const int ox = ox0 + x;
  AT* ob = out + (long long)n * OH * OW * out_ld + c;
#pragma unroll
  for (int j = 0; j < RPG; ++j) {
    const int oy = oy0 + j0 + j;
    if (oy < OH && ox < OW) {
      st4(ob + ((long long)oy * OW + ox) * out_ld, acc[j]);
      s1 += acc[j];
      s2 += acc[j] * acc[j];
    }
  }
  if (stats) {
    // lanes l, l + Q, ... of a 16-lane DPP row share the channel quad: rotate-and-add inside the row (v_add_dpp
    // row_ror), then each of the 16 rows of the workgroup stores its totals in its own LDS slot (no LDS atomics)
    const int lane = tid & 63;
    float* slot = red + ((tid >> 6) * 4 + (lane >> 4)) * 2 * CB;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float a = s1[k], b = s2[k];
      if (Q <= 4) { a += dpp_mov<0x124>(a); b += dpp_mov<0x124>(b); }     // row_ror:4
      a += dpp_mov<0x128>(a); b += dpp_mov<0x128>(b);                     // row_ror:8
      if ((lane & 15) < Q) {
        slot[2 * (4 * q + k)] = a;
        slot[2 * (4 * q + k) + 1] = b;
      }
    }
    __syncthreads();
    double* st = stats + ((blockIdx.x + 3 * blockIdx.y) % nrep) * rep_stride + 2 * cb * CB;
    if (tid < 2 * CB) {
      float v = 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) v += red[r * 2 * CB + tid];
      atomicAdd(&st[tid], (double)v);
    }
  }
}

template <int S, int CB, int DW_TH, typename AT>
static void launch_dw_lds(const ConvArgs& a, hipStream_t s) {
  constexpr int IH = S * (DW_TH - 1) + 3, IW = S * (DW_TW - 1) + 3;
  const int tiles_x = (a.OW + DW_TW - 1) / DW_TW, tiles_y = (a.OH + DW_TH - 1) / DW_TH;
  const size_t smem = ((size_t)IH * IW * (S == 1 ? CB : CB + JN_DW_S2_PAD) + 16 * 2 * CB) * sizeof(float);
  dim3 grid((unsigned)(tiles_x * tiles_y * (a.cin / CB)), (unsigned)a.N);
  hipLaunchKernelGGL((dw3x3_lds_kernel<S, CB, DW_TH, AT>), grid, dim3(256), smem, s, (const AT*)a.in, a.in_ld, a.itab, a.w,
                     (AT*)a.out, a.out_ld, a.cin, a.H, a.W, a.OH, a.OW, tiles_x, tiles_y, a.stats, a.stats_rep_stride,
                     a.skip_flag, a.skip_when, a.stats_nrep > 0 ? a.stats_nrep : JN_NREP);
}

// ------------------------------------------------------------------------------------
// Eval-mode DWConv in ONE kernel: depthwise 3x3 (stride S) -> BN + SiLU (fixed affine from the running statistics)
// -> pointwise 1x1.  The depthwise output never goes to HBM: per 16-channel chunk the workgroup stages the activated
// input tile (+halo) in LDS, forms the depthwise outputs of its TH x 16 pixels, activates them into the MFMA operand
// tile and accumulates W_pw[:, chunk] . X.  (Train mode cannot do this: the BN between the two convs needs the
// statistics of the whole batch first.)  WM = waves along the pixels: 4 -> 128 pixels (S = 1, TH = 8), 2 -> 64 pixels
// (S = 2, TH = 4; the wave pairs split the channel tiles).
// ------------------------------------------------------------------------------------
template <int S, int WM, int CT, typename AT>
__global__ __launch_bounds__(256) void dwpw_eval_kernel(
    const AT* __restrict__ in, int in_ld, ChanTab it, const float* __restrict__ w_dw, ChanTab mt,
    const float* __restrict__ w_pw, AT* __restrict__ out, int out_ld, int C, int Nc, int H, int W, int OH, int OW,
    int tiles_x, int tiles_y, const int* __restrict__ skip_flag, int skip_when, const AT* __restrict__ res, int res_ld,
    ChanTab rtab, ChanTab ptab) {
  if (skip_flag && *skip_flag >= skip_when) return;
  constexpr int CB = 16, Q = 4, TW = 16, TH = 2 * WM, BM = TH * TW;
  constexpr int IH = S * (TH - 1) + 3, IW = S * (TW - 1) + 3;
  constexpr int PS = S == 1 ? CB : CB + 4;            // input tile pixel stride (see dw3x3_lds_kernel)
  constexpr int LDX = CB + 4;                         // MFMA operand rows: 16-B aligned, banks spread
  constexpr int RPG = TH / 4;                         // output rows per thread in the depthwise phase
  constexpr int CTW = CT * WM / 4;                    // channel tiles per wave
  constexpr int NT = (IH * IW * Q + 255) / 256, NW = (16 * CT * Q + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* Ti = sm;                                     // [IH*IW][PS]  activated input chunk
  float* Xs = Ti + IH * IW * PS;                      // [BM][LDX]    activated depthwise outputs
  float* Ws = Xs + BM * LDX;                          // [16*CT][LDX] pointwise weight chunk
  // per-channel constants of ALL chunks, staged once: input table (3 rows), depthwise taps (9), dconv BN table (2) —
  // read from global memory at the top of every chunk they were a dependent L2 round trip (two per chunk)
  float* Cst = Ws + 16 * CT * LDX;                    // [14][C]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < C; i += 256) {
    Cst[i] = it.sc[i]; Cst[C + i] = it.sh[i]; Cst[2 * C + i] = it.fl[i];
#pragma unroll
    for (int t = 0; t < 9; ++t) Cst[(3 + t) * C + i] = w_dw[t * C + i];
    Cst[12 * C + i] = mt.sc[i]; Cst[13 * C + i] = mt.sh[i];
  }
  const int lm = lane & 15, g = lane >> 4;
  const int wm = wave % WM, wn = wave / WM;
  const int tile = blockIdx.x % (tiles_x * tiles_y), n = blockIdx.x / (tiles_x * tiles_y);
  const int oy0 = (tile / tiles_x) * TH, ox0 = (tile % tiles_x) * TW;
  const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
  const int q = tid & 3;
  const AT* inb = in + (long long)n * H * W * in_ld;
  f32x4 acc[2][CTW];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int c = 0; c < CTW; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 rt[NT], rw[NW];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int i = tid + 256 * j, p = i >> 2, r = p / IW, cx = p - r * IW;
      const int iy = iy0 + r, ix = ix0 + cx;
      rt[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < IH * IW * Q && iy >= 0 && iy < H && ix >= 0 && ix < W) rt[j] = ld4(inb + ((long long)iy * W + ix) * in_ld + k0 + 4 * q);
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int i = tid + 256 * j, r = i >> 2;
      rw[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < 16 * CT * Q && r < Nc) rw[j] = *reinterpret_cast<const f32x4*>(w_pw + (long long)r * C + k0 + 4 * q);
    }
  };
  fetch(0);
  __syncthreads();                                             // Cst is in place
  const int x = (tid >> 2) & 15, j0 = (tid >> 6) * RPG;       // depthwise phase: column, first row of this thread
  for (int k0 = 0; k0 < C; k0 += CB) {
    const int c = k0 + 4 * q;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(Cst + c), sh = *reinterpret_cast<const f32x4*>(Cst + C + c),
                fl = *reinterpret_cast<const f32x4*>(Cst + 2 * C + c);
    if (k0) __syncthreads();                          // the previous chunk's MFMAs have read Xs / Ws
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int i = tid + 256 * j, p = i >> 2, r = p / IW, cx = p - r * IW;
      const int iy = iy0 + r, ix = ix0 + cx;
      if (i < IH * IW * Q) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = tf4(rt[j], sc, sh, fl);
        *reinterpret_cast<f32x4*>(Ti + p * PS + 4 * q) = v;
      }
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int i = tid + 256 * j;
      if (i < 16 * CT * Q) *reinterpret_cast<f32x4*>(Ws + (i >> 2) * LDX + 4 * q) = rw[j];
    }
    __syncthreads();
    if (k0 + CB < C) fetch(k0 + CB);
    {   // depthwise 3x3 on the chunk, then the dconv layer's BN + SiLU -> MFMA operand tile
      f32x4 wv[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4*>(Cst + (3 + t) * C + c);
      const f32x4 msc = *reinterpret_cast<const f32x4*>(Cst + 12 * C + c), msh = *reinterpret_cast<const f32x4*>(Cst + 13 * C + c);
      f32x4 d[RPG];
#pragma unroll
      for (int j = 0; j < RPG; ++j) d[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      constexpr int R = S * (RPG - 1) + 3;
      const float* sp = Ti + ((j0 * S) * IW + x * S) * PS + 4 * q;
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(sp + (r * IW + kx) * PS);
#pragma unroll
          for (int j = 0; j < RPG; ++j) {
            const int ky = r - j * S;
            if (ky >= 0 && ky < 3) d[j] += v * wv[ky * 3 + kx];
          }
        }
#pragma unroll
      for (int j = 0; j < RPG; ++j) {
        f32x4 a;
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = silu(fmaf(d[j][k], msc[k], msh[k]));
        *reinterpret_cast<f32x4*>(Xs + ((j0 + j) * TW + x) * LDX + 4 * q) = a;
      }
    }
    __syncthreads();
    {
      const f32x4 xb0 = *reinterpret_cast<const f32x4*>(Xs + (wm * 32 + lm) * LDX + 4 * g);
      const f32x4 xb1 = *reinterpret_cast<const f32x4*>(Xs + (wm * 32 + 16 + lm) * LDX + 4 * g);
#pragma unroll
      for (int cc = 0; cc < CTW; ++cc) {
        const f32x4 wa = *reinterpret_cast<const f32x4*>(Ws + ((wn * CTW + cc) * 16 + lm) * LDX + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc[0][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j], xb0[j], acc[0][cc], 0, 0, 0);
          acc[1][cc] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j], xb1[j], acc[1][cc], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int pix = wm * 32 + 16 * p + lm;
    const int oy = oy0 + pix / TW, ox = ox0 + pix % TW;
    if (oy >= OH || ox >= OW) continue;
    AT* op = out + (((long long)n * OH + oy) * OW + ox) * out_ld;
#pragma unroll
    for (int cc = 0; cc < CTW; ++cc) {
      const int nn = (wn * CTW + cc) * 16 + 4 * g;
      if (nn >= Nc) continue;
      f32x4 v = acc[p][cc];
      if (res) {
        // bottleneck shortcut folded in (eval): out = silu(bn(z)) + T(res), a materialised activation
        const f32x4 psc = *reinterpret_cast<const f32x4*>(ptab.sc + nn), psh = *reinterpret_cast<const f32x4*>(ptab.sh + nn);
        const f32x4 rv = tf4(ld4(res + (((long long)n * OH + oy) * OW + ox) * res_ld + nn), *reinterpret_cast<const f32x4*>(rtab.sc + nn),
                             *reinterpret_cast<const f32x4*>(rtab.sh + nn), *reinterpret_cast<const f32x4*>(rtab.fl + nn));
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = silu(fmaf(v[k], psc[k], psh[k])) + rv[k];
      }
      st4(op + nn, v);
    }
  }
}

template <int S, int WM, int CT, typename AT>
static void launch_dwpw_t(const DwPwArgs& a, hipStream_t s) {
  constexpr int TH = 2 * WM, IH = S * (TH - 1) + 3, IW = S * 15 + 3, PS = S == 1 ? 16 : 20;
  const int tiles_x = (a.OW + 15) / 16, tiles_y = (a.OH + TH - 1) / TH;
  const size_t smem = ((size_t)IH * IW * PS + (size_t)TH * 16 * 20 + (size_t)16 * CT * 20 + (size_t)14 * a.C) * sizeof(float);
  hipLaunchKernelGGL((dwpw_eval_kernel<S, WM, CT, AT>), dim3(tiles_x * tiles_y * a.N), dim3(256), smem, s, (const AT*)a.in,
                     a.in_ld, a.itab, a.w_dw, a.mtab, a.w_pw, (AT*)a.out, a.out_ld, a.C, a.cout, a.H, a.W, a.OH, a.OW, tiles_x,
                     tiles_y, a.skip_flag, a.skip_when, (const AT*)a.res, a.res_ld, a.rtab, a.ptab);
}

bool dwpw_supported(int C, int cout, int stride) {
  // measured at B = 64: a win up to 64 depthwise channels (<= 4 chunks); at 128 channels (14x14 maps) the serial
  // chunk loop of the few workgroups is slower than the two separate kernels
  if (C % 16 || cout % 16 || cout > 128 || C > 64) return false;
  if (stride == 2 && (cout / 16) % 2) return false;         // the wave pairs split the channel tiles
  const int ct = cout / 16;
  return ct == 1 || ct == 2 || ct == 4 || ct == 8;
}

int launch_dwpw(const DwPwArgs& a, hipStream_t s) {
  const int ct = a.cout / 16;
#define JN_DP(S_, WM_, CT_) { if (a.dtype == JN_BF16) launch_dwpw_t<S_, WM_, CT_, bf16_t>(a, s); else launch_dwpw_t<S_, WM_, CT_, float>(a, s); return 0; }
  if (a.stride == 1) {
    if (ct == 1) JN_DP(1, 4, 1) if (ct == 2) JN_DP(1, 4, 2) if (ct == 4) JN_DP(1, 4, 4) if (ct == 8) JN_DP(1, 4, 8)
  } else {
    if (ct == 2) JN_DP(2, 2, 2) if (ct == 4) JN_DP(2, 2, 4) if (ct == 8) JN_DP(2, 2, 8)
  }
#undef JN_DP
  return -1;
}

int launch_dw(const ConvArgs& a, hipStream_t s) {
  if (a.cin % 16 == 0 && a.N <= 65535) {
    const bool bf = a.in_dtype == JN_BF16;
    if (a.stride == 1) {
      if (a.cin % 32 == 0) { if (bf) launch_dw_lds<1, 32, 8, bf16_t>(a, s); else launch_dw_lds<1, 32, 8, float>(a, s); }
      else { if (bf) launch_dw_lds<1, 16, 8, bf16_t>(a, s); else launch_dw_lds<1, 16, 8, float>(a, s); }
    } else {
      // stride 2: 4-row tiles halve the LDS halo tile (9 x 33 pixels): twice the workgroups per CU
      // 32-channel blocks where the layer has them (round 3: a pixel of the halo tile is then a whole 128-byte line instead
      // of half of one, and the launch has half the workgroups): 112 -> 56 layer 45.3 -> 39.8 us, 56 -> 28 25.4 -> 21.9 us;
      if (!bf && a.cin % 32 == 0) launch_dw_lds<2, 32, 4, float>(a, s);
      else if (bf) launch_dw_lds<2, 16, 4, bf16_t>(a, s); else launch_dw_lds<2, 16, 4, float>(a, s);
    }
    return 0;
  }
  const int YS = (a.OH + 3) / 4;
  const long long total = (long long)a.N * YS * a.OW * (a.cin / 4);
  const unsigned blocks = (unsigned)((total + 255) / 256);
  const size_t smem = a.stats ? (size_t)2 * a.cin * sizeof(float) : 0;
#define JN_DW(S_, T_)                                                                                              \
  hipLaunchKernelGGL((dw3x3_kernel<S_, T_>), dim3(blocks), dim3(256), smem, s, (const T_*)a.in, a.in_ld, a.itab, a.w, \
                     (T_*)a.out, a.out_ld, a.cin, a.H, a.W, a.OH, a.OW, a.N, a.stats, a.stats_rep_stride, a.skip_flag, \
                     a.skip_when)
  if (a.in_dtype == JN_BF16) { if (a.stride == 1) JN_DW(1, bf16_t); else JN_DW(2, bf16_t); }
  else { if (a.stride == 1) JN_DW(1, float); else JN_DW(2, float); }
#undef JN_DW
  return 0;
}

// ------------------------------------------------------------------------------------
// pointwise 1x1 conv: z[m][n] = sum_k T(x[m][k]) * w[n][k];  optional bias + act epilogue for the
// BN-free layers (embed_fpn.0, head predictors).  D = W * X^T per 16x16 tile: channel on the
// row, pixel on the lane -> each lane owns 4 consecutive channels of one pixel (dwordx4 store).
// ------------------------------------------------------------------------------------

// WT: `w` is stored [K][Nc] (the forward weight of the layer whose data-gradient is computed) and is
// read transposed; accumulate: out += result (gradient buffers with several contributors).
// WM = waves along the pixel dimension: 4 -> 128 pixels x 16*CT channels per workgroup, every wave all
// channel tiles; 2 -> 64 pixels, the two wave pairs split the channel tiles (more workgroups for the
// 14x14 / 28x28 layers).  K chunks are software-pipelined: chunk i+1 is fetched into registers while
// the MFMAs of chunk i run.
// BF = false: fp32 LDS tiles, v_mfma_f32_16x16x4_f32 (exact fp32, k-permuted ds_read_b128 fragments).
// BF = true : bf16 LDS tiles, v_mfma_f32_16x16x32_bf16 (8 consecutive k per lane = one ds_read_b128),
//             fp32 accumulation; weights are converted from the fp32 master copy while staging.
template <int CT, int PW_KC, bool WT, int WM, typename IT, typename OT, bool BF>
__global__ __launch_bounds__(256) void pw_mfma_kernel(
    const IT* __restrict__ x, int x_ld, ChanTab it, const float* __restrict__ w, const float* __restrict__ bias,
    OT* __restrict__ out, int out_ld, long long M, int K, int Nc, int act, int accumulate,
    double* __restrict__ stats, long long rep_stride, const int* __restrict__ skip_flag, int skip_when,
    long long x_slot, long long out_slot, long long tab_slot, int nrep) {
  if (skip_flag && *skip_flag >= skip_when) return;
  x += blockIdx.z * x_slot; out += blockIdx.z * out_slot;        // step-batched launches (gradients)
  it.sc += blockIdx.z * tab_slot; it.sh += blockIdx.z * tab_slot; it.fl += blockIdx.z * tab_slot;
  using LT = typename std::conditional<BF, bf16_t, float>::type;   // LDS element type
  constexpr int PW_LD = PW_KC + (BF ? 8 : 4);   // row stride in elements: 16-B aligned rows, banks spread
  constexpr int BM = 32 * WM;             // pixels per workgroup
  constexpr int CTW = CT * WM / 4;        // channel tiles per wave
  constexpr int NX = BM * (PW_KC / 4) / 256, NW = (16 * CT * (PW_KC / 4) + 255) / 256;
  constexpr int KSTEP = BF ? 32 : 16;     // k consumed per fragment read
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  LT* Xs = reinterpret_cast<LT*>(smem_raw);     // [BM][PW_LD]
  LT* Ws = Xs + BM * PW_LD;                      // [16*CT][PW_LD]
  float* red = reinterpret_cast<float*>(Ws + 16 * CT * PW_LD);   // [WM][16*CT][2]: one slot set per pixel-wave
  // (scale, shift, flag) of the K input channels in LDS: the staging below used to read them from global memory per
  // staged quad, a dependent L2 round trip inside every chunk's critical path
  float* Tb = red + WM * 32 * CT;                                 // [3][K4]
  const int K4 = (K + 3) & ~3;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int lm = lane & 15, g = lane >> 4;
  const long long m0 = (long long)blockIdx.x * BM;
  const int n0 = blockIdx.y * (16 * CT);

  f32x4 acc[2][CTW];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int c = 0; c < CTW; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  f32x4 xr[NX], wr[NW];
  // chunk width rounded up to the fragment step (K = 16 layers on the bf16 path are zero-padded to 32)
  auto chunk_q4 = [&](int k0) { const int kc = (K - k0 < PW_KC) ? (K - k0) : PW_KC; return ((kc + KSTEP - 1) / KSTEP * KSTEP) >> 2; };
  auto fetch = [&](int k0) {
    const int q4 = chunk_q4(k0);
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int i = tid + 256 * j, r = i / q4, q = i - r * q4;
      xr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < BM * q4 && m0 + r < M && k0 + 4 * q < K) xr[j] = ld4(x + (m0 + r) * x_ld + k0 + 4 * q);
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int i = tid + 256 * j, r = i / q4, q = i - r * q4;
      wr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (i < 16 * CT * q4 && n0 + r < Nc && k0 + 4 * q < K) {
        if (WT) {
          const float* wp = w + (long long)(k0 + 4 * q) * Nc + n0 + r;
          wr[j] = f32x4{wp[0], wp[Nc], wp[2 * Nc], wp[3 * Nc]};
        } else {
          wr[j] = *reinterpret_cast<const f32x4*>(w + (long long)(n0 + r) * K + k0 + 4 * q);
        }
      }
    }
  };
  auto stage = [&](int k0) {
    const int q4 = chunk_q4(k0);
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int i = tid + 256 * j, r = i / q4, q = i - r * q4;
      if (i < BM * q4) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int kk = k0 + 4 * q;
        if (m0 + r < M && kk < K)
          v = tf4(xr[j], *reinterpret_cast<const f32x4*>(Tb + kk), *reinterpret_cast<const f32x4*>(Tb + K4 + kk),
                  *reinterpret_cast<const f32x4*>(Tb + 2 * K4 + kk));
        st4(Xs + r * PW_LD + 4 * q, v);
      }
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int i = tid + 256 * j, r = i / q4, q = i - r * q4;
      if (i < 16 * CT * q4) st4(Ws + r * PW_LD + 4 * q, wr[j]);
    }
  };

  fetch(0);
  tab_to_lds(Tb, K4, K, it, tid, 256);    // after the first chunk's loads are in flight (deferred entries read the batch sums)
  __syncthreads();                        // Tb is in place
  for (int k0 = 0; k0 < K; k0 += PW_KC) {
    const int kc = chunk_q4(k0) << 2;     // multiple of KSTEP
    if (k0) __syncthreads();
    stage(k0);
    __syncthreads();
    if (k0 + PW_KC < K) fetch(k0 + PW_KC);
    if constexpr (BF) {
      const LT* xrow0 = Xs + (wm * 32 + lm) * PW_LD + 8 * g;
      const LT* xrow1 = xrow0 + 16 * PW_LD;
      const LT* wrow = Ws + (wn * CTW * 16 + lm) * PW_LD + 8 * g;
      for (int kk = 0; kk < kc; kk += 32) {
        const bf16x8 xb0 = *reinterpret_cast<const bf16x8*>(xrow0 + kk);
        const bf16x8 xb1 = *reinterpret_cast<const bf16x8*>(xrow1 + kk);
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
          const bf16x8 wa = *reinterpret_cast<const bf16x8*>(wrow + c * 16 * PW_LD + kk);
          acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb0, acc[0][c], 0, 0, 0);
          acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb1, acc[1][c], 0, 0, 0);
        }
      }
    } else {
      const LT* xrow0 = Xs + (wm * 32 + lm) * PW_LD + 4 * g;
      const LT* xrow1 = xrow0 + 16 * PW_LD;
      const LT* wrow = Ws + (wn * CTW * 16 + lm) * PW_LD + 4 * g;
      for (int kk = 0; kk < kc; kk += 16) {
        const f32x4 xb0 = *reinterpret_cast<const f32x4*>(xrow0 + kk);
        const f32x4 xb1 = *reinterpret_cast<const f32x4*>(xrow1 + kk);
        f32x4 wa[CTW];
#pragma unroll
        for (int c = 0; c < CTW; ++c) wa[c] = *reinterpret_cast<const f32x4*>(wrow + c * 16 * PW_LD + kk);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int c = 0; c < CTW; ++c) {
            acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][j], xb0[j], acc[0][c], 0, 0, 0);
            acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][j], xb1[j], acc[1][c], 0, 0, 0);
          }
        }
      }
    }
  }
  f32x4 s1[CTW], s2[CTW];
#pragma unroll
  for (int c = 0; c < CTW; ++c) { s1[c] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[c] = s1[c]; }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const long long m = m0 + wm * 32 + p * 16 + lm;
    if (m >= M) continue;
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
      const int n = n0 + (wn * CTW + c) * 16 + 4 * g;
      if (n >= Nc) continue;
      f32x4 v = acc[p][c];
      if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
      if (act != ACT_NONE) v = f32x4{act_apply(v.x, act), act_apply(v.y, act), act_apply(v.z, act), act_apply(v.w, act)};
      if (accumulate) v += ld4(out + m * out_ld + n);
      st4(out + m * out_ld + n, v);
      s1[c] += v;
      s2[c] += v * v;
    }
  }
  if (stats) {
#ifndef JN_DBG_NO_LDS_STATS
    wave_stats_to_lds<CTW>(s1, s2, red + wm * 32 * CT + 2 * (wn * CTW * 16), lane, Nc - n0 - wn * CTW * 16);
#endif
    __syncthreads();
#ifndef JN_DBG_NO_GLOBAL_STATS
    if (tid < 32 * CT && n0 + (tid >> 1) < Nc) {
      float v = 0.0f;
#pragma unroll
      for (int q = 0; q < WM; ++q) v += red[q * 32 * CT + tid];
      atomicAdd(&stats[(blockIdx.x % nrep) * rep_stride + 2 * n0 + tid], (double)v);
    }
#endif
  }
}

// Narrow 1x1 layers (K = 16 / 32 / 64 input channels, fp32, train or eval forward): the whole K fits one LDS chunk, so
// the generic kernel above has nothing to pipeline inside a workgroup — its 3000-12000 workgroups all load, then all
// compute, then all store.  This variant is persistent over pixel tiles: the weight tile is staged ONCE, the next
// tile's pixels are fetched into registers while the MFMAs and stores of the current one run, the per-thread table
// quad sits in registers and the BatchNorm sums stay in registers across tiles (one set of fp64 atomics per workgroup).
// LDS row padding (floats): rows of K + 8 floats are 32 x odd bytes for K = 16 / 32 / 64, the stride at which the
// ds_read_b128 fragment reads of a 16-lane group cover the 64 banks exactly once (MI355X_MICROARCH.md, LDS)
#ifndef PWN_PAD
#define PWN_PAD 8
#endif
template <int CT, int KC>
__global__ __launch_bounds__(256) void pw_narrow_kernel(
    const float* __restrict__ x, int x_ld, ChanTab it, const float* __restrict__ w, float* __restrict__ out, int out_ld,
    long long M, int Nc, double* __restrict__ stats, long long rep_stride, const int* __restrict__ skip_flag,
    int skip_when, int nrep) {
  if (skip_flag && *skip_flag >= skip_when) return;
  // waves along the pixel dimension: 2 (64-pixel tiles, wave pairs split the channel tiles) or, for a single channel
  // tile, 4 (128-pixel tiles)
  constexpr int WMW = (CT % 2 == 0) ? 2 : 4;
  constexpr int LD = KC + PWN_PAD, BM = 32 * WMW, CTW = CT * WMW / 4, Q4 = KC / 4, NX = BM * Q4 / 256, NW = (16 * CT * Q4 + 255) / 256;
  static_assert(CTW >= 1 && NX >= 1, "pw_narrow tile mapping");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* Xs = reinterpret_cast<float*>(smem_raw);     // [BM][LD]
  float* Ws = Xs + BM * LD;                            // [16*CT][LD]
  float* red = Ws + 16 * CT * LD;                      // [WMW][16*CT][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WMW, wn = wave / WMW;
  const int lm = lane & 15, g = lane >> 4;
  const int n0 = blockIdx.y * (16 * CT);
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int i = tid + 256 * j, r = i / Q4, q = i - r * Q4;
    if (i < 16 * CT * Q4) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n0 + r < Nc) v = *reinterpret_cast<const f32x4*>(w + (long long)(n0 + r) * KC + 4 * q);
      *reinterpret_cast<f32x4*>(Ws + r * LD + 4 * q) = v;
    }
  }
  const int q = tid % Q4, r0 = tid / Q4;               // this thread's channel quad and first row of a tile (fixed)
  f32x4 t_sc, t_sh, t_fl;
  if (it.dsrc) {                                       // deferred entries (jn_tab.h): KC threads derive them, via the idle statistics slots
    static_assert(WMW * 32 * CT >= 3 * KC, "statistics slots hold the table");
    tab_to_lds(red, KC, KC, it, tid, 256);
    __syncthreads();
    t_sc = *reinterpret_cast<const f32x4*>(red + 4 * q); t_sh = *reinterpret_cast<const f32x4*>(red + KC + 4 * q);
    t_fl = *reinterpret_cast<const f32x4*>(red + 2 * KC + 4 * q);
  } else {
    t_sc = *reinterpret_cast<const f32x4*>(it.sc + 4 * q); t_sh = *reinterpret_cast<const f32x4*>(it.sh + 4 * q);
    t_fl = *reinterpret_cast<const f32x4*>(it.fl + 4 * q);
  }
  const long long n_tiles = (M + BM - 1) / BM;
  f32x4 xr[NX];
  auto fetch = [&](long long m0) {
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const long long m = m0 + r0 + (256 / Q4) * j;
      xr[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (m < M) xr[j] = *reinterpret_cast<const f32x4*>(x + m * x_ld + 4 * q);
    }
  };
  f32x4 s1[CTW], s2[CTW];
#pragma unroll
  for (int c = 0; c < CTW; ++c) { s1[c] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[c] = s1[c]; }
  long long tile = blockIdx.x;
  if (tile < n_tiles) fetch(tile * BM);
  for (; tile < n_tiles; tile += gridDim.x) {
    const long long m0 = tile * BM;
    __syncthreads();                     // the previous tile's fragment reads are done (and Ws is in place)
#pragma unroll
    for (int j = 0; j < NX; ++j) {
      const int r = r0 + (256 / Q4) * j;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m0 + r < M) v = tf4(xr[j], t_sc, t_sh, t_fl);
      *reinterpret_cast<f32x4*>(Xs + r * LD + 4 * q) = v;
    }
    __syncthreads();
    if (tile + gridDim.x < n_tiles) fetch((tile + gridDim.x) * BM);
    f32x4 acc[2][CTW];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int c = 0; c < CTW; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* xrow0 = Xs + (wm * 32 + lm) * LD + 4 * g;
    const float* xrow1 = xrow0 + 16 * LD;
    const float* wrow = Ws + (wn * CTW * 16 + lm) * LD + 4 * g;
#pragma unroll
    for (int kk = 0; kk < KC; kk += 16) {
      const f32x4 xb0 = *reinterpret_cast<const f32x4*>(xrow0 + kk);
      const f32x4 xb1 = *reinterpret_cast<const f32x4*>(xrow1 + kk);
      f32x4 wa[CTW];
#pragma unroll
      for (int c = 0; c < CTW; ++c) wa[c] = *reinterpret_cast<const f32x4*>(wrow + c * 16 * LD + kk);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
          acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][j], xb0[j], acc[0][c], 0, 0, 0);
          acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][j], xb1[j], acc[1][c], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const long long m = m0 + wm * 32 + p * 16 + lm;
      if (m >= M) continue;
#pragma unroll
      for (int c = 0; c < CTW; ++c) {
        const int n = n0 + (wn * CTW + c) * 16 + 4 * g;
        if (n >= Nc) continue;
        *reinterpret_cast<f32x4*>(out + m * out_ld + n) = acc[p][c];
        s1[c] += acc[p][c];
        s2[c] += acc[p][c] * acc[p][c];
      }
    }
  }
  if (stats) {
    __syncthreads();
    wave_stats_to_lds<CTW>(s1, s2, red + wm * 32 * CT + 2 * (wn * CTW * 16), lane, Nc - n0 - wn * CTW * 16);
    __syncthreads();
    if (tid < 32 * CT && n0 + (tid >> 1) < Nc) {
      float v = 0.0f;
#pragma unroll
      for (int qq = 0; qq < WMW; ++qq) v += red[qq * 32 * CT + tid];
      atomicAdd(&stats[(blockIdx.x % nrep) * rep_stride + 2 * n0 + tid], (double)v);
    }
  }
}

template <int CT, int KC>
static void launch_pw_narrow_t(const ConvArgs& a, long long M, hipStream_t s) {
  constexpr int BM = (CT % 2 == 0) ? 64 : 128;
  const long long n_tiles = (M + BM - 1) / BM;
  // persistent workgroups: whole resident rounds (what the occupancy API says fits, x 256 CUs).  Round 1 had swept a fixed
  // 1536; the kernel's registers have changed since (84 VGPRs at <2, 32>: five workgroups per CU, so 1536 was a round of 1280 and a
  // tail of 256) — forward 1.842 -> 1.822 ms per pass (tools/ab_fwd.sh, 1024 / 1280 / 1536 / 2048 / resident swept)
  static int cap = 0;
  if (!cap) {
    int per_cu = 0;
    const size_t sm = (size_t)(BM + 16 * CT) * (KC + PWN_PAD) * sizeof(float) + (BM / 32) * 32 * CT * sizeof(float);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&pw_narrow_kernel<CT, KC>), 256, sm) != hipSuccess || per_cu < 1) per_cu = 4;
    cap = per_cu * 256;
  }
  const unsigned gx = (unsigned)std::min<long long>(n_tiles, cap);
  dim3 grid(gx, (unsigned)((a.cout + 16 * CT - 1) / (16 * CT)));
  const size_t smem = (size_t)(BM + 16 * CT) * (KC + PWN_PAD) * sizeof(float) + (BM / 32) * 32 * CT * sizeof(float);
  hipLaunchKernelGGL((pw_narrow_kernel<CT, KC>), grid, dim3(256), smem, s, (const float*)a.in, a.in_ld, a.itab, a.w,
                     (float*)a.out, a.out_ld, M, a.cout, a.stats, a.stats_rep_stride, a.skip_flag, a.skip_when,
                     a.stats_nrep > 0 ? a.stats_nrep : JN_NREP);
}

// true when the persistent narrow kernel took the launch
static bool launch_pw_narrow(const ConvArgs& a, hipStream_t s) {
  static const bool off = std::getenv("JN_NO_PW_NARROW") != nullptr;
  if (off || a.bf16_mfma || a.in_dtype != JN_F32 || a.out_dtype != JN_F32 || a.w_transposed || a.accumulate || a.bias ||
      a.act != ACT_NONE || a.n_slots > 1)
    return false;
  const long long M = (long long)a.N * a.H * a.W;
  const int nt = (a.cout + 15) / 16;
  // small maps have too few tiles to walk (JN_PWN_MIN_M: test hook, read per launch so that a test can force the kernel
  // onto small maps)
  const char* mm = std::getenv("JN_PWN_MIN_M");
  if (M < (mm ? std::atoll(mm) : 65536)) return false;
#define JN_PWN(CT_, KC_) if (nt == CT_ && a.cin == KC_) { launch_pw_narrow_t<CT_, KC_>(a, M, s); return true; }
  JN_PWN(1, 16) JN_PWN(2, 16) JN_PWN(2, 32) JN_PWN(4, 32) JN_PWN(4, 64) JN_PWN(8, 64)
#undef JN_PWN
  return false;
}

template <int CT, int KC, bool WT, int WM, typename IT, typename OT, bool BF>
static void launch_pw_kc(const ConvArgs& a, long long M, hipStream_t s) {
  constexpr int BM = 32 * WM;
  dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((a.cout + 16 * CT - 1) / (16 * CT)), a.n_slots > 1 ? a.n_slots : 1);
  const size_t smem = (size_t)(BM + 16 * CT) * (KC + (BF ? 8 : 4)) * (BF ? 2 : 4) + WM * 32 * CT * sizeof(float) +
                      (size_t)3 * ((a.cin + 3) & ~3) * sizeof(float);
  hipLaunchKernelGGL((pw_mfma_kernel<CT, KC, WT, WM, IT, OT, BF>), grid, dim3(256), smem, s, (const IT*)a.in, a.in_ld,
                     a.itab, a.w, a.bias, (OT*)a.out, a.out_ld, M, a.cin, a.cout, a.act, a.accumulate, a.stats,
                     a.stats_rep_stride, a.skip_flag, a.skip_when, a.in_slot_stride, a.out_slot_stride, a.tab_slot_stride,
                     a.stats_nrep > 0 ? a.stats_nrep : JN_NREP);
}

template <int CT, bool WT, int WM, typename IT, typename OT, bool BF>
static void launch_pw_cfg(const ConvArgs& a, long long M, hipStream_t s) {
  // K chunk: 64 wherever the tiles fit in 64 KB of LDS (fewer load -> LDS -> MFMA round trips); layers with
  // K <= 32 / K <= 16 get narrower LDS rows: half / a quarter of the LDS per workgroup = more workgroups per CU
  constexpr int KC = BF ? 64 : ((CT > 4) ? 32 : 64);
  if constexpr (!BF && CT <= 4) {
    if constexpr (WM >= 2) {
      if (a.cin <= 16) { launch_pw_kc<CT, 16, WT, WM, IT, OT, BF>(a, M, s); return; }
    }
    if (a.cin <= 32) { launch_pw_kc<CT, 32, WT, WM, IT, OT, BF>(a, M, s); return; }
  }
  launch_pw_kc<CT, KC, WT, WM, IT, OT, BF>(a, M, s);
}

template <int CT, typename IT, typename OT, bool BF>
static void launch_pw_ct(const ConvArgs& a, long long M, hipStream_t s) {
  // 64-pixel workgroups (two wave pairs split the channel tiles) whenever CT is even: half the LDS per
  // workgroup -> more workgroups per CU, which is what hides the load -> LDS -> MFMA latency of these short kernels
  // (measured: 56x56 layers 50 -> 42 us, 112x112 60 -> 55 us against 128-pixel workgroups)
  constexpr bool can_split = (CT % 2 == 0), can_split4 = (CT % 4 == 0);
  const long long Mtot = M * (a.n_slots > 1 ? a.n_slots : 1);
  const bool tiny_m = Mtot <= 16384;                  // 14x14 maps at 64 patches
  if (a.w_transposed) {              // data gradients (step-batched, large): 128-pixel workgroups unless the map is small
    // (JN_PW_WT_SMALL_M: test hook, read per launch — 0 forces the large-map variant onto the small parity shapes)
    const char* sm_env = std::getenv("JN_PW_WT_SMALL_M");
    if (can_split && Mtot <= (sm_env ? std::atoll(sm_env) : 65536)) launch_pw_cfg<CT, true, can_split ? 2 : 4, IT, OT, BF>(a, M, s);
    else launch_pw_cfg<CT, true, 4, IT, OT, BF>(a, M, s);
  } else {
    // 14x14 maps with K <= 128: 32-pixel workgroups (the four waves split the channel tiles) quadruple the
    // workgroup count; with a longer K the re-staged weight tile costs more than the parallelism gains
    if (can_split4 && tiny_m && a.cin <= 128) launch_pw_cfg<CT, false, can_split4 ? 1 : 4, IT, OT, BF>(a, M, s);
    else launch_pw_cfg<CT, false, can_split ? 2 : 4, IT, OT, BF>(a, M, s);
  }
}

template <typename IT, typename OT, bool BF>
static void launch_pw_types(const ConvArgs& a, hipStream_t s) {
  const long long M = (long long)a.N * a.H * a.W;
  const int nt = (a.cout + 15) / 16;
  if (nt == 1) launch_pw_ct<1, IT, OT, BF>(a, M, s);
  else if (nt == 2) launch_pw_ct<2, IT, OT, BF>(a, M, s);
  else if (nt == 3) launch_pw_ct<3, IT, OT, BF>(a, M, s);
  else if (nt == 4 || (nt % 8 != 0 && nt % 4 == 0)) launch_pw_ct<4, IT, OT, BF>(a, M, s);
  else launch_pw_ct<8, IT, OT, BF>(a, M, s);
}

// Type combinations in use: fp32 mode (f32, f32, fp32 MFMA); bf16 mode: activations (bf16 -> bf16),
// embed_fpn.0 (bf16 -> f32) and gradients (f32 -> f32), all on the bf16 MFMA.
// Pixel-stationary kernel (kernels_pwxs.hip) for the forward 1x1 layers with K, N >= 64 on at most JN_XS_MAX_M pixels per
// launch (the 56x56 / 28x28 / 14x14 maps of the headline batch: 10 - 30 % under the weight-stationary kernel on every one
// of them, profiles/r03_pwxsbench.txt); JN_NO_PW_XS=1 keeps them on the weight-stationary kernel.
static int launch_pw_small_maps(const ConvArgs& a, hipStream_t s) {
  static const bool off = std::getenv("JN_NO_PW_XS") != nullptr;
  static const long long max_m = std::getenv("JN_XS_MAX_M") ? std::atoll(std::getenv("JN_XS_MAX_M")) : 262144;
  if (off || (long long)a.N * a.H * a.W > max_m) return -1;
  if (pw_x1_supported(a)) return launch_pw_x1(a, s);      // bf16 inference mode
  if (!pw_xs_supported(a)) return -1;
  // fp32 operands as three bf16 planes on the bf16 matrix pipe where that is the faster kernel (a.w_x3 is null under
  // JN_NO_PW_X3=1: fp32 pipe)
  if (pw_x3_preferred(a) && launch_pw_x3(a, 0, s) == 0) return 0;
  return launch_pw_xs(a, 0, s);
}

// The two layers whose output feeds a nearest x2 upsample (lateral_conv0 256 -> 128, reduce_conv1 128 -> 64) on the
// small-map routes of the headline batch: their kernel can store the upsampled copy itself (a.up_out).  Mirrors the
// route choice of launch_pw_small_maps; any other shape / route keeps the separate upsample launch.
bool pw_fused_upsample_supported(const ConvArgs& a) {
  static const bool off = std::getenv("JN_NO_PW_XS") != nullptr || std::getenv("JN_NO_FUSED_UPSAMPLE") != nullptr;
  static const long long max_m = std::getenv("JN_XS_MAX_M") ? std::atoll(std::getenv("JN_XS_MAX_M")) : 262144;
  if (off || (long long)a.N * a.H * a.W > max_m || !pw_xs_supported(a)) return false;
  if (a.cin == 128 && a.cout == 64) return pw_x3_preferred(a);          // pw_x3_kernel<128, 1, 2, 4>
  return a.cin == 256 && a.cout == 128;                                  // pw_xs_kernel<256, 2, 2, 4>
}

int launch_pw(const ConvArgs& a, hipStream_t s) {
  if (launch_pw_small_maps(a, s) == 0) return 0;           // small maps, forward: pixel-stationary kernel (kernels_pwxs.hip)
  if (a.up_out) return -1;                                 // (the caller asked pw_fused_upsample_supported first)
  if (launch_pw_wide(a, s) == 0) return 0;                // K, N >= 64: weight-stationary kernel (kernels_pwres.hip)
  if (launch_pw_narrow(a, s)) return 0;
  if (!a.bf16_mfma) {
    if (a.in_dtype == JN_F32 && a.out_dtype == JN_F32) { launch_pw_types<float, float, false>(a, s); return 0; }
    return -1;
  }
  if (a.in_dtype == JN_BF16 && a.out_dtype == JN_BF16) launch_pw_types<bf16_t, bf16_t, true>(a, s);
  else if (a.in_dtype == JN_BF16 && a.out_dtype == JN_F32) launch_pw_types<bf16_t, float, true>(a, s);
  else if (a.in_dtype == JN_F32 && a.out_dtype == JN_F32) launch_pw_types<float, float, true>(a, s);
  else return -1;
  return 0;
}

// ------------------------------------------------------------------------------------
// SPP: slices 1..3 of `cat` = maxpool 5 / 9 / 13 (stride 1, -inf padding) of the ACTIVATION of
// slice 0 (raw z + table).  mp9 = mp5(mp5), mp13 = mp5(mp9); each mp5 is separable.
// ------------------------------------------------------------------------------------
template <typename AT>
__global__ __launch_bounds__(256) void spp_kernel(AT* __restrict__ cat, int ld, int h, int H, int W, int cb,
                                                  ChanTab it, const int* __restrict__ skip_flag, int skip_when) {
  if (skip_flag && *skip_flag >= skip_when) return;
  extern __shared__ float sp[];
  const int HW = H * W;
  float* A = sp;
  float* Bf = sp + HW * cb;
  __shared__ float tb[3 * 64];                         // (scale, shift, flag) of the workgroup's channels (cb <= 64)
  const int n = blockIdx.y, c0 = blockIdx.x * cb;
  AT* base = cat + (long long)n * HW * ld + c0;
  const int tid = threadIdx.x;
  if (tid < cb) {
    float a, b, f;
    tab_entry(it, c0 + tid, a, b, f);
    tb[tid] = a; tb[64 + tid] = b; tb[128 + tid] = f;
  }
  __syncthreads();
  for (int e = tid; e < HW * cb; e += 256) {
    const int c = e % cb;
    const float z = ld1(base + (long long)(e / cb) * ld + c);
    A[e] = tb[128 + c] != 0.0f ? silu(fmaf(z, tb[c], tb[64 + c])) : z;
  }
  __syncthreads();
  for (int stage = 1; stage <= 3; ++stage) {
    for (int e = tid; e < HW * cb; e += 256) {
      const int p = e / cb, c = e % cb, y = p / W, xx = p - y * W;
      float m = -INFINITY;
      for (int d = -2; d <= 2; ++d) {
        const int x2 = xx + d;
        if (x2 >= 0 && x2 < W) m = fmaxf(m, A[(y * W + x2) * cb + c]);
      }
      Bf[e] = m;
    }
    __syncthreads();
    for (int e = tid; e < HW * cb; e += 256) {
      const int p = e / cb, c = e % cb, y = p / W, xx = p - y * W;
      float m = -INFINITY;
      for (int d = -2; d <= 2; ++d) {
        const int y2 = y + d;
        if (y2 >= 0 && y2 < H) m = fmaxf(m, Bf[(y2 * W + xx) * cb + c]);
      }
      A[e] = m;
      st1(base + (long long)p * ld + stage * h + c, m);
    }
    __syncthreads();
  }
}

// fp32, 16-byte form (round 4): a thread owns channel QUADS — 16-byte loads / stores (64 B contiguous per pixel instead of 32
// in 4-byte pieces), ds_read_b128 windows, the (y, x) of a thread's elements computed once (they are the same in all seven
// phases; the scalar kernel spends two integer divisions per element and phase).  16 channels per workgroup.
template <int CBQ>
__global__ __launch_bounds__(256) void spp4_kernel(float* __restrict__ cat, int ld, int h, int H, int W, ChanTab it,
                                                   const int* __restrict__ skip_flag, int skip_when) {
  if (skip_flag && *skip_flag >= skip_when) return;
  extern __shared__ __attribute__((aligned(16))) float sp[];
  constexpr int NI = 8;                                // elements per thread at most: H * W * CBQ <= 2048
  const int HW = H * W, NEL = HW * CBQ;
  f32x4* A = reinterpret_cast<f32x4*>(sp);
  f32x4* Bf = A + NEL;
  __shared__ float tb[3 * 4 * CBQ];
  const int n = blockIdx.y, c0 = blockIdx.x * 4 * CBQ;
  float* base = cat + (long long)n * HW * ld + c0;
  const int tid = threadIdx.x;
  if (tid < 4 * CBQ) {
    float a, b, f;
    tab_entry(it, c0 + tid, a, b, f);
    tb[tid] = a; tb[4 * CBQ + tid] = b; tb[8 * CBQ + tid] = f;
  }
  int py[NI], px[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int e = tid + 256 * i, p = e / CBQ;
    py[i] = p / W; px[i] = p - py[i] * W;
  }
  __syncthreads();
  const int q = tid % CBQ;                             // 256 % CBQ == 0: the channel quad is the same for all of a thread's elements
  const f32x4 sc = *reinterpret_cast<const f32x4*>(tb + 4 * q), sh = *reinterpret_cast<const f32x4*>(tb + 4 * CBQ + 4 * q),
              fl = *reinterpret_cast<const f32x4*>(tb + 8 * CBQ + 4 * q);
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int e = tid + 256 * i;
    if (e < NEL) A[e] = tf4_tab(*reinterpret_cast<const f32x4*>(base + (long long)(e / CBQ) * ld + 4 * q), sc, sh, fl);
  }
  __syncthreads();
  const f32x4 ninf = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
  auto max4 = [](f32x4 a, f32x4 b) { return f32x4{fmaxf(a[0], b[0]), fmaxf(a[1], b[1]), fmaxf(a[2], b[2]), fmaxf(a[3], b[3])}; };
  for (int stage = 1; stage <= 3; ++stage) {
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int e = tid + 256 * i;
      if (e < NEL) {
        f32x4 m = ninf;
#pragma unroll
        for (int d = -2; d <= 2; ++d) {
          const int x2 = px[i] + d;
          if (x2 >= 0 && x2 < W) m = max4(m, A[e + d * CBQ]);
        }
        Bf[e] = m;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int e = tid + 256 * i;
      if (e < NEL) {
        f32x4 m = ninf;
#pragma unroll
        for (int d = -2; d <= 2; ++d) {
          const int y2 = py[i] + d;
          if (y2 >= 0 && y2 < H) m = max4(m, Bf[e + d * W * CBQ]);
        }
        A[e] = m;
        *reinterpret_cast<f32x4*>(base + (long long)(e / CBQ) * ld + stage * h + 4 * q) = m;
      }
    }
    __syncthreads();
  }
}

int launch_spp(void* cat, int dtype, int ld, int h, int H, int W, int N, ChanTab it, const int* skip_flag,
               int skip_when, hipStream_t s) {
  static const bool no_v4 = std::getenv("JN_NO_SPP_V4") != nullptr;
  if (dtype == JN_F32 && !no_v4 && h % 16 == 0 && ld % 4 == 0 && H * W * 4 <= 2048) {
    hipLaunchKernelGGL(spp4_kernel<4>, dim3(h / 16, N), dim3(256), (size_t)H * W * 4 * 2 * sizeof(f32x4), s, (float*)cat, ld, h, H, W, it,
                       skip_flag, skip_when);
    return 0;
  }
  // channels per block: 2 * HW * cb floats of LDS; 8 -> 1024 workgroups at B = 64 (measured 40.9 us with 32 or 16, 31.9 us with 8)
  const int cb0 = 8;
  int cb = cb0 > 64 ? 64 : cb0;
  while (cb > 4 && (size_t)H * W * cb * 2 * sizeof(float) > 48 * 1024) cb >>= 1;
  dim3 grid(h / cb, N);
  const size_t smem = (size_t)H * W * cb * 2 * sizeof(float);
  if (dtype == JN_BF16)
    hipLaunchKernelGGL(spp_kernel<bf16_t>, grid, dim3(256), smem, s, (bf16_t*)cat, ld, h, H, W, cb, it, skip_flag, skip_when);
  else
    hipLaunchKernelGGL(spp_kernel<float>, grid, dim3(256), smem, s, (float*)cat, ld, h, H, W, cb, it, skip_flag, skip_when);
  return 0;
}

// nearest x2 upsample of raw values: out[n][y][x][c] = in[n][y/2][x/2][c]
template <typename AT>
__global__ __launch_bounds__(256) void upsample_kernel(const AT* __restrict__ in, int in_ld,
                                                       AT* __restrict__ out, int out_ld, int C, int H, int W,
                                                       long long total, const int* __restrict__ skip_flag,
                                                       int skip_when) {
  if (skip_flag && *skip_flag >= skip_when) return;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int C4 = C >> 2, OW = 2 * W, OH = 2 * H;
  const int c4 = (int)(idx % C4);
  const int ox = (int)((idx / C4) % OW);
  const int oy = (int)((idx / ((long long)C4 * OW)) % OH);
  const long long n = idx / ((long long)C4 * OW * OH);
  const f32x4 v = ld4(in + ((n * H + (oy >> 1)) * W + (ox >> 1)) * in_ld + 4 * c4);
  st4(out + ((n * OH + oy) * OW + ox) * out_ld + 4 * c4, v);
}

int launch_upsample(const void* in, int in_ld, void* out, int out_ld, int dtype, int C, int H, int W, int N,
                    const int* skip_flag, int skip_when, hipStream_t s) {
  const long long total = (long long)N * 4 * H * W * (C / 4);
  const dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == JN_BF16)
    hipLaunchKernelGGL(upsample_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)in, in_ld, (bf16_t*)out, out_ld, C, H,
                       W, total, skip_flag, skip_when);
  else
    hipLaunchKernelGGL(upsample_kernel<float>, grid, dim3(256), 0, s, (const float*)in, in_ld, (float*)out, out_ld, C, H,
                       W, total, skip_flag, skip_when);
  return 0;
}

// Bottleneck shortcut: out = T_res(res) + T_z(z), materialised (table flag 0 on the output).  Both tables go through LDS
// (deferred entries are derived from the batch sums once per workgroup), the workgroups stride over the elements.
template <typename AT>
__global__ __launch_bounds__(256) void addact_kernel(const AT* __restrict__ z, int z_ld, ChanTab zt,
                                                     const AT* __restrict__ res, int res_ld, ChanTab rt,
                                                     AT* __restrict__ out, int out_ld, int C, long long M,
                                                     const int* __restrict__ skip_flag, int skip_when) {
  if (skip_flag && *skip_flag >= skip_when) return;
  extern __shared__ __attribute__((aligned(16))) float tb[];      // [2][3][C]
  tab_to_lds(tb, C, C, zt, threadIdx.x, 256);          // (two plain loops: selecting between the two by-value structs at run time
  tab_to_lds(tb + 3 * C, C, C, rt, threadIdx.x, 256);  //  would send both to scratch memory)
  __syncthreads();
  const int C4 = C >> 2;
  const long long total = M * C4;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int c = (int)(idx % C4) * 4;
    const long long m = idx / C4;
    const f32x4 a = tf4(ld4(z + m * z_ld + c), *reinterpret_cast<const f32x4*>(tb + c),
                        *reinterpret_cast<const f32x4*>(tb + C + c), *reinterpret_cast<const f32x4*>(tb + 2 * C + c));
    const f32x4 r = tf4(ld4(res + m * res_ld + c), *reinterpret_cast<const f32x4*>(tb + 3 * C + c),
                        *reinterpret_cast<const f32x4*>(tb + 4 * C + c), *reinterpret_cast<const f32x4*>(tb + 5 * C + c));
    st4(out + m * out_ld + c, a + r);
  }
}

int launch_addact(const void* z, int z_ld, ChanTab zt, const void* res, int res_ld, ChanTab rt, void* out, int out_ld,
                  int dtype, int C, long long M, const int* skip_flag, int skip_when, hipStream_t s) {
  const long long total = M * (C / 4);
  const dim3 grid((unsigned)std::min<long long>((total + 255) / 256, 4096));      // (512 ... 4096 workgroups: no difference, measured)
  const size_t smem = (size_t)6 * C * sizeof(float);
  if (dtype == JN_BF16)
    hipLaunchKernelGGL(addact_kernel<bf16_t>, grid, dim3(256), smem, s, (const bf16_t*)z, z_ld, zt, (const bf16_t*)res, res_ld,
                       rt, (bf16_t*)out, out_ld, C, M, skip_flag, skip_when);
  else
    hipLaunchKernelGGL(addact_kernel<float>, grid, dim3(256), smem, s, (const float*)z, z_ld, zt, (const float*)res, res_ld,
                       rt, (float*)out, out_ld, C, M, skip_flag, skip_when);
  return 0;
}

// Train-mode BatchNorm2d (eps 1e-3, momentum 0.03): batch mean / biased variance
// from the fp64 sums -> (scale, shift) of the layer's output channels (and of an alias slice
// that holds an upsampled copy), saved (mean, invstd) for the backward pass, running stats.
__global__ void bn_finalize_kernel(const double* __restrict__ stats, long long rep_stride, double count,
                                   const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ run_mean,
                                   float* __restrict__ run_var, float* __restrict__ save, ChanTab t0, ChanTab t1,
                                   int C, float eps, float momentum, const int* __restrict__ skip_flag,
                                   int skip_when) {
  // 8 lanes per channel, 4 replicas each (independent loads), then three DPP-free xor steps inside the octet.  Every
  // load of the kernel — the skip flag, the sums, the affine — is requested before the first one is waited for: the
  // kernel is three dependent round trips long otherwise (flag -> sums -> gamma / beta), and it runs 26 times per pass
  const int sub = threadIdx.x & 7;
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
  const int cc = c < C ? c : C - 1;
  const int flag = skip_flag ? *skip_flag : skip_when - 1;
  const float gm = gamma[cc], bt = beta[cc];
  double v1[JN_NREP / 8], v2[JN_NREP / 8];
#pragma unroll
  for (int k = 0; k < JN_NREP / 8; ++k) { v1[k] = stats[(sub + 8 * k) * rep_stride + 2 * cc]; v2[k] = stats[(sub + 8 * k) * rep_stride + 2 * cc + 1]; }
  if (flag >= skip_when) return;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int k = 0; k < JN_NREP / 8; ++k) { s1 += v1[k]; s2 += v2[k]; }
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
  if (sub != 0 || c >= C) return;
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gm * invstd;
  const float sh = bt - (float)mean * sc;
  t0.sc[c] = sc; t0.sh[c] = sh; t0.fl[c] = 1.0f;
  if (t1.sc) { t1.sc[c] = sc; t1.sh[c] = sh; t1.fl[c] = 1.0f; }
  if (save) { save[2 * c] = (float)mean; save[2 * c + 1] = invstd; }
  if (run_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    run_mean[c] = (1.0f - momentum) * run_mean[c] + momentum * (float)mean;
    run_var[c] = (1.0f - momentum) * run_var[c] + momentum * (float)unbiased;
  }
}

int launch_bn_finalize(const double* stats, long long rep_stride, double count, const float* gamma, const float* beta, float* run_mean,
                       float* run_var, float* save, ChanTab t0, ChanTab t1, int C, float eps, float momentum,
                       const int* skip_flag, int skip_when, hipStream_t s) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((8 * C + 255) / 256), dim3(256), 0, s, stats, rep_stride, count, gamma, beta, run_mean,
                     run_var,
                     save, t0, t1, C, eps, momentum, skip_flag, skip_when);
  return 0;
}

// One launch at the end of a train-mode pass: saved (mean, invstd) and running statistics of EVERY BatchNorm layer of the
// pass, and the table of the layers whose table was deferred (ChanTab, jn_kernels.h) — the arithmetic of
// bn_finalize_kernel per channel (thread = channel).  The per-layer finalize launches that remain (layers above the
// deferral limit) then only write their table: one global round trip shorter each.
__global__ __launch_bounds__(256) void bn_finalize_all_kernel(BnAllArgs a) {
  if (a.skip_flag && *a.skip_flag >= a.skip_when) return;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.n_stat) return;
  const float hw = a.hw[i];
  if (hw <= 0.0f) return;                                  // not a layer of this pass (detection head)
  // layers above the deferral limit spread their sums over all JN_NREP replicas and had their table written by their own
  // (table-only) finalize launch; their saved statistics and running averages are formed here with everybody else's
  const int nrep = (long long)a.N * (long long)hw > a.defer_max_m ? JN_NREP : JN_NREP_DEFER;
  const double count = (double)a.N * (double)hw;
  float sc, sh, mean, invstd; double var;
  bn_from_sums(a.stats, a.rep_stride, nrep, i, count, a.params[a.goff[i]], a.params[a.boff[i]], a.eps, sc, sh, mean, invstd, var);
  const int t0 = a.t0[i], t1 = a.t1[i];
  a.tab[t0] = sc; a.tab[a.tab_channels + t0] = sh; a.tab[2 * a.tab_channels + t0] = 1.0f;
  if (t1 >= 0) { a.tab[t1] = sc; a.tab[a.tab_channels + t1] = sh; a.tab[2 * a.tab_channels + t1] = 1.0f; }
  a.save[2 * i] = mean; a.save[2 * i + 1] = invstd;
  const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
  float* rm = a.run_mean[i]; float* rv = a.run_var[i];
  *rm = (1.0f - a.momentum) * *rm + a.momentum * mean;
  *rv = (1.0f - a.momentum) * *rv + a.momentum * (float)unbiased;
}

long long jn_defer_max_m() {
  static const long long v = std::getenv("JN_DEFER_MAX_M") ? std::atoll(std::getenv("JN_DEFER_MAX_M")) : JN_DEFER_MAX_M;
  return v;
}

int launch_bn_finalize_all(const BnAllArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(bn_finalize_all_kernel, dim3((a.n_stat + 255) / 256), dim3(256), 0, s, a);
  return 0;
}

// NHWC slice (raw z + table) -> contiguous NCHW activations (boundary export for the parity API)
template <typename AT>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const AT* __restrict__ in, int in_ld, ChanTab it,
                                                           float* __restrict__ out, int C, int HW, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int p = (int)(idx % HW);
  const int c = (int)((idx / HW) % C);
  const long long n = idx / ((long long)HW * C);
  const float z = ld1(in + (n * HW + p) * in_ld + c);
  out[idx] = it.fl[c] != 0.0f ? silu(fmaf(z, it.sc[c], it.sh[c])) : z;
}

int launch_nhwc_to_nchw(const void* in, int dtype, int in_ld, ChanTab it, float* out, int C, int HW, int N,
                        hipStream_t s) {
  const long long total = (long long)N * C * HW;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == JN_BF16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)in, in_ld, it, out, C, HW, total);
  else
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, s, (const float*)in, in_ld, it, out, C, HW, total);
  return 0;
}

// ------------------------------------------------------------------------------------
// embed_fpn.3 Linear over the flattened (c, h, w) map, as split-K partial sums:
// part[n][ks][o] = sum_{k in slice ks} e[n][k] * Wt[k][o];  k = hw*C + c (weights pre-permuted).
// The KS partials + bias are summed in fixed order by the decode kernel (deterministic).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void efpn_linear_kernel(const float* __restrict__ e, const float* __restrict__ wt,
                                                          float* __restrict__ part, int K, int Co, int KS,
                                                          const int* __restrict__ skip_flag, int skip_when) {
  if (skip_flag && *skip_flag >= skip_when) return;
  extern __shared__ float red[];   // [slices][Co]
  const int n = blockIdx.y, ks = blockIdx.x;
  const int kper = (K + KS - 1) / KS;
  const int kb = ks * kper, ke = min(K, kb + kper);
  const int tid = threadIdx.x;
  const int nsl = 256 / Co;                 // thread slices inside the block
  const int o = tid % Co, sl = tid / Co;
  float acc = 0.0f;
  if (sl < nsl) {
    const float* ep = e + (long long)n * K;
    for (int k = kb + sl; k < ke; k += nsl) acc = fmaf(ep[k], wt[(long long)k * Co + o], acc);
    red[sl * Co + o] = acc;
  }
  __syncthreads();
  if (tid < Co) {
    float s = 0.0f;
    for (int i = 0; i < nsl; ++i) s += red[i * Co + tid];
    part[((long long)n * KS + ks) * Co + tid] = s;
  }
}

// MFMA form (Co % 16 == 0): workgroup = (K slice, 64 agents, 48 outputs); the weight slice is read ONCE per agent
// block instead of once per agent (the GEMV above re-read 1.8 MB of weights 64 times: 92 us -> ~15 us at B = 64).
constexpr int EL_KC = 64, EL_LDE = EL_KC + 4, EL_LDW = 48 + 4;

__global__ __launch_bounds__(256) void efpn_linear_mfma_kernel(const float* __restrict__ e, const float* __restrict__ wt,
                                                               float* __restrict__ part, int N, int K, int Co, int KS,
                                                               int kper, const int* __restrict__ skip_flag,
                                                               int skip_when) {
  if (skip_flag && *skip_flag >= skip_when) return;
  __shared__ __attribute__((aligned(16))) float Es[64 * EL_LDE];
  __shared__ __attribute__((aligned(16))) float Ws[EL_KC * EL_LDW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, g = lane >> 4;
  const int ks = blockIdx.x, n0 = blockIdx.y * 64, o0 = blockIdx.z * 48;
  const int kb = ks * kper, ke = min(K, kb + kper);
  f32x4 acc[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int k0 = kb; k0 < ke; k0 += EL_KC) {
    if (k0 != kb) __syncthreads();
    for (int i = tid; i < 64 * (EL_KC / 4); i += 256) {
      const int r = i / (EL_KC / 4), q = i % (EL_KC / 4);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int k = k0 + 4 * q;
      if (n0 + r < N && k < ke) v = *reinterpret_cast<const f32x4*>(e + (long long)(n0 + r) * K + k);   // K % 4 == 0, ke % 4 == 0
      *reinterpret_cast<f32x4*>(Es + r * EL_LDE + 4 * q) = v;
    }
    for (int i = tid; i < EL_KC * 12; i += 256) {
      const int r = i / 12, q = i % 12;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (k0 + r < ke && o0 + 4 * q < Co) v = *reinterpret_cast<const f32x4*>(wt + (long long)(k0 + r) * Co + o0 + 4 * q);
      *reinterpret_cast<f32x4*>(Ws + r * EL_LDW + 4 * q) = v;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < EL_KC; kk += 16) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(Es + (16 * wave + lm) * EL_LDE + kk + 4 * g);   // A[i = agent][k perm]
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* wr = Ws + (kk + 4 * g + j) * EL_LDW + lm;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], wr[16 * c], acc[c], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + 16 * wave + 4 * g + r, o = o0 + 16 * c + lm;
      if (n < N && o < Co) part[((long long)n * KS + ks) * Co + o] = acc[c][r];
    }
}

int launch_efpn_linear(const float* e, const float* wt, float* part, int N, int K, int Co, int KS,
                       const int* skip_flag, int skip_when, hipStream_t s) {
  if (Co % 16 == 0 && K % 4 == 0) {
    const int kper = ((K + KS - 1) / KS + 3) / 4 * 4;
    dim3 grid(KS, (N + 63) / 64, (Co + 47) / 48);
    hipLaunchKernelGGL(efpn_linear_mfma_kernel, grid, dim3(256), 0, s, e, wt, part, N, K, Co, KS, kper, skip_flag, skip_when);
    return 0;
  }
  dim3 grid(KS, N);
  const size_t smem = (size_t)(256 / Co) * Co * sizeof(float);
  hipLaunchKernelGGL(efpn_linear_kernel, grid, dim3(256), smem, s, e, wt, part, K, Co, KS, skip_flag, skip_when);
  return 0;
}

}  // namespace jnr
