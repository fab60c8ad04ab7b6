// Convolution kernels of the YOLOX PAFPN patch encoder for gfx950 (MI355X).
// Activations: NHWC fp32.  Eval-mode BatchNorm is folded into weight + bias at load
// time; bias + SiLU (+ residual) are applied in each kernel's epilogue so every
// activation is written once and read once per consumer.
//
//   stem_kernel     Focus (space-to-depth) + dense 3x3 conv == 6x6 stride-2 conv on the
//                   image; reads the patch straight out of the big image at the agent's
//                   position (the gather is fused: SURVEY.md §8 a2 + a3), LDS halo tile.
//   dw3x3_kernel    depthwise 3x3, stride 1/2, float4 over channels, 4-row strips.
//   pw_mfma_kernel  1x1 conv as a GEMM on v_mfma_f32_16x16x4_f32 (exact fp32), X and W
//                   tiles staged in LDS, one dwordx4 store per lane per 16x16 tile.
//   spp_kernel      SPP max-pools 5/9/13 as a cascade of separable 5-pools in LDS.
//   upsample_kernel nearest x2 into a channel slice of the consumer's concat buffer.
#include <hip/hip_runtime.h>

#include "jn_kernels.h"

namespace jnr {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == ACT_SILU) return v / (1.0f + __expf(-v));
  if (act == ACT_RELU) return fmaxf(v, 0.0f);
  if (act == ACT_SIGMOID) return 1.0f / (1.0f + __expf(-v));
  return v;
}

// ------------------------------------------------------------------------------------
// stem: out[n][oy][ox][oc] = silu(b[oc] + sum_{c,dy,dx} w[c][dy][dx][oc] * img[c][2oy-2+dy][2ox-2+dx])
// ------------------------------------------------------------------------------------
constexpr int ST_TY = 16, ST_TX = 32;
constexpr int ST_IH = 2 * ST_TY + 4, ST_IW = 2 * ST_TX + 4;   // 36 x 68

__global__ __launch_bounds__(256) void stem_kernel(
    const float* __restrict__ src, const long long* __restrict__ pos, long long sample_stride,
    long long chan_stride, int row_stride, int P, const float* __restrict__ w,
    const float* __restrict__ bias, float* __restrict__ out, int out_ld, int cout, int ocg,
    const int* __restrict__ skip_flag, int skip_when) {
  if (skip_flag && *skip_flag >= skip_when) return;
  __shared__ __attribute__((aligned(16))) float tile[3][ST_IH][ST_IW];
  const int tid = threadIdx.x;
  const int n = blockIdx.z / ocg, og = blockIdx.z % ocg;
  const int OH = P / 2;
  const int oy0 = blockIdx.y * ST_TY, ox0 = blockIdx.x * ST_TX;
  const float* base = src + (long long)n * sample_stride;
  if (pos) base += pos[2 * n] * (long long)P * row_stride + pos[2 * n + 1] * (long long)P;
  for (int i = tid; i < 3 * ST_IH * ST_IW; i += 256) {
    const int c = i / (ST_IH * ST_IW), r = (i / ST_IW) % ST_IH, q = i % ST_IW;
    const int iy = 2 * oy0 - 2 + r, ix = 2 * ox0 - 2 + q;
    float v = 0.0f;
    if (iy >= 0 && iy < P && ix >= 0 && ix < P) v = base[c * chan_stride + (long long)iy * row_stride + ix];
    tile[c][r][q] = v;
  }
  __syncthreads();
  const int txp = tid & 15, ty = tid >> 4;
  float acc0[16], acc1[16];
#pragma unroll
  for (int o = 0; o < 16; ++o) acc0[o] = acc1[o] = bias[og * 16 + o];
  const float* wg = w + og * 16;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int dy = 0; dy < 6; ++dy) {
      const f32x4* rp = reinterpret_cast<const f32x4*>(&tile[c][2 * ty + dy][4 * txp]);
      const f32x4 a = rp[0], b = rp[1];
      const float in[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
      for (int dx = 0; dx < 6; ++dx) {
        const float* wp = wg + ((c * 6 + dy) * 6 + dx) * cout;
#pragma unroll
        for (int o = 0; o < 16; ++o) {
          const float wv = wp[o];
          acc0[o] = fmaf(in[dx], wv, acc0[o]);
          acc1[o] = fmaf(in[dx + 2], wv, acc1[o]);
        }
      }
    }
  }
  const int oy = oy0 + ty, ox = ox0 + 2 * txp;
  if (oy < OH) {
    float* op = out + (((long long)n * OH + oy) * OH + ox) * out_ld + og * 16;
    if (ox < OH) {
#pragma unroll
      for (int o = 0; o < 16; o += 4) {
        f32x4 v = {act_apply(acc0[o], ACT_SILU), act_apply(acc0[o + 1], ACT_SILU),
                   act_apply(acc0[o + 2], ACT_SILU), act_apply(acc0[o + 3], ACT_SILU)};
        *reinterpret_cast<f32x4*>(op + o) = v;
      }
    }
    if (ox + 1 < OH) {
#pragma unroll
      for (int o = 0; o < 16; o += 4) {
        f32x4 v = {act_apply(acc1[o], ACT_SILU), act_apply(acc1[o + 1], ACT_SILU),
                   act_apply(acc1[o + 2], ACT_SILU), act_apply(acc1[o + 3], ACT_SILU)};
        *reinterpret_cast<f32x4*>(op + out_ld + o) = v;
      }
    }
  }
}

int launch_stem(const StemArgs& a, hipStream_t s) {
  const int OH = a.P / 2;
  const int ocg = a.cout / 16;
  dim3 grid((OH + ST_TX - 1) / ST_TX, (OH + ST_TY - 1) / ST_TY, a.N * ocg);
  hipLaunchKernelGGL(stem_kernel, grid, dim3(256), 0, s, a.src, (const long long*)a.positions, a.sample_stride,
                     a.chan_stride, a.row_stride, a.P, a.w, a.bias, a.out, a.out_ld, a.cout, ocg,
                     a.skip_flag, a.skip_when);
  return 0;
}

// ------------------------------------------------------------------------------------
// depthwise 3x3 (pad 1), stride S; thread = 4 channels x 4 output rows of one column.
// ------------------------------------------------------------------------------------
template <int S>
__global__ __launch_bounds__(256) void dw3x3_kernel(
    const float* __restrict__ in, int in_ld, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, int out_ld, int C, int H, int W, int OH, int OW, int N, int act,
    const int* __restrict__ skip_flag, int skip_when) {
  if (skip_flag && *skip_flag >= skip_when) return;
  const int C4 = C >> 2;
  const int YS = (OH + 3) >> 2;
  const long long total = (long long)N * YS * OW * C4;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c4 = (int)(idx % C4);
  const int ox = (int)((idx / C4) % OW);
  const int ys = (int)((idx / ((long long)C4 * OW)) % YS);
  const int n = (int)(idx / ((long long)C4 * OW * YS));
  const int c = c4 * 4;
  f32x4 wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4*>(w + t * C + c);
  const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + c);
  f32x4 acc[4] = {bv, bv, bv, bv};
  const int oy0 = ys * 4;
  constexpr int R = 3 * S + 3;
  const float* inb = in + (long long)n * H * W * in_ld + c;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int iy = oy0 * S - 1 + r;
    if (iy < 0 || iy >= H) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox * S - 1 + kx;
      if (ix < 0 || ix >= W) continue;
      const f32x4 v = *reinterpret_cast<const f32x4*>(inb + ((long long)iy * W + ix) * in_ld);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ky = r - j * S;
        if (ky >= 0 && ky < 3) acc[j] += v * wv[ky * 3 + kx];
      }
    }
  }
  float* ob = out + (long long)n * OH * OW * out_ld + c;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int oy = oy0 + j;
    if (oy < OH) {
      f32x4 v = {act_apply(acc[j].x, act), act_apply(acc[j].y, act), act_apply(acc[j].z, act), act_apply(acc[j].w, act)};
      *reinterpret_cast<f32x4*>(ob + ((long long)oy * OW + ox) * out_ld) = v;
    }
  }
}

int launch_dw(const ConvArgs& a, hipStream_t s) {
  const int YS = (a.OH + 3) / 4;
  const long long total = (long long)a.N * YS * a.OW * (a.cin / 4);
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (a.stride == 1)
    hipLaunchKernelGGL(dw3x3_kernel<1>, dim3(blocks), dim3(256), 0, s, a.in, a.in_ld, a.w, a.bias, a.out, a.out_ld,
                       a.cin, a.H, a.W, a.OH, a.OW, a.N, a.act, a.skip_flag, a.skip_when);
  else
    hipLaunchKernelGGL(dw3x3_kernel<2>, dim3(blocks), dim3(256), 0, s, a.in, a.in_ld, a.w, a.bias, a.out, a.out_ld,
                       a.cin, a.H, a.W, a.OH, a.OW, a.N, a.act, a.skip_flag, a.skip_when);
  return 0;
}

// ------------------------------------------------------------------------------------
// pointwise 1x1 conv: out[m][n] = act(bias[n] + sum_k x[m][k] * w[n][k]) (+ res[m][n])
// D = W * X^T per 16x16 tile on v_mfma_f32_16x16x4_f32: channel n on the row, pixel m on the
// lane, so each lane owns 4 consecutive channels of one pixel (one dwordx4 store).
// ------------------------------------------------------------------------------------
constexpr int PW_BM = 128;     // pixels per block (4 waves x 2 tiles of 16)

template <int CT, int PW_KC>
__global__ __launch_bounds__(256) void pw_mfma_kernel(
    const float* __restrict__ x, int x_ld, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ out, int out_ld, const float* __restrict__ res, int res_ld,
    long long M, int K, int Nc, int act, const int* __restrict__ skip_flag, int skip_when) {
  if (skip_flag && *skip_flag >= skip_when) return;
  constexpr int PW_LD = PW_KC + 4;        // K chunk staged in LDS (+4 floats: bank spread, 16-B rows)
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Xs = smem;                       // [PW_BM][PW_LD]
  float* Ws = smem + PW_BM * PW_LD;       // [16*CT][PW_LD]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lm = lane & 15, g = lane >> 4;
  const long long m0 = (long long)blockIdx.x * PW_BM;
  const int n0 = blockIdx.y * (16 * CT);

  f32x4 acc[2][CT];
#pragma unroll
  for (int p = 0; p < 2; ++p)
#pragma unroll
    for (int c = 0; c < CT; ++c) acc[p][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int k0 = 0; k0 < K; k0 += PW_KC) {
    const int kc = (K - k0 < PW_KC) ? (K - k0) : PW_KC;   // multiple of 16
    const int q4 = kc >> 2;
    if (k0) __syncthreads();
    for (int i = tid; i < PW_BM * q4; i += 256) {
      const int r = i / q4, q = i - r * q4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (m0 + r < M) v = *reinterpret_cast<const f32x4*>(x + (m0 + r) * x_ld + k0 + 4 * q);
      *reinterpret_cast<f32x4*>(Xs + r * PW_LD + 4 * q) = v;
    }
    for (int i = tid; i < 16 * CT * q4; i += 256) {
      const int r = i / q4, q = i - r * q4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n0 + r < Nc) v = *reinterpret_cast<const f32x4*>(w + (long long)(n0 + r) * K + k0 + 4 * q);
      *reinterpret_cast<f32x4*>(Ws + r * PW_LD + 4 * q) = v;
    }
    __syncthreads();
    const float* xrow0 = Xs + (wave * 32 + lm) * PW_LD + 4 * g;
    const float* xrow1 = xrow0 + 16 * PW_LD;
    const float* wrow = Ws + lm * PW_LD + 4 * g;
    for (int kk = 0; kk < kc; kk += 16) {
      const f32x4 xb0 = *reinterpret_cast<const f32x4*>(xrow0 + kk);
      const f32x4 xb1 = *reinterpret_cast<const f32x4*>(xrow1 + kk);
      f32x4 wa[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) wa[c] = *reinterpret_cast<const f32x4*>(wrow + c * 16 * PW_LD + kk);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          acc[0][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][j], xb0[j], acc[0][c], 0, 0, 0);
          acc[1][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[c][j], xb1[j], acc[1][c], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const long long m = m0 + wave * 32 + p * 16 + lm;
    if (m >= M) continue;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const int n = n0 + c * 16 + 4 * g;
      if (n >= Nc) continue;
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
      f32x4 v = acc[p][c] + bv;
      v = f32x4{act_apply(v.x, act), act_apply(v.y, act), act_apply(v.z, act), act_apply(v.w, act)};
      if (res) v += *reinterpret_cast<const f32x4*>(res + m * res_ld + n);
      *reinterpret_cast<f32x4*>(out + m * out_ld + n) = v;
    }
  }
}

template <int CT>
static void launch_pw_ct(const ConvArgs& a, long long M, hipStream_t s) {
  constexpr int KC = (CT > 4) ? 32 : 64;
  dim3 grid((unsigned)((M + PW_BM - 1) / PW_BM), (unsigned)((a.cout + 16 * CT - 1) / (16 * CT)));
  const size_t smem = (size_t)(PW_BM + 16 * CT) * (KC + 4) * sizeof(float);
  hipLaunchKernelGGL((pw_mfma_kernel<CT, KC>), grid, dim3(256), smem, s, a.in, a.in_ld, a.w, a.bias, a.out, a.out_ld,
                     a.res, a.res_ld, M, a.cin, a.cout, a.act, a.skip_flag, a.skip_when);
}

int launch_pw(const ConvArgs& a, hipStream_t s) {
  const long long M = (long long)a.N * a.H * a.W;
  const int nt = (a.cout + 15) / 16;
  if (nt == 1) launch_pw_ct<1>(a, M, s);
  else if (nt == 2) launch_pw_ct<2>(a, M, s);
  else if (nt == 3) launch_pw_ct<3>(a, M, s);
  else if (nt == 4 || (nt % 8 != 0 && nt % 4 == 0)) launch_pw_ct<4>(a, M, s);
  else launch_pw_ct<8>(a, M, s);
  return 0;
}

// ------------------------------------------------------------------------------------
// SPP: slices 1..3 of `cat` = maxpool 5 / 9 / 13 (stride 1, -inf padding) of slice 0.
// mp9 = mp5(mp5), mp13 = mp5(mp9); each mp5 is separable.  Block = (image, 32 channels).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spp_kernel(float* __restrict__ cat, int ld, int h, int H, int W, int cb,
                                                  const int* __restrict__ skip_flag, int skip_when) {
  if (skip_flag && *skip_flag >= skip_when) return;
  extern __shared__ float sp[];
  const int HW = H * W;
  float* A = sp;
  float* Bf = sp + HW * cb;
  const int n = blockIdx.y, c0 = blockIdx.x * cb;
  float* base = cat + (long long)n * HW * ld + c0;
  const int tid = threadIdx.x;
  for (int e = tid; e < HW * cb; e += 256) A[e] = base[(long long)(e / cb) * ld + (e % cb)];
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
      base[(long long)p * ld + stage * h + c] = m;
    }
    __syncthreads();
  }
}

int launch_spp(float* cat, int ld, int h, int H, int W, int N, const int* skip_flag, int skip_when, hipStream_t s) {
  int cb = 32;                                   // channels per block: keep 2 * HW * cb floats under 48 KB
  while (cb > 4 && (size_t)H * W * cb * 2 * sizeof(float) > 48 * 1024) cb >>= 1;
  dim3 grid(h / cb, N);
  const size_t smem = (size_t)H * W * cb * 2 * sizeof(float);
  hipLaunchKernelGGL(spp_kernel, grid, dim3(256), smem, s, cat, ld, h, H, W, cb, skip_flag, skip_when);
  return 0;
}

// nearest x2 upsample: out[n][y][x][c] = in[n][y/2][x/2][c]
__global__ __launch_bounds__(256) void upsample_kernel(const float* __restrict__ in, int in_ld,
                                                       float* __restrict__ out, int out_ld, int C, int H, int W,
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
  const f32x4 v = *reinterpret_cast<const f32x4*>(in + ((n * H + (oy >> 1)) * W + (ox >> 1)) * in_ld + 4 * c4);
  *reinterpret_cast<f32x4*>(out + ((n * OH + oy) * OW + ox) * out_ld + 4 * c4) = v;
}

int launch_upsample(const float* in, int in_ld, float* out, int out_ld, int C, int H, int W, int N,
                    const int* skip_flag, int skip_when, hipStream_t s) {
  const long long total = (long long)N * 4 * H * W * (C / 4);
  hipLaunchKernelGGL(upsample_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, in_ld, out, out_ld, C, H,
                     W, total, skip_flag, skip_when);
  return 0;
}

// NHWC slice -> contiguous NCHW (boundary export for the parity API; not on the hot loop)
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ in, int in_ld,
                                                           float* __restrict__ out, int C, int HW, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int p = (int)(idx % HW);
  const int c = (int)((idx / HW) % C);
  const long long n = idx / ((long long)HW * C);
  out[idx] = in[(n * HW + p) * in_ld + c];
}

int launch_nhwc_to_nchw(const float* in, int in_ld, float* out, int C, int HW, int N, hipStream_t s) {
  const long long total = (long long)N * C * HW;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, in_ld, out, C, HW,
                     total);
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

int launch_efpn_linear(const float* e, const float* wt, float* part, int N, int K, int Co, int KS,
                       const int* skip_flag, int skip_when, hipStream_t s) {
  dim3 grid(KS, N);
  const size_t smem = (size_t)(256 / Co) * Co * sizeof(float);
  hipLaunchKernelGGL(efpn_linear_kernel, grid, dim3(256), smem, s, e, wt, part, K, Co, KS, skip_flag, skip_when);
  return 0;
}

}  // namespace jnr
