// On-device detection augmentation (Trainer.init_detection, src/trainer.py:176-186): the per-patch
// chain  colour gains -> grayscale -> 3x3 Gaussian blur (reflect border) -> plasma shadow -> additive Gaussian noise ->
// 3x3 motion blur (zero border)  in ONE pass over the patches: 4 B read + 4 B written per element instead of five kornia ops (>= 40 B).
// Every patch carries its own parameters (AUG_NPARAM floats, sampled by the host mirror); an op a patch did not draw is
// the identity (gains 1, centre weight 1, delta kernel, std 0).  HBM-bound stencil: 16x64 output tile per workgroup,
// input tile + 2-pixel halo staged in LDS with the colour ops applied, the blurred + noised tile in a second LDS tile.
#include <hip/hip_runtime.h>

#include "jn_kernels.h"

namespace jnr {

constexpr int AUG_TH = 16, AUG_TW = 64;      // output tile: 64-pixel rows = one 256 B segment per wave
constexpr int AUG_XH = AUG_TH + 4, AUG_XW = AUG_TW + 4, AUG_GH = AUG_TH + 2, AUG_GW = AUG_TW + 2;

__device__ __forceinline__ unsigned aug_hash(unsigned v) {       // murmur3 fmix32: 32-bit ops only (64-bit multiplies are slow here)
  v ^= v >> 16; v *= 0x85EBCA6Bu;
  v ^= v >> 13; v *= 0xC2B2AE35u;
  return v ^ (v >> 16);
}
// standard normal from the element's global index (counter-based: tiles agree on their shared halo); Box-Muller
__device__ __forceinline__ float aug_normal(unsigned long long seed, unsigned long long idx) {
  const unsigned lo = (unsigned)idx, hi = (unsigned)(idx >> 32);
  const unsigned k0 = (unsigned)seed ^ (hi * 0x9E3779B9u), k1 = (unsigned)(seed >> 32) + 0x7F4A7C15u;
  const unsigned a = aug_hash(aug_hash(lo ^ k0) + k1), b = aug_hash(aug_hash(lo + k1) ^ k0 ^ 0x5BD1E995u);
  const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f), u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2);
}

// RandomPlasmaShadow (src/trainer.py:180-182): kornia shades the image where a plasma fractal (diamond-square) falls
// below `shade_quantity`.  A recursive generator does not fit a tiled one-pass kernel, so the fractal is restated as
// fractional-Brownian value noise with the same octave structure: lattice spacing P/2, P/4, ... (AUG_OCT octaves),
// octave o weighted roughness^o, every lattice value a counter-based hash of (patch seed, octave, iy, ix) — any pixel
// of any tile evaluates it independently.  f in [0, 1]; the per-sample min-max normalisation of kornia's map becomes a
// fixed stretch about 0.5 by the field's analytic spread (host side: params[18] = 1 / (5 sigma)).
constexpr int AUG_OCT = 7;
__device__ __forceinline__ float aug_lattice(unsigned seed, int o, int iy, int ix) {
  const unsigned h = aug_hash(aug_hash((unsigned)ix * 0x9E3779B1u ^ (unsigned)iy * 0x85EBCA77u ^ (unsigned)(o + 1) * 0xC2B2AE3Du) ^ seed);
  return (float)(h >> 8) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float aug_plasma(unsigned seed, int y, int x, int P, float rough, float stretch) {
  float f = 0.0f, wsum = 0.0f, wgt = 1.0f;
#pragma unroll
  for (int o = 0; o < AUG_OCT; ++o) {
    const float cell = (float)P / (float)(2 << o);          // lattice spacing of the octave
    const float fy = (float)y / cell, fx = (float)x / cell;
    const int iy = (int)fy, ix = (int)fx;
    const float ty = fy - (float)iy, tx = fx - (float)ix;
    const float a = aug_lattice(seed, o, iy, ix), b = aug_lattice(seed, o, iy, ix + 1);
    const float c = aug_lattice(seed, o, iy + 1, ix), d = aug_lattice(seed, o, iy + 1, ix + 1);
    const float top = a + (b - a) * tx, bot = c + (d - c) * tx;
    f += wgt * (top + (bot - top) * ty);
    wsum += wgt;
    wgt *= rough;
  }
  f = 0.5f + (f / wsum - 0.5f) * stretch;
  return fminf(fmaxf(f, 0.0f), 1.0f);
}

__device__ __forceinline__ int reflect(int i, int n) {      // kornia / F.pad "reflect": -1 -> 1, n -> n - 2
  if (i < 0) i = -i;
  if (i >= n) i = 2 * n - 2 - i;
  return i;
}

__global__ __launch_bounds__(256) void augment_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                      const float* __restrict__ params, const float* __restrict__ noise,
                                                      unsigned long long seed, int P, int tiles_x) {
  __shared__ float Xs[3][AUG_XH][AUG_XW + 1];
  __shared__ float Gs[3][AUG_GH][AUG_GW + 1];
  __shared__ float Sh[AUG_GH][AUG_GW + 1];                  // plasma-shadow multiplier of the tile + halo 1
  __shared__ float prm[AUG_NPARAM];
  const int n = blockIdx.y, tile = blockIdx.x, ty0 = (tile / tiles_x) * AUG_TH, tx0 = (tile % tiles_x) * AUG_TW;
  const int tid = threadIdx.x;
  if (tid < AUG_NPARAM) prm[tid] = params[(long long)n * AUG_NPARAM + tid];
  __syncthreads();
  const float r_gain = prm[0], b_gain = prm[1], gray = prm[2], w0 = prm[3], w1 = prm[4], nstd = prm[5];
  const long long patch0 = (long long)n * 3 * P * P;
  const float* src = in + patch0;
  float* dst = out + patch0;
  // ---- stage: colour ops on the input tile + halo 2 (reflect-indexed) ----
  for (int i = tid; i < AUG_XH * AUG_XW; i += 256) {
    const int yy = i / AUG_XW, xx = i - yy * AUG_XW;
    const int y = reflect(ty0 - 2 + yy, P), x = reflect(tx0 - 2 + xx, P);
    float r = 0.f, g = 0.f, b = 0.f;
    if (y >= 0 && y < P && x >= 0 && x < P) {              // (reflect of a coordinate beyond 2P is not needed: halo <= 2 < P)
      const int o = y * P + x, PP = P * P;                 // offsets inside one patch fit 32 bits
      r = src[o]; g = src[PP + o]; b = src[2 * PP + o];
      if (r_gain != 1.0f || b_gain != 1.0f) {             // Planckian jitter: gains on red / blue, clamp to [0, 1]
        r = fminf(fmaxf(r * r_gain, 0.0f), 1.0f); g = fminf(fmaxf(g, 0.0f), 1.0f); b = fminf(fmaxf(b * b_gain, 0.0f), 1.0f);
      }
      if (gray != 0.0f) { const float l = 0.299f * r + 0.587f * g + 0.114f * b; r = g = b = l; }
    }
    Xs[0][yy][xx] = r; Xs[1][yy][xx] = g; Xs[2][yy][xx] = b;
  }
  // ---- plasma shadow multiplier (after the blur, before the noise: the reference's op order) ----
  const float sh_int = prm[15], sh_qty = prm[16], sh_rough = prm[17], sh_stretch = prm[18];
  if (sh_int != 0.0f) {
    const unsigned pseed = aug_hash((unsigned)seed ^ (unsigned)(seed >> 32) ^ ((unsigned)n * 0x9E3779B9u + 0x7F4A7C15u));
    for (int i = tid; i < AUG_GH * AUG_GW; i += 256) {
      const int yy = i / AUG_GW, xx = i - yy * AUG_GW;
      const int y = ty0 - 1 + yy, x = tx0 - 1 + xx;
      float m = 1.0f;
      if (y >= 0 && y < P && x >= 0 && x < P && aug_plasma(pseed, y, x, P, sh_rough, sh_stretch) < sh_qty) m = 1.0f + sh_int;
      Sh[yy][xx] = m;
    }
  }
  __syncthreads();
  // ---- Gaussian 3x3 (separable weights w1 w0 w1) + noise on tile + halo 1; outside the image = 0 (motion-blur border) ----
  for (int i = tid; i < 3 * AUG_GH * AUG_GW; i += 256) {
    const int c = i / (AUG_GH * AUG_GW), rem = i - c * AUG_GH * AUG_GW, yy = rem / AUG_GW, xx = rem - yy * AUG_GW;
    const int y = ty0 - 1 + yy, x = tx0 - 1 + xx;
    float v = 0.0f;
    if (y >= 0 && y < P && x >= 0 && x < P) {
      const float (*X)[AUG_XW + 1] = Xs[c];
      if (w1 == 0.0f) {                                    // no blur drawn for this patch (uniform per workgroup)
        v = X[yy + 1][xx + 1];
      } else {
        const float top = w1 * X[yy][xx] + w0 * X[yy][xx + 1] + w1 * X[yy][xx + 2];
        const float mid = w1 * X[yy + 1][xx] + w0 * X[yy + 1][xx + 1] + w1 * X[yy + 1][xx + 2];
        const float bot = w1 * X[yy + 2][xx] + w0 * X[yy + 2][xx + 1] + w1 * X[yy + 2][xx + 2];
        v = w1 * top + w0 * mid + w1 * bot;
      }
      if (sh_int != 0.0f) v *= Sh[yy][xx];
      if (nstd != 0.0f) {
        const int e = c * P * P + y * P + x;
        v += nstd * (noise ? noise[patch0 + e] : aug_normal(seed, (unsigned long long)(patch0 + e)));
      }
    }
    Gs[c][yy][xx] = v;
  }
  __syncthreads();
  // ---- motion blur: per-patch 3x3 kernel (cross-correlation, zero border) ----
  float mk[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) mk[t] = prm[6 + t];
  const bool delta = mk[4] == 1.0f;                        // kernels are normalised: centre 1 = no motion blur drawn
  for (int i = tid; i < 3 * AUG_TH * AUG_TW; i += 256) {
    const int c = i / (AUG_TH * AUG_TW), rem = i - c * AUG_TH * AUG_TW, yy = rem / AUG_TW, xx = rem - yy * AUG_TW;
    const int y = ty0 + yy, x = tx0 + xx;
    if (y >= P || x >= P) continue;
    const float (*G)[AUG_GW + 1] = Gs[c];
    float v = 0.0f;
    if (delta) {
      v = G[yy + 1][xx + 1];
    } else {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) v += mk[ky * 3 + kx] * G[yy + ky][xx + kx];
    }
    dst[c * P * P + y * P + x] = v;
  }
}

int launch_augment(const float* in, float* out, const float* params, const float* noise, unsigned long long seed, int N, int P,
                   hipStream_t s) {
  const int tiles_x = (P + AUG_TW - 1) / AUG_TW, tiles_y = (P + AUG_TH - 1) / AUG_TH;
  hipLaunchKernelGGL(augment_kernel, dim3(tiles_x * tiles_y, N), dim3(256), 0, s, in, out, params, noise, seed, P, tiles_x);
  return 0;
}

}  // namespace jnr
