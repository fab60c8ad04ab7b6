// "Normalize on read" helpers shared by the forward kernels: a = flag ? silu(z * scale + shift) : z, and the
// per-channel (scale, shift, flag) entries — from the table, or (deferred entries, see ChanTab in jn_kernels.h)
// straight from the batch sums of the producing layer.
#pragma once
#include <hip/hip_runtime.h>

#include "jn_kernels.h"
#include "jn_types.h"

namespace jnr {

__device__ __forceinline__ float silu_tab(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

__device__ __forceinline__ f32x4 tf4_tab(f32x4 z, f32x4 sc, f32x4 sh, f32x4 fl) {
  f32x4 r;
  r.x = fl.x != 0.0f ? silu_tab(fmaf(z.x, sc.x, sh.x)) : z.x;
  r.y = fl.y != 0.0f ? silu_tab(fmaf(z.y, sc.y, sh.y)) : z.y;
  r.z = fl.z != 0.0f ? silu_tab(fmaf(z.z, sc.z, sh.z)) : z.z;
  r.w = fl.w != 0.0f ? silu_tab(fmaf(z.w, sc.w, sh.w)) : z.w;
  return r;
}

// Train-mode BatchNorm (eps 1e-3) of one channel from its fp64 (sum, sumsq) replicas: the arithmetic of
// bn_finalize_kernel / bn_finalize_all_kernel, so that a consumer that derives the entry itself and the table written
// at the end of the pass agree bit for bit.
__device__ __forceinline__ void bn_from_sums(const double* __restrict__ st, long long rep_stride, int nrep, int idx, double count,
                                             float gamma, float beta, float eps, float& sc, float& sh, float& mean_f,
                                             float& invstd_f, double& var_out) {
  double s1 = 0.0, s2 = 0.0;
  for (int r = 0; r < nrep; ++r) {
    const double2 v = *reinterpret_cast<const double2*>(st + r * rep_stride + 2 * idx);
    s1 += v.x; s2 += v.y;
  }
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  sc = gamma * invstd;
  sh = beta - (float)mean * sc;
  mean_f = (float)mean; invstd_f = invstd; var_out = var;
}

// Entry of channel c of the view.  With deferred fields present, the four descriptor values are fetched together (one
// round trip), then either the table entry or the batch sums + BatchNorm affine (a second one).
__device__ __forceinline__ bool tab_run(const ChanTab& t, const ChanTab::Run& r, int c, float& sc, float& sh, float& fl) {
  if (c < r.c0 || c >= r.c1 || r.stat0 < 0) return false;
  const int d = c - r.c0;
  const float gamma = t.dparams[r.g0 + d], beta = t.dparams[r.b0 + d];
  float m, is; double var;
  bn_from_sums(t.dstats, t.drep_stride, JN_NREP_DEFER, r.stat0 + d, (double)t.dN * (double)r.hw, gamma, beta, 1e-3f, sc, sh, m, is, var);
  fl = 1.0f;
  return true;
}

__device__ __forceinline__ void tab_entry(const ChanTab& t, int c, float& sc, float& sh, float& fl) {
  if (t.nseg > 0) {
    if (tab_run(t, t.r0, c, sc, sh, fl)) return;
    if (t.nseg > 1 && tab_run(t, t.r1, c, sc, sh, fl)) return;
    if (t.nseg > 2 && tab_run(t, t.r2, c, sc, sh, fl)) return;
    if (t.nseg > 3 && tab_run(t, t.r3, c, sc, sh, fl)) return;
    sc = t.sc[c]; sh = t.sh[c]; fl = t.fl[c];
    return;
  }
  if (t.dsrc) {
    const int src = t.dsrc[c], go = t.dgoff[c], bo = t.dboff[c];
    const float hw = t.dhw[c];
    if (src >= 0 && (long long)t.dN * (long long)hw <= t.dmax) {
      const float gamma = t.dparams[go], beta = t.dparams[bo];
      float m, is; double var;
      bn_from_sums(t.dstats, t.drep_stride, JN_NREP_DEFER, src, (double)t.dN * (double)hw, gamma, beta, 1e-3f, sc, sh, m, is, var);
      fl = 1.0f;
      return;
    }
  }
  sc = t.sc[c]; sh = t.sh[c]; fl = t.fl[c];
}

// (scale, shift, flag) of channels [0, K) of the view -> Tb[3][K4] in LDS (entries K .. K4 are the identity).  The caller
// synchronises.
__device__ __forceinline__ void tab_to_lds(float* Tb, int K4, int K, const ChanTab& t, int tid, int nthreads) {
  for (int c = tid; c < K4; c += nthreads) {
    float sc = 1.0f, sh = 0.0f, fl = 0.0f;
    if (c < K) tab_entry(t, c, sc, sh, fl);
    Tb[c] = sc; Tb[K4 + c] = sh; Tb[2 * K4 + c] = fl;
  }
}

}  // namespace jnr
