// Detector training loss: the YOLOX head's get_losses for one class — SimOTA assignment
// (centre-radius candidates, dynamic-k by the summed top-10 IoUs, lowest-cost matching, one gt per anchor), IoU loss
// (1 - iou^2, weight 5), objectness BCE over all anchors, class BCE against the matched IoU, L1 on the raw regression —
// and d loss / d raw predictor outputs.  Restated from the published YOLOX v0.3.0 yolo_head.py (call site
// src/models/yolox.py:58-73; the package itself is not under /root/reference: parity is against oracle/yolox_ref.py).
// One workgroup per patch: a patch has <= 27 candidate anchors per ground-truth box (3 x 3 cells x 3 levels).
#include <hip/hip_runtime.h>

#include "jn_kernels.h"
#include "jn_types.h"

namespace jnr {

__device__ __forceinline__ float silu_l(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
__device__ __forceinline__ f32x4 tf4_d(f32x4 z, f32x4 sc, f32x4 sh, f32x4 fl) {
  f32x4 r;
  r.x = fl.x != 0.0f ? silu_l(fmaf(z.x, sc.x, sh.x)) : z.x;
  r.y = fl.y != 0.0f ? silu_l(fmaf(z.y, sc.y, sh.y)) : z.y;
  r.z = fl.z != 0.0f ? silu_l(fmaf(z.z, sc.z, sh.z)) : z.z;
  r.w = fl.w != 0.0f ? silu_l(fmaf(z.w, sc.w, sh.w)) : z.w;
  return r;
}

constexpr int DL_MAXG = 8;          // ground-truth boxes per patch
constexpr int DL_MAXC = 256;        // candidate anchors per patch (27 per box)

__device__ __forceinline__ float sigm(float v) { return 1.0f / (1.0f + expf(-v)); }
__device__ __forceinline__ float softplus(float v) { return fmaxf(v, 0.0f) + log1pf(expf(-fabsf(v))); }

struct DlBox { float cx, cy, w, h; };

__device__ __forceinline__ float iou_cxcywh(const DlBox& a, const DlBox& b) {
  const float tlx = fmaxf(a.cx - a.w * 0.5f, b.cx - b.w * 0.5f), tly = fmaxf(a.cy - a.h * 0.5f, b.cy - b.h * 0.5f);
  const float brx = fminf(a.cx + a.w * 0.5f, b.cx + b.w * 0.5f), bry = fminf(a.cy + a.h * 0.5f, b.cy + b.h * 0.5f);
  const float en = (tlx < brx && tly < bry) ? 1.0f : 0.0f;
  const float inter = (brx - tlx) * (bry - tly) * en;
  return inter / (a.w * a.h + b.w * b.h - inter);
}

// acc: per-launch accumulators [0] iou loss, [1] obj loss, [2] cls loss, [3] l1 loss, [4] num_fg, [5] num_gt (floats)
__global__ __launch_bounds__(256) void yolox_loss_kernel(const float* __restrict__ raw, const float* __restrict__ labels,
                                                         int nb, DetGeom geo, float* __restrict__ d_raw,
                                                         float* __restrict__ acc, int use_l1) {
  __shared__ DlBox gt[DL_MAXG];
  __shared__ int cand[DL_MAXC];                 // anchor index of each candidate, ascending
  __shared__ unsigned char geo_ok[DL_MAXG][DL_MAXC];
  __shared__ float ious[DL_MAXG][DL_MAXC], cost[DL_MAXG][DL_MAXC];
  __shared__ unsigned char match[DL_MAXG][DL_MAXC];
  __shared__ int c_gt[DL_MAXC];                 // matched gt of a foreground candidate, -1 otherwise
  __shared__ int s_ng, s_nc, wave_cnt[4];
  __shared__ float part[6];
  const int n = blockIdx.x, tid = threadIdx.x, A = geo.A;
  const float* r = raw + (long long)n * A * 6;
  float* dr = d_raw + (long long)n * A * 6;
  if (tid == 0) {
    // number of objects = rows with a positive sum; the FIRST ng rows are taken as the boxes (as published)
    int ng = 0;
    for (int k = 0; k < nb; ++k) {
      const float* l = labels + ((long long)n * nb + k) * 5;
      if (l[0] + l[1] + l[2] + l[3] + l[4] > 0.0f) ++ng;
    }
    if (ng > DL_MAXG) ng = DL_MAXG;
    for (int k = 0; k < ng; ++k) {
      const float* l = labels + ((long long)n * nb + k) * 5;
      gt[k] = DlBox{l[1], l[2], l[3], l[4]};
    }
    s_ng = ng; s_nc = 0;
  }
  if (tid < 6) part[tid] = 0.0f;
  __syncthreads();
  const int ng = s_ng;
  auto level_of = [&](int a, int& gx, int& gy, float& st) {
    int l = 0;
    while (l < 2 && a >= geo.a0[l + 1]) ++l;
    const int p = a - geo.a0[l];
    gx = p % geo.W[l]; gy = p / geo.W[l]; st = (float)geo.stride[l];
  };
  // ---- candidates: anchors whose centre lies within 1.5 strides of some gt centre (ordered compaction) ----
  for (int base = 0; base < A && ng > 0; base += 256) {
    const int a = base + tid;
    bool ok = false;
    if (a < A) {
      int gx, gy; float st;
      level_of(a, gx, gy, st);
      const float xc = (gx + 0.5f) * st, yc = (gy + 0.5f) * st, rad = 1.5f * st;
      for (int g = 0; g < ng; ++g) {
        const float m = fminf(fminf(xc - (gt[g].cx - rad), yc - (gt[g].cy - rad)), fminf(gt[g].cx + rad - xc, gt[g].cy + rad - yc));
        ok |= m > 0.0f;
      }
    }
    const unsigned long long bal = __ballot(ok);
    const int lane = tid & 63, wv = tid >> 6;
    if (lane == 0) wave_cnt[wv] = __popcll(bal);
    __syncthreads();
    int off = s_nc;
    for (int w = 0; w < wv; ++w) off += wave_cnt[w];
    const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
    if (ok && pos < DL_MAXC) cand[pos] = a;
    __syncthreads();
    if (tid == 0) s_nc = min(DL_MAXC, s_nc + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3]);
    __syncthreads();
  }
  const int nc = s_nc;
  // ---- pairwise IoU / cost ----
  for (int e = tid; e < ng * nc; e += 256) {
    const int g = e / nc, c = e - g * nc, a = cand[c];
    int gx, gy; float st;
    level_of(a, gx, gy, st);
    const float* o = r + a * 6;
    const DlBox pb{(o[0] + gx) * st, (o[1] + gy) * st, expf(o[2]) * st, expf(o[3]) * st};
    const float xc = (gx + 0.5f) * st, yc = (gy + 0.5f) * st, rad = 1.5f * st;
    const float m = fminf(fminf(xc - (gt[g].cx - rad), yc - (gt[g].cy - rad)), fminf(gt[g].cx + rad - xc, gt[g].cy + rad - yc));
    const bool in = m > 0.0f;
    const float iou = iou_cxcywh(gt[g], pb);
    const float p = sqrtf(sigm(o[5]) * sigm(o[4]));
    const float cls_cost = fminf(-logf(p), 100.0f);              // F.binary_cross_entropy(p, 1): log clamped at -100
    geo_ok[g][c] = in;
    ious[g][c] = iou;
    cost[g][c] = cls_cost + 3.0f * (-logf(iou + 1e-8f)) + (in ? 0.0f : 1e6f);
    match[g][c] = 0;
  }
  __syncthreads();
  // ---- dynamic k and the k cheapest candidates of every gt (one thread per gt; <= 27 real candidates each) ----
  if (tid < ng && nc > 0) {
    const int g = tid;
    float top[10];
    int nt = 0;
    for (int c = 0; c < nc; ++c) {                               // ten largest IoUs, kept sorted descending
      const float v = ious[g][c];
      int j;
      if (nt < 10) j = nt++;
      else if (v > top[9]) j = 9;
      else continue;
      while (j > 0 && top[j - 1] < v) { top[j] = top[j - 1]; --j; }
      top[j] = v;
    }
    float sum = 0.0f;
    for (int j = 0; j < nt; ++j) sum += top[j];
    int k = (int)sum;
    if (k < 1) k = 1;
    if (k > nc) k = nc;
    for (int it = 0; it < k; ++it) {                             // k smallest costs, lowest index first on ties
      int best = -1;
      float bv = INFINITY;
      for (int c = 0; c < nc; ++c)
        if (!match[g][c] && cost[g][c] < bv) { bv = cost[g][c]; best = c; }
      if (best >= 0) match[g][best] = 1;
    }
  }
  __syncthreads();
  // ---- one gt per anchor: an anchor claimed by several gts goes to the cheapest ----
  for (int c = tid; c < nc; c += 256) {
    int cnt = 0, first = -1, amin = 0;
    float bv = INFINITY;
    for (int g = 0; g < ng; ++g) {
      if (match[g][c]) { ++cnt; if (first < 0) first = g; }
      if (cost[g][c] < bv) { bv = cost[g][c]; amin = g; }
    }
    c_gt[c] = cnt == 0 ? -1 : (cnt == 1 ? first : amin);
  }
  __syncthreads();
  // ---- objectness over all anchors (target 1 on foreground anchors), zero the other gradients ----
  float l_obj = 0.0f;
  for (int a = tid; a < A; a += 256) {
    const float o = r[a * 6 + 4];
    l_obj += softplus(o);
    float* d = dr + a * 6;
    d[0] = 0.0f; d[1] = 0.0f; d[2] = 0.0f; d[3] = 0.0f; d[4] = sigm(o); d[5] = 0.0f;
  }
  __syncthreads();
  // ---- foreground anchors: IoU, class and L1 terms ----
  float l_iou = 0.0f, l_cls = 0.0f, l_l1 = 0.0f, n_fg = 0.0f;
  for (int c = tid; c < nc; c += 256) {
    const int g = c_gt[c];
    if (g < 0) continue;
    const int a = cand[c];
    int gx, gy; float st;
    level_of(a, gx, gy, st);
    const float* o = r + a * 6;
    float* d = dr + a * 6;
    n_fg += 1.0f;
    l_obj -= o[4];                       // BCE-with-logits target 1: softplus(o) - o
    d[4] -= 1.0f;
    const DlBox t = gt[g];
    const float pw = expf(o[2]) * st, ph = expf(o[3]) * st, pcx = (o[0] + gx) * st, pcy = (o[1] + gy) * st;
    // IoU loss 1 - iou^2 (eps 1e-16 in the union) and its gradient through the decoded box
    const float ptlx = pcx - pw * 0.5f, ptly = pcy - ph * 0.5f, pbrx = pcx + pw * 0.5f, pbry = pcy + ph * 0.5f;
    const float ttlx = t.cx - t.w * 0.5f, ttly = t.cy - t.h * 0.5f, tbrx = t.cx + t.w * 0.5f, tbry = t.cy + t.h * 0.5f;
    const float tlx = fmaxf(ptlx, ttlx), tly = fmaxf(ptly, ttly), brx = fminf(pbrx, tbrx), bry = fminf(pbry, tbry);
    const bool en = tlx < brx && tly < bry;
    const float iw = brx - tlx, ih = bry - tly;
    const float inter = en ? iw * ih : 0.0f;
    const float ap = pw * ph, ag = t.w * t.h;
    const float uni = ap + ag - inter + 1e-16f;
    const float iou = inter / uni;
    l_iou += 1.0f - iou * iou;
    // d inter / d (tl, br); torch.max / torch.min route the gradient to the selected operand (ties: split evenly)
    float di_tlx = en ? -ih : 0.0f, di_tly = en ? -iw : 0.0f, di_brx = en ? ih : 0.0f, di_bry = en ? iw : 0.0f;
    auto sel_max = [](float p, float q) { return p > q ? 1.0f : (p == q ? 0.5f : 0.0f); };
    auto sel_min = [](float p, float q) { return p < q ? 1.0f : (p == q ? 0.5f : 0.0f); };
    const float s_tlx = sel_max(ptlx, ttlx), s_tly = sel_max(ptly, ttly), s_brx = sel_min(pbrx, tbrx), s_bry = sel_min(pbry, tbry);
    const float di_cx = di_tlx * s_tlx + di_brx * s_brx, di_cy = di_tly * s_tly + di_bry * s_bry;
    const float di_w = -0.5f * di_tlx * s_tlx + 0.5f * di_brx * s_brx, di_h = -0.5f * di_tly * s_tly + 0.5f * di_bry * s_bry;
    // iou = inter / uni, uni = ap + ag - inter
    const float inv = 1.0f / (uni * uni);
    auto d_iou = [&](float di, float dap) { return (di * uni - inter * (dap - di)) * inv; };
    const float k_l = -2.0f * iou * 5.0f;                         // d (5 * (1 - iou^2)) / d iou
    const float g_cx = k_l * d_iou(di_cx, 0.0f), g_cy = k_l * d_iou(di_cy, 0.0f);
    const float g_w = k_l * d_iou(di_w, ph), g_h = k_l * d_iou(di_h, pw);
    d[0] += g_cx * st; d[1] += g_cy * st; d[2] += g_w * pw; d[3] += g_h * ph;
    // class BCE-with-logits against the (detached) IoU of the match
    const float tc = ious[g][c];
    l_cls += softplus(o[5]) - tc * o[5];
    d[5] += sigm(o[5]) - tc;
    if (use_l1) {
      const float lt[4] = {t.cx / st - gx, t.cy / st - gy, logf(t.w / st + 1e-8f), logf(t.h / st + 1e-8f)};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float df = o[k] - lt[k];
        l_l1 += fabsf(df);
        d[k] += df > 0.0f ? 1.0f : (df < 0.0f ? -1.0f : 0.0f);
      }
    }
  }
  atomicAdd(&part[0], l_iou); atomicAdd(&part[1], l_obj); atomicAdd(&part[2], l_cls); atomicAdd(&part[3], l_l1);
  atomicAdd(&part[4], n_fg);
  __syncthreads();
  if (tid < 5) atomicAdd(&acc[tid], part[tid]);
  if (tid == 5) atomicAdd(&acc[5], (float)ng);
}

// metrics[0..5] = total, 5 * iou, obj, cls, l1, num_fg / max(num_gts, 1); scale[0] = loss_scale / max(num_fg, 1)
__global__ void yolox_loss_finalize_kernel(const float* __restrict__ acc, float loss_scale, float* __restrict__ metrics,
                                           float* __restrict__ scale) {
  const float den = fmaxf(acc[4], 1.0f);
  const float iou = 5.0f * acc[0] / den, obj = acc[1] / den, cls = acc[2] / den, l1 = acc[3] / den;
  metrics[0] = iou + obj + cls + l1; metrics[1] = iou; metrics[2] = obj; metrics[3] = cls; metrics[4] = l1;
  metrics[5] = acc[4] / fmaxf(acc[5], 1.0f);
  scale[0] = loss_scale / den;
}

int launch_yolox_loss(const float* raw, const float* labels, int N, int nb, const DetGeom& geo, float* d_raw, float* acc,
                      int use_l1, float loss_scale, float* metrics, float* scale, hipStream_t s) {
  (void)hipMemsetAsync(acc, 0, 8 * sizeof(float), s);
  hipLaunchKernelGGL(yolox_loss_kernel, dim3(N), dim3(256), 0, s, raw, labels, nb, geo, d_raw, acc, use_l1);
  hipLaunchKernelGGL(yolox_loss_finalize_kernel, dim3(1), dim3(1), 0, s, acc, loss_scale, metrics, scale);
  return 0;
}

// ---- backward of the three predictor convs of one level --------------------------------------------------------
// d_raw rows (scaled by *scale): rows 0..4 read reg_feat, row 5 reads cls_feat.  One thread per anchor forms the data
// gradients g_reg[p][c] = sum_{r<5} d[r] w[r][c], g_cls[p][c] = d[5] w[5][c]; weight / bias gradients are reduced per
// workgroup in LDS and added with one atomic per entry.
template <typename AT>
__global__ __launch_bounds__(256) void head_pred_bwd_kernel(const float* __restrict__ d_raw, const float* __restrict__ scale,
                                                            const AT* __restrict__ reg, int reg_ld, ChanTab rt,
                                                            const AT* __restrict__ cls, int cls_ld, ChanTab ct,
                                                            const float* __restrict__ wp, float* __restrict__ g_reg,
                                                            float* __restrict__ g_cls, float* __restrict__ g_wp,
                                                            float* __restrict__ g_bp, int hid, int HW, int A, int a0, int N) {
  extern __shared__ float sw[];          // [6][hid] weights, then [6][hid] weight-gradient partials, then [6] bias partials
  float* gw = sw + 6 * hid;
  float* gb = gw + 6 * hid;
  for (int i = threadIdx.x; i < 6 * hid; i += 256) { sw[i] = wp[i]; gw[i] = 0.0f; }
  if (threadIdx.x < 6) gb[threadIdx.x] = 0.0f;
  __syncthreads();
  const float sc = scale[0];
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const bool active = idx < (long long)N * HW;
  float d[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (active) {
    const int p = (int)(idx % HW);
    const long long n = idx / HW;
    const float* dp = d_raw + (n * A + a0 + p) * 6;
#pragma unroll
    for (int j = 0; j < 6; ++j) d[j] = dp[j] * sc;
  }
  // bias gradients: wave sums, then LDS
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    float v = d[j];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) atomicAdd(&gb[j], v);
  }
  for (int k = 0; k < hid; k += 4) {
    f32x4 rv = {0.f, 0.f, 0.f, 0.f}, cv = {0.f, 0.f, 0.f, 0.f};
    if (active) {
      rv = tf4_d(ld4(reg + idx * reg_ld + k), *reinterpret_cast<const f32x4*>(rt.sc + k), *reinterpret_cast<const f32x4*>(rt.sh + k),
                 *reinterpret_cast<const f32x4*>(rt.fl + k));
      cv = tf4_d(ld4(cls + idx * cls_ld + k), *reinterpret_cast<const f32x4*>(ct.sc + k), *reinterpret_cast<const f32x4*>(ct.sh + k),
                 *reinterpret_cast<const f32x4*>(ct.fl + k));
      f32x4 gr, gc;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float a = 0.0f;
#pragma unroll
        for (int j = 0; j < 5; ++j) a = fmaf(d[j], sw[j * hid + k + q], a);
        gr[q] = a;
        gc[q] = d[5] * sw[5 * hid + k + q];
      }
      *reinterpret_cast<f32x4*>(g_reg + idx * reg_ld + k) = gr;
      *reinterpret_cast<f32x4*>(g_cls + idx * cls_ld + k) = gc;
    }
    // weight gradients: dw[j][k + q] += d[j] * a[k + q], wave-reduced
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        float v = d[j] * (j < 5 ? rv[q] : cv[q]);
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((threadIdx.x & 63) == 0) atomicAdd(&gw[j * hid + k + q], v);
      }
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 6 * hid; i += 256) atomicAdd(&g_wp[i], gw[i]);
  if (threadIdx.x < 6) atomicAdd(&g_bp[threadIdx.x < 4 ? threadIdx.x : 4 * (threadIdx.x - 3)], gb[threadIdx.x]);
}

int launch_head_pred_bwd(const float* d_raw, const float* scale, const void* reg, int reg_ld, ChanTab rt, const void* cls,
                         int cls_ld, ChanTab ct, int dtype, const float* wp, float* g_reg, float* g_cls, float* g_wp,
                         float* g_bp, int hid, int HW, int A, int a0, int N, hipStream_t s) {
  if (dtype != JN_F32) return -1;
  const long long total = (long long)N * HW;
  const size_t smem = ((size_t)12 * hid + 8) * sizeof(float);
  hipLaunchKernelGGL(head_pred_bwd_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), smem, s, d_raw, scale,
                     (const float*)reg, reg_ld, rt, (const float*)cls, cls_ld, ct, wp, g_reg, g_cls, g_wp, g_bp, hid, HW, A, a0, N);
  return 0;
}

}  // namespace jnr
