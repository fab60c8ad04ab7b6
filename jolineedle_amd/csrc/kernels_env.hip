// Environment primitives: patch gather (bit-exact), bbox -> patch-grid masks, reset,
// step/reward, and the rollout prologue / epilogue (masks roll + suffix-sum returns).
// Integer / bool work is exact; see jn_device.h for the per-agent step.
#include <hip/hip_runtime.h>

#include "jn_device.h"

namespace jnr {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// out[b, c, r, :] = images[b, c, y*P + r, x*P : x*P + P]   (src/env/general_env.py:285-306)
template <bool VEC>
__global__ __launch_bounds__(256) void gather_kernel(const float* __restrict__ images,
                                                     const long long* __restrict__ pos, float* __restrict__ out,
                                                     long long out_sample_stride, int C, int H, int W, int P,
                                                     long long total, const int* __restrict__ skip_flag,
                                                     int skip_when, const long long* __restrict__ img_idx) {
  if (skip_flag && *skip_flag >= skip_when) return;
  constexpr int V = VEC ? 4 : 1;
  const int PV = P / V;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int q = (int)(idx % PV);
    const int r = (int)((idx / PV) % P);
    const int c = (int)((idx / ((long long)PV * P)) % C);
    const int b = (int)(idx / ((long long)PV * P * C));
    const long long y = pos[2 * b], x = pos[2 * b + 1];
    const long long im = img_idx ? img_idx[b] : b;   // indexed form: several patches per image; < 0 = zero patch
    float* dp = out + b * out_sample_stride + ((long long)c * P + r) * P + q * V;
    if (im < 0) {
      if (VEC) *reinterpret_cast<f32x4*>(dp) = f32x4{0.f, 0.f, 0.f, 0.f};
      else *dp = 0.f;
      continue;
    }
    const float* sp = images + ((im * C + c) * H + y * P + r) * W + x * P + q * V;
    if (VEC) *reinterpret_cast<f32x4*>(dp) = *reinterpret_cast<const f32x4*>(sp);
    else *dp = *sp;
  }
}

int launch_gather(const float* images, const int64_t* positions, float* out, long long out_sample_stride, int B,
                  int C, int H, int W, int P, const int* skip_flag, int skip_when, hipStream_t s,
                  const int64_t* image_index) {
  const bool vec = (P % 4 == 0) && (W % 4 == 0) && (out_sample_stride % 4 == 0) &&
                   (((uintptr_t)images | (uintptr_t)out) % 16 == 0);
  const long long total = (long long)B * C * P * (P / (vec ? 4 : 1));
  const unsigned blocks = (unsigned)std::min<long long>((total + 255) / 256, 256 * 32);
  if (vec)
    hipLaunchKernelGGL(gather_kernel<true>, dim3(blocks), dim3(256), 0, s, images, (const long long*)positions, out,
                       out_sample_stride, C, H, W, P, total, skip_flag, skip_when, (const long long*)image_index);
  else
    hipLaunchKernelGGL(gather_kernel<false>, dim3(blocks), dim3(256), 0, s, images, (const long long*)positions, out,
                       out_sample_stride, C, H, W, P, total, skip_flag, skip_when, (const long long*)image_index);
  return 0;
}

// convert_bboxes_to_masks (src/env/general_env.py:360-379) on the patch grid: a box covers
// pixels x1..x2, y1..y2 inclusive (kornia "xyxy_plus"), clipped to the image; a patch is
// marked when it holds at least one covered pixel.  One thread per image.
__global__ void bbox_masks_kernel(const long long* __restrict__ bboxes, uint8_t* __restrict__ masks,
                                  int32_t* __restrict__ n_tiles, int B, int nb, int H, int W, int P) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const int Gh = H / P, Gw = W / P;
  uint8_t* m = masks + (long long)b * Gh * Gw;
  for (int i = 0; i < Gh * Gw; ++i) m[i] = 0;
  for (int k = 0; k < nb; ++k) {
    const long long* bb = bboxes + ((long long)b * nb + k) * 4;
    long long x1 = bb[0] < 0 ? 0 : bb[0], y1 = bb[1] < 0 ? 0 : bb[1];
    long long x2 = bb[2] + 1 > W ? W : bb[2] + 1, y2 = bb[3] + 1 > H ? H : bb[3] + 1;   // exclusive
    if (x2 <= x1 || y2 <= y1) continue;
    for (int gy = (int)(y1 / P); gy <= (int)((y2 - 1) / P); ++gy)
      for (int gx = (int)(x1 / P); gx <= (int)((x2 - 1) / P); ++gx) m[gy * Gw + gx] = 1;
  }
  int cnt = 0;
  for (int i = 0; i < Gh * Gw; ++i) cnt += m[i];
  n_tiles[b] = cnt;
}

int launch_bbox_masks(const int64_t* bboxes, uint8_t* masks, int32_t* n_tiles, int B, int nb, int H, int W, int P,
                      hipStream_t s) {
  hipLaunchKernelGGL(bbox_masks_kernel, dim3((B + 63) / 64), dim3(64), 0, s, (const long long*)bboxes, masks, n_tiles,
                     B, nb, H, W, P);
  return 0;
}

// reset (src/env/general_env.py:144-170): zero state, place agents, mark the start tile.
__global__ void env_reset_kernel(EnvPtrs e, const long long* __restrict__ start, uint64_t seed) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= e.B) return;
  int y, x;
  if (start) { y = (int)start[2 * b]; x = (int)start[2 * b + 1]; }
  else {
    const uint4 r = philox4x32(seed, (uint32_t)b, 0u, 0x52455345u, 0u);
    y = (int)(r.x % (uint32_t)e.Gh); x = (int)(r.y % (uint32_t)e.Gw);
  }
  e.positions[2 * b] = y; e.positions[2 * b + 1] = x;
  uint8_t* v = e.visited + (long long)b * e.Gh * e.Gw;
  for (int i = 0; i < e.Gh * e.Gw; ++i) v[i] = 0;
  const int tile = y * e.Gw + x;
  v[tile] = 1;
  e.found[b] = e.bbox_masks[(long long)b * e.Gh * e.Gw + tile] ? 1 : 0;
  e.steps[b] = 0;
  e.has_stopped[b] = 0;
}

int launch_env_reset(const EnvPtrs& e, const int64_t* start_positions, uint64_t seed, hipStream_t s) {
  hipLaunchKernelGGL(env_reset_kernel, dim3((e.B + 63) / 64), dim3(64), 0, s, e, (const long long*)start_positions, seed);
  return 0;
}

__global__ void env_step_kernel(EnvPtrs e, const long long* __restrict__ actions, float* __restrict__ rewards,
                                uint8_t* __restrict__ terminated, uint8_t* __restrict__ truncated) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= e.B) return;
  int a = (int)actions[b];
  a = min(max(a, 0), 8);
  const EnvStepResult r = env_step_one(e, b, a);
  if (rewards) rewards[b] = r.reward;
  if (terminated) terminated[b] = r.terminated;
  if (truncated) truncated[b] = r.truncated;
}

int launch_env_step(const EnvPtrs& e, const int64_t* actions, float* rewards, uint8_t* terminated,
                    uint8_t* truncated, hipStream_t s) {
  hipLaunchKernelGGL(env_step_kernel, dim3((e.B + 63) / 64), dim3(64), 0, s, e, (const long long*)actions, rewards,
                     terminated, truncated);
  return 0;
}

// Rollout prologue (src/reinforce.py:123-139): BOS action 0, masks[:,0] = True, positions[:,0].
__global__ void rollout_begin_kernel(EnvPtrs e, RolloutBuffers r, long long* prev_action, int32_t* cache_len,
                                     int32_t* n_done) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b == 0)
    for (int t = 0; t <= e.T; ++t) n_done[t] = 0;
  if (b >= e.B) return;
  prev_action[b] = 0;
  cache_len[b] = 0;
  if (r.masks) r.masks[(long long)b * (e.T + 1)] = 1;
  if (r.positions) {
    r.positions[((long long)b * (e.T + 1)) * 2] = e.positions[2 * b];
    r.positions[((long long)b * (e.T + 1)) * 2 + 1] = e.positions[2 * b + 1];
  }
}

int launch_rollout_begin(const EnvPtrs& e, const RolloutBuffers& r, int64_t* prev_action, int32_t* cache_len,
                         int32_t* n_done, hipStream_t s) {
  hipLaunchKernelGGL(rollout_begin_kernel, dim3((e.B + 63) / 64), dim3(64), 0, s, e, r, (long long*)prev_action, cache_len,
                     n_done);
  return 0;
}

// Rollout epilogue (src/reinforce.py:186-202): logit_masks = roll(masks[:,1:], 1) with column 0
// forced True; returns[t] = sum_{k>=t} rewards[k] * logit_masks[k], accumulated from the end
// (the order of the reference's flip -> cumsum -> flip).  S = steps actually executed.
__global__ void rollout_epilogue_kernel(RolloutBuffers r, const int32_t* __restrict__ n_done, int B, int T,
                                        int stop_early) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int S = T;
  if (stop_early)
    for (int t = 1; t <= T; ++t)
      if (n_done[t] >= B) { S = t; break; }
  float run = 0.0f;
  for (int t = S - 1; t >= 0; --t) {
    const bool lm = (t == 0) ? true : (r.masks[(long long)b * (T + 1) + t] != 0);
    if (r.logit_masks) r.logit_masks[(long long)b * T + t] = lm;
    run = run + r.rewards[(long long)b * T + t] * (lm ? 1.0f : 0.0f);
    if (r.returns) r.returns[(long long)b * T + t] = run;
  }
}

int launch_rollout_epilogue(const RolloutBuffers& r, const int32_t* n_done, int B, int T, int stop_early,
                            hipStream_t s) {
  hipLaunchKernelGGL(rollout_epilogue_kernel, dim3((B + 63) / 64), dim3(64), 0, s, r, n_done, B, T, stop_early);
  return 0;
}

}  // namespace jnr
