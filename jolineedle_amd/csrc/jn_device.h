// Device helpers shared by kernels_env.hip and kernels_gpt.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "jn_kernels.h"

namespace jnr {

// src/env/common.py:17-27 — (dy, dx) per action id LEFT..STOP.
__device__ __constant__ const int8_t kActionDy[9] = {0, 0, -1, 1, -1, -1, 1, 1, 0};
__device__ __constant__ const int8_t kActionDx[9] = {-1, 1, 0, 0, -1, 1, -1, 1, 0};

// Philox4x32-10 keyed by seed, counter (a, b, c, d) -> 4 x u32.
__device__ inline uint4 philox4x32(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}
__device__ inline float u01(uint32_t x) { return (x >> 8) * (1.0f / 16777216.0f); }   // [0, 1)

// Dropout of the decision transformer (src/models/gpt.py:65-66, 100, 107, 124, 314: embd / attn / resid, p = --dropout):
// keep-scale (0 or 1 / (1 - p)) of element `idx` at `site` of `layer` for token `tok` of agent `b`, a pure function of the
// rollout's dropout seed — the teacher-forced backward regenerates it, no mask is stored.  Sites: 0 embedding (after
// final_emb), 1 attention probabilities (idx = head * Tmax + key), 2 residual after attn.c_proj, 3 residual after mlp.c_proj.
__device__ inline float drop_scale(uint64_t seed, int b, int tok, int layer, int site, int idx, float p) {
  const uint4 r = philox4x32(seed, (uint32_t)b, (uint32_t)tok, 0x44520000u | ((uint32_t)layer << 4) | (uint32_t)site, (uint32_t)(idx >> 2));
  const int k = idx & 3;
  const uint32_t w = k == 0 ? r.x : k == 1 ? r.y : k == 2 ? r.z : r.w;
  return u01(w) >= p ? 1.0f / (1.0f - p) : 0.0f;
}

struct EnvStepResult { float reward; bool terminated, truncated; int y, x; };

// One agent's NeedleGeneralEnv.step (src/env/general_env.py:172-233, 321-358, 235-246):
// move + clamp, sticky STOP, reward from `visited` BEFORE this step's update, visit, count.
__device__ inline EnvStepResult env_step_one(const EnvPtrs& e, int b, int action) {
  int y = (int)e.positions[2 * b] + kActionDy[action];
  int x = (int)e.positions[2 * b + 1] + kActionDx[action];
  y = min(max(y, 0), e.Gh - 1);
  x = min(max(x, 0), e.Gw - 1);
  e.positions[2 * b] = y; e.positions[2 * b + 1] = x;
  bool stopped = e.has_stopped[b] | (action == 8);
  e.has_stopped[b] = stopped;
  const int tile = (b * e.Gh + y) * e.Gw + x;
  const bool on_box = e.bbox_masks[tile], seen = e.visited[tile];
  const float hit = (on_box && !seen) ? 1.0f : 0.0f;
  const float cost = (float)(-1.0 / (double)e.T);
  int found = e.found[b];
  const int total = e.n_bbox_tiles[b];
  float stop_eval = 0.0f;
  if (e.stop) {
    const int se = (found == total) ? found : (found - total);
    stop_eval = stopped ? (float)se : 0.0f;
  }
  EnvStepResult r;
  r.reward = (hit + cost) + stop_eval;
  if (on_box && !seen) found += 1;
  e.found[b] = found;
  e.visited[tile] = 1;
  const int steps = e.steps[b] + 1;
  e.steps[b] = steps;
  r.truncated = steps >= e.T;
  r.terminated = e.stop ? stopped : (found == total);
  r.y = y; r.x = x;
  return r;
}

}  // namespace jnr
