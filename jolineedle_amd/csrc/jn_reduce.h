// Cross-lane reductions shared by the MFMA kernels' BatchNorm-statistics epilogues.
#pragma once
#include <hip/hip_runtime.h>

#include "jn_types.h"

namespace jnr {

// Sum over the 16 lanes of a DPP row (lanes that share lane>>4): four v_add_f32_dpp (quad_perm [1,0,3,2],
// quad_perm [2,3,0,1], row_half_mirror, row_mirror).  __shfl_xor would go through ds_bpermute (LDS crossbar) and
// made this reduction 8 % of a train-mode pass.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  return v;
}

// Epilogue helper of the MFMA kernels: s1/s2[t] hold sum / sumsq over this lane's pixels of channels
// (16 t + 4g .. +3).  Stores the wave's per-channel totals into ITS OWN LDS slots `red` [nch][2] (plain stores:
// LDS float atomics are slow); the caller sums the slots of the waves that share channels.
template <int NT>
__device__ __forceinline__ void wave_stats_to_lds(const f32x4 (&s1)[NT], const f32x4 (&s2)[NT], float* red,
                                                  int lane, int nch) {
  const int lm = lane & 15, g = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float a = row16_sum(s1[t][r]);
      const float b = row16_sum(s2[t][r]);
      const int ch = 16 * t + 4 * g + r;
      if (lm == 0 && ch < nch) {
        red[2 * ch] = a;
        red[2 * ch + 1] = b;
      }
    }
  }
}

}  // namespace jnr
