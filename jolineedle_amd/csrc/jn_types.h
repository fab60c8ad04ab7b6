// Storage types of activation buffers and the 4-wide load/store helpers every kernel uses.
// fp32 = the reference's dtype (parity mode); bf16 = MI355X performance mode (activations stored in
// bf16, accumulation / statistics / weights master copy in fp32, 1x1 convs on bf16 MFMA).
#pragma once
#include <hip/hip_runtime.h>

namespace jnr {

using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef __bf16 bf16_t;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const bf16_t* p) {
  return __builtin_convertvector(*reinterpret_cast<const bf16x4*>(p), f32x4);
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, f32x4 v) {
  *reinterpret_cast<bf16x4*>(p) = __builtin_convertvector(v, bf16x4);
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return (float)*p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = (bf16_t)v; }

// One MFMA operand fragment of v_mfma_f32_16x16x32_bf16 (8 consecutive k of the lane's row / column) out of a [k][channel]
// bf16 tile in LDS: two ds_read_b64_tr_b16.  Per 16-lane group the hardware reads 4 k-rows x 16 channels and hands every
// lane ITS channel's four values; lane 4 q + p of the group supplies the address of row q, channels 4 p .. 4 p + 3
// (8-byte aligned).  r0 / r1: this lane's addresses for k 0..3 / 4..7 of its octet.  EXEC must be all ones.
typedef __attribute__((ext_vector_type(4))) short tr_s4;
__device__ __forceinline__ bf16x8 tr_frag(const bf16_t* r0, const bf16_t* r1) {
  const tr_s4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_s4 __attribute__((address_space(3)))*)r0);
  const tr_s4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tr_s4 __attribute__((address_space(3)))*)r1);
  typedef __attribute__((ext_vector_type(8))) short s8;
  const s8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(bf16x8, v);
}

}  // namespace jnr
