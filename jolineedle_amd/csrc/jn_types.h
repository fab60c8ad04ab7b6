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

}  // namespace jnr
