// common.h -- shared device helpers for libplipmi (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plipmi {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// OCP e4m3fn operand of the experimental fp8 GEMM (storage tag only: the arithmetic is the scaled f8f6f4 MFMA)
struct fp8_t { unsigned char v; };
typedef __attribute__((ext_vector_type(8))) int i32x8;

constexpr int kWave = 64;  // CDNA wavefront

// ---- scalar conversions -------------------------------------------------
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }  // RNE (v_cvt_pk_bf16_f32)

// 4 consecutive outputs: fp32 -> 16-byte store, bf16 -> 8-byte store
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
__device__ __forceinline__ void store4(bf16_t* p, float a, float b, float c, float d) {
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
  bf16x4 v = {(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
  *reinterpret_cast<bf16x4*>(p) = v;
}
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const bf16_t* p) {
  typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}

// ---- wavefront reductions (64 lanes) -----------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// QuickGELU: x * sigmoid(1.702 x)  (transformers/activations.py:117-123)
// fp32 engine: libm exp + IEEE divide; bf16 engine: v_exp_f32 + v_rcp_f32 (1 ulp), far below bf16 rounding
template <bool kAccurate> __device__ __forceinline__ float quick_gelu(float x) {
  if constexpr (kAccurate) return x / (1.0f + expf(-1.702f * x));
  else return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
}

}  // namespace plipmi
