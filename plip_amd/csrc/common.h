// common.h -- shared device helpers for libplipmi (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plipmi {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
// IEEE half: the OTHER 16-bit operand type of the matrix cores (v_mfma_f32_32x32x16_f16, same rate as bf16).  11
// significand bits against bf16's 8: the PLIPMI_F16 engine's operand roundings are 8x smaller, which is what the cosine
// bar needs on the text tower; range +-65504 -- the reference's own CUDA path runs its CLIP in
// this type (clip.model.convert_weights; reproducibility/embedders/factory.py:21 -> clip.load on "cuda").
typedef _Float16 f16_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;


constexpr int kWave = 64;  // CDNA wavefront

// ---- scalar conversions -------------------------------------------------
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return (float)x; }
__device__ __forceinline__ float to_f32(f16_t x) { return (float)x; }
// fp32 -> half: values beyond the type's range SATURATE (+-65504) instead of becoming inf, so one outlier activation costs
// accuracy on its own row, not NaNs through every softmax it reaches
__device__ __forceinline__ float sat_f16(float x) { return __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f); }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) { return (bf16_t)x; }  // RNE (v_cvt_pk_bf16_f32)
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float x) { return (f16_t)sat_f16(x); }   // RNE (v_cvt_f16_f32)

// the two 16-bit operand types: vector forms and the MFMA that multiplies them
template <typename T> struct half_traits;
template <> struct half_traits<bf16_t> {
  using x8 = bf16x8; using x4 = bf16x4;
  static constexpr const char* name = "bf16";
  __device__ __forceinline__ static f32x16 mfma32(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  __device__ __forceinline__ static f32x4 mfma16(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct half_traits<f16_t> {
  using x8 = f16x8; using x4 = f16x4;
  static constexpr const char* name = "f16";
  __device__ __forceinline__ static f32x16 mfma32(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  __device__ __forceinline__ static f32x4 mfma16(x8 a, x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <typename T> constexpr bool is_half_v = sizeof(T) == 2;

// 4 consecutive outputs: fp32 -> 16-byte store, bf16 -> 8-byte store
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}
__device__ __forceinline__ void store4(bf16_t* p, float a, float b, float c, float d) {
  bf16x4 v = {(bf16_t)a, (bf16_t)b, (bf16_t)c, (bf16_t)d};
  *reinterpret_cast<bf16x4*>(p) = v;
}
__device__ __forceinline__ void store4(f16_t* p, float a, float b, float c, float d) {
  f16x4 v = {from_f32<f16_t>(a), from_f32<f16_t>(b), from_f32<f16_t>(c), from_f32<f16_t>(d)};
  *reinterpret_cast<f16x4*>(p) = v;
}
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const bf16_t* p) {
  bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ float4 load4(const f16_t* p) {
  f16x4 v = *reinterpret_cast<const f16x4*>(p);
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

// Sum over the 16 lanes of a DPP row (lanes 16r .. 16r+15), result in every lane of the row: two quad
// permutes, then row_half_mirror and row_mirror (after the quad steps every quad is uniform, so the mirrors act
// as the 4- and 8-lane butterfly steps).  Pure VALU (no LDS crossbar, unlike __shfl_xor / ds_bpermute).
// Call with all 64 lanes active.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);  // row_half_mirror
  v += dpp_mov<0x140>(v);  // row_mirror
  return v;
}

// the same over the 8 lanes of a half row (lanes 8h .. 8h+7)
__device__ __forceinline__ float row8_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  return v;
}

// LayerNorm statistics travel as per-row PARTIALS over 64-column slices: {sum, M2 = sum (x - sum/64)^2}.  Folding
// NS slices with Chan's update gives the row mean and the (biased) variance without ever forming E[x^2] - mean^2.
//
// The LayerNorm-folded engines keep their residual stream as TWO planes: hi = the value rounded to the engine's 16-bit operand
// type -- which IS the next GEMM's A operand, no separate copy exists -- and lo = a signed 8-BIT remainder (round 6; rounds 2-5
// carried 16 bits, the exact fp32 value).  A residual GEMM's epilogue then moves 3 + 3 bytes per element instead of 4 + 4: those
// epilogues are bound by the bytes they move (profiles/r06_bytes_plan.txt (c)).
//   bf16: hi = nearest bf16 (ties away from zero) = the upper half of the bit pattern after rounding; the remainder of the bit pattern
//         r = bits(x) - (hi << 16) lies in [-32768, 32767]; lo = round(r / 256) clamped to [-128, 127]:
//         bits(x') = (hi << 16) + (lo << 8), |x' - x| <= 2^-16 |x| (8 + 8 significand bits; the operand plane itself keeps 8) --
//         2^-15 in the corner r >= 32640, where the remainder rounds to +128 and is stored as +127 (0.2 % of values).
//   f16:  hi = nearest f16 (ties to even, saturating at +-65504); lo = round((x - hi) / 2^(E - 18)) clamped to [-128, 127] with E the
//         exponent of hi (at least -14): x - hi is at most half an f16 ulp = 2^(E - 11) = 128 units.  |x' - x| <= 2^-19 |x|
//         (11 + 8 bits) for 2^-14 <= |x| <= 65504; beyond, the stream saturates (the reference's own f16 CUDA path would hold inf).
// The stream is NOT the exact fp32 value any more: every residual update rounds it to 16 (bf16) / 19 (f16) significand bits -- three
// orders of magnitude below the 2^-9 / 2^-12 rounding of the operands it feeds.  hi is always the correctly rounded operand of the
// value the epilogue computed in fp32 (the remainder is rounded, never the operand).
//
// Memory layout of the lo plane [rows, D] (bytes): blocks of 16 rows x 8 columns = 128 bytes, [row & 7][row >> 3 & 1][column & 7]
// inside a block, blocks column-major inside a 16-row band: a GEMM epilogue lane that owns 8 consecutive columns of rows r and
// r + 8 of a band reads / writes ONE 16-byte piece, and a wave's 64 pieces are 1 KB of contiguous memory (8-byte pieces would go to
// the fabric one by one through the write-through stores: MI355X_MICROARCH.md, "stores of each flavour").  Planes are allocated
// for a whole number of bands.
__host__ __device__ __forceinline__ size_t lo_plane_bytes(size_t rows, size_t D) { return (rows + 15) / 16 * 16 * D; }
// byte offset of element (m, n); the 8 columns n & ~7 .. of row m are contiguous from lo_plane_off(m, n & ~7, D)
__device__ __forceinline__ size_t lo_plane_off(size_t m, unsigned n, unsigned D) {
  return (m >> 4) * 16 * (size_t)D + (size_t)(n >> 3) * 128 + (m & 7) * 16 + ((m >> 3) & 1) * 8 + (n & 7);
}
template <typename H> __device__ __forceinline__ void split_f32(float x, unsigned& hi16, unsigned& lo8);   // lo8: the remainder's two's complement in the low 8 bits
template <typename H> __device__ __forceinline__ float join_f32(unsigned hi16, int lo8);                  // lo8 sign-extended
template <> __device__ __forceinline__ void split_f32<bf16_t>(float x, unsigned& hi16, unsigned& lo8) {
  const unsigned u = __builtin_bit_cast(unsigned, x), t = u + 0x8000u;
  hi16 = t >> 16;
  const int r = (int)(u - (t & 0xffff0000u));                    // [-32768, 32767]
  const int q = (r + 128) >> 8;                                  // round to nearest, [-128, 128]
  lo8 = (unsigned)(q > 127 ? 127 : q) & 0xffu;
}
template <> __device__ __forceinline__ float join_f32<bf16_t>(unsigned hi16, int lo8) {
  return __builtin_bit_cast(float, (hi16 << 16) + (unsigned)(lo8 << 8));
}
__device__ __forceinline__ unsigned f16_plane_exp(float hf) {   // biased fp32 exponent of the f16 value, floor 127 - 14
  const unsigned eb = (__builtin_bit_cast(unsigned, hf) >> 23) & 0xffu;
  return eb < 113u ? 113u : eb;
}
template <> __device__ __forceinline__ void split_f32<f16_t>(float x, unsigned& hi16, unsigned& lo8) {
  const f16_t h = (f16_t)sat_f16(x);
  const float hf = (float)h;
  const float scale = __builtin_bit_cast(float, (272u - f16_plane_exp(hf)) << 23);   // 2^(18 - E)
  const float r = __builtin_amdgcn_fmed3f(__builtin_rintf((x - hf) * scale), -128.0f, 127.0f);
  hi16 = (unsigned)__builtin_bit_cast(unsigned short, h);
  lo8 = (unsigned)(int)r & 0xffu;
}
template <> __device__ __forceinline__ float join_f32<f16_t>(unsigned hi16, int lo8) {
  const float hf = (float)__builtin_bit_cast(f16_t, (unsigned short)hi16);
  const float unit = __builtin_bit_cast(float, (f16_plane_exp(hf) - 18u) << 23);     // 2^(E - 18)
  return fmaf((float)lo8, unit, hf);
}
// byte k (0 .. 3) of a word, sign-extended
__device__ __forceinline__ int sbyte(unsigned w, int k) { return (int)(w << (24 - 8 * k)) >> 24; }
// 4 consecutive columns n .. n + 3 (n % 4 == 0) of row m: 8 bytes of hi, 4 bytes of lo
template <typename H>
__device__ __forceinline__ void store4_split(unsigned short* hi, unsigned char* lo, size_t m, unsigned n, unsigned D, float a, float b, float c, float d) {
  unsigned h0, h1, h2, h3, l0, l1, l2, l3;
  split_f32<H>(a, h0, l0); split_f32<H>(b, h1, l1); split_f32<H>(c, h2, l2); split_f32<H>(d, h3, l3);
  *reinterpret_cast<uint2*>(hi + m * D + n) = make_uint2(h0 | (h1 << 16), h2 | (h3 << 16));
  *reinterpret_cast<unsigned*>(lo + lo_plane_off(m, n, D)) = l0 | (l1 << 8) | (l2 << 16) | (l3 << 24);
}
template <typename H>
__device__ __forceinline__ float4 load4_split(const unsigned short* hi, const unsigned char* lo, size_t m, unsigned n, unsigned D) {
  const uint2 h = *reinterpret_cast<const uint2*>(hi + m * D + n);
  const unsigned l = *reinterpret_cast<const unsigned*>(lo + lo_plane_off(m, n, D));
  return make_float4(join_f32<H>(h.x & 0xffffu, sbyte(l, 0)), join_f32<H>(h.x >> 16, sbyte(l, 1)),
                     join_f32<H>(h.y & 0xffffu, sbyte(l, 2)), join_f32<H>(h.y >> 16, sbyte(l, 3)));
}

constexpr int kLnSlice = 64;
// The fold is cut in two so that a kernel prologue can put its LDS-DMA fills BETWEEN the loads and their use (round 6): the loads of a
// batch of (up to) 16 slices -- 8 float4, indices clamped, unconditional, issued back to back (a per-load predicate makes hipcc branch
// around each load and wait for it separately: one L2 / HBM round trip per load) -- and the update sequence over them.  ln_combine is
// the two run one after the other; all three are ONE source of arithmetic, so a row's rstd is the same bits wherever it is folded.
__device__ __forceinline__ void ln_load16(const float* __restrict__ st, int ns, int j0, float4 (&v)[8]) {
  // ns is even (widths are multiples of 128): two slices per 16-byte load
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int j = j0 + 2 * k < ns ? j0 + 2 * k : ns - 2;
    v[k] = *reinterpret_cast<const float4*>(st + 2 * j);
  }
}
// The arithmetic is spelled out operation by operation with contraction OFF: the fold is inlined into several kernels (gemm.h, the
// small-M kernel, the fused text kernel) whose results must agree bit for bit (packed captions = padded captions, fused = two kernels),
// and hipcc's default contraction fuses `a * b + c` differently from one inlining context to the next (round 6: a twin of this function
// with compile-time bounds differed in the last bit of rstd).
__device__ __forceinline__ void ln_fold16(const float4 (&v)[8], int ns, int j0, float& n, float& mu, float& m2) {
#pragma clang fp contract(off)
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    if (j0 + 2 * k < ns) {   // wave-uniform
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const float sj = h ? v[k].z : v[k].x, qj = h ? v[k].w : v[k].y;
        const float mj = sj * (1.0f / kLnSlice);                       // exact (power of two)
        const float nn = n + (float)kLnSlice;                           // exact (small integers)
        const float d = mj - mu;
        const float w = (float)kLnSlice * __builtin_amdgcn_rcpf(nn);   // nn = 64 (j + 1): 1-ulp reciprocal of a small integer
        mu = __builtin_fmaf(d, w, mu);
        const float t = (d * d) * n;
        m2 = m2 + __builtin_fmaf(t, w, qj);
        n = nn;
      }
    }
  }
}
__device__ __forceinline__ float ln_rstd(float m2, float inv_d, float eps) {
  return __builtin_amdgcn_rsqf(__builtin_fmaf(m2, inv_d, eps));   // v_rsq_f32, 1 ulp: far below the bf16 rounding of what it scales
}
__device__ __forceinline__ void ln_combine(const float* __restrict__ st, int ns, float inv_d, float eps, float& mean,
                                           float& rstd) {
  float n = 0.f, mu = 0.f, m2 = 0.f;
  for (int j0 = 0; j0 < ns; j0 += 16) {
    float4 v[8];
    ln_load16(st, ns, j0, v);
    ln_fold16(v, ns, j0, n, mu, m2);
  }
  mean = mu;
  rstd = ln_rstd(m2, inv_d, eps);
}

// QuickGELU: x * sigmoid(1.702 x)  (transformers/activations.py:117-123)
// fp32 engine: libm exp + IEEE divide; bf16 engine: v_exp_f32 + v_rcp_f32 (1 ulp), far below bf16 rounding
template <bool kAccurate> __device__ __forceinline__ float quick_gelu(float x) {
  if constexpr (kAccurate) return x / (1.0f + expf(-1.702f * x));
  else return x * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * x));
}

}  // namespace plipmi
