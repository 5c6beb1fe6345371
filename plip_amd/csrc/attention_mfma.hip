// attention_mfma.hip -- bf16 MFMA attention for the short CLIP sequences (S <= 128:
// 50 vision tokens, 77 text tokens), one workgroup per (image|caption, head), one
// wavefront per block of 32 queries, v_mfma_f32_32x32x16_bf16 for both products.
//
//   scores^T = K Q^T   (operands swapped): a lane owns ONE query column (lane&31) and
//                      16 keys per 32-key tile, so the softmax row reductions are a
//                      register max/sum plus one exchange with lane^32 -- no LDS.
//   O^T = V^T P^T      the contraction index (key) may be permuted freely as long as
//                      both MFMA operands use the same permutation.  Choosing the
//                      permutation that the QK^T accumulator layout already has
//                      (slot jj of lane group g  <->  key (jj&3) + 4g + 8(jj>>2) + 16s)
//                      makes the P operand a plain register pack: no cross-lane shuffle.
//                      V^T comes from an LDS copy of V transposed at staging time
//                      (row stride S_pad+4 bf16 -> conflict-free ds_read_b64).
//   Softmax statistics, the running sum and the 1/l normalisation are fp32
//   (modeling_clip.py:271); P is rounded to bf16 only as the MFMA operand.
#include "kernels.h"

namespace plipmi {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <int KT>  // 32-key tiles: S <= 32*KT
__global__ __launch_bounds__(64 * KT) void attention_mfma_kernel(const bf16_t* __restrict__ qkv,
                                                                 bf16_t* __restrict__ out, int S, int H, int causal,
                                                                 const int64_t* __restrict__ key_mask) {
  constexpr int SP = 32 * KT;   // padded sequence
  constexpr int VLD = SP + 4;   // Vt row stride (bf16): (SP/2 + 2) dwords, odd multiple of 2 -> all 64 banks
  // Q and K rows are staged through LDS in whole 128-byte lines (8 lanes x 16 B per row) instead of being loaded
  // fragment-shaped (16 B from each of 32 rows per instruction, which costs the texture-address unit 2x the
  // time for the same bytes); the LDS image uses the GEMM's XOR swizzle so the fragment ds_read_b128 are
  // conflict-free.  V is transposed on the way in.
  __shared__ __attribute__((aligned(16))) char Qs[SP * 128];
  __shared__ __attribute__((aligned(16))) char Ks[SP * 128];
  __shared__ __attribute__((aligned(16))) bf16_t Vt[64 * VLD];
  __shared__ unsigned long long mk[4];

  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int D = H * 64, ld = 3 * D;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bf16_t* base = qkv + (size_t)b * S * ld + h * 64;

  // key validity bits (sequence padding and the tokenizer's attention_mask): key = tid
  {
    const bool ok = tid < S && (key_mask == nullptr || key_mask[(size_t)b * S + tid] != 0);
    const unsigned long long bits = __ballot(ok);
    if (lane == 0) mk[wave] = bits;
    if (KT < 4 && tid < 4 - KT) mk[KT + tid] = 0ull;
  }
  for (int e = tid; e < SP * 8; e += 64 * KT) {
    const int row = e >> 3, c = e & 7;
    const int rg = row < S ? row : S - 1;
    const bf16_t* src = base + (size_t)rg * ld + c * 8;
    const u32x4 q16 = *reinterpret_cast<const u32x4*>(src);
    const u32x4 k16 = *reinterpret_cast<const u32x4*>(src + D);
    const bf16x8 v8 = *reinterpret_cast<const bf16x8*>(src + 2 * D);
    const int off = row * 128 + ((c ^ ((row >> 1) & 7)) << 4);
    *reinterpret_cast<u32x4*>(Qs + off) = q16;
    *reinterpret_cast<u32x4*>(Ks + off) = k16;
#pragma unroll
    for (int i = 0; i < 8; ++i) Vt[(c * 8 + i) * VLD + row] = v8[i];
  }
  __syncthreads();

  const int q0 = wave * 32;
  if (q0 >= S) return;
  const int lrow = lane & 31, hi = lane >> 5;
  const int qidx = q0 + lrow;
  const int lsw = (lrow >> 1) & 7;

  // Q fragments (B operand): Q[query = lrow][d = 16ks + 8hi .. +7]
  u32x4 qf[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    qf[ks] = *reinterpret_cast<const u32x4*>(Qs + (q0 + lrow) * 128 + (((ks * 2 + hi) ^ lsw) << 4));

  // scores^T tiles
  f32x16 sc[KT];
  const unsigned long long m0 = mk[0], m1 = mk[1];
  float rmax = -INFINITY;
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const bool live = !(causal && 32 * t > q0 + 31);  // wave-uniform: tile entirely above the diagonal
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[t][r] = 0.f;
    if (live) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const u32x4 kf = *reinterpret_cast<const u32x4*>(Ks + (32 * t + lrow) * 128 + (((ks * 2 + hi) ^ lsw) << 4));
        sc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[ks]),
                                                        sc[t], 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi;
      const unsigned long long word = key < 64 ? m0 : m1;
      const bool ok = live && ((word >> (key & 63)) & 1ull) && (!causal || key <= qidx);
      sc[t][r] = ok ? sc[t][r] : -INFINITY;
      rmax = fmaxf(rmax, sc[t][r]);
    }
  }
  rmax = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
  const float m_use = (rmax == -INFINITY) ? 0.f : rmax;
  float rsum = 0.f;
#pragma unroll
  for (int t = 0; t < KT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sc[t][r] = __expf(sc[t][r] - m_use);
      rsum += sc[t][r];
    }
  rsum += __shfl_xor(rsum, 32, 64);

  // O^T = V^T P^T
  f32x16 acc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    if (causal && 32 * t > q0 + 31) continue;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 pf;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) pf[jj] = (bf16_t)sc[t][8 * s2 + jj];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const bf16_t* vr = Vt + (dt * 32 + lrow) * VLD + 32 * t + 16 * s2 + 4 * hi;
        const u32x2 v0 = *reinterpret_cast<const u32x2*>(vr);
        const u32x2 v1 = *reinterpret_cast<const u32x2*>(vr + 8);
        const u32x4 vf = {v0[0], v0[1], v1[0], v1[1]};
        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), pf, acc[dt], 0, 0, 0);
      }
    }
  }
  // The accumulator layout gives a lane one query ROW (like the GEMM): write the normalised 32 x 64 bf16 tile
  // into this wave's own (already consumed) Q rows of the LDS image, then store whole 128-byte rows.
  {
    const float inv = 1.0f / rsum;
    char* orow_lds = Qs + (q0 + lrow) * 128;
    typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int d = dt * 32 + 8 * q4 + 4 * hi;                 // 4 consecutive head-dim columns
        const int c = d >> 3;                                    // 16-byte chunk, half (d&4) inside it
        const bf16x4 v = {(bf16_t)(acc[dt][4 * q4 + 0] * inv), (bf16_t)(acc[dt][4 * q4 + 1] * inv),
                          (bf16_t)(acc[dt][4 * q4 + 2] * inv), (bf16_t)(acc[dt][4 * q4 + 3] * inv)};
        *reinterpret_cast<bf16x4*>(orow_lds + ((c ^ lsw) << 4) + (d & 4) * 2) = v;
      }
    __builtin_amdgcn_wave_barrier();
    const int c = lane & 7;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = q0 + it * 8 + (lane >> 3);
      const u32x4 v = *reinterpret_cast<const u32x4*>(Qs + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
      if (r < S) *reinterpret_cast<u32x4*>(out + ((size_t)b * S + r) * D + h * 64 + c * 8) = v;
    }
  }
}

hipError_t launch_attention_mfma(const void* qkv, void* out, int B, int S, int H, int causal, const int64_t* key_mask,
                                 hipStream_t s) {
  if (S > 128 || S <= 0) return hipErrorNotSupported;
  const int KT = (S + 31) / 32;
  const dim3 grid(B * H), block(64 * KT);
#define PLIPMI_ATT(K) \
  hipLaunchKernelGGL(attention_mfma_kernel<K>, grid, block, 0, s, (const bf16_t*)qkv, (bf16_t*)out, S, H, causal, key_mask)
  switch (KT) {
    case 1: PLIPMI_ATT(1); break;
    case 2: PLIPMI_ATT(2); break;
    case 3: PLIPMI_ATT(3); break;
    default: PLIPMI_ATT(4); break;
  }
#undef PLIPMI_ATT
  return hipGetLastError();
}

}  // namespace plipmi
