// attention.hip -- softmax(Q K^T [+ causal / padding mask]) V per (image|caption, head)
// over the fused qkv activation (CLIPAttention.forward, modeling_clip.py:298-335 with
// eager_attention_forward :259-277; softmax statistics in fp32 like :271).
//
// head_dim is 64 for every CLIP/PLIP variant and the sequences are tiny (50 vision
// tokens, 77 text tokens), so a whole (b, h) problem fits one workgroup:
//   * attention_valu_kernel  -- exact fp32 arithmetic (the fp32 engine's kernel, and the
//                               on-device checker for the MFMA kernel): one query row per
//                               lane, K/V tiles of 64 keys broadcast from LDS, flash-style
//                               online softmax in chunks of 8 keys.
//   * attention_mfma_kernel  -- bf16 MFMA kernel (attention_mfma.hip).
#include "kernels.h"

namespace plipmi {

hipError_t launch_attention_mfma(const void* qkv, void* out, int dtype, int B, int S, int H, int causal, const int64_t* key_mask,
                                 hipStream_t s, const int* cu);

constexpr int kDh = 64;      // head dim
constexpr int kKeyTile = 64;  // keys staged in LDS per step

template <typename T, int kMaxThreads>
__global__ __launch_bounds__(kMaxThreads) void attention_valu_kernel(const T* __restrict__ qkv, T* __restrict__ out, int S,
                                                              int H, int causal, const int64_t* __restrict__ key_mask) {
  __shared__ __attribute__((aligned(16))) float Ks[kKeyTile][kDh];
  __shared__ __attribute__((aligned(16))) float Vs[kKeyTile][kDh];
  __shared__ int Mk[kKeyTile];

  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int D = H * kDh, ld = 3 * D;
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int i = tid;                       // query row of this lane
  const int ic = i < S ? i : S - 1;        // clamped (lanes past S compute a duplicate, never store)
  const int wave_last = min(S - 1, (__builtin_amdgcn_readfirstlane(tid >> 6)) * 64 + 63);
  const T* base = qkv + (size_t)b * S * ld + h * kDh;

  float q[kDh], acc[kDh];
  {
    const T* qr = base + (size_t)ic * ld;
#pragma unroll
    for (int d = 0; d < kDh; d += 4) {
      const float4 t = load4(qr + d);
      q[d] = t.x; q[d + 1] = t.y; q[d + 2] = t.z; q[d + 3] = t.w;
    }
  }
#pragma unroll
  for (int d = 0; d < kDh; ++d) acc[d] = 0.f;
  float m = -INFINITY, l = 0.f;

  // keys needed by anyone in the block: all of them, or up to the last row under the causal mask
  const int keys_block = S;
  for (int j0 = 0; j0 < keys_block; j0 += kKeyTile) {
    __syncthreads();  // previous tile fully consumed
    for (int e = tid; e < kKeyTile * (kDh / 4); e += nthreads) {
      const int j = e / (kDh / 4), d = (e - j * (kDh / 4)) * 4;
      const int jg = min(j0 + j, S - 1);
      const T* kr = base + (size_t)jg * ld + D + d;
      *reinterpret_cast<float4*>(&Ks[j][d]) = load4(kr);
      *reinterpret_cast<float4*>(&Vs[j][d]) = load4(kr + D);
    }
    for (int j = tid; j < kKeyTile; j += nthreads) {
      const int jg = j0 + j;
      Mk[j] = (jg < S) && (key_mask == nullptr || key_mask[(size_t)b * S + jg] != 0);
    }
    __syncthreads();

    // wave-uniform number of keys of this tile that any row of this wave may attend to
    int nk = min(kKeyTile, S - j0);
    if (causal) nk = min(nk, wave_last - j0 + 1);
    for (int c0 = 0; c0 < nk; c0 += 8) {
      float s[8];
      float cmax = -INFINITY;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int j = c0 + jj;  // < 64 always (nk <= 64 and tiles are 64 rows)
        const float4* kr = reinterpret_cast<const float4*>(&Ks[j][0]);
        float a = 0.f;
#pragma unroll
        for (int d4 = 0; d4 < kDh / 4; ++d4) {
          const float4 kv = kr[d4];
          a = fmaf(q[4 * d4 + 0], kv.x, a);
          a = fmaf(q[4 * d4 + 1], kv.y, a);
          a = fmaf(q[4 * d4 + 2], kv.z, a);
          a = fmaf(q[4 * d4 + 3], kv.w, a);
        }
        const bool ok = Mk[j] && (!causal || (j0 + j) <= i);
        s[jj] = ok ? a : -INFINITY;
        cmax = fmaxf(cmax, s[jj]);
      }
      const float m_new = fmaxf(m, cmax);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;  // everything masked so far: keep zeros
      const float corr = expf(m - m_use);
      m = m_new;
      l *= corr;
#pragma unroll
      for (int d = 0; d < kDh; ++d) acc[d] *= corr;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const float p = expf(s[jj] - m_use);
        l += p;
        const float4* vr = reinterpret_cast<const float4*>(&Vs[c0 + jj][0]);
#pragma unroll
        for (int d4 = 0; d4 < kDh / 4; ++d4) {
          const float4 vv = vr[d4];
          acc[4 * d4 + 0] = fmaf(p, vv.x, acc[4 * d4 + 0]);
          acc[4 * d4 + 1] = fmaf(p, vv.y, acc[4 * d4 + 1]);
          acc[4 * d4 + 2] = fmaf(p, vv.z, acc[4 * d4 + 2]);
          acc[4 * d4 + 3] = fmaf(p, vv.w, acc[4 * d4 + 3]);
        }
      }
    }
  }
  if (i < S) {
    const float inv = 1.0f / l;
    T* orow = out + ((size_t)b * S + i) * D + h * kDh;
#pragma unroll
    for (int d = 0; d < kDh; d += 4) store4(orow + d, acc[d] * inv, acc[d + 1] * inv, acc[d + 2] * inv, acc[d + 3] * inv);
  }
}

hipError_t launch_attention(const void* qkv, void* out, int dtype, int B, int S, int H, int causal,
                            const int64_t* key_mask, int impl, hipStream_t s, const int* cu) {
  if (B <= 0) return hipSuccess;
  if (impl == 1) {
    if (dtype != 1 && dtype != 2) return hipErrorInvalidValue;
    return launch_attention_mfma(qkv, out, dtype, B, S, H, causal, key_mask, s, cu);
  }
  if (cu) return hipErrorInvalidValue;   // packed rows are an MFMA-kernel form
  const int threads = ((S + 63) / 64) * 64;
  if (threads > 1024) return hipErrorInvalidValue;  // S <= 1024 (ViT-L/14@336 has 577 tokens)
  const dim3 grid(B * H), block(threads);
  // <= 256 threads: the register allocator may use 256 VGPRs (q[64] + acc[64] live); the 1024-thread
  // build (S > 256, e.g. ViT-L/14@336) trades some spills for the larger block.
#define PLIPMI_ATTN(T, MAXT) \
  hipLaunchKernelGGL((attention_valu_kernel<T, MAXT>), grid, block, 0, s, (const T*)qkv, (T*)out, S, H, causal, key_mask)
  if (threads <= 256) { if (dtype == 1) PLIPMI_ATTN(bf16_t, 256); else if (dtype == 2) PLIPMI_ATTN(f16_t, 256); else PLIPMI_ATTN(float, 256); }
  else                { if (dtype == 1) PLIPMI_ATTN(bf16_t, 1024); else if (dtype == 2) PLIPMI_ATTN(f16_t, 1024); else PLIPMI_ATTN(float, 1024); }
#undef PLIPMI_ATTN
  return hipGetLastError();
}

}  // namespace plipmi
