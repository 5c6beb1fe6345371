// gemm_skinny.hip -- the NT GEMM for SMALL M (latency regime):  C = epilogue(A[M,K] . W[N,K]^T), 16-bit operands (bf16 / f16).
//
// Where the big-tile kernel (gemm.h) walks K one 64-deep tile after the other on a handful of workgroups -- M = 256
// rows x N = 768 is 12 workgroups of 128x128, each 48 dependent K iterations for fc2 = 35-55 us -- this kernel spends
// the chip's width on N, on M and on K at once:
//   * one workgroup (NW = 4 or 8 waves) owns a 32 x 64 output tile and its waves split K NW ways (fixed-order reduction
//     through LDS: deterministic, independent of M); a wave requests NB 64-deep blocks of both operands (all of its share
//     when that is <= 3 blocks) before the first MFMA, so its K walk is 1-3 memory round trips, not one per block;
//   * operands come straight from L2 as MFMA fragments -- no LDS staging, nothing to synchronise in the K loop.  The k
//     index inside an MFMA may be permuted freely as long as both operands share the permutation, so a lane fetches 64
//     CONTIGUOUS bytes of its row per 64-deep block (lanes l and l+32 together one whole 128-byte line) and feeds
//     MFMA t of the block from the t-th 16-byte piece;
//   * the same fused epilogues as the big kernel's LayerNorm-folded family: rstd * acc + c2 (-> bf16, optional
//     QuickGELU), and the in-place fp32 residual update that also emits bf16(x) and the row's {sum, centred M2} of the
//     64-column slice (= exactly this tile's width).
// Used by the 16-bit engines (a) for the LAST block of each tower, whose out_proj / fc1 / fc2 only ever matter for the pooled
// row of each sample (engine.hip: run_last_block_pooled): M = batch size, at every batch size; and (b), round 4, for EVERY GEMM
// of a small batch (engine.hip: latency path, batches of at most plipmi_engine::latency_batch samples -- the reference drives
// its heads at batch 8, plip.py:90-91): the big tiles walk K serially whatever M is (fc2: 48 dependent K tiles = 45 us for
// 400 rows), this kernel splits it 8 ways.  A row's result never depends on the other rows of ITS call, so embeddings are
// bit-identical across batch sizes within each of the two regimes; between them they differ by fp32 summation order.
#include "gemm.h"

namespace plipmi {

namespace {

constexpr int SK_BM = 32, SK_BN = 64, SK_PITCH = 68;   // floats per LDS row of a partial tile (272 B: conflict-free b128)

template <typename T, int EPI, int NW, int NB>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_kernel(const GemmParams p) {
  using X8 = typename half_traits<T>::x8;
  __shared__ __attribute__((aligned(16))) float part[NW][SK_BM][SK_PITCH];
  __shared__ float rs_s[SK_BM];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 31, hi = lane >> 5;
  const int n0 = blockIdx.x * SK_BN, m0 = blockIdx.y * SK_BM;
  const int kw = p.K / NW;                                   // this wave's share of K (a multiple of 64 * NB)
  const int mr = m0 + lrow < p.M ? m0 + lrow : p.M - 1;      // M edge: re-read the last row, stores are masked
  const T* ap = reinterpret_cast<const T*>(p.A) + (size_t)mr * p.lda + wave * kw + 32 * hi;
  const T* w0 = reinterpret_cast<const T*>(p.W) + (size_t)(n0 + lrow) * p.ldw + wave * kw + 32 * hi;
  const T* w1 = w0 + (size_t)32 * p.ldw;

  if constexpr (epi_is_ln(EPI)) {                            // rstd of the tile's rows, while the first operands travel
    if (tid < SK_BM) {
      const int r = m0 + tid < p.M ? m0 + tid : p.M - 1;
      float mu, rs;
      ln_combine(p.ln_stats + (size_t)r * p.ln_ns * 2, p.ln_ns, p.ln_inv_d, p.ln_eps, mu, rs);
      rs_s[tid] = rs;
    }
  }

  f32x16 acc0, acc1;
#pragma unroll
  for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
#pragma unroll 1
  for (int kk = 0; kk < kw; kk += 64 * NB) {
    u32x4 xa[NB][4], wa[NB][4], wb[NB][4];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        xa[b][t] = *reinterpret_cast<const u32x4*>(ap + kk + 64 * b + 8 * t);
        wa[b][t] = *reinterpret_cast<const u32x4*>(w0 + kk + 64 * b + 8 * t);
        wb[b][t] = *reinterpret_cast<const u32x4*>(w1 + kk + 64 * b + 8 * t);
      }
    // keep every request of the batch AHEAD of the first MFMA (left alone the scheduler sinks each load next to its use:
    // one memory round trip per block instead of one per batch)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        acc0 = half_traits<T>::mfma32(__builtin_bit_cast(X8, wa[b][t]), __builtin_bit_cast(X8, xa[b][t]), acc0);
        acc1 = half_traits<T>::mfma32(__builtin_bit_cast(X8, wb[b][t]), __builtin_bit_cast(X8, xa[b][t]), acc1);
      }
  }
  // partial tile of this wave, row-major: acc[4q+e] = C[m = lrow][n = 8q + 4hi + e] of its 32-column half
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    *reinterpret_cast<f32x4*>(&part[wave][lrow][8 * q + 4 * hi]) = f32x4{acc0[4 * q], acc0[4 * q + 1], acc0[4 * q + 2], acc0[4 * q + 3]};
    *reinterpret_cast<f32x4*>(&part[wave][lrow][32 + 8 * q + 4 * hi]) = f32x4{acc1[4 * q], acc1[4 * q + 1], acc1[4 * q + 2], acc1[4 * q + 3]};
  }
  __syncthreads();
  // threads 0..255 -> row r, 8 consecutive columns; the NW K partials are added in a fixed order (pairs, then pairs of pairs)
  if (tid >= 256) return;                          // waves 4.. (NW = 8) have handed their partials over
  const int r = tid >> 3, c8 = (tid & 7) * 8;
  float v[8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    f32x4 sum4[NW / 4];
#pragma unroll
    for (int g = 0; g < NW / 4; ++g) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(&part[4 * g + 0][r][c8 + 4 * h]), b = *reinterpret_cast<const f32x4*>(&part[4 * g + 1][r][c8 + 4 * h]);
      const f32x4 c = *reinterpret_cast<const f32x4*>(&part[4 * g + 2][r][c8 + 4 * h]), d = *reinterpret_cast<const f32x4*>(&part[4 * g + 3][r][c8 + 4 * h]);
#pragma unroll
      for (int e = 0; e < 4; ++e) sum4[g][e] = (a[e] + b[e]) + (c[e] + d[e]);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[4 * h + e] = NW == 8 ? sum4[0][e] + sum4[NW / 4 - 1][e] : sum4[0][e];
  }
  const int m = m0 + r, n = n0 + c8;
  const bool in_range = m < p.M;
  // (EPI_PATCH: p.bias is the position table [(np + 1), N]; its row 0 is read here and unused)
  const float4 b0 = *reinterpret_cast<const float4*>(p.bias + n), b1 = *reinterpret_cast<const float4*>(p.bias + n + 4);
  const float bias[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
  if constexpr (epi_is_colwise(EPI)) {
    const float rs = epi_is_ln(EPI) ? rs_s[r] : 1.0f;
    X8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float y = fmaf(rs, v[e], bias[e]);
      if constexpr (EPI == EPI_BIAS_QGELU || EPI == EPI_QGELU_LN) y = quick_gelu<false>(y);
      o[e] = from_f32<T>(y);
    }
    if (in_range) *reinterpret_cast<X8*>(reinterpret_cast<T*>(p.C) + (size_t)m * p.ldc + n) = o;
  } else if constexpr (EPI == EPI_PATCH) {
    // patch row m = img * np + pp -> token row img * (np + 1) + 1 + pp of the fp32 embedding rows, plus its position row
    const int mm = in_range ? m : p.M - 1;
    const int img = mm / p.np, pp = mm - img * p.np;
    const float4 p0 = *reinterpret_cast<const float4*>(p.bias + (size_t)(1 + pp) * p.N + n), p1 = *reinterpret_cast<const float4*>(p.bias + (size_t)(1 + pp) * p.N + n + 4);
    float* crow = reinterpret_cast<float*>(p.C) + ((size_t)img * (p.np + 1) + 1 + pp) * p.ldc + n;
    if (in_range) {
      *reinterpret_cast<float4*>(crow) = make_float4(v[0] + p0.x, v[1] + p0.y, v[2] + p0.z, v[3] + p0.w);
      *reinterpret_cast<float4*>(crow + 4) = make_float4(v[4] + p1.x, v[5] + p1.y, v[6] + p1.z, v[7] + p1.w);
    }
  } else if constexpr (EPI == EPI_RESID_SPLIT) {
    // the residual stream as its two planes (gemm.h EPI_RESID_SPLIT; common.h split_f32): join, (x + bias) + product, split, statistics of
    // the tile's 64-column slice -- a lane owns 8 consecutive columns of a row, 8 lanes the slice, as in the big kernel
    const size_t mr = (size_t)(in_range ? m : p.M - 1);
    const size_t off = mr * p.ldc + n;
    const size_t loff = lo_plane_off(mr, (unsigned)n, (unsigned)p.ldc);     // this row's 8 bytes of a band's 16-byte piece
    const u32x4 h = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(p.xb_out) + off);
    const uint2 l = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned char*>(p.lo_io) + loff);
    float o[8], s = 0.f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[2 * e] = join_f32<T>(h[e] & 0xffffu, sbyte(e < 2 ? l.x : l.y, (2 * e) & 3));
      o[2 * e + 1] = join_f32<T>(h[e] >> 16, sbyte(e < 2 ? l.x : l.y, (2 * e + 1) & 3));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (o[e] + bias[e]) + v[e];
    s = ((o[0] + o[1]) + (o[2] + o[3])) + ((o[4] + o[5]) + (o[6] + o[7]));
    const float ssum = row8_sum(s);
    const float mj = ssum * (1.0f / kLnSlice);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = o[e] - mj; q = fmaf(d, d, q); }
    const float m2 = row8_sum(q);
    if (in_range) {
      u32x4 ho;
      unsigned lb[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned ha, hb;
        split_f32<T>(o[2 * e], ha, lb[2 * e]);
        split_f32<T>(o[2 * e + 1], hb, lb[2 * e + 1]);
        ho[e] = ha | (hb << 16);
      }
      *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(p.xb_out) + off) = ho;
      *reinterpret_cast<uint2*>(reinterpret_cast<unsigned char*>(p.lo_io) + loff) =
          make_uint2(lb[0] | (lb[1] << 8) | (lb[2] << 16) | (lb[3] << 24), lb[4] | (lb[5] << 8) | (lb[6] << 16) | (lb[7] << 24));
      if ((tid & 7) == 0) *reinterpret_cast<float2*>(p.st_out + ((size_t)m * (p.N / kLnSlice) + n0 / kLnSlice) * 2) = make_float2(ssum, m2);
    }
  } else {
    static_assert(epi_is_resid(EPI), "skinny epilogues: bias / QuickGELU (optionally LayerNorm-folded), patch rows and the residual forms");
    float* crow = reinterpret_cast<float*>(p.C) + (size_t)(in_range ? m : p.M - 1) * p.ldc + n;
    const float4 x0 = *reinterpret_cast<const float4*>(crow), x1 = *reinterpret_cast<const float4*>(crow + 4);
    const float xo[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    float o[8], s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { o[e] = xo[e] + (v[e] + bias[e]); s += o[e]; }   // same association as the big kernel: x + (acc + bias)
    if (in_range) {
      *reinterpret_cast<float4*>(crow) = make_float4(o[0], o[1], o[2], o[3]);
      *reinterpret_cast<float4*>(crow + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
    if constexpr (EPI == EPI_RESID_EMIT) {
      const float ssum = row8_sum(s);                          // the tile is one 64-column slice wide: 8 lanes per row
      const float mj = ssum * (1.0f / kLnSlice);
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = o[e] - mj; q += d * d; }
      const float m2 = row8_sum(q);
      if (in_range) {
        X8 ob;
#pragma unroll
        for (int e = 0; e < 8; ++e) ob[e] = from_f32<T>(o[e]);
        *reinterpret_cast<X8*>(reinterpret_cast<T*>(p.xb_out) + (size_t)m * p.ldc + n) = ob;
        if ((tid & 7) == 0) *reinterpret_cast<float2*>(p.st_out + ((size_t)m * (p.N / kLnSlice) + n0 / kLnSlice) * 2) = make_float2(ssum, m2);
      }
    }
  }
}

template <typename T, int EPI, int NW, int NB>
int launch_skinny_nb(const GemmParams& p, hipStream_t s) {
  const dim3 grid(p.N / SK_BN, (p.M + SK_BM - 1) / SK_BM);
  hipLaunchKernelGGL((gemm_skinny_kernel<T, EPI, NW, NB>), grid, dim3(64 * NW), 0, s, p);
  return (int)hipGetLastError();
}
// K split: 8 waves when each still gets whole 64-deep blocks, else 4; blocks requested together: 3, 2 or 1.  The choice
// depends on K only (never on M): a row's bits do not depend on the batch it arrives in.
template <typename T, int EPI>
int launch_skinny(const GemmParams& p, hipStream_t s) {
  const int blocks = p.K / 64;
  if (blocks % 8 == 0) {
    const int nb = blocks / 8;
    if (nb % 3 == 0) return launch_skinny_nb<T, EPI, 8, 3>(p, s);
    if (nb % 2 == 0) return launch_skinny_nb<T, EPI, 8, 2>(p, s);
    return launch_skinny_nb<T, EPI, 8, 1>(p, s);
  }
  const int nb = blocks / 4;
  if (nb % 3 == 0) return launch_skinny_nb<T, EPI, 4, 3>(p, s);
  if (nb % 2 == 0) return launch_skinny_nb<T, EPI, 4, 2>(p, s);
  return launch_skinny_nb<T, EPI, 4, 1>(p, s);
}

}  // namespace

bool gemm_skinny_supports(int epi, int M, int N, int K) {
  const bool epi_ok = epi == EPI_BIAS || epi == EPI_BIAS_QGELU || epi == EPI_BIAS_RESID || epi == EPI_BIAS_LN ||
                      epi == EPI_QGELU_LN || epi == EPI_RESID_EMIT || epi == EPI_RESID_SPLIT || epi == EPI_PATCH;
  return epi_ok && M > 0 && N % SK_BN == 0 && K % 256 == 0;
}

template <typename T>
static int launch_skinny_epi(int epi, const GemmParams& p, hipStream_t s) {
  switch (epi) {
    case EPI_BIAS: return launch_skinny<T, EPI_BIAS>(p, s);
    case EPI_BIAS_QGELU: return launch_skinny<T, EPI_BIAS_QGELU>(p, s);
    case EPI_BIAS_RESID: return launch_skinny<T, EPI_BIAS_RESID>(p, s);
    case EPI_BIAS_LN: return launch_skinny<T, EPI_BIAS_LN>(p, s);
    case EPI_QGELU_LN: return launch_skinny<T, EPI_QGELU_LN>(p, s);
    case EPI_RESID_EMIT: return launch_skinny<T, EPI_RESID_EMIT>(p, s);
    case EPI_RESID_SPLIT: return launch_skinny<T, EPI_RESID_SPLIT>(p, s);
    case EPI_PATCH: return launch_skinny<T, EPI_PATCH>(p, s);
    default: return (int)hipErrorInvalidValue;
  }
}

int gemm_launch_skinny(int dtype, int epi, const GemmParams& p, hipStream_t s, const char** kernel_name) {
  if (p.M <= 0) return 0;
  if ((dtype != 1 && dtype != 2) || !gemm_skinny_supports(epi, p.M, p.N, p.K) || p.lda % 8 || p.ldw % 8) return (int)hipErrorInvalidValue;
  static const char* names[2][EPI_COUNT] = {
      {"gemm_skinny<bf16,32x64_splitk,bias>", "gemm_skinny<bf16,32x64_splitk,bias_qgelu>", "gemm_skinny<bf16,32x64_splitk,bias_resid>",
       nullptr, "gemm_skinny<bf16,32x64_splitk,patch>", "gemm_skinny<bf16,32x64_splitk,ln_bias>", "gemm_skinny<bf16,32x64_splitk,ln_qgelu>",
       "gemm_skinny<bf16,32x64_splitk,resid_emit>", "gemm_skinny<bf16,32x64_splitk,resid_split>"},
      {"gemm_skinny<f16,32x64_splitk,bias>", "gemm_skinny<f16,32x64_splitk,bias_qgelu>", "gemm_skinny<f16,32x64_splitk,bias_resid>",
       nullptr, "gemm_skinny<f16,32x64_splitk,patch>", "gemm_skinny<f16,32x64_splitk,ln_bias>", "gemm_skinny<f16,32x64_splitk,ln_qgelu>",
       "gemm_skinny<f16,32x64_splitk,resid_emit>", "gemm_skinny<f16,32x64_splitk,resid_split>"}};
  if (kernel_name) *kernel_name = names[dtype - 1][epi];
  return dtype == 1 ? launch_skinny_epi<bf16_t>(epi, p, s) : launch_skinny_epi<f16_t>(epi, p, s);
}

}  // namespace plipmi
