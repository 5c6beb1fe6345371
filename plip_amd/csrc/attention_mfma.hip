// attention_mfma.hip -- MFMA attention for the two 16-bit engines (bf16 / f16).  Short CLIP sequences (S <= 128: 50 vision tokens, 77 text tokens)
// take the single-pass kernel: one workgroup per (image|caption, head), one wavefront per block of 32 queries,
// v_mfma_f32_32x32x16_bf16 / _f16 for both products.  Longer ones (ViT-B/16, ViT-L/14[@336]) take the chunked
// online-softmax kernel further down, built from the same pieces.
//
//   scores^T = K Q^T   (operands swapped): a lane owns ONE query column (lane&31) and
//                      16 keys per 32-key tile, so the softmax row reductions are a
//                      register max/sum plus one exchange with lane^32 -- no LDS.
//   O^T = V^T P^T      the contraction index (key) may be permuted freely as long as
//                      both MFMA operands use the same permutation.  Choosing the
//                      permutation that the QK^T accumulator layout already has
//                      (slot jj of lane group g  <->  key (jj&3) + 4g + 8(jj>>2) + 16s)
//                      makes the P operand a plain register pack: no cross-lane shuffle.
//                      V^T comes from a row-major LDS copy of V through the hardware transpose read
//                      ds_read_b64_tr_b16 (two [keys][32] images with 64-byte rows -> conflict-free).
//   Softmax statistics, the running sum and the 1/l normalisation are fp32
//   (modeling_clip.py:271); P is rounded to the operand type only as the MFMA operand.
#include "kernels.h"

namespace plipmi {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) short i16x4;

// V stays row-major in LDS (whole 16-byte pieces straight from the global rows) as two [keys][32 columns] images,
// one per 32-wide head-dim tile, 64-byte rows.  ds_read_b64_tr_b16 then hands lane (d = lane&31, g = lane>>5) the
// four keys k0+4g .. k0+4g+3 of column d -- exactly one half of the V^T MFMA fragment under the key permutation
// described above -- with every 32-lane half of the instruction touching each of the 64 banks once (4 rows x 64 B).
// Per-lane byte offset inside an image; the (tile, sub-block, image) terms are compile-time immediates.
__device__ __forceinline__ int vtr_lane_offset(int lrow, int hi) {
  return (((lrow & 15) >> 2) + 4 * hi) * 64 + ((lrow & 3) * 4 + (lrow >> 4) * 16) * 2;
}
template <typename H>
__device__ __forceinline__ typename half_traits<H>::x8 vtr_fragment(const char* vs, int byte_off) {
  typedef __attribute__((address_space(3))) i16x4* lds_ptr;
  const i16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(vs + byte_off));
  const i16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(vs + byte_off + 8 * 64));  // keys +8
  typedef __attribute__((ext_vector_type(8))) short i16x8;
  const i16x8 v = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
  return __builtin_bit_cast(typename half_traits<H>::x8, v);
}

// One (sample, head) problem per workgroup.  (Two problems per workgroup with a register prefetch of the next K / V rows
// was measured slower in round 2 -- registers: three waves per SIMD instead of six to eight -- and is gone.)
// (amdgpu_waves_per_eu(4, 4): with the default heuristics hipcc parks the MFMA accumulators in AGPRs and pays 96 v_accvgpr
// copies per wave to get the scores and the outputs back into VGPRs; told it may use 128 registers it needs 77 (S <= 64) / 93
// (S <= 96) plain VGPRs, no AGPRs, 10-12 % fewer VALU instructions, and one more wave per SIMD than before.)
template <typename HT, int KT>  // 32-key tiles: S <= 32*KT
__global__ __launch_bounds__(64 * KT) __attribute__((amdgpu_waves_per_eu(4, 4))) void attention_mfma_kernel(const HT* __restrict__ qkv, HT* __restrict__ out, int S, int H,
                                                                 int causal, const int64_t* __restrict__ key_mask,
                                                                 const int* __restrict__ cu /* packed rows: sample b owns rows
                                                                 cu[b] .. cu[b+1]-1 of qkv / out (nullptr: b*S .. b*S+S-1) */) {
  using X8 = typename half_traits<HT>::x8;
  using X4 = typename half_traits<HT>::x4;
  constexpr int SP = 32 * KT;   // padded sequence
  constexpr int NT = 64 * KT;
  constexpr int NP = SP * 8 / NT;   // 16-byte pieces of K (and of V) per thread: 4
  // K rows are staged through LDS in whole 128-byte lines (8 lanes x 16 B per row) instead of being loaded
  // fragment-shaped (16 B from each of 32 rows per instruction, which costs the texture-address unit 2x the
  // time for the same bytes); the LDS image uses the GEMM's XOR swizzle so the fragment ds_read_b128 are
  // conflict-free.  V keeps its row-major form (two [SP][32] images) and is transposed by the LDS read itself.
  // The second V image starts 64 bytes past a 128-byte boundary: the eight 16-byte pieces of one V row (four per image)
  // then cover all 32 write banks once instead of hitting the same 16 twice (SQ_LDS_BANK_CONFLICT, round 1: half of
  // the 3.9e5 conflict cycles per launch; the other half was the output staging below).
  constexpr int VIMG = SP * 64 + 64;
  __shared__ __attribute__((aligned(16))) char Ks[SP * 128];
  __shared__ __attribute__((aligned(16))) char Vs[2 * VIMG];
  __shared__ unsigned long long mk[4];

  const int D = H * 64, ld = 3 * D;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q0 = wave * 32;
  const int lrow = lane & 31, hi = lane >> 5;
  const int qidx = q0 + lrow;
  const int lsw = (lrow >> 1) & 7;
  const int bh = blockIdx.x;

  u32x4 kreg[NP], vreg[NP];
  {
    const int b = bh / H, h = bh - b * H;
    // packed captions: this problem's rows start at cu[b] and there are cu[b+1] - cu[b] of them (queries AND keys); the
    // tokenizer mask keeps its [B, S] layout.  Keys past the caption's last row are masked like sequence padding, so a
    // query's result is the one the padded layout gives it, bit for bit (a masked key contributes an exact zero).
    const int row0 = __builtin_amdgcn_readfirstlane(cu ? cu[b] : b * S);
    const int Sb = __builtin_amdgcn_readfirstlane(cu ? cu[b + 1] - row0 : S);
    const bool active = q0 < Sb;
    const HT* base = qkv + (size_t)row0 * ld + h * 64;
#pragma unroll
    for (int i = 0; i < NP; ++i) {   // this thread's 16-byte pieces of the problem's K and V rows -> registers
      const int e = tid + i * NT, row = e >> 3, c = e & 7;
      // wave-uniform base + 32-bit lane offset (a problem spans < 4 GiB): scalar-base global loads, no 64-bit lane arithmetic
      const unsigned off = ((unsigned)(row < Sb ? row : Sb - 1) * (unsigned)ld + (unsigned)(c * 8)) * (unsigned)sizeof(HT);
      kreg[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(base + D) + off);
      vreg[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(base + 2 * D) + off);
    }
    // key validity bits (sequence padding and the tokenizer's attention_mask): key = tid
    {
      const bool ok = tid < Sb && (key_mask == nullptr || key_mask[(size_t)b * S + tid] != 0);
      const unsigned long long bits = __ballot(ok);
      if (lane == 0) mk[wave] = bits;
      if (KT < 4 && tid < 4 - KT) mk[KT + tid] = 0ull;
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int e = tid + i * NT, row = e >> 3, c = e & 7;
      *reinterpret_cast<u32x4*>(Ks + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = kreg[i];
      *reinterpret_cast<u32x4*>(Vs + (c >> 2) * VIMG + row * 64 + (c & 3) * 16) = vreg[i];
    }
    // Q fragments (B operand) straight from global memory: Q[query = lrow][d = 16ks + 8hi .. +7].  Only this wave reads
    // these rows, so an LDS image of Q would only cost residency (8-16 KB per workgroup = a third of its LDS).
    u32x4 qf[4];
    {
      const unsigned qoff = ((unsigned)(qidx < Sb ? qidx : Sb - 1) * (unsigned)ld + (unsigned)(hi * 8)) * (unsigned)sizeof(HT);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(base) + qoff + ks * 16 * sizeof(HT));
    }
    __syncthreads();

    // scores^T tiles
    f32x16 sc[KT];
    // key validity words as scalars (wave-uniform): 32 keys per tile
    const unsigned long long m0 = mk[0], m1 = mk[1];
    const unsigned vw[4] = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)m0),
                            (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(m0 >> 32)),
                            (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)m1),
                            (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(m1 >> 32))};
    float rmax = -INFINITY;
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      const bool live = active && !(causal && 32 * t > q0 + 31);  // wave-uniform: tile entirely above the diagonal
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[t][r] = 0.f;
      if (live) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const u32x4 kf = *reinterpret_cast<const u32x4*>(Ks + (32 * t + lrow) * 128 + (((ks * 2 + hi) ^ lsw) << 4));
          sc[t] = half_traits<HT>::mfma32(__builtin_bit_cast(X8, kf), __builtin_bit_cast(X8, qf[ks]), sc[t]);
        }
      }
      // The keys this lane may use in the tile, as ONE 32-bit word: validity (padding / attention_mask) AND, under the causal
      // mask, keys 32t + j <= query  <=>  j <= d = query - 32t.  Slot r of the accumulator holds key 32t + 4hi + (r&3) + 8(r>>2):
      // after a shift by 4hi its bit sits at a compile-time position, so masking a score is a 1-bit field extract (0 / -1),
      // an AND that turns it into 0.0 / -inf, and an add -- no compares, no 64-bit shifts, no branches.  (The two-operation form,
      // a bit select against -inf, was miscompiled by this hipcc: its v_bitop3_b32 folding mixed the elements' masks.)
      // (round 6) a tile every query of the wave may use entirely -- all 32 keys valid and, under the causal mask, wholly below the
      // diagonal: nothing to mask (wave-uniform test; the 50-token vision tower's first key tile is always one)
      if (live && vw[t] == 0xffffffffu && (!causal || 32 * t + 31 <= q0)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rmax = fmaxf(rmax, sc[t][r]);
        continue;
      }
      unsigned bits = live ? vw[t] : 0u;
      if (causal) {
        const int d = qidx - 32 * t;
        bits &= d < 0 ? 0u : (d >= 31 ? 0xffffffffu : (2u << d) - 1u);
      }
      const unsigned nbits = ~(bits >> (4 * hi));        // 1 = masked
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)nbits, (r & 3) + 8 * (r >> 2), 1);   // masked ? 0xffffffff : 0
        sc[t][r] += __builtin_bit_cast(float, m & 0xff800000u);                                       // + (-inf) or + 0
        rmax = fmaxf(rmax, sc[t][r]);
      }
    }
    __syncthreads();  // every wave has its scores: the K rows may now be reused as output staging
    if (active) {
      rmax = fmaxf(rmax, __shfl_xor(rmax, 32, 64));
      const float m_use = (rmax == -INFINITY) ? 0.f : rmax;
      const float m2 = m_use * 1.4426950408889634f;       // exp(x - m) = 2^(x log2 e - m log2 e): one FMA + v_exp_f32 per score
      float rsum = 0.f;
#pragma unroll
      for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sc[t][r] = __builtin_amdgcn_exp2f(fmaf(sc[t][r], 1.4426950408889634f, -m2));
          rsum += sc[t][r];
        }
      rsum += __shfl_xor(rsum, 32, 64);

      // O^T = V^T P^T
      f32x16 acc[2];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
      const char* vlane = Vs + vtr_lane_offset(lrow, hi);
#pragma unroll
      for (int t = 0; t < KT; ++t) {
        if (causal && 32 * t > q0 + 31) continue;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          X8 pf;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) pf[jj] = (HT)sc[t][8 * s2 + jj];
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const X8 vf = vtr_fragment<HT>(vlane, dt * VIMG + (32 * t + 16 * s2) * 64);
            acc[dt] = half_traits<HT>::mfma32(vf, pf, acc[dt]);
          }
        }
      }
      // The accumulator layout gives a lane one query ROW (like the GEMM): write the normalised 32 x 64 bf16 tile
      // into the K rows of its own query range (every wave is past its scores), then store whole 128-byte rows.
      // Rows 2k and 2k+1 share a swizzle slot; odd rows keep their two 8-byte halves exchanged, so the 16 lanes of a
      // ds_write_b64 group (16 consecutive rows, same column chunk) touch 16 distinct 8-byte bank pairs.
      const float inv = 1.0f / rsum;
      char* orow_lds = Ks + (q0 + lrow) * 128;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int d = dt * 32 + 8 * q4 + 4 * hi;                 // 4 consecutive head-dim columns
          const int c = d >> 3;                                    // 16-byte chunk, half (d&4) inside it
          const X4 v = {from_f32<HT>(acc[dt][4 * q4 + 0] * inv), from_f32<HT>(acc[dt][4 * q4 + 1] * inv),
                        from_f32<HT>(acc[dt][4 * q4 + 2] * inv), from_f32<HT>(acc[dt][4 * q4 + 3] * inv)};
          *reinterpret_cast<X4*>(orow_lds + ((c ^ lsw) << 4) + ((((d >> 2) ^ lrow) & 1) << 3)) = v;
        }
      __builtin_amdgcn_wave_barrier();
      const int c = lane & 7;
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = q0 + it * 8 + (lane >> 3);
        const u32x4 raw = *reinterpret_cast<const u32x4*>(Ks + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
        const u32x4 v = (r & 1) ? u32x4{raw[2], raw[3], raw[0], raw[1]} : raw;
        if (r < Sb) *reinterpret_cast<u32x4*>(out + ((size_t)row0 + r) * D + h * 64 + c * 8) = v;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// S > 128 (ViT-B/16: 197 tokens, ViT-L/14: 257, ViT-L/14@336: 577): the same two MFMA products, streamed over
// 128-key chunks with an online softmax.  One workgroup = 128 queries of one (sample, head) (4 waves x 32), grid
// (B*H, ceil(S/128)).  Because a lane owns one query column in BOTH accumulator layouts (scores^T and O^T), the
// running max / running sum / rescale of the online softmax are per-lane scalars: no cross-lane traffic beyond the
// one lane^32 exchange per chunk.  K and V^T chunks are staged through LDS exactly like the short-sequence kernel.
// Round 4: the NEXT chunk's K / V rows travel global -> registers while the current chunk is multiplied (32 VGPRs of
// staging; the LDS write waits behind the chunk's barrier), so a workgroup no longer stands still for a global round trip
// per chunk; and the short kernel's per-score arithmetic -- one 32-bit mask word per tile, additive -inf, exp as FMA + v_exp_f32.
// ---------------------------------------------------------------------------------------------
template <typename HT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void attention_flash_kernel(const HT* __restrict__ qkv, HT* __restrict__ out,
                                                              int S, int H, int causal,
                                                              const int64_t* __restrict__ key_mask) {
  using X8 = typename half_traits<HT>::x8;
  using X4 = typename half_traits<HT>::x4;
  constexpr int SP = 128;
  constexpr int NP = SP * 8 / 256;   // 16-byte pieces of a chunk's K (and V) rows per thread: 4
  constexpr float kLog2e = 1.4426950408889634f;
  __shared__ __attribute__((aligned(16))) char Qs[SP * 128];
  __shared__ __attribute__((aligned(16))) char Ks[SP * 128];
  __shared__ __attribute__((aligned(16))) char Vs[2 * SP * 64];
  __shared__ unsigned long long mk[2];

  const int bh = blockIdx.x, b = bh / H, h = bh - b * H;
  const int qbase = blockIdx.y * SP;
  const int D = H * 64, ld = 3 * D;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const HT* base = qkv + (size_t)b * S * ld + h * 64;     // wave-uniform (blockIdx only)

  int nchunks = (S + SP - 1) / SP;
  if (causal) nchunks = min(nchunks, (qbase + SP - 1) / SP + 1);  // chunks entirely above the diagonal

  // this thread's pieces of a chunk: rows (tid + i*256) >> 3, 16-byte column c = tid & 7 (the same for every i)
  u32x4 kreg[NP], vreg[NP];
  auto fetch = [&](int kbase) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int row = (tid + i * 256) >> 3, c = tid & 7;
      const int rg = kbase + row < S ? kbase + row : S - 1;       // rows past the sequence end re-read the last row (masked)
      // wave-uniform base + 32-bit lane offset (a sample's qkv rows span < 4 GiB): scalar-base loads, no 64-bit lane arithmetic
      const unsigned off = ((unsigned)rg * (unsigned)ld + (unsigned)(c * 8)) * (unsigned)sizeof(HT);
      kreg[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(base + D) + off);
      vreg[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(base + 2 * D) + off);
    }
  };
  fetch(0);

  {  // the query block, whole 128-byte lines, GEMM swizzle: all four loads of a thread in flight together (rows past the
     // sequence end re-read its last row; their outputs are never stored)
    u32x4 q16[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int row = (tid + i * 256) >> 3, c = tid & 7;
      const int rg = qbase + row < S ? qbase + row : S - 1;
      q16[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(base) +
                                               ((unsigned)rg * (unsigned)ld + (unsigned)(c * 8)) * (unsigned)sizeof(HT));
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int row = (tid + i * 256) >> 3, c = tid & 7;
      *reinterpret_cast<u32x4*>(Qs + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = q16[i];
    }
  }

  const int q0 = wave * 32;  // inside the block
  const bool active = qbase + q0 < S;
  const int lrow = lane & 31, hi = lane >> 5;
  const int qidx = qbase + q0 + lrow;
  const int lsw = (lrow >> 1) & 7;

  const char* vlane = Vs + vtr_lane_offset(lrow, hi);
  u32x4 qf[4];
  f32x16 acc[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;  // l_run: this lane's half of the row sum

  for (int ch = 0; ch < nchunks; ++ch) {
    const int kbase = ch * SP;
    __syncthreads();  // everyone is done reading the previous chunk
    if (tid < SP) {
      const int key = kbase + tid;
      const bool ok = key < S && (key_mask == nullptr || key_mask[(size_t)b * S + key] != 0);
      const unsigned long long bits = __ballot(ok);
      if (lane == 0) mk[wave] = bits;
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {   // the chunk fetched during the previous iteration (or the prologue) -> LDS
      const int row = (tid + i * 256) >> 3, c = tid & 7;
      *reinterpret_cast<u32x4*>(Ks + row * 128 + ((c ^ ((row >> 1) & 7)) << 4)) = kreg[i];
      *reinterpret_cast<u32x4*>(Vs + (c >> 2) * (SP * 64) + row * 64 + (c & 3) * 16) = vreg[i];
    }
    if (ch + 1 < nchunks) fetch(kbase + SP);   // in flight while this chunk is multiplied
    __syncthreads();
    if (!active) continue;
    if (ch == 0) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *reinterpret_cast<const u32x4*>(Qs + (q0 + lrow) * 128 + (((ks * 2 + hi) ^ lsw) << 4));
    }
    // two 64-key halves per staged chunk: 32 score registers live instead of 64 (3 waves/SIMD), one online-softmax
    // update per half
    const unsigned long long mhalf[2] = {mk[0], mk[1]};
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      if (kbase + 64 * hf >= S) break;                           // wave-uniform: past the sequence end
      if (causal && kbase + 64 * hf > qbase + q0 + 31) break;  // wave-uniform: the rest is above the diagonal
      f32x16 sc[2];
      float cmax = -INFINITY;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * hf + tt;
        const bool live = kbase + 32 * t < S && !(causal && kbase + 32 * t > qbase + q0 + 31);  // wave-uniform
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[tt][r] = 0.f;
        if (live) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const u32x4 kf = *reinterpret_cast<const u32x4*>(Ks + (32 * t + lrow) * 128 + (((ks * 2 + hi) ^ lsw) << 4));
            sc[tt] = half_traits<HT>::mfma32(__builtin_bit_cast(X8, kf), __builtin_bit_cast(X8, qf[ks]), sc[tt]);
          }
        }
        // the keys this lane may use in the tile as ONE 32-bit word (validity AND causal limit), then per score a 1-bit field
        // extract, an AND that makes it 0.0 / -inf, an add -- the short kernel's form (see there for why not a bit select)
        unsigned bits = live ? (unsigned)(mhalf[hf] >> (32 * tt)) : 0u;
        if (causal) {
          const int d = qidx - (kbase + 32 * t);
          bits &= d < 0 ? 0u : (d >= 31 ? 0xffffffffu : (2u << d) - 1u);
        }
        if (__ballot(bits != 0xffffffffu) == 0ull) {
          // every key of the tile is usable by every query of the wave (the common case away from the sequence end and the
          // diagonal): no masking arithmetic at all
#pragma unroll
          for (int r = 0; r < 16; ++r) cmax = fmaxf(cmax, sc[tt][r]);
        } else {
          const unsigned nbits = ~(bits >> (4 * hi));        // 1 = masked; slot r holds key 32t + 4hi + (r&3) + 8(r>>2)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)nbits, (r & 3) + 8 * (r >> 2), 1);
            sc[tt][r] += __builtin_bit_cast(float, m & 0xff800000u);
            cmax = fmaxf(cmax, sc[tt][r]);
          }
        }
      }
      cmax = fmaxf(cmax, __shfl_xor(cmax, 32, 64));
      const float m_new = fmaxf(m_run, cmax);
      const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
      const float m2 = m_use * kLog2e;
      const bool moved = m_new != m_run;    // (NaN-free: -inf != -inf is false, a first finite maximum is a move)
      // a lane whose maximum did not move rescales by EXACTLY 1 -- exp2 of the FMA's rounding residual is 1 +- 3e-6, and l_run
      // (always rescaled) would otherwise drift against acc (rescaled only when some lane moved): ADVICE r4
      const float scale = !moved ? 1.f : (m_run == -INFINITY) ? 0.f : __builtin_amdgcn_exp2f(fmaf(m_run, kLog2e, -m2));  // acc, l_run are 0 while m_run = -inf
      m_run = m_new;
      if (__ballot(moved) != 0ull) {        // wave-uniform: once the running maxima have settled the 32 rescaling multiplies go
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[dt][r] *= scale;
      }
      float csum = 0.f;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int t = 2 * hf + tt;
        if (kbase + 32 * t >= S || (causal && kbase + 32 * t > qbase + q0 + 31)) continue;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          X8 pf;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            const float pv = __builtin_amdgcn_exp2f(fmaf(sc[tt][8 * s2 + jj], kLog2e, -m2));   // exp(x - m): one FMA + v_exp_f32
            csum += pv;
            pf[jj] = (HT)pv;
          }
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const X8 vf = vtr_fragment<HT>(vlane, dt * (SP * 64) + (32 * t + 16 * s2) * 64);
            acc[dt] = half_traits<HT>::mfma32(vf, pf, acc[dt]);
          }
        }
      }
      l_run = l_run * scale + csum;
    }
  }
  if (!active) return;  // no workgroup barrier below
  {
    const float rsum = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / rsum;
    char* orow_lds = Qs + (q0 + lrow) * 128;  // this wave's own Q rows: consumed into qf long ago
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int d = dt * 32 + 8 * q4 + 4 * hi;
        const int c = d >> 3;
        const X4 v = {from_f32<HT>(acc[dt][4 * q4 + 0] * inv), from_f32<HT>(acc[dt][4 * q4 + 1] * inv),
                      from_f32<HT>(acc[dt][4 * q4 + 2] * inv), from_f32<HT>(acc[dt][4 * q4 + 3] * inv)};
        *reinterpret_cast<X4*>(orow_lds + ((c ^ lsw) << 4) + (d & 4) * 2) = v;
      }
    __builtin_amdgcn_wave_barrier();
    const int c = lane & 7;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int rl = q0 + it * 8 + (lane >> 3);
      const int r = qbase + rl;
      const u32x4 v = *reinterpret_cast<const u32x4*>(Qs + rl * 128 + ((c ^ ((rl >> 1) & 7)) << 4));
      if (r < S) *reinterpret_cast<u32x4*>(out + ((size_t)b * S + r) * D + h * 64 + c * 8) = v;
    }
  }
}

template <typename HT>
static hipError_t launch_attention_mfma_t(const void* qkv, void* out, int B, int S, int H, int causal, const int64_t* key_mask,
                                          hipStream_t s, const int* cu) {
  if (S > 128) {
    if (cu) return hipErrorInvalidValue;   // packed rows: short-sequence kernel only (captions are 77 tokens at most)
    hipLaunchKernelGGL(attention_flash_kernel<HT>, dim3(B * H, (S + 127) / 128), dim3(256), 0, s, (const HT*)qkv,
                       (HT*)out, S, H, causal, key_mask);
    return hipGetLastError();
  }
  const int KT = (S + 31) / 32;
  const dim3 grid(B * H), block(64 * KT);
#define PLIPMI_ATT(K) \
  hipLaunchKernelGGL((attention_mfma_kernel<HT, K>), grid, block, 0, s, (const HT*)qkv, (HT*)out, S, H, causal, key_mask, cu)
  switch (KT) {
    case 1: PLIPMI_ATT(1); break;
    case 2: PLIPMI_ATT(2); break;
    case 3: PLIPMI_ATT(3); break;
    default: PLIPMI_ATT(4); break;
  }
#undef PLIPMI_ATT
  return hipGetLastError();
}

hipError_t launch_attention_mfma(const void* qkv, void* out, int dtype, int B, int S, int H, int causal, const int64_t* key_mask,
                                 hipStream_t s, const int* cu) {
  if (S <= 0 || (dtype != 1 && dtype != 2)) return hipErrorInvalidValue;
  return dtype == 1 ? launch_attention_mfma_t<bf16_t>(qkv, out, B, S, H, causal, key_mask, s, cu)
                    : launch_attention_mfma_t<f16_t>(qkv, out, B, S, H, causal, key_mask, s, cu);
}

}  // namespace plipmi
