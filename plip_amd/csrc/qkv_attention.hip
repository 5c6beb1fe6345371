// qkv_attention.hip -- the text tower's q/k/v projection WITH the scaled-dot-product attention in its epilogue: one kernel,
// the fused activation `qkv` [B*S, 3D] never exists in memory (north_star: "fused QKV projection + scaled-dot-product
// attention with LDS-staged K/V tiles").  Replaces, per block, the LayerNorm-folded q/k/v GEMM
// (gemm.h EPI_BIAS_LN; modeling_clip.py:293-296 with layer_norm1 :370 folded in) + attention_mfma_kernel (:259-335) for
// sequences of 65 .. 80 tokens (CLIP's 77-token captions).
//
// Tile = FOUR captions x ONE head: 4 x 80 padded rows = 320 rows of the residual stream's operand plane against the head's
// 192 weight rows (q_h | k_h | v_h, 64 each).  bs = 256, 8 heads: 64 x 8 = 512 workgroups = exactly two rounds of 256 CUs
// (the vision tower's 50 tokens do not tile like this: 258- / 516-tile covers, DESIGN.md section 4.4.1 -- it keeps the two kernels).
//   * K loop: 8 waves as 4 x 2, wave tile 80 x 96 = five 16-row x six 16-column v_mfma_f32_16x16x32 tiles (120 accumulator
//     registers), BK = 64, two LDS stages fed by LDS-DMA, fragments of the next K step read between the MFMAs of this one,
//     the tile's barrier in front of its second step (the ring form of gemm.h on two stages).  Same K order per output as
//     every gemm.h tile, so q/k/v are the bits the unfused GEMM writes.
//   * epilogue 1: y = rstd[m] * acc + c2[n] (LayerNorm folded, gemm.h EPI_BIAS_LN), rounded to the operand type, written to
//     LDS as the Q / K / V images of the four captions (the staging LDS is free by then): Q, K [336][128 B] with the GEMM's
//     XOR swizzle, V as two row-major [336][64 B] images for ds_read_b64_tr_b16 -- the layouts of attention_mfma.hip.
//   * epilogue 2: two waves per caption; under the causal mask query block 0 needs one 32-key tile, block 1 two, block 2
//     three, so one wave takes blocks 0 + 1 and the other block 2 (3 + 3 tile products).  Arithmetic per query block is
//     attention_mfma_kernel's, instruction for instruction (scores^T = K Q^T, additive -inf masks from one 32-bit word per
//     tile, exp2 softmax in fp32, O^T = V^T P^T from registers) -- the attention output is BIT-IDENTICAL to the two-kernel
//     path's (tests/test_gpu_attention.py::test_fused_qkv_attention_matches_the_two_kernels).
#include <mutex>

#include "gemm.h"
#include "kernels.h"

namespace plipmi {

namespace {

constexpr int kCPT = 4;            // captions per tile
constexpr int kSPad = 80;          // rows per caption in the tile (S <= 80)
constexpr int kBM = kCPT * kSPad;  // 320
constexpr int kBN = 192;           // q_h | k_h | v_h
constexpr int kNT = 512;
constexpr int kABytes = kBM * 128, kWBytes = kBN * 128, kStage = kABytes + kWBytes;   // 64 KB per stage
constexpr int kImgRows = kBM + 16;                      // the last caption's third key tile reads 16 rows past the tile
constexpr int kQs = 0, kKs = kImgRows * 128, kVs = 2 * kImgRows * 128;
constexpr int kVImg = kImgRows * 64 + 64;               // second V image 64 B past a 128-B boundary (attention_mfma.hip)
constexpr int kImgBytes = kVs + 2 * kVImg;
static_assert(kImgBytes <= 2 * kStage, "the Q / K / V images reuse the staging LDS");
constexpr int kLnRows = 2 * kStage;                     // rstd of the tile's rows, 320 floats
constexpr int kMaskWords = kLnRows + kBM * 4;           // key validity: 4 captions x 4 x 32 bits
constexpr int kC2 = kMaskWords + kCPT * 4 * 4;           // the head's 192 biases (c2 of q_h | k_h | v_h)
constexpr int kLdsBytes = kC2 + kBN * 4;

struct QkvAttnParams {
  const void* A;        // [B*S, D] operand plane of the residual stream (LayerNorm input, rounded)
  const void* W;        // [3D, D] q | k | v weights with LayerNorm's gain and centring folded in (q rows pre-scaled by 1/8)
  const float* c2;      // [3D]
  const float* stats;   // [B*S, D/64, 2] partials of the LayerNorm input rows
  void* out;            // [B*S, D] attention output
  const int64_t* key_mask;   // [B, S] or nullptr
  int B, S, H, D, causal;
  float ln_inv_d, ln_eps;
  // test hook (plipmi_qkv_attention): 8 x u64 per workgroup {start, prologue done, K loop done, images written, end (s_memtime),
  // tile id, 0, start | lifetime << 32 (s_memrealtime, 100 MHz)}; nullptr on the product path
  unsigned long long* trace = nullptr;
};

typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) short i16x4;

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

template <typename T>
__global__ __launch_bounds__(kNT) __attribute__((amdgpu_waves_per_eu(2, 2))) void qkv_attention_kernel(const QkvAttnParams p) {
  using X8 = typename half_traits<T>::x8;
  using X4 = typename half_traits<T>::x4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int S = p.S, D = p.D;

  // ---- tile assignment: an XCD (blockIdx % 8) works through whole caption groups, every head of a group back to back, so
  // the group's 320 A rows are fetched into that XCD's L2 once
  const int ngrp = (p.B + kCPT - 1) / kCPT, nblk = ngrp * p.H;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, xi = bid >> 3, xq = nblk >> 3, xr = nblk & 7;
  const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
  const int grp = lid / p.H, head = lid - grp * p.H;
  const int cap0 = grp * kCPT;

  // ---- staging addresses (gemm.h layout: 128-byte LDS rows, chunk c of row r in slot c ^ ((r >> 1) & 7))
  const int srow = tid >> 3;
  const int schunk = (tid & 7) ^ ((srow >> 1) & 7);
  constexpr int PA = kBM * 8 / kNT, PW = kBN * 8 / kNT;   // 5 + 3 LDS-DMA pieces per thread and K tile
  const i32x4 rs_a = make_buffer_rsrc(p.A), rs_w = make_buffer_rsrc(p.W);
  unsigned a_off[PA], w_off[PW];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    const int r = i * (kNT / 8) + srow;                   // tile row: caption r / 80, position r % 80
    int c = cap0 + r / kSPad, pos = r % kSPad;
    c = c < p.B ? c : p.B - 1;                            // captions past the batch / positions past S: re-read a live row,
    pos = pos < S ? pos : S - 1;                          // nobody stores what is computed from it
    a_off[i] = (unsigned)(((size_t)(c * S + pos) * D + schunk * 8) * sizeof(T));
  }
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int r = i * (kNT / 8) + srow;                   // 0..63 q_h, 64..127 k_h, 128..191 v_h
    const int row = (r >> 6) * D + head * 64 + (r & 63);
    w_off[i] = (unsigned)(((size_t)row * D + schunk * 8) * sizeof(T));
  }
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
  unsigned koff = 0;
  auto fill_a = [&](int buf) __attribute__((always_inline)) {      // 4 + 1 requests
    const unsigned base = lds0 + buf * kStage + wave * 1024;
    glds16_buf_n<4, 0, kNT * 16, 2 * kNT * 16, 3 * kNT * 16>(base, koff, rs_a, a_off[0], rs_a, a_off[1], rs_a, a_off[2], rs_a, a_off[3]);
    glds16_buf_n<1, 4 * kNT * 16>(base, koff, rs_a, a_off[4], rs_a, a_off[4], rs_a, a_off[4], rs_a, a_off[4]);
  };
  auto fill_w = [&](int buf) __attribute__((always_inline)) {      // 3 requests; the tile's last: advance K
    const unsigned base = lds0 + buf * kStage + wave * 1024;
    glds16_buf_n<3, kABytes, kABytes + kNT * 16, kABytes + 2 * kNT * 16>(base, koff, rs_w, w_off[0], rs_w, w_off[1], rs_w, w_off[2], rs_w, w_off[2]);
    koff += 128;
  };

  // ---- fragments: 16 rows x 32 k per ds_read_b128 (lane = row l16, 16-byte chunk 4 s + g16 of the row)
  const int l16 = lane & 15, g16 = lane >> 4;
  int foff16[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) foff16[s2] = l16 * 128 + (((4 * s2 + g16) ^ (l16 >> 1)) << 4);
  constexpr int MI2 = 5, NI2 = 6;
  const int a_tile = wm * 80 * 128, w_tile = kABytes + wn * 96 * 128;
  f32x4 acc[MI2][NI2];
#pragma unroll
  for (int i = 0; i < MI2; ++i)
#pragma unroll
    for (int j = 0; j < NI2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 xf[2][MI2], wf[2][NI2];
  auto read_frag = [&](const char* sb, int s2, int b, int r) __attribute__((always_inline)) {   // r: 0..4 x, 5..10 w
    if (r < MI2) xf[b][r] = *reinterpret_cast<const u32x4*>(sb + a_tile + r * 16 * 128 + foff16[s2]);
    else wf[b][r - MI2] = *reinterpret_cast<const u32x4*>(sb + w_tile + (r - MI2) * 16 * 128 + foff16[s2]);
  };
  // one K step: 30 MFMAs in serpentine order (one operand changes per MFMA); meanwhile the 11 fragments of step (sbn, sn)
  // are read into the other buffer, one per MFMA slot, and (fill >= 0) the next tile's 8 LDS-DMA requests go out in two batches
  auto step = [&](int b, const char* sbn, int sn, int fill) __attribute__((always_inline)) {
#pragma unroll
    for (int ii = 0; ii < MI2; ++ii)
#pragma unroll
      for (int jj = 0; jj < NI2; ++jj) {
        const int n = ii * NI2 + jj;
        const int js = (ii & 1) ? NI2 - 1 - jj : jj;
        acc[ii][js] = half_traits<T>::mfma16(__builtin_bit_cast(X8, wf[b][js]), __builtin_bit_cast(X8, xf[b][ii]), acc[ii][js]);
        if (sn >= 0 && n < MI2 + NI2) read_frag(sbn, sn, b ^ 1, n);
        if (fill >= 0 && n == 12) fill_a(fill);
        if (fill >= 0 && n == 20) fill_w(fill);
        __builtin_amdgcn_sched_barrier(0);
      }
  };

  // ---- prologue: tiles 0 and 1 on their way, rstd of the tile's rows and the captions' key-validity words parked in LDS meanwhile
  const int KT = D / 64;
  unsigned long long* trace = p.trace ? p.trace + (size_t)bid * 8 : nullptr;
  unsigned long long trace_real0 = 0;
  if (trace && tid == 0) {
    trace[0] = __builtin_amdgcn_s_memtime();
    trace_real0 = __builtin_amdgcn_s_memrealtime();
    trace[5] = lid;
  }
  // (round 6) Prologue order.  The compiler waits for ITS loads -- statistics partials, biases, mask words -- with counted vmcnt and cannot
  // see the inline-asm LDS-DMA fills: any such wait placed behind fills drains them first (3.45 us of prologue in round 5 with both
  // tiles' fills in front of this block).  Tile 0's fills go out first (the barrier needs them anyway), the block's own loads next,
  // tile 1's fills behind them: 3.0 us.  (Requesting the statistics BEFORE tile 0's fills, as the GEMM kernels now do, measured level
  // to +1 us here -- 32 more live registers across the fill in a 229-register kernel -- and is not used.)
  fill_a(0);
  fill_w(0);
  {
    float* ln_rows = reinterpret_cast<float*>(smem + kLnRows);
    const int ns = D / kLnSlice;
    for (int r = tid; r < kBM; r += kNT) {
      int c = cap0 + r / kSPad, pos = r % kSPad;
      c = c < p.B ? c : p.B - 1;
      pos = pos < S ? pos : S - 1;
      float mu, rs;
      ln_combine(p.stats + (size_t)(c * S + pos) * ns * 2, ns, p.ln_inv_d, p.ln_eps, mu, rs);
      ln_rows[r] = rs;
    }
    // the head's biases: epilogue 1 reads them from LDS (no global round trips between the K loop and the image writes)
    if (tid < kBN) reinterpret_cast<float*>(smem + kC2)[tid] = p.c2[(tid >> 6) * D + head * 64 + (tid & 63)];
    // key validity bits (sequence padding and the tokenizer's attention_mask), 32 keys per word: wave w covers caption w >> 1,
    // keys 64 (w & 1) + lane
    unsigned* mkw = reinterpret_cast<unsigned*>(smem + kMaskWords);
    const int c = cap0 + (wave >> 1), key = 64 * (wave & 1) + lane;
    const bool ok = c < p.B && key < S && (p.key_mask == nullptr || p.key_mask[(size_t)c * S + key] != 0);
    const unsigned long long bits = __ballot(ok);
    if (lane == 0) {
      mkw[(wave >> 1) * 4 + 2 * (wave & 1)] = (unsigned)bits;
      mkw[(wave >> 1) * 4 + 2 * (wave & 1) + 1] = (unsigned)(bits >> 32);
    }
  }
  if (KT > 1) { fill_a(1); fill_w(1); }
  if (KT > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PW) : "memory");   // tile 0 has landed (in-order retirement), tile 1 may still fly
  else wait_vm0();
  __syncthreads();
  if (trace && tid == 0) trace[1] = __builtin_amdgcn_s_memtime();

  // ---- K loop.  Two stages, and the fill of tile kt+2 goes out right BEHIND the barrier of iteration kt -- every wave's reads of
  // tile kt's stage have returned by then -- so a tile has a whole iteration to arrive, not half of one (the A rows were just written
  // by the previous kernel and come from HBM / the Infinity Cache, ~2 us away).  Measured against the fill in the next iteration's
  // first step: 47.6 vs 48.5 us per launch warm, 46.1 vs 47.0 cold (profiles/r05_fused_text_attention.txt).
#pragma unroll
  for (int r = 0; r < MI2 + NI2; ++r) read_frag(smem, 0, 0, r);
  __builtin_amdgcn_sched_barrier(0);
  for (int kt = 0; kt < KT; ++kt) {
    const char* sc = smem + (kt & 1) * kStage;
    const char* sn = smem + ((kt & 1) ^ 1) * kStage;
    step(0, sc, 1, -1);
    // this wave's pieces of tile kt+1 have landed (the only requests outstanding) and its last reads of tile kt have returned:
    // publish, then the second step's MFMAs (registers only) with the next tile's first fragments read between them
    wait_vm0();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    step(1, sn, kt + 1 < KT ? 0 : -1, kt + 2 < KT ? (kt & 1) : -1);
  }
  if (trace && tid == 0) trace[2] = __builtin_amdgcn_s_memtime();

  // ---- epilogue 1: q / k / v of the four captions -> LDS images in the operand type
  __syncthreads();   // every wave is past its last fragment read: the staging LDS is free
  {
    const float* ln_rows = reinterpret_cast<const float*>(smem + kLnRows);
    float rs[MI2];
#pragma unroll
    for (int ii = 0; ii < MI2; ++ii) rs[ii] = ln_rows[wm * 80 + ii * 16 + l16];
#pragma unroll
    for (int jj = 0; jj < NI2; ++jj) {
      const int n = wn * 96 + jj * 16 + 4 * g16;          // column of the tile: segment n / 64 (q, k, v), head dim d = n % 64
      const int seg = n >> 6, d = n & 63;
      const float4 cb = *reinterpret_cast<const float4*>(smem + kC2 + (seg * 64 + d) * 4);
      const int chunk = d >> 3, half = (d >> 2) & 1;
#pragma unroll
      for (int ii = 0; ii < MI2; ++ii) {
        const int row = wm * 80 + ii * 16 + l16;
        const f32x4 c = acc[ii][jj];
        const X4 pk = {from_f32<T>(fmaf(rs[ii], c[0], cb.x)), from_f32<T>(fmaf(rs[ii], c[1], cb.y)),
                       from_f32<T>(fmaf(rs[ii], c[2], cb.z)), from_f32<T>(fmaf(rs[ii], c[3], cb.w))};
        char* dst;
        if (seg < 2) dst = smem + (seg == 0 ? kQs : kKs) + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4) + half * 8;
        else dst = smem + kVs + (chunk >> 2) * kVImg + row * 64 + (chunk & 3) * 16 + half * 8;
        *reinterpret_cast<X4*>(dst) = pk;
      }
    }
    // rows 320 .. 335 of the K and V images: masked keys of the last caption's third tile -- they must be finite
    // (masking is additive, and a probability of exactly 0 times a NaN value row is a NaN)
    if (tid < 16 * 8) *reinterpret_cast<u32x4*>(smem + kKs + (kBM + (tid >> 3)) * 128 + (tid & 7) * 16) = u32x4{0u, 0u, 0u, 0u};
    else if (tid < 16 * 16) {
      const int t2 = tid - 128;
      *reinterpret_cast<u32x4*>(smem + kVs + (t2 >> 6) * kVImg + (kBM + ((t2 >> 2) & 15)) * 64 + (t2 & 3) * 16) = u32x4{0u, 0u, 0u, 0u};
    }
  }
  __syncthreads();
  if (trace && tid == 0) trace[3] = __builtin_amdgcn_s_memtime();

  // ---- epilogue 2: attention, two waves per caption
  // caption of the tile, and which of its two waves: 0 = query blocks 0 + 1 (the longer instruction stream), 1 = block 2.  Waves w and
  // w + 4 of a workgroup share a SIMD (dispatch order): the roles are dealt so that every SIMD hosts one wave of each kind (round 6;
  // wave & 1 put both long waves of captions 0 and 2 on SIMD 0 and both of captions 1 and 3 on SIMD 2)
  const int cl = wave >> 1, whalf = (wave ^ (wave >> 2)) & 1;
  const int cap = cap0 + cl;
  if (cap >= p.B && !trace) return;                       // wave-uniform (no barrier follows on the product path)
  if (cap < p.B) {
  const int lrow = lane & 31, hi = lane >> 5;
  const int lsw = (lrow >> 1) & 7;                        // query / key tiles start at multiples of 32: the rows' swizzle term
  const unsigned* mkw = reinterpret_cast<const unsigned*>(smem + kMaskWords) + cl * 4;
  const unsigned vw[3] = {(unsigned)__builtin_amdgcn_readfirstlane((int)mkw[0]), (unsigned)__builtin_amdgcn_readfirstlane((int)mkw[1]),
                          (unsigned)__builtin_amdgcn_readfirstlane((int)mkw[2])};
  char* Qs = smem + kQs + cl * kSPad * 128;               // caption-relative images (80 rows per caption: (row >> 1) & 7 keeps its
  const char* Ks = smem + kKs + cl * kSPad * 128;         // meaning, 80 is a multiple of 16)
  const char* Vs = smem + kVs + cl * kSPad * 64;
  constexpr int KTL = 3;                                  // 32-key tiles (65 .. 80 tokens: three query blocks, three key tiles)
  // NB query blocks of 32, starting at block `first`, worked through TOGETHER: their MFMA chains and softmax passes are
  // independent, and with two waves per SIMD that interleaving is what covers the LDS and MFMA latencies.  Per block the
  // arithmetic is attention_mfma_kernel's, operation for operation.
  auto attend = [&](auto nb_c, int first) __attribute__((always_inline)) {
    constexpr int NB = decltype(nb_c)::value;
    u32x4 qf[NB][4];
    f32x16 sc[NB][KTL];
    float rmax[NB], m2[NB], rsum[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int qidx = 32 * (first + b) + lrow;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
        qf[b][ks] = *reinterpret_cast<const u32x4*>(Qs + qidx * 128 + (((ks * 2 + hi) ^ lsw) << 4));
      rmax[b] = -INFINITY;
    }
#pragma unroll
    for (int t = 0; t < KTL; ++t)
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int q0 = 32 * (first + b);
        const bool live = !(p.causal && 32 * t > q0 + 31);  // wave-uniform: tile entirely above the diagonal
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[b][t][r] = 0.f;
        if (live) {
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const u32x4 kf = *reinterpret_cast<const u32x4*>(Ks + (32 * t + lrow) * 128 + (((ks * 2 + hi) ^ lsw) << 4));
            sc[b][t] = half_traits<T>::mfma32(__builtin_bit_cast(X8, kf), __builtin_bit_cast(X8, qf[b][ks]), sc[b][t]);
          }
        }
      }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int q0 = 32 * (first + b), qidx = q0 + lrow;
#pragma unroll
      for (int t = 0; t < KTL; ++t) {
        const bool live = !(p.causal && 32 * t > q0 + 31);
        // (round 6) a tile entirely above the diagonal: its scores would all become -inf and change no maximum; the exp loop below
        // zeroes them and the P V products skip the tile -- no mask arithmetic for it (three of the six tile passes of the wave that
        // owns query blocks 0 + 1)
        if (!live) continue;                                 // wave-uniform
        // a tile every query of the block may use entirely (below the diagonal, all 32 keys valid): nothing to mask (wave-uniform)
        if (vw[t] == 0xffffffffu && (!p.causal || 32 * t + 31 <= q0)) {
#pragma unroll
          for (int r = 0; r < 16; ++r) rmax[b] = fmaxf(rmax[b], sc[b][t][r]);
          continue;
        }
        unsigned bits = vw[t];
        if (p.causal) {
          const int d = qidx - 32 * t;
          bits &= d < 0 ? 0u : (d >= 31 ? 0xffffffffu : (2u << d) - 1u);
        }
        const unsigned nbits = ~(bits >> (4 * hi));        // 1 = masked
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)nbits, (r & 3) + 8 * (r >> 2), 1);   // masked ? 0xffffffff : 0
          sc[b][t][r] += __builtin_bit_cast(float, m & 0xff800000u);                                    // + (-inf) or + 0
          rmax[b] = fmaxf(rmax[b], sc[b][t][r]);
        }
      }
      rmax[b] = fmaxf(rmax[b], __shfl_xor(rmax[b], 32, 64));
      const float m_use = (rmax[b] == -INFINITY) ? 0.f : rmax[b];
      m2[b] = m_use * 1.4426950408889634f;
      rsum[b] = 0.f;
#pragma unroll
      for (int t = 0; t < KTL; ++t) {
        const bool live = !(p.causal && 32 * t > q0 + 31);
        if (!live) {                                       // every score of the tile is -inf: exp gives exact zeros
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[b][t][r] = 0.f;
          continue;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          sc[b][t][r] = __builtin_amdgcn_exp2f(fmaf(sc[b][t][r], 1.4426950408889634f, -m2[b]));
          rsum[b] += sc[b][t][r];
        }
      }
      rsum[b] += __shfl_xor(rsum[b], 32, 64);
    }
    f32x16 oacc[NB][2];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[b][dt][r] = 0.f;
    const char* vlane = Vs + vtr_lane_offset(lrow, hi);
#pragma unroll
    for (int t = 0; t < KTL; ++t)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          if (p.causal && 32 * t > 32 * (first + b) + 31) continue;
          X8 pf;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) pf[jj] = (T)sc[b][t][8 * s2 + jj];
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const X8 vf = vtr_fragment<T>(vlane, dt * kVImg + (32 * t + 16 * s2) * 64);
            oacc[b][dt] = half_traits<T>::mfma32(vf, pf, oacc[b][dt]);
          }
        }
    // normalised 32 x 64 output tiles -> this wave's OWN Q rows (it holds their fragments in registers, nobody else reads
    // them), then whole 128-byte rows to memory.  Odd rows keep their two 8-byte halves exchanged (bank spread of the
    // transposing ds_write_b64, attention_mfma.hip).
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int q0 = 32 * (first + b), qidx = q0 + lrow;
      const float inv = 1.0f / rsum[b];
      char* orow_lds = Qs + qidx * 128;
      // (rows 80 .. 95 of the third block are the NEXT caption's first Q rows, which its own wave may not have read yet: the
      //  queries there do not exist, their lanes write nothing)
      if (qidx < kSPad) {
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const int d = dt * 32 + 8 * q4 + 4 * hi;
            const int c = d >> 3;
            const X4 v = {from_f32<T>(oacc[b][dt][4 * q4 + 0] * inv), from_f32<T>(oacc[b][dt][4 * q4 + 1] * inv),
                          from_f32<T>(oacc[b][dt][4 * q4 + 2] * inv), from_f32<T>(oacc[b][dt][4 * q4 + 3] * inv)};
            *reinterpret_cast<X4*>(orow_lds + ((c ^ lsw) << 4) + ((((d >> 2) ^ lrow) & 1) << 3)) = v;
          }
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int c = lane & 7;
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = 32 * (first + b) + it * 8 + (lane >> 3);
        const u32x4 raw = *reinterpret_cast<const u32x4*>(Qs + r * 128 + ((c ^ ((r >> 1) & 7)) << 4));
        const u32x4 v = (r & 1) ? u32x4{raw[2], raw[3], raw[0], raw[1]} : raw;
        if (r < S) *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(p.out) + ((size_t)cap * S + r) * D + head * 64 + c * 8) = v;
      }
  };
  // 65 .. 80 tokens = three query blocks; under the causal mask block 0 needs one key tile, block 1 two, block 2 three
  if (whalf == 0) attend(std::integral_constant<int, 2>{}, 0);
  else attend(std::integral_constant<int, 1>{}, 2);
  }
  if (trace) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      trace[4] = __builtin_amdgcn_s_memtime();
      trace[7] = (trace_real0 & 0xffffffffull) | ((__builtin_amdgcn_s_memrealtime() - trace_real0) << 32);
    }
  }
}

template <typename T>
hipError_t launch_t(const QkvAttnParams& p, hipStream_t s) {
  auto kern = qkv_attention_kernel<T>;
  static std::once_flag once;
  static hipError_t rc = hipSuccess;
  std::call_once(once, [&]() {
    rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
  });
  if (rc != hipSuccess) return rc;
  const int nblk = ((p.B + kCPT - 1) / kCPT) * p.H;
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(kNT), kLdsBytes, s, p);
  return hipGetLastError();
}

}  // namespace

bool qkv_attention_supports(int dtype, int B, int S, int H, int D) {
  // sequences that fill the 80-row caption slots well (CLIP's 77 tokens); operands below 4 GiB (32-bit buffer offsets)
  return (dtype == 1 || dtype == 2) && B > 0 && S > 64 && S <= kSPad && H > 0 && D == H * 64 && D % 128 == 0 &&
         (size_t)B * S * D * 2 < (1ull << 32) && (size_t)3 * D * D * 2 < (1ull << 32);
}

// Whether the fused kernel is the faster form at this batch: its 4-caption x 1-head workgroups walk K loop, image writes and
// attention one after the other, alone on their CU -- below ~0.6 workgroups per CU the two-kernel path (many small attention
// workgroups, 128x128 GEMM tiles) wins (ViT-B/32 text tower, bf16: B = 64: +4.4 %, B = 96: -3.9 %, B = 256: -10.5 % --
// tools/exp/r05_fuse_small_batches.py, profiles/r05_fused_text_attention.txt).  Same bits either way.
bool qkv_attention_pays(int B, int H, int num_cus) { return ((B + kCPT - 1) / kCPT) * H * 5 >= num_cus * 3; }

hipError_t launch_qkv_attention(int dtype, const void* A, const void* W, const float* c2, const float* stats, float ln_inv_d,
                                float ln_eps, void* out, int B, int S, int H, int causal, const int64_t* key_mask, hipStream_t s,
                                unsigned long long* trace) {
  const int D = H * 64;
  if (!qkv_attention_supports(dtype, B, S, H, D)) return hipErrorInvalidValue;
  QkvAttnParams p;
  p.A = A; p.W = W; p.c2 = c2; p.stats = stats; p.out = out; p.key_mask = key_mask;
  p.B = B; p.S = S; p.H = H; p.D = D; p.causal = causal; p.ln_inv_d = ln_inv_d; p.ln_eps = ln_eps; p.trace = trace;
  return dtype == 1 ? launch_t<bf16_t>(p, s) : launch_t<f16_t>(p, s);
}

}  // namespace plipmi
