// gemm.h -- the NT GEMM that carries >98 % of the PLIP forward's FLOPs.
//
//   C = epilogue( A[M,K] * W[N,K]^T )         A, W row-major, K contiguous
//
// which is exactly nn.Linear (HF stores Linear weights [out,in]) -- q/k/v, out_proj,
// fc1, fc2 (modeling_clip.py:293-296,343-344), the unfolded patch-embed conv
// (:148-154) and, with A = image embeds / W = text embeds, the logits (:814).
//
// gfx950 design
//   * MFMA: v_mfma_f32_16x16x32_bf16 / _f16 in the production tiles of the two 16-bit engines (variants 2, 3, 6: less
//     power per FLOP than 32x32x16, see the note above the kernel), v_mfma_f32_32x32x16 in the other 16-bit tiles, or
//     v_mfma_f32_32x32x2_f32 (exact fp32, fmaf-chain numerics).  Operands are SWAPPED -- the weight
//     fragment is the MFMA "A" operand and the activation fragment the "B"
//     operand -- so a lane ends up with 4 CONSECUTIVE output columns of one
//     output row per accumulator quad: 16-byte fp32 / 8-byte bf16 epilogue stores
//     and a float4 bias load instead of 2-byte scalar stores.
//   * LDS tile rows are always 128 bytes (BK = 64 bf16 / 32 fp32) = eight 16-byte
//     chunks; chunk c of row r lives at slot c ^ ((r>>1)&7).  With that XOR every
//     ds_read_b128 lane group (MI355X_MICROARCH LDS table) touches 16 distinct
//     16-byte slots of the 256-byte bank row: conflict-free fragment reads.
//   * global -> LDS staging through LDS-DMA (`buffer_load_dwordx4 ... lds`, or `global_load_lds_dwordx4` for
//     operands of 4 GiB and more): the LDS image is lane-linear, so the swizzle is applied to the per-lane SOURCE address.
//   * two LDS stages (one tile of lookahead) or, where three fit in the 160 KB (the 160x256 tile), a ring of three
//     (two tiles of lookahead, counted vmcnt); one barrier per K tile either way.  With operands resident in the
//     Infinity Cache the two are equal; with operands coming from HBM -- how the engine's kernels find the activations
//     the previous kernel wrote -- the second tile of lookahead is what covers the longer fill (profiles/r03_gemm_cold.txt).
//   * blockIdx -> tile map is XCD-aware: hardware round-robins blocks over the 8
//     XCDs, so block b is given logical tile (b%8)*ceil(n/8)+b/8 (bijective form)
//     and each XCD's private L2 sees a contiguous strip of M tiles sweeping N.
#pragma once
#include <type_traits>
#include "common.h"

namespace plipmi {

// first-class vector (HIP's uint4 struct keeps staging arrays in scratch)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

enum Epilogue : int {
  EPI_BIAS = 0,        // C(T)   = acc + bias[n]
  EPI_BIAS_QGELU = 1,  // C(T)   = quickgelu(acc + bias[n])
  EPI_BIAS_RESID = 2,  // C(f32) += acc + bias[n]            (in-place residual stream)
  EPI_SCALE = 3,       // C(f32) = alpha * acc
  EPI_PATCH = 4,       // C(f32)[img*(np+1)+1+p, n] = acc + pos[(1+p), n]   (m = img*np + p)
  // LayerNorm folded into the GEMMs on either side of it (16-bit engines; modeling_clip.py:370-381: LN -> Linear):
  //   the Linear's weights carry LayerNorm's gain AND its centring (W' = W * g with each row's mean over k removed, so
  //   x . W'^T == (x - mean(x)) . (W * g)^T: the mean subtraction happens inside the contraction), its bias carries
  //   LayerNorm's bias (c2 = W b + bias), the A operand is the bf16 residual stream itself, and only the row's rstd
  //   enters in the epilogue:   y = rstd[m] * acc + c2[n]
  EPI_BIAS_LN = 5,     // C(bf16) = that                                   (LN1 -> q/k/v)
  EPI_QGELU_LN = 6,    // C(bf16) = quickgelu(that)                        (LN2 -> fc1)
  // ... and the producer of the NEXT LayerNorm's input: the in-place residual update also emits the bf16 copy of the
  // new rows (the next GEMM's A operand) and their statistics as per-64-column partials {sum, centred M2}
  EPI_RESID_EMIT = 7,  // C(f32) += acc + bias[n];  xb(bf16) = C;  st[m, n/64] = {sum, M2}
  // The same update on a residual stream kept as TWO planes (common.h split_f32): hi = the fp32 value rounded to the operand type, lo =
  // an 8-bit remainder -- the stream at 16 (bf16) / 19 (f16) significand bits, its hi plane IS the next GEMM's A operand, and the
  // epilogue moves 6 bytes per element (3 in, 3 out) instead of the 10 of EPI_RESID_EMIT (which writes fp32 and a separate 16-bit copy).
  EPI_RESID_SPLIT = 8, // {hi,lo} += acc + bias[n] (xb_out = hi, lo_io = lo);  st[m, n/64] = {sum, M2}
  EPI_COUNT = 9
};
constexpr bool epi_is_ln(int e) { return e == EPI_BIAS_LN || e == EPI_QGELU_LN; }
constexpr bool epi_is_colwise(int e) { return e == EPI_BIAS || e == EPI_BIAS_QGELU || epi_is_ln(e); }
constexpr bool epi_is_resid(int e) { return e == EPI_BIAS_RESID || e == EPI_RESID_EMIT || e == EPI_RESID_SPLIT; }
constexpr bool epi_emits_stats(int e) { return e == EPI_RESID_EMIT || e == EPI_RESID_SPLIT; }


struct GemmParams {
  const void* A;
  const void* W;
  void* C;
  const float* bias;  // [N] (EPI_BIAS*) or position embedding [(np+1), N] (EPI_PATCH)
  int M, N, K;
  int lda, ldw, ldc;  // in elements
  float alpha;
  int np;             // patches per image (EPI_PATCH)
  // EPI_*_LN consumers: per-row statistics partials [M, ln_ns, 2] fp32 over 64-column slices of the LayerNorm input
  // (ln_combine; bias carries c2), 1/D and eps of that LayerNorm
  const float* ln_stats = nullptr;
  int ln_ns = 0;
  float ln_inv_d = 0.f, ln_eps = 0.f;
  // EPI_RESID_EMIT producer: bf16 copy of the updated rows [M, ldc] and their partial statistics [M, N/64, 2].
  // EPI_RESID_SPLIT: xb_out is the hi plane (16-bit, read AND written), lo_io the lo plane (8-bit, blocked layout, read and written)
  void* xb_out = nullptr;
  float* st_out = nullptr;
  void* lo_io = nullptr;
  // EPI_RESID_SPLIT: write the updated planes in the OTHER 16-bit type's split format (read them in T's): the last f16 block
  // of a mixed text tower (plipmi_config.text_f16_layers) hands the stream to the bf16 blocks without a re-coding pass.
  // (The epilogue splits the value it computed in fp32: no double rounding through T's format.)
  int planes_other = 0;
  // ADDR 2 (the patch GEMM, im2col ON LOAD; modeling_clip.py:148-154,209-210): A is not read as [M, K] rows -- row m = patch (img, gi, gj)
  // and column k = (c, u, v) are GATHERED from the fp32 NCHW pixels while a K tile is staged: four consecutive pixels of one image row
  // per lane into registers, rounded to the operand type, written to the A stage (LDS-DMA copies bytes, it cannot convert: this is the
  // register-staged converting A path).  pix = the pixels, img_hw = image side, patch_log2 = log2 of the patch side (4 or 5).
  const float* pix = nullptr;
  int img_hw = 0, patch_log2 = 0;
  // ADDR 3: the same gather from native uint8 HWC tiles [B, H, W, 3] (plipmi_encode_image_u8; reproducibility/embedders/transform.py:45-52
  // on 224 x 224 tiles reduces to (u8 / 255 - mean) / std): a lane loads the 12 bytes of four RGB pixels, takes the K tile's channel and
  // normalises with ONE fma per pixel -- fl(b * A_c + B_c) rounds to the same bf16 / f16 as the unfold kernel's (b / 255 - mean_c) * (1 / std_c)
  // for every byte value and channel (exhaustive: tests/test_host.py::test_u8_normalisation_by_one_fma_is_exact_after_rounding)
  const unsigned char* tiles = nullptr;
  // Row count known only on the device (packed captions, kernels.h launch_text_pack): when set, the kernel processes
  // min(*m_dev, M) rows -- M then only sizes the grid; workgroups whose tile starts past the live rows exit at once
  const int* m_dev = nullptr;
  // Tile raster: the N tiles are cut in column groups `gw` tiles wide; logical tile ids run group by group,
  // M-major inside a group.  An XCD's contiguous id range is then a compact (rows x gw) patch whose W panels
  // (gw*BN rows of W) stay resident in its 4 MiB L2 while the A row panels stream through once.
  // 0 = one group (N-major sweep of whole rows).  Set by gemm_launch.
  int gw = 0;
  // test hook (plipmi_gemm_nt_traced): per workgroup 8 x u64 {start, prologue done, main loop done, epilogue
  // done, logical tile id, HW_ID, k tiles, 0}, s_memtime ticks.  nullptr on the product path.
  unsigned long long* trace = nullptr;
};

// LDS-DMA (global_load_lds_dwordx4): each lane's 16 bytes at `gsrc` land at
// `lds_wave_base + lane*16` (wave-uniform base in M0).  Issued through inline asm
// on purpose: with the builtin, hipcc cannot tell that the DMA fills the OTHER
// LDS buffer and drains it (s_waitcnt vmcnt(0)) in front of the first ds_read of
// the current one, which serialises load and compute.  The asm form is invisible
// to its wait-count pass; the kernel waits vmcnt(0) itself right before the barrier
// that publishes the buffer (cdna_hip_programming.md 5.7 item 1).  M0 is saved and
// restored inside the statement because the compiler owns it.
__device__ __forceinline__ void glds16(const char* gsrc, unsigned lds_wave_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_wave_base)
      : "memory");
}
// The same LDS-DMA through the buffer path: a 128-bit resource descriptor in SGPRs (base, size) plus ONE 32-bit
// per-lane byte offset instead of a 64-bit per-lane address (ADDR = 1 kernels).
typedef __attribute__((ext_vector_type(4))) int i32x4;
__device__ __forceinline__ i32x4 make_buffer_rsrc(const void* base) {
  const unsigned long long a = (unsigned long long)base;
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(a & 0xffffffffull));
  r[1] = __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffull));  // stride 0: raw buffer
  r[2] = -1;                                                              // num_records: whole address range
  r[3] = 0x00020000;                                                      // gfx9-family raw dword buffer
  return r;
}
// soff: wave-uniform byte offset (the K position) in an SGPR -- the per-lane offsets never change inside the K loop
__device__ __forceinline__ void glds16_buf(const i32x4 rsrc, unsigned voff, unsigned soff, unsigned lds_wave_base) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_wave_base)
      : "memory");
}
// N pieces under ONE M0 save/restore; the LDS destination of piece e is lds_base + OFFe (compile-time), formed by
// the s_add that writes M0.  Operand order per piece: resource, lane offset.
template <int N, int OFF0, int OFF1 = 0, int OFF2 = 0, int OFF3 = 0>
__device__ __forceinline__ void glds16_buf_n(unsigned lds_base, unsigned soff, const i32x4 r0, unsigned v0,
                                             const i32x4 r1, unsigned v1, const i32x4 r2, unsigned v2,
                                             const i32x4 r3, unsigned v3) {
  unsigned keep;
  if constexpr (N == 1) {
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_add_u32 m0, %1, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %2, %5 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_base), "s"(r0), "v"(v0), "i"(OFF0), "s"(soff) : "memory", "scc");
  } else if constexpr (N == 2) {
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_add_u32 m0, %1, %6\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %2, %8 offen lds\n\t"
                 "s_add_u32 m0, %1, %7\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %4, %8 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_base), "s"(r0), "v"(v0), "s"(r1), "v"(v1), "i"(OFF0), "i"(OFF1), "s"(soff)
                 : "memory", "scc");
  } else if constexpr (N == 3) {
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_add_u32 m0, %1, %8\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %2, %11 offen lds\n\t"
                 "s_add_u32 m0, %1, %9\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %4, %11 offen lds\n\t"
                 "s_add_u32 m0, %1, %10\n\ts_nop 0\n\tbuffer_load_dwordx4 %7, %6, %11 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_base), "s"(r0), "v"(v0), "s"(r1), "v"(v1), "s"(r2), "v"(v2), "i"(OFF0), "i"(OFF1), "i"(OFF2), "s"(soff)
                 : "memory", "scc");
  } else {
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_add_u32 m0, %1, %10\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %2, %14 offen lds\n\t"
                 "s_add_u32 m0, %1, %11\n\ts_nop 0\n\tbuffer_load_dwordx4 %5, %4, %14 offen lds\n\t"
                 "s_add_u32 m0, %1, %12\n\ts_nop 0\n\tbuffer_load_dwordx4 %7, %6, %14 offen lds\n\t"
                 "s_add_u32 m0, %1, %13\n\ts_nop 0\n\tbuffer_load_dwordx4 %9, %8, %14 offen lds\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_base), "s"(r0), "v"(v0), "s"(r1), "v"(v1), "s"(r2), "v"(v2), "s"(r3), "v"(v3), "i"(OFF0), "i"(OFF1),
                   "i"(OFF2), "i"(OFF3), "s"(soff)
                 : "memory", "scc");
  }
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// One 16-byte chunk of K per lane -> one (16-bit operand types) or four (fp32) MFMAs.
template <typename T>
__device__ __forceinline__ void mma16(f32x16& acc, const u32x4& wfrag, const u32x4& xfrag) {
  if constexpr (sizeof(T) == 2) {
    using X8 = typename half_traits<T>::x8;
    acc = half_traits<T>::mfma32(__builtin_bit_cast(X8, wfrag), __builtin_bit_cast(X8, xfrag), acc);
  } else {
    // lane group g = lane>>5 holds k = 4*(2*kq+g)+j, j=0..3, for BOTH operands, so
    // MFMA j multiplies matching k's (any k permutation shared by A and B is valid).
    f32x4 w = __builtin_bit_cast(f32x4, wfrag), x = __builtin_bit_cast(f32x4, xfrag);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[j], x[j], acc, 0, 0, 0);
  }
}

// 16-byte output store, written THROUGH the XCD's L2 (sc0 sc1): the line goes to the fabric now, while other workgroups
// are still in their K loops, instead of staying dirty until the end-of-kernel write-back every launch otherwise ends with
// (MI355X_MICROARCH.md, "boundary": + B / 6 TB/s for B dirty bytes): -0.5 ... -4 us per launch on the eight production
// shapes against plain stores (profiles/r03_gemm_tiles.txt, measured while a process-wide switch existed: commit ba1f4b8;
// as a run-time flag the choice itself cost 1 % of the two-stream step).  The asm store ends with s_nop 1: hipcc does not
// know the statement reads its data registers after issue (cdna_hip_programming.md 5.7 item 1).
__device__ __forceinline__ void store16(void* ptr, const u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
}

// Epilogue split in a LOAD half (bias / residual / position rows; issued back to
// back for a whole 32x32 tile so the loads overlap) and a STORE half.
template <typename T, int EPI>
struct EpilogueOp {
  static constexpr bool kAccurate = sizeof(T) == 4;
  using OutT = std::conditional_t<sizeof(T) == 4, float, T>;
  __device__ __forceinline__ static float4 load(const GemmParams& p, int m, int n0) {
    if constexpr (epi_is_colwise(EPI)) {
      return *reinterpret_cast<const float4*>(p.bias + n0);
    } else if constexpr (EPI == EPI_RESID_SPLIT) {
      return make_float4(0.f, 0.f, 0.f, 0.f);   // the split-plane epilogue loads its planes itself (16-byte pieces)
    } else if constexpr (epi_is_resid(EPI)) {
      const float4 b = *reinterpret_cast<const float4*>(p.bias + n0);
      const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.C) + (size_t)m * p.ldc + n0);
      return make_float4(r.x + b.x, r.y + b.y, r.z + b.z, r.w + b.w);
    } else if constexpr (EPI == EPI_PATCH) {
      const int img = m / p.np, pp = m - img * p.np;
      return *reinterpret_cast<const float4*>(p.bias + (size_t)(1 + pp) * p.N + n0);
    } else {
      return make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // 4 consecutive columns n0..n0+3 of output row m
  __device__ __forceinline__ static void store(const GemmParams& p, int m, int n0, float v0, float v1, float v2,
                                               float v3, const float4 add) {
    if constexpr (epi_is_colwise(EPI)) {
      v0 += add.x; v1 += add.y; v2 += add.z; v3 += add.w;
      if constexpr (EPI == EPI_BIAS_QGELU) {
        v0 = quick_gelu<kAccurate>(v0); v1 = quick_gelu<kAccurate>(v1);
        v2 = quick_gelu<kAccurate>(v2); v3 = quick_gelu<kAccurate>(v3);
      }
      store4(reinterpret_cast<OutT*>(p.C) + (size_t)m * p.ldc + n0, v0, v1, v2, v3);
    } else if constexpr (epi_is_resid(EPI)) {
      store4(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n0, add.x + v0, add.y + v1, add.z + v2, add.w + v3);
    } else if constexpr (EPI == EPI_SCALE) {
      store4(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n0, p.alpha * v0, p.alpha * v1, p.alpha * v2,
             p.alpha * v3);
    } else {  // EPI_PATCH: patch row m = img*np + pp goes to token row img*(np+1) + 1 + pp
      const int img = m / p.np, pp = m - img * p.np;
      float* c = reinterpret_cast<float*>(p.C) + ((size_t)img * (p.np + 1) + 1 + pp) * p.ldc + n0;
      store4(c, v0 + add.x, v1 + add.y, v2 + add.z, v3 + add.w);
    }
  }
};

// ADDR 2 helpers (im2col on load): a 16-byte pixel load into registers that hipcc does not count, and the wait that hands the
// registers back to it -- they pass THROUGH the wait statement, so no use of them is scheduled above it (cdna_hip_programming.md
// 5.7 item 1, VGPR destinations, form ii).  s_nop 4: the scalar offset may come straight from SALU arithmetic.
__device__ __forceinline__ void pix_load16(u32x4& dst, unsigned voff, const i32x4 rsrc, unsigned soff) {
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// (the count is chosen by a wave-uniform branch around OPERAND-FREE wait statements; the registers then pass through ONE
//  unconditional empty statement behind them -- a register-tied statement on each side of a branch makes hipcc merge the two
//  register sets with v_mov copies in FRONT of the waits, i.e. it reads the destinations before the data has landed)
__device__ __forceinline__ void tie_regs5(u32x4& a, u32x4& b, u32x4& c, u32x4& d, u32x4& e) {
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : : "memory");
}
// ADDR 3: twelve bytes = four RGB pixels of a uint8 tile
typedef __attribute__((ext_vector_type(3))) unsigned u32x3;
__device__ __forceinline__ void pix_load12(u32x3& dst, unsigned voff, const i32x4 rsrc, unsigned soff) {
  asm volatile("s_nop 4\n\tbuffer_load_dwordx3 %0, %1, %2, %3 offen" : "=v"(dst) : "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void tie_regs5(u32x3& a, u32x3& b, u32x3& c, u32x3& d, u32x3& e) {
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : : "memory");
}
// CLIP normalisation of a byte of channel c as one fma: A_c = fl((1 / std_c) / 255), B_c = fl(-mean_c / std_c)  (bit patterns, so that no
// compiler's constant folding can move them)
__device__ __forceinline__ float u8_norm(float b, int c) {
  const unsigned A[3] = {0x3c6f2e3cu, 0x3c75e324u, 0x3c68fb47u}, Bc[3] = {0xbfe568dbu, 0xbfe044b8u, 0xbfbd77d7u};
  return fmaf(b, __builtin_bit_cast(float, A[c]), __builtin_bit_cast(float, Bc[c]));
}

// BM x BN block tile, WM x WN waves.  A wave owns (BN / WN) columns and a run of the tile's rows.  32x32 MFMA forms (fp32
// engine, SCHED 0 .. 6): BM / 32 blocks are dealt to the WM wave rows MI = ceil(BM / 32 / WM) at a time, so the LAST wave row may
// hold fewer (160 x 256 on 2 x 4 waves: 3 + 2 blocks; waves w and w + 4 of a workgroup share a SIMD -- MI355X_MICROARCH.md, LDS
// section: dispatch order 0->2->1->3 -- so with WN = 4 every SIMD hosts one wave of each wave row and the MFMA work per SIMD stays
// even).  16x16x32 ring form (SCHED 7): wave rows of BM / WM rows in 16-row blocks, every wave row the same (kHalf below).
// SCHED 0: fragment reads / MFMAs in compiler order, the whole fill issued at the top of the iteration;
//       1: reads of K-step ks+1 pinned in front of the MFMAs of step ks (register double buffering), fill at the top;
//       6: as 1, and the next fill's LDS-DMA requests are packed into the first 3 K steps of the iteration, one batch
//          in front of each step's MFMA group, instead of queueing all of them on the texture-address unit at once.
// NSTAGE 2: the fill runs ONE K tile ahead, the end-of-iteration wait is vmcnt(0);
//        3: three LDS stages, the fill runs TWO K tiles ahead and the wait is a counted vmcnt (in-order retirement: the
//           older tile has landed, the newest may still fly).  Needs 3 * (BM + BN) * 128 B of the 160 KB.  The iteration's
//           barrier sits IN FRONT of its last K step's MFMAs: behind it a wave first requests the next tile's first
//           fragments, then issues the MFMA group it still holds in registers -- the LDS round trip every wave starts a tile
//           with runs under matrix work instead of in front of it (1893 -> 1768 cycles per K tile).
//           (With two stages AND the 32x32 burst schedule the same move buys 0-2 % per kernel and nothing on the step; the
//           streamed 16x16x32 two-stage form, SCHED 8, makes it pay: 2965 -> 2608 cycles per K tile, profiles/r04_gemm_m16.txt.)
// ADDR 0: 64-bit per-lane global addresses (any operand size); 1: buffer resource + 32-bit lane offset (< 4 GiB);
//      2 / 3: as 1 for W, the A tile gathered from fp32 pixels / uint8 tiles through registers (im2col on load, the patch GEMM).
// waves per SIMD the kernel is built for: 2 (LDS caps residency there, so let the allocator use 256 VGPRs)
// index sets of the staged fill: [p0, p1) without [g0, g1)
constexpr int count_outside(int p0, int p1, int g0, int g1) {
  int c = 0;
  for (int i = p0; i < p1; ++i) c += (i >= g0 && i < g1) ? 0 : 1;
  return c;
}
constexpr int nth_outside(int p0, int p1, int g0, int g1, int k) {
  for (int i = p0; i < p1; ++i) {
    if (i >= g0 && i < g1) continue;
    if (k == 0) return i;
    --k;
  }
  return p0;
}
// SCHED 7 / 8 (16-bit operand types): the K loop in v_mfma_f32_16x16x32 instead of 32x32x16 -- the same FLOPs per cycle with a
//    quarter of the accumulator registers per instruction.  Under the chip's power budget bare random-data streams of it sustain
//    1880-1980 TFLOP/s against 1600-1720 for the 32x32 form (profiles/r04_mfma_power_ceiling.txt; the vendor library's kernels
//    are MI16x16 throughout), and in the K loop the shader clock settles ~0.1 GHz higher.  The K step is ONE hand-placed stream
//    -- MFMA, fragment read, MFMA, read ... with the fill's LDS-DMA batches behind the reads -- pinned with a scheduling fence
//    per slot.  7: ring of three stages; 8: two stages, the tile's barrier in front of its last MFMA groups.
//    profiles/r04_gemm_m16.txt: q/k/v 957 -> 1022, fc1 1076 -> 1131, fc2 1145 -> 1176 TFLOP/s, the step -2 ... -3 %.
template <typename T, int BM, int BN, int WM, int WN, int EPI, int SCHED = 0, int ADDR = 0, int NSTAGE = 2>
__global__ __launch_bounds__(WM* WN * 64) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gemm_nt_kernel(const GemmParams p) {
  constexpr int NT = WM * WN * 64;
  // SCHED 7 (16x16x32 form on the ring of three): wave rows are dealt in 16-ROW blocks -- BM / WM rows each, a
  // multiple of 16 but not necessarily of 32 (160 rows on two wave rows: 80 rows = five 16-row MFMA tiles per wave row, all wave
  // rows equal).  The epilogues still walk 32-row slabs; a wave row's last slab may then be a half slab (kHalf paths below).
  // Round 5: the ring tile used to deal 32-row blocks 3 + 2 (96 x 64 and 64 x 64 wave tiles; the short waves read a block nobody
  // multiplied and waited at every barrier); 80 x 64 everywhere: 1876 -> 1825 cycles per K tile on fc2, 9 fragment reads per 20
  // MFMAs instead of 10 per 20, cold-operand launches -6 ... -12 %, the step -0.4 % / -0.9 % (profiles/r05_ring_even_dealing.txt).
  constexpr bool kHalf = SCHED == 7;
  constexpr int RB = BM / 32;                  // 32-row blocks of the tile
  constexpr int MI = kHalf ? (BM / WM + 31) / 32 : (RB + WM - 1) / WM;   // ... per wave row (the last one may hold fewer)
  constexpr bool kUneven = !kHalf && RB % WM != 0;
  constexpr int TM = kHalf ? BM / WM : MI * 32, TN = BN / WN;
  static_assert(!kHalf || (BM % WM == 0 && TM % 16 == 0), "SCHED 7: wave rows of whole 16-row MFMA tiles");
  constexpr int NI = TN / 32;
  constexpr int ELEMS16 = 16 / sizeof(T);  // elements per 16-byte chunk
  using OutT = std::conditional_t<sizeof(T) == 4, float, T>;
  static_assert(!(epi_is_ln(EPI) || epi_emits_stats(EPI)) || sizeof(T) == 2, "LayerNorm folding is a 16-bit-engine form");
  // five forms, each reachable from gemm_default_variant (gemm_inst.h): 0 / 1 the 128x128 tiles and the fp32 engine, 6 the 16-bit
  // 192x256 / 160x256 two-stage tiles, 7 the ring, 8 the 256x256 / 320x256 tiles of the 16-bit engines
  static_assert(SCHED == 0 || SCHED == 1 || SCHED == 6 || SCHED == 7 || SCHED == 8,
                "schedules: 0, 1, 6 (fill in three parts), 7 / 8 (16x16x32 form: ring of three / two stages)");
  static_assert(NSTAGE == 2 || NSTAGE == 3, "two LDS stages or a ring of three");
  constexpr bool kSpread = SCHED >= 6;
  constexpr bool kM16 = SCHED >= 7;
  static_assert(!kM16 || sizeof(T) == 2, "the 16x16x32 form: 16-bit operands");
  static_assert(!kM16 || NSTAGE == (SCHED == 7 ? 3 : 2), "schedule 7 runs on the ring, 8 on two stages");
  static_assert(!kSpread || ADDR >= 1, "the spread fill batches buffer-form requests");
  constexpr bool kGather = ADDR >= 2;      // A gathered from fp32 pixels (2) / uint8 tiles (3) through registers (W: buffer-form LDS-DMA as ADDR 1)
  constexpr bool kGatherU8 = ADDR == 3;
  static_assert(!kGather || (SCHED == 7 && EPI == EPI_PATCH), "im2col on load: the ring tile's patch epilogue only");
  constexpr int BK = 8 * ELEMS16;          // 128-byte rows
  constexpr int A_BYTES = BM * 128, W_BYTES = BN * 128, STAGE = A_BYTES + W_BYTES;
  static_assert(NSTAGE * STAGE + (epi_is_ln(EPI) ? BM * 4 : 0) <= 160 * 1024, "LDS stages exceed the CU's 160 KB");
  // 16-byte chunks per thread per tile.  A piece = one wave instruction = 8 rows; when BM * 8 is not a multiple of the
  // thread count the last A piece exists for the first waves only (wave-uniform test a_piece(i))
  constexpr int PA = kGather ? 0 : (BM * 8 + NT - 1) / NT, PW = BN * 8 / NT;   // (gathered A: no A pieces in the LDS-DMA fill)
  constexpr int PA_MIN = kGather ? 0 : BM * 8 / NT;      // pieces every wave issues (counted vmcnt of the three-stage ring)
  constexpr int NAL = BM * 16 / NT;        // kGather: four-pixel loads (16 B of fp32 / 12 B of RGB bytes) per thread and K tile
  static_assert(!kGather || (BM * 16) % NT == 0, "gathered A: whole passes of four-pixel loads");
  static_assert(BM % 32 == 0 && TN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA tile");
  static_assert((BM * 8) % 64 == 0 && (BN * 8) % NT == 0, "staging passes must be whole wave pieces");
  static_assert((NT / 8) % 16 == 0, "swizzle term must not depend on the staging pass");
  static_assert(PA == PA_MIN || (NT / 8) % 8 == 0, "partial last A pass: whole waves in or out");

  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int mi_w = kUneven ? (RB - wm * MI < MI ? RB - wm * MI : MI) : MI;   // this wave's 32-row blocks (wave-uniform)
  auto a_piece = [&](int i) -> bool {      // does this wave own A piece i of a tile?
    return PA == PA_MIN || i < PA_MIN || i * (NT / 8) + wave * 8 < BM;
  };

  // ---- XCD-aware tile assignment (bijective for any block count) -------------
  const int nbn = p.N / BN;
  const int nbm = (p.M + BM - 1) / BM;
  const int nblk = nbm * nbn;
  const int bid = blockIdx.x;
  const int xcd = bid & 7, xi = bid >> 3, xq = nblk >> 3, xr = nblk & 7;
  const int lid = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
  const int gw = p.gw > 0 ? p.gw : nbn, tpg = nbm * gw;  // column group width, tiles per group
  const int cgrp = lid / tpg, crem = lid - cgrp * tpg;
  const int m0 = (crem / gw) * BM, n0 = (cgrp * gw + crem % gw) * BN;
  int Mrt = p.M;   // live rows
  if (p.m_dev) {
    const int md = __builtin_amdgcn_readfirstlane(*p.m_dev);
    Mrt = md < p.M ? md : p.M;
    if (m0 >= Mrt) return;   // workgroup-uniform
  }

  // ---- staging addresses --------------------------------------------------
  // thread -> LDS chunk position q = pass*NT + tid: row = q>>3, slot = q&7, and the
  // K chunk that belongs in that slot is slot ^ ((row>>1)&7) (pass-independent).
  const int srow = tid >> 3;
  const int schunk = (tid & 7) ^ ((srow >> 1) & 7);
  const char* a_src[PA ? PA : 1];
  const char* w_src[PW];
#pragma unroll
  for (int i = 0; i < PA; ++i) {
    int r = m0 + i * (NT / 8) + srow;
    r = r < Mrt ? r : Mrt - 1;  // M edge: re-read the last row, stores are masked
    a_src[i] = reinterpret_cast<const char*>(p.A) + ((size_t)r * p.lda + schunk * ELEMS16) * sizeof(T);
  }
#pragma unroll
  for (int i = 0; i < PW; ++i) {
    const int r = n0 + i * (NT / 8) + srow;
    w_src[i] = reinterpret_cast<const char*>(p.W) + ((size_t)r * p.ldw + schunk * ELEMS16) * sizeof(T);
  }
  const unsigned lds0 =
      __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
  i32x4 rs_a, rs_w;
  unsigned a_off[PA ? PA : 1], w_off[PW];
  if constexpr (ADDR >= 1) {
    rs_a = make_buffer_rsrc(p.A);
    rs_w = make_buffer_rsrc(p.W);
#pragma unroll
    for (int i = 0; i < PA; ++i) a_off[i] = (unsigned)(a_src[i] - reinterpret_cast<const char*>(p.A));
#pragma unroll
    for (int i = 0; i < PW; ++i) w_off[i] = (unsigned)(w_src[i] - reinterpret_cast<const char*>(p.W));
  }
  unsigned koff = 0;  // ADDR 1: K byte offset of the tile being fetched (SGPR); advanced when its last piece is out
  auto dma_a = [&](int i, unsigned lds) {
    if (!a_piece(i)) return;
    if constexpr (ADDR >= 1) glds16_buf(rs_a, a_off[i], koff, lds);
    else { glds16(a_src[i], lds); a_src[i] += 128; }
  };
  auto dma_w = [&](int i, unsigned lds) {
    if constexpr (ADDR >= 1) glds16_buf(rs_w, w_off[i], koff, lds);
    else { glds16(w_src[i], lds); w_src[i] += 128; }
    if constexpr (ADDR >= 1) { if (i == PW - 1) koff += 128; }  // W pieces follow the A pieces: PW-1 is a tile's last
  };

  auto stage_issue = [&](int buf) {  // the whole tile at once; reads a_src/w_src (koff), then advances them by one K tile
    const unsigned base = lds0 + buf * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < PA; ++i) dma_a(i, base + i * NT * 16);
#pragma unroll
    for (int i = 0; i < PW; ++i) dma_w(i, base + A_BYTES + i * NT * 16);
  };
  // the same fill cut in parts, one per K step of the MFMA block (SCHED 5 / 6: packed into the first 2 / 3 K steps, so the
  // last request has most of the iteration -- not a quarter of it -- to land before the end-of-iteration wait)
  constexpr int kFillParts = 3;
  auto stage_issue_part = [&](int buf, int part) {
    const unsigned base = lds0 + buf * STAGE + wave * 1024;
    if constexpr (kM16) {
      return;   // the hand-placed schedules issue through fill_part_placed below
    } else if constexpr (ADDR >= 1 && PA == PA_MIN) {
      constexpr int PER = (PA + PW + kFillParts - 1) / kFillParts;
      // piece idx of the tile: resource, lane offset and (compile-time) LDS offset
      auto RS = [&](int idx) -> const i32x4& { return idx < PA ? rs_a : rs_w; };
      auto VO = [&](int idx) { return idx < PA ? a_off[idx < PA ? idx : 0] : w_off[idx - PA < PW ? (idx >= PA ? idx - PA : 0) : 0]; };
      constexpr auto LO = [](int idx) constexpr { return idx < PA ? idx * NT * 16 : A_BYTES + (idx - PA) * NT * 16; };
      auto go = [&](auto part_c) {
        constexpr int P0 = decltype(part_c)::value * PER;
        constexpr int N = (P0 + PER <= PA + PW) ? PER : (PA + PW - P0 > 0 ? PA + PW - P0 : 0);
        if constexpr (N > 0) {
          constexpr int I0 = P0, I1 = P0 + (N > 1 ? 1 : 0), I2 = P0 + (N > 2 ? 2 : 0), I3 = P0 + (N > 3 ? 3 : 0);
          glds16_buf_n<N, LO(I0), LO(I1), LO(I2), LO(I3)>(base, koff, RS(I0), VO(I0), RS(I1), VO(I1), RS(I2), VO(I2),
                                                            RS(I3), VO(I3));
          if constexpr (P0 + N == PA + PW) koff += 128;
        }
      };
      static_assert(PER <= 4, "glds16_buf_n batches at most four pieces");
      if (part == 0) go(std::integral_constant<int, 0>{});
      else if (part == 1) go(std::integral_constant<int, 1>{});
      else if (part == 2) go(std::integral_constant<int, 2>{});
      return;
    } else if constexpr (ADDR >= 1) {
      // partial last A pass: part 0 = this wave's A pieces (2 or 3 single requests), the W pieces split over the other parts
      static_assert(PW % 2 == 0 && PW <= 8, "W pieces are batched in two halves");
      constexpr int HW = PW / 2;
      constexpr auto WO = [](int i) constexpr { return A_BYTES + i * NT * 16; };
      if (part == 0) {
#pragma unroll
        for (int i = 0; i < PA; ++i)
          if (a_piece(i)) glds16_buf(rs_a, a_off[i], koff, base + i * NT * 16);
      } else if (part == 1) {
        glds16_buf_n<HW, WO(0), WO(HW > 1 ? 1 : 0), WO(HW > 2 ? 2 : 0), WO(HW > 3 ? 3 : 0)>(
            base, koff, rs_w, w_off[0], rs_w, w_off[HW > 1 ? 1 : 0], rs_w, w_off[HW > 2 ? 2 : 0], rs_w, w_off[HW > 3 ? 3 : 0]);
      } else if (part == 2) {
        glds16_buf_n<HW, WO(HW), WO(HW + (HW > 1 ? 1 : 0)), WO(HW + (HW > 2 ? 2 : 0)), WO(HW + (HW > 3 ? 3 : 0))>(
            base, koff, rs_w, w_off[HW], rs_w, w_off[HW + (HW > 1 ? 1 : 0)], rs_w, w_off[HW + (HW > 2 ? 2 : 0)], rs_w,
            w_off[HW + (HW > 3 ? 3 : 0)]);
        koff += 128;
      }
      return;
    }
  };

  // Hand-placed schedules (SCHED 7 / 8): the tile's PA + PW requests are dealt to the first kParts K steps, PER per step, one
  // statement of up to four requests (A pieces that not every wave owns go singly behind their wave-uniform test).
  constexpr int kParts = kFillParts;
  constexpr int PER = (PA + PW + kParts - 1) / kParts;
  static_assert(!kM16 || ADDR >= 1, "hand-placed schedules use the buffer-form LDS-DMA");
  auto piece_lds = [](int idx) constexpr { return idx < PA ? idx * NT * 16 : A_BYTES + (idx - PA) * NT * 16; };
  auto fill_batched = [&](int buf, auto part_c) __attribute__((always_inline)) {
    constexpr int part = decltype(part_c)::value;
    constexpr int P0 = part * PER, P1 = (P0 + PER < PA + PW) ? P0 + PER : PA + PW;
    if constexpr (ADDR >= 1 && P0 < P1) {
      const unsigned base = lds0 + buf * STAGE + wave * 1024;
      constexpr int G0 = PA_MIN, G1 = PA;   // [G0, G1): A pieces only the first waves own
#pragma unroll
      for (int idx = (P0 > G0 ? P0 : G0); idx < (P1 < G1 ? P1 : G1); ++idx)
        if (a_piece(idx)) glds16_buf(rs_a, a_off[idx < PA ? idx : 0], koff, base + idx * NT * 16);
      constexpr int NU = count_outside(P0, P1, G0, G1);
      auto RS = [&](int idx) -> const i32x4& { return idx < PA ? rs_a : rs_w; };
      auto VO = [&](int idx) { return idx < PA ? a_off[idx < PA ? idx : 0] : w_off[idx >= PA && idx < PA + PW ? idx - PA : 0]; };
      auto batch = [&](auto b_c) {
        constexpr int b = decltype(b_c)::value;
        constexpr int n = NU - 4 * b >= 4 ? 4 : NU - 4 * b;
        if constexpr (n > 0) {
          constexpr int I0 = nth_outside(P0, P1, G0, G1, 4 * b), I1 = n > 1 ? nth_outside(P0, P1, G0, G1, 4 * b + 1) : I0,
                        I2 = n > 2 ? nth_outside(P0, P1, G0, G1, 4 * b + 2) : I0, I3 = n > 3 ? nth_outside(P0, P1, G0, G1, 4 * b + 3) : I0;
          glds16_buf_n<n, piece_lds(I0), piece_lds(I1), piece_lds(I2), piece_lds(I3)>(base, koff, RS(I0), VO(I0), RS(I1), VO(I1),
                                                                                    RS(I2), VO(I2), RS(I3), VO(I3));
        }
      };
      static_assert(NU <= 12, "three batches of four per K step");
      batch(std::integral_constant<int, 0>{});
      batch(std::integral_constant<int, 1>{});
      batch(std::integral_constant<int, 2>{});
      if constexpr (P1 == PA + PW) koff += 128;
    }
  };
  auto fill_part_placed = [&](int buf, int part) __attribute__((always_inline)) {
    if (part == 0) fill_batched(buf, std::integral_constant<int, 0>{});
    else if (part == 1) fill_batched(buf, std::integral_constant<int, 1>{});
    else if (part == 2) fill_batched(buf, std::integral_constant<int, 2>{});
  };

  // ---- ADDR 2: the A tile gathered from fp32 pixels (im2col on load) ------------------------------------------------
  // A K tile of 64 columns = 64 / P patch rows u of one channel c (P = 32: two rows, P = 16: four), 64 consecutive k = (c, u, v).
  // Thread -> load q = pass * NT + tid: tile row q >> 4, four-pixel group q & 15 of the row's 64 columns.  The per-lane byte offset
  // into the pixels never changes inside the K loop; the tile's (c, u0) is a wave-uniform scalar offset.  Two register sets: the
  // loads of tile kt+2 travel while tile kt+1's values are converted and written to its LDS stage (K loop unrolled by two, so
  // the sets are compile-time).
  i32x4 rs_p;
  unsigned pix_off[kGather ? NAL : 1];
  int ga_dst[kGather ? NAL : 1];
  using GA = std::conditional_t<kGatherU8, u32x3, u32x4>;
  GA ga[2][kGather ? NAL : 1];
  if constexpr (kGather) {
    rs_p = kGatherU8 ? make_buffer_rsrc(p.tiles) : make_buffer_rsrc(p.pix);
    const int P = 1 << p.patch_log2, g = p.img_hw >> p.patch_log2, f4_per_row = P >> 2;   // patch side, patches per image side
#pragma unroll
    for (int i = 0; i < NAL; ++i) {
      const int q = i * NT + tid, row = q >> 4, f4 = q & 15;
      int r = m0 + row;
      r = r < Mrt ? r : Mrt - 1;
      const int img = r / p.np, pp = r - img * p.np, gi = pp / g, gj = pp - gi * g;
      const int j = f4 / f4_per_row, gq = f4 - j * f4_per_row;        // patch row inside the K tile, four-pixel group inside it
      if constexpr (kGatherU8)   // HWC bytes: pixel (img, y, x) at ((img * H + y) * W + x) * 3; the channel is picked after the load
        pix_off[i] = (unsigned)((((size_t)img * p.img_hw + gi * P + j) * p.img_hw + gj * P + gq * 4) * 3);
      else
        pix_off[i] = (unsigned)((((size_t)img * 3 * p.img_hw + gi * P + j) * p.img_hw + gj * P + gq * 4) * 4);
      const int kl = j * P + gq * 4;                                   // column inside the K tile: 16-byte chunk kl >> 3, half (kl >> 2) & 1
      ga_dst[i] = row * 128 + ((((kl >> 3) ^ ((row >> 1) & 7))) << 4) + ((kl >> 2) & 1) * 8;
    }
  }
  // scalar byte offset of K tile t inside an image: channel c = t / (P * P / 64), first patch row u0 = (t % (P * P / 64)) * (64 / P)
  auto gather_soff = [&](int t) __attribute__((always_inline)) -> unsigned {
    const int tpc_log2 = 2 * p.patch_log2 - 6;
    const int c = t >> tpc_log2, u0 = (t & ((1 << tpc_log2) - 1)) << (6 - p.patch_log2);
    if constexpr (kGatherU8) return (unsigned)(u0 * p.img_hw * 3);
    return (unsigned)((c * p.img_hw + u0) * p.img_hw * 4);
  };
  constexpr int NALX = kGather ? NAL : 1;
  auto gather_load = [&](GA (&set)[NALX], int t) __attribute__((always_inline)) {
    if constexpr (kGather) {
      const unsigned soff = gather_soff(t);
#pragma unroll
      for (int i = 0; i < NAL; ++i) {
        if constexpr (kGatherU8) pix_load12(set[i], pix_off[i], rs_p, soff);
        else pix_load16(set[i], pix_off[i], rs_p, soff);
      }
    }
  };
  // (leave: how many of this wave's vector-memory requests may still be outstanding -- one tile's, or none)
  auto gather_wait = [&](GA (&set)[NALX], bool one_tile) __attribute__((always_inline)) {
    if constexpr (kGather) {
      static_assert(NAL == 5, "the wait statement names five register sets");
      if (one_tile) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NAL + PW) : "memory");
      else wait_vm0();
      tie_regs5(set[0], set[1], set[2], set[3], set[4]);
    }
  };
  // t: the K tile the set holds (uint8 tiles: its channel picks the bytes and the normalisation constants; wave-uniform)
  auto gather_store = [&](GA (&set)[NALX], int stage, int t) __attribute__((always_inline)) {   // -> operand type (the unfold kernels' rounding), into the A stage
    if constexpr (kGather && sizeof(T) == 2) {
      using Th = std::conditional_t<sizeof(T) == 2, T, bf16_t>;
      using X4h = typename half_traits<Th>::x4;
      if constexpr (kGatherU8) {
        const int c = t >> (2 * p.patch_log2 - 6);
        auto put = [&](auto c_c) __attribute__((always_inline)) {
          constexpr int C = decltype(c_c)::value;        // the lane's four pixels are bytes C, C + 3, C + 6, C + 9 of its twelve
#pragma unroll
          for (int i = 0; i < NAL; ++i) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              constexpr int dummy = 0; (void)dummy;
              const int byte = C + 3 * e;
              v[e] = u8_norm((float)((set[i][byte >> 2] >> (8 * (byte & 3))) & 0xffu), C);
            }
            const X4h pk = {from_f32<Th>(v[0]), from_f32<Th>(v[1]), from_f32<Th>(v[2]), from_f32<Th>(v[3])};
            *reinterpret_cast<X4h*>(smem + stage * STAGE + ga_dst[i]) = pk;
          }
        };
        if (c == 0) put(std::integral_constant<int, 0>{});
        else if (c == 1) put(std::integral_constant<int, 1>{});
        else put(std::integral_constant<int, 2>{});
      } else {
#pragma unroll
        for (int i = 0; i < NAL; ++i) {
          const f32x4 v = __builtin_bit_cast(f32x4, set[i]);
          const X4h pk = {from_f32<Th>(v[0]), from_f32<Th>(v[1]), from_f32<Th>(v[2]), from_f32<Th>(v[3])};
          *reinterpret_cast<X4h*>(smem + stage * STAGE + ga_dst[i]) = pk;
        }
      }
    }
  };

  // ---- fragment read offsets (lane-constant) -----------------------------------
  const int lrow = lane & 31, lgrp = lane >> 5;
  const int lsw = (lrow >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) foff[ks] = lrow * 128 + (((ks * 2 + lgrp) ^ lsw) << 4);
  const int a_tile = wm * TM * 128;
  const int w_tile = A_BYTES + wn * TN * 128;

  f32x16 acc[kM16 ? 1 : MI][kM16 ? 1 : NI];
#pragma unroll
  for (int i = 0; i < (kM16 ? 1 : MI); ++i)
#pragma unroll
    for (int j = 0; j < (kM16 ? 1 : NI); ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  // 16x16x32 form: the wave's 32x32 blocks as four 16x16 tiles each.  acc4[2i+b][2j+a][e] = C[m = 32i + 16b + (lane & 15)]
  // [n = 32j + 16a + 4 (lane >> 4) + e]: a lane holds TWO rows of a block (b = 0, 1) and, per row, 4 consecutive columns in each
  // 16-column half -- again whole 16-byte fp32 / 8-byte 16-bit pieces of an output row.
  constexpr int MI2 = kHalf ? TM / 16 : 2 * MI, NI2 = 2 * NI;
  // rows of a wave row's 32-row slab i that exist (kHalf: the last slab may be a half slab)
  auto slab_rows = [](int i) constexpr { return kHalf ? (TM - 32 * i < 32 ? TM - 32 * i : 32) : 32; };
  f32x4 acc4[kM16 ? MI2 : 1][kM16 ? NI2 : 1];
#pragma unroll
  for (int i = 0; i < (kM16 ? MI2 : 1); ++i)
#pragma unroll
    for (int j = 0; j < (kM16 ? NI2 : 1); ++j) acc4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int l16 = lane & 15, g16 = lane >> 4;

  // epilogue operands in the row-contiguous layout of the transposed store (16 lanes x 16 B per output row)
  // Register budget (256 per lane at two waves per SIMD): accumulators + K-loop fragments + one operand block must
  // fit for the early request, accumulators + two operand blocks + the transposed values for the double buffer;
  // the 192x256 / 160x256 tiles afford both, 320x256 and the 4x2-wave 256x256 tile neither (they would spill).
  constexpr int kAccRegs = MI * NI * 16, kBlkRegs = (NI / 2) * 32;
  constexpr bool kRowOperand = (epi_is_resid(EPI) || EPI == EPI_PATCH) && sizeof(T) == 2 &&
                               kAccRegs + 2 * (MI + NI) * 4 + kBlkRegs + 24 <= 256;
  constexpr int kAddBufs = (kAccRegs + 2 * kBlkRegs + 32 + 24 <= 256) ? 2 : 1;
  const int rd_row = lane >> 4, rd_col = (lane & 15) * 4;
  float4 add[kAddBufs][NI / 2][8];
  bool add_ready = false;
  auto load_block = [&](int i, float4 (&dst)[NI / 2][8]) {
    if constexpr (EPI == EPI_RESID_SPLIT) {
      // the residual planes in 16-byte pieces: a lane owns 8 consecutive columns of a row (8 lanes = one 64-column slice), 8 rows per
      // pass.  dst[jp][it] = hi piece of pass it (8 x 16-bit operand type); dst[jp][4 + pr] = lo piece of the pass PAIR pr: the 8-bit
      // remainders of the lane's columns in rows r and r + 8 of a 16-row band (common.h lo_plane_off) -- 6 loads per slab, not 8
#pragma unroll
      for (int jp = 0; jp < NI / 2; ++jp) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          if (it * 8 >= slab_rows(i)) continue;   // half slab: rows 16.. belong to the next wave row
          int m = m0 + wm * TM + i * 32 + it * 8 + (lane >> 3);
          m = m < Mrt ? m : Mrt - 1;
          const size_t off = (size_t)m * p.ldc + n0 + wn * TN + jp * 64 + (lane & 7) * 8;
          dst[jp][it] = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned short*>(p.xb_out) + off);
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          if (pr * 16 >= slab_rows(i)) continue;
          int band = (m0 + wm * TM + i * 32 + pr * 16) >> 4;
          band = band < ((Mrt - 1) >> 4) ? band : ((Mrt - 1) >> 4);      // bands past the live rows: re-read the last one, stores are masked
          const size_t off = (size_t)band * 16 * p.ldc + (size_t)(((n0 + wn * TN + jp * 64) >> 3) + (lane & 7)) * 128 + (lane >> 3) * 16;
          dst[jp][4 + pr] = *reinterpret_cast<const float4*>(reinterpret_cast<const unsigned char*>(p.lo_io) + off);
        }
      }
    } else {
#pragma unroll
      for (int jp = 0; jp < NI / 2; ++jp)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          if (it * 4 >= slab_rows(i)) continue;
          const int m = m0 + wm * TM + i * 32 + it * 4 + rd_row;
          dst[jp][it] = EpilogueOp<T, EPI>::load(p, m < Mrt ? m : Mrt - 1, n0 + wn * TN + jp * 64 + rd_col);
        }
    }
  };

  // ---- hand-placed K step, 16x16x32 form (SCHED 7 / 8) ---------------------------------------------------------
  // A K tile is TWO steps of 32.  Per step the wave needs MI2 activation fragments (16 rows x 32 k: lane = row l16, 16-byte
  // chunk 4s + g16 of the 128-byte LDS row; the XOR swizzle keeps every ds_read_b128 lane group on 16 distinct slots) and NI2
  // weight fragments, and issues MI2 * NI2 MFMAs.
  //  * two LDS stages: the operand with FEWER fragments (always 4 here) is KEPT in registers for the step (double-buffered
  //    across steps), the other is STREAMED -- one fragment per group of 4 MFMAs (below).
  //  * ring of three (barrier IN FRONT of the tile's last step): behind that barrier nobody may read the tile's stage any more
  //    -- the next iteration's fill overwrites it -- so every fragment of a step is read during the step before it (both operands
  //    double-buffered whole), one read per MFMA slot, and the last step's slots carry the NEXT tile's first fragments.
  // A scheduling fence closes every slot.
  constexpr bool kStream16 = NSTAGE == 2;
  constexpr bool kKeepX = MI2 <= NI2;                 // (streamed form) keep the activation fragments, stream the weights -- or the reverse
  constexpr int NK = kKeepX ? MI2 : NI2, NS = kKeepX ? NI2 : MI2;
  static_assert(!kM16 || !kStream16 || (NK == 4 && NS >= 4), "streamed 16x16x32 form: four kept fragments, at least four streamed ones");
  static_assert(!kM16 || !kUneven, "the 16x16x32 forms deal wave rows in 16-row blocks: every wave row holds the same MI2 of them");
  int foff16[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) foff16[s2] = l16 * 128 + (((4 * s2 + g16) ^ (l16 >> 1)) << 4);
  using H16 = std::conditional_t<sizeof(T) == 2, T, bf16_t>;
  using X8h = typename half_traits<H16>::x8;
  auto x_frag = [&](const char* sb, int s2, int ii) __attribute__((always_inline)) {
    return *reinterpret_cast<const u32x4*>(sb + a_tile + ii * 16 * 128 + foff16[s2]);
  };
  auto w_frag = [&](const char* sb, int s2, int jj) __attribute__((always_inline)) {
    return *reinterpret_cast<const u32x4*>(sb + w_tile + jj * 16 * 128 + foff16[s2]);
  };
  auto mma16x16 = [&](int ii, int jj, const u32x4& wf, const u32x4& xf) __attribute__((always_inline)) {
    if constexpr (kM16) acc4[ii][jj] = half_traits<H16>::mfma16(__builtin_bit_cast(X8h, wf), __builtin_bit_cast(X8h, xf), acc4[ii][jj]);
  };
  u32x4 kf16[(kM16 && kStream16) ? 2 : 1][4];
  auto keep_frag = [&](const char* sb, int s2, int u) __attribute__((always_inline)) { return kKeepX ? x_frag(sb, s2, u) : w_frag(sb, s2, u); };
  auto stream_frag = [&](const char* sb, int s2, int t) __attribute__((always_inline)) { return kKeepX ? w_frag(sb, s2, t) : x_frag(sb, s2, t); };
  // ---- streamed form (SCHED 8).  The streamed fragments of a tile form ONE sequence g = step * NS + t over both steps, read two
  //      groups ahead (one group = 4 MFMAs = 64 pipe cycles, 128 with the SIMD's other wave in between: one group of lookahead
  //      does not cover an LDS round trip under load).  The tile's barrier sits IN FRONT of its last PB groups: every LDS read of
  //      a tile still lies between the tile's two barriers, but the last PB groups' stream fragments are read early (two fragments per group over the groups
  //      before them, then one group of slack for the LDS round trip), so behind the barrier 4 PB MFMAs run from registers while the
  //      NEXT tile's kept fragments and first two stream fragments arrive -- barrier wait and LDS round trip under matrix work.
  //      PB = 4 on both tiles (320x256, 160 accumulator registers: 3407 -> 3249 cycles per K tile against PB = 2; its plain epilogues
  //      spill 3-17 registers outside the loop either way, the LayerNorm-folded fc1 epilogue none).
  constexpr int G16 = 2 * NS;
  constexpr int PB16 = 4;
  constexpr int GD16 = G16 - 2 * PB16;                 // first group that reads two stream fragments
  u32x4 sfr[(kM16 && kStream16) ? G16 : 1];
  auto prime16 = [&](const char* sb) __attribute__((always_inline)) {
    sfr[0] = stream_frag(sb, 0, 0);
#pragma unroll
    for (int u = 0; u < 4; ++u) kf16[0][u] = keep_frag(sb, 0, u);
    sfr[1] = stream_frag(sb, 0, 1);
  };
  // groups [g0, g1) of the tile in stage sb; sbn: the next tile's stage for the groups behind the barrier (has_next)
  auto groups16 = [&](const char* sb, int g0, int g1, int fill_buf, int nparts, const char* sbn, bool has_next) __attribute__((always_inline)) {
#pragma unroll
    for (int g = 0; g < G16; ++g) {
      if (g < g0 || g >= g1) continue;
      const int s2 = g / NS, t = g % NS;
      const bool post = g >= G16 - PB16;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int us = (g & 1) ? 3 - u : u;             // serpentine over the kept fragments: one operand changes per MFMA
        if (kKeepX) mma16x16(us, t, sfr[g], kf16[s2][us]);
        else mma16x16(t, us, kf16[s2][us], sfr[g]);
        if (!post) {
          if (g < GD16) {
            if (u == 0 && g + 2 < G16) sfr[g + 2] = stream_frag(sb, (g + 2) / NS, (g + 2) % NS);
          } else {
            const int f = GD16 + 2 + 2 * (g - GD16) + (u >> 1);
            if ((u == 0 || u == 2) && f < G16) sfr[f] = stream_frag(sb, f / NS, f % NS);
          }
          if (s2 == 0 && u == 1 && t < 4) kf16[1][t] = keep_frag(sb, 1, t);
          if (s2 == 0 && u == 3 && fill_buf >= 0 && t < nparts) fill_part_placed(fill_buf, t);
        } else if (has_next && u < 3) {
          const int r = (g - (G16 - PB16)) * 3 + u;     // 0: stream 0, 1..4: kept 0..3, 5: stream 1
          if (r == 0) sfr[0] = stream_frag(sbn, 0, 0);
          else if (r <= 4) kf16[0][r - 1] = keep_frag(sbn, 0, r - 1);
          else if (r == 5) sfr[1] = stream_frag(sbn, 0, 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // ---- fully double-buffered form (ring of three)
  u32x4 xf16[(kM16 && !kStream16) ? 2 : 1][MI2], wf16[(kM16 && !kStream16) ? 2 : 1][NI2];
  constexpr int NR16 = MI2 + NI2;                      // fragment reads of a step, in the order below
  static_assert(!kM16 || kStream16 || NR16 <= MI2 * NI2, "ring form: one fragment read per MFMA slot");
  // read r of a step: the x fragments, then the w's
  auto read16 = [&](const char* sb, int s2, int b, int r) __attribute__((always_inline)) {
    if (r < MI2) xf16[b][r] = x_frag(sb, s2, r);
    else wf16[b][r - MI2] = w_frag(sb, s2, r - MI2);
  };
  auto read16_all = [&](const char* sb, int s2, int b) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < NR16; ++r) read16(sb, s2, b, r);
  };
  // b: fragment buffer of this step; (sbn, sn): the step whose fragments are read meanwhile into b ^ 1 (sn < 0: none)
  auto step16_full = [&](int b, const char* sbn, int sn, int fill_buf, int nparts) __attribute__((always_inline)) {
#pragma unroll
    for (int ii = 0; ii < MI2; ++ii) {
#pragma unroll
      for (int jj = 0; jj < NI2; ++jj) {
        const int n = ii * NI2 + jj;
        const int js = (ii & 1) ? NI2 - 1 - jj : jj;   // serpentine: one operand changes per MFMA, also at a row change (+0.5-1.5 % warm)
        mma16x16(ii, js, wf16[b][js], xf16[b][ii]);
        if (sn >= 0 && n < NR16) read16(sbn, sn, b ^ 1, n);
        if (fill_buf >= 0 && jj == NI2 - 1 && ii >= 1 && ii - 1 < nparts) fill_part_placed(fill_buf, ii - 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  auto compute = [&](int buf, int fill_buf) {
    const char* sb = smem + buf * STAGE;
    // fragments of K-step ks+1 are fetched from LDS before the MFMAs of step ks issue.  (Rows past an uneven wave row's
    // last block are read too -- in-bounds LDS bytes nobody multiplies -- so the reads stay branch-free: skipping them behind a
    // wave-uniform branch measured 47.2 vs 44.6 us per launch of the residual GEMMs in the step.)
    u32x4 xf[2][MI], wf[2][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i) xf[0][i] = *reinterpret_cast<const u32x4*>(sb + a_tile + i * 32 * 128 + foff[0]);
#pragma unroll
    for (int j = 0; j < NI; ++j) wf[0][j] = *reinterpret_cast<const u32x4*>(sb + w_tile + j * 32 * 128 + foff[0]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < 3) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
          xf[(ks + 1) & 1][i] = *reinterpret_cast<const u32x4*>(sb + a_tile + i * 32 * 128 + foff[ks + 1]);
#pragma unroll
        for (int j = 0; j < NI; ++j)
          wf[(ks + 1) & 1][j] = *reinterpret_cast<const u32x4*>(sb + w_tile + j * 32 * 128 + foff[ks + 1]);
      }
      if constexpr (kSpread) {
        if (fill_buf >= 0 && ks < kFillParts) stage_issue_part(fill_buf, ks);
      }
      if constexpr (SCHED >= 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        if (kUneven && i >= mi_w) break;   // wave-uniform
#pragma unroll
        for (int j = 0; j < NI; ++j) mma16<T>(acc[i][j], wf[ks & 1][j], xf[ks & 1][i]);
      }
      if constexpr (SCHED >= 1) __builtin_amdgcn_sched_barrier(0);
    }
  };

  // rstd of this tile's BM LayerNorm-input rows goes to LDS once, while the first K tile is in flight: the epilogue
  // then reads one float per row instead of walking the partials (a chain of L2 round trips per row) before the stores.
  // (Round 6: requesting the partials BEFORE the first fill and folding them behind it -- so that the compiler's vmcnt for these loads
  //  does not drain the inline-asm fills first -- measured level on fc1 / q/k/v and on the step, and is not done.)
  auto stage_ln_rows = [&]() {
    if constexpr (epi_is_ln(EPI)) {
      float* ln_rows = reinterpret_cast<float*>(smem + NSTAGE * STAGE);
      for (int r = tid; r < BM; r += NT) {
        const int mr = m0 + r < Mrt ? m0 + r : Mrt - 1;
        float mu, rs;
        ln_combine(p.ln_stats + (size_t)mr * p.ln_ns * 2, p.ln_ns, p.ln_inv_d, p.ln_eps, mu, rs);
        ln_rows[r] = rs;
      }
    }
  };

  // ---- main loop: tiles kt+1 (and kt+2) stream in while tile kt is multiplied; one barrier per tile.
  const int KT = p.K / BK;
  unsigned long long* trace = p.trace ? p.trace + (size_t)bid * 8 : nullptr;
  unsigned long long trace_real0 = 0;
  if (trace && tid == 0) {
    trace[0] = __builtin_amdgcn_s_memtime();
    trace_real0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz, one counter for the whole device (s_memtime is per XCD)
    trace[4] = lid;
    trace[5] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4 /* HW_ID [31:0] */) |
               ((unsigned long long)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20 /* XCC_ID [3:0] */) << 32);
    trace[6] = (unsigned long long)KT |
               ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 6 /* LDS_ALLOC [31:0] */) << 32);
  }
  if constexpr (NSTAGE == 3) {
    // Ring of three.  Invariants at the top of iteration kt: tile kt is visible and its K-step-0 fragments are in registers;
    // tile kt+1 is landing or landed; the stage of tile kt-1 is free (every wave's reads of it had returned before the
    // barrier of iteration kt-1), so the fill of tile kt+2 goes there.  The wait in front of the barrier leaves one tile's
    // requests of this wave outstanding (vmcnt retires in order): tile kt+1 has landed, tile kt+2 may still fly.
    constexpr int kLeave = PA_MIN + PW;
    u32x4 xf[2][MI], wf[2][NI];
    auto read_frags = [&](int stage, int ks, int b) {
      const char* sb = smem + stage * STAGE;
#pragma unroll
      for (int i = 0; i < MI; ++i) xf[b][i] = *reinterpret_cast<const u32x4*>(sb + a_tile + i * 32 * 128 + foff[ks]);
#pragma unroll
      for (int j = 0; j < NI; ++j) wf[b][j] = *reinterpret_cast<const u32x4*>(sb + w_tile + j * 32 * 128 + foff[ks]);
    };
    auto mma_step = [&](int b) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        if (kUneven && i >= mi_w) break;   // wave-uniform
#pragma unroll
        for (int j = 0; j < NI; ++j) mma16<T>(acc[i][j], wf[b][j], xf[b][i]);
      }
    };
    if constexpr (kGather) {
      // tile 0 -> set 0 / stage 0, tile 1 -> set 1 / stage 1; tile 0's pixels are rounded and written before the first barrier
      gather_load(ga[0], 0);
      stage_issue(0);
      if (KT > 1) { gather_load(ga[1], 1); stage_issue(1); }
      gather_wait(ga[0], KT > 1);
      gather_store(ga[0], 0, 0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
    stage_issue(0);
    if (KT > 1) stage_issue(1);
    stage_ln_rows();
    if (KT > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLeave) : "memory");  // tile 0 landed, tile 1 may be in flight
    else wait_vm0();
    }
    __syncthreads();
    if (trace && tid == 0) trace[1] = __builtin_amdgcn_s_memtime();
    int cur = 0, nxt = 1, nxt2 = 2;  // stages of tiles kt, kt+1, kt+2
    if constexpr (kM16 && kGather) {
      // (this branch replaces the prologue above: see the `if constexpr (!kGather)` around it)
      using C0 = std::integral_constant<int, 0>; using C1 = std::integral_constant<int, 1>;
      // iteration kt (tile kt in stage cur, set S = kt & 1 free: tile kt's values were written to LDS an iteration ago)
      // (fetch_c = 0: the loop's odd last iteration, kt = KT - 2, has no tile kt + 2 -- said at compile time, so that no pixel load
      //  whose registers nobody reads afterwards is emitted there: hipcc gave five such dead loads ONE destination and reused it
      //  straight away, harmless only because the run-time test never took them; tests/test_isa_audit.py found it)
      auto iter = [&](auto s_c, auto fetch_c, int kt) __attribute__((always_inline)) {
        constexpr int S = decltype(s_c)::value;
        constexpr bool kMayFetch = decltype(fetch_c)::value != 0;
        const int fb = (kMayFetch && kt + 2 < KT) ? nxt2 : -1;
        if constexpr (kMayFetch) { if (fb >= 0) gather_load(ga[S], kt + 2); }
        const char* sc = smem + cur * STAGE;
        step16_full(0, sc, 1, fb, kParts);            // W pieces of tile kt+2 ride in this step
        // tile kt+1: its pixels (set S ^ 1) and this wave's W pieces have arrived -- only tile kt+2's requests may be outstanding;
        // round and write its A rows, publish, then the last step's MFMAs with the next tile's first fragment reads
        gather_wait(ga[S ^ 1], fb >= 0);
        gather_store(ga[S ^ 1], nxt, kt + 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        step16_full(1, smem + nxt * STAGE, 0, -1, 0);
        cur = nxt; nxt = nxt2; nxt2 = 3 - cur - nxt;
      };
      read16_all(smem, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      int kt = 0;
      for (; kt + 1 < KT - 1; kt += 2) { iter(C0{}, C1{}, kt); iter(C1{}, C1{}, kt + 1); }
      if (kt < KT - 1) iter(C0{}, C0{}, kt);
      if constexpr (kRowOperand) {
        load_block(0, add[0]);
        add_ready = true;
        __builtin_amdgcn_sched_barrier(0);
      }
      const char* sc = smem + cur * STAGE;
      step16_full(0, sc, 1, -1, 0);
      step16_full(1, sc, -1, -1, 0);
    } else if constexpr (kM16) {
      read16_all(smem, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      for (int kt = 0; kt < KT - 1; ++kt) {
        const int fb = kt + 2 < KT ? nxt2 : -1;
        const char* sc = smem + cur * STAGE;
        step16_full(0, sc, 1, fb, kParts);            // the whole fill of tile kt+2 rides in this step (the counted wait below)
        // this wave's pieces of tile kt+1 have landed and its last reads of tile kt have returned: publish, then the last step's
        // MFMAs (registers only) with the next tile's first fragment reads between them
        if (fb >= 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLeave) : "memory");
        else wait_vm0();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        step16_full(1, smem + nxt * STAGE, 0, -1, 0);
        cur = nxt; nxt = nxt2; nxt2 = 3 - cur - nxt;
      }
      if constexpr (kRowOperand) {
        load_block(0, add[0]);
        add_ready = true;
        __builtin_amdgcn_sched_barrier(0);
      }
      const char* sc = smem + cur * STAGE;
      step16_full(0, sc, 1, -1, 0);
      step16_full(1, sc, -1, -1, 0);
    } else {
    read_frags(0, 0, 0);
    for (int kt = 0; kt < KT - 1; ++kt) {
      const bool fetch = kt + 2 < KT;
      if constexpr (!kSpread) { if (fetch) stage_issue(nxt2); }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks < 3) {
          read_frags(cur, ks + 1, (ks + 1) & 1);
          if constexpr (kSpread) { if (fetch && ks < kFillParts) stage_issue_part(nxt2, ks); }
        } else {
          // this wave's pieces of tile kt+1 have landed and its last reads of tile kt have returned: publish, then fetch
          // the next tile's first fragments while the MFMAs below run
          if (fetch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLeave) : "memory");
          else wait_vm0();
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __syncthreads();
          read_frags(nxt, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma_step(ks & 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      cur = nxt; nxt = nxt2; nxt2 = 3 - cur - nxt;   // rotate: the stage tile kt leaves becomes tile kt+3's
    }
    if constexpr (kRowOperand) {  // the residual / position rows of the first 32-row block travel during the last K tile
      load_block(0, add[0]);
      add_ready = true;
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < 3) read_frags(cur, ks + 1, (ks + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_step(ks & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    }
  } else if constexpr (kM16) {
    // two stages, 16x16x32 form: the fill of tile kt+1 rides in the first groups of tile kt; one barrier per tile, in front of
    // the tile's last PB16 groups
    stage_issue(0);
    stage_ln_rows();
    wait_vm0();
    __syncthreads();
    if (trace && tid == 0) trace[1] = __builtin_amdgcn_s_memtime();
    prime16(smem);
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < KT - 1; ++kt) {
      const char* sc = smem + (kt & 1) * STAGE;
      const char* sn = smem + ((kt & 1) ^ 1) * STAGE;
      groups16(sc, 0, G16 - PB16, (kt & 1) ^ 1, kParts, sn, false);
      wait_vm0();                                       // this wave's pieces of tile kt+1 ...
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // ... and its last reads of tile kt
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      groups16(sc, G16 - PB16, G16, -1, 0, sn, true);
    }
    if constexpr (kRowOperand) {  // the residual / position rows of the first 32-row block travel during the last K tile
      load_block(0, add[0]);
      add_ready = true;
      __builtin_amdgcn_sched_barrier(0);
    }
    groups16(smem + ((KT - 1) & 1) * STAGE, 0, G16, -1, 0, smem, false);
  } else {
    stage_issue(0);
    stage_ln_rows();
    wait_vm0();
    __syncthreads();
    if (trace && tid == 0) trace[1] = __builtin_amdgcn_s_memtime();
    for (int kt = 0; kt < KT - 1; ++kt) {
      const int cur = kt & 1;
      if constexpr (!kSpread) {
        stage_issue(cur ^ 1);
      }
      compute(cur, cur ^ 1);
      wait_vm0();
      __syncthreads();
    }
    if constexpr (kRowOperand) {  // the residual / position rows of the first 32-row block travel during the last K step
      load_block(0, add[0]);
      add_ready = true;
      __builtin_amdgcn_sched_barrier(0);
    }
    compute((KT - 1) & 1, -1);
  }
  if (trace && tid == 0) trace[2] = __builtin_amdgcn_s_memtime();

  // ---- epilogue -------------------------------------------------------------------------------
  // The accumulator layout gives a lane ONE output row: acc[i][j][4q+e] = C[m = .. + lrow][n = .. + 8q + 4*lgrp + e].
  // Stored straight from there every store instruction touches 32 different rows (64 scattered 16-byte
  // pieces), and the in-kernel timeline showed that costing 22-34k cycles per 256x256 tile -- a third of the
  // workgroup's lifetime.  So each wave transposes its sub-tile through a private LDS slab (32 rows x 64
  // columns fp32, row pitch 272 B: conflict-free ds_write_b128) and writes it back ROW-contiguous: 16 lanes
  // cover 256 B (fp32) / 128 B (16-bit) of one row, so loads/stores are whole cache lines.
  constexpr int SLAB_PITCH = 64 * 4 + 16;
  constexpr int SLAB_BYTES = 32 * SLAB_PITCH;
  static_assert(WM * WN * SLAB_BYTES <= NSTAGE * STAGE, "epilogue slabs must fit in the staging buffers");
  static_assert(NI % 2 == 0, "epilogue handles two 32-column MFMA tiles per slab");
  __syncthreads();  // every wave is done reading the last K tile: the staging LDS can be reused
  char* slab = smem + wave * SLAB_BYTES;
  if constexpr (sizeof(T) == 2 && epi_is_colwise(EPI)) {
    // 16-bit outputs whose epilogue is column-wise (bias, QuickGELU): finish the arithmetic in the ACCUMULATOR layout
    // -- the bias of a lane's 4 x 4 columns per MFMA tile is loaded once per tile column, not once per output row --
    // round there, and transpose HALF the bytes: 8 ds_write_b64 + 4 ds_read_b128 + 4 16-byte global stores
    // per 32 x 64 slab instead of 8 ds_write_b128 + 8 ds_read_b128 + 8 bias loads + 8 8-byte stores.
    // 16-bit slab: 32 rows x 128 B, no padding.  16-byte chunk c of row r sits in slot c ^ (r & 7) and, for rows with
    // bit 3 set, its two 8-byte halves are swapped: the transposing ds_write_b64 (16 consecutive rows, same column)
    // then covers all 32 write banks once, the row-contiguous ds_read_b128 all 64 read banks once
    // (SQ_LDS_BANK_CONFLICT = 0); the half swap is undone in registers, statically per store iteration.
    constexpr int HP = 128;
    using X4 = typename half_traits<OutT>::x4;
    // bias of this lane's columns: 32x32 form 4 columns at 8q + 4 lgrp of each 32-column block, 16x16 form at 16a + 4 g16
    float4 bq[NI][4];
#pragma unroll
    for (int j = 0; j < NI; ++j)
#pragma unroll
      for (int q = 0; q < (kM16 ? 2 : 4); ++q)
        bq[j][q] = *reinterpret_cast<const float4*>(p.bias + n0 + wn * TN + j * 32 + (kM16 ? 16 * q + 4 * g16 : 8 * q + 4 * lgrp));
    const int hr_row = lane >> 3, hr_chunk = lane & 7;  // 8 lanes x 16 B = one 128-byte output row piece
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (kUneven && i >= mi_w) break;   // wave-uniform
      float ln_rs = 1.f, ln_rs2[2] = {1.f, 1.f};
      if constexpr (epi_is_ln(EPI)) {  // this lane's row(s) of the block: rstd of the LayerNorm input row (staged at kernel start)
        if constexpr (kM16) {
          ln_rs2[0] = *reinterpret_cast<const float*>(smem + NSTAGE * STAGE + (wm * TM + i * 32 + l16) * 4);
          if (slab_rows(i) > 16)
            ln_rs2[1] = *reinterpret_cast<const float*>(smem + NSTAGE * STAGE + (wm * TM + i * 32 + 16 + l16) * 4);
        } else {
          ln_rs = *reinterpret_cast<const float*>(smem + NSTAGE * STAGE + (wm * TM + i * 32 + lrow) * 4);
        }
      }
#pragma unroll
      for (int jp = 0; jp < NI / 2; ++jp) {
        if constexpr (kM16) {
          // 16x16 tiles (b = row half, a = column half) of the 32 x 64 slab: row 16b + l16, columns jj*32 + 16a + 4 g16 .. +3 ->
          // 16-byte chunk jj*4 + 2a + (g16 >> 1), 8-byte half g16 & 1; same swizzle as below (the 16 lanes of a ds_write_b64
          // group are again 16 consecutive rows of one column)
#pragma unroll
          for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
              for (int b = 0; b < 2; ++b) {
                if (16 * b >= slab_rows(i)) continue;
                const int j = 2 * jp + jj;
                const f32x4 c = acc4[2 * i + b < MI2 ? 2 * i + b : 0][2 * j + a];
                float v0, v1, v2, v3;
                if constexpr (epi_is_ln(EPI)) {
                  v0 = fmaf(ln_rs2[b], c[0], bq[j][a].x); v1 = fmaf(ln_rs2[b], c[1], bq[j][a].y);
                  v2 = fmaf(ln_rs2[b], c[2], bq[j][a].z); v3 = fmaf(ln_rs2[b], c[3], bq[j][a].w);
                } else {
                  v0 = c[0] + bq[j][a].x; v1 = c[1] + bq[j][a].y; v2 = c[2] + bq[j][a].z; v3 = c[3] + bq[j][a].w;
                }
                if constexpr (EPI == EPI_BIAS_QGELU || EPI == EPI_QGELU_LN) {
                  v0 = quick_gelu<false>(v0); v1 = quick_gelu<false>(v1);
                  v2 = quick_gelu<false>(v2); v3 = quick_gelu<false>(v3);
                }
                const X4 pk = {from_f32<OutT>(v0), from_f32<OutT>(v1), from_f32<OutT>(v2), from_f32<OutT>(v3)};
                const int row = 16 * b + l16, chunk = jj * 4 + 2 * a + (g16 >> 1), half = g16 & 1;
                *reinterpret_cast<X4*>(slab + row * HP + ((chunk ^ (row & 7)) << 4) + ((half ^ ((row >> 3) & 1)) << 3)) = pk;
              }
        } else {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int j = 2 * jp + jj;
            float v0, v1, v2, v3;
            if constexpr (epi_is_ln(EPI)) {
              v0 = fmaf(ln_rs, acc[i][j][4 * q + 0], bq[j][q].x); v1 = fmaf(ln_rs, acc[i][j][4 * q + 1], bq[j][q].y);
              v2 = fmaf(ln_rs, acc[i][j][4 * q + 2], bq[j][q].z); v3 = fmaf(ln_rs, acc[i][j][4 * q + 3], bq[j][q].w);
            } else {
              v0 = acc[i][j][4 * q + 0] + bq[j][q].x; v1 = acc[i][j][4 * q + 1] + bq[j][q].y;
              v2 = acc[i][j][4 * q + 2] + bq[j][q].z; v3 = acc[i][j][4 * q + 3] + bq[j][q].w;
            }
            if constexpr (EPI == EPI_BIAS_QGELU || EPI == EPI_QGELU_LN) {
              v0 = quick_gelu<false>(v0); v1 = quick_gelu<false>(v1);
              v2 = quick_gelu<false>(v2); v3 = quick_gelu<false>(v3);
            }
            const X4 pk = {from_f32<OutT>(v0), from_f32<OutT>(v1), from_f32<OutT>(v2), from_f32<OutT>(v3)};
            *reinterpret_cast<X4*>(slab + lrow * HP + (((jj * 4 + q) ^ (lrow & 7)) << 4) +
                                   ((lgrp ^ ((lrow >> 3) & 1)) << 3)) = pk;
          }
        }
        __builtin_amdgcn_wave_barrier();
        u32x4 o[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          if (it * 8 >= slab_rows(i)) continue;
          const u32x4 raw = *reinterpret_cast<const u32x4*>(slab + (it * 8 + hr_row) * HP + ((hr_chunk ^ hr_row) << 4));
          o[it] = (it & 1) ? u32x4{raw[2], raw[3], raw[0], raw[1]} : raw;  // rows 8..15, 24..31: halves were swapped
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          if (it * 8 >= slab_rows(i)) continue;
          int m = m0 + wm * TM + i * 32 + it * 8 + hr_row;
          const bool in_range = m < Mrt;
          if (in_range)
            store16(reinterpret_cast<OutT*>(p.C) + (size_t)m * p.ldc + n0 + wn * TN + jp * 64 + hr_chunk * 8, o[it]);
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    if (trace) {
      __builtin_amdgcn_s_waitcnt(0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (tid == 0) {
        trace[3] = __builtin_amdgcn_s_memtime();
        trace[7] = (trace_real0 & 0xffffffffull) | ((__builtin_amdgcn_s_memrealtime() - trace_real0) << 32);
      }
    }
    return;
  }
  // every bias / residual / position row a 32-row block needs is requested one block ahead (block 0 before the last
  // K step), so the memory latency of the residual stream (MALL/HBM) is covered by the previous block's transpose
  if (!add_ready) load_block(0, add[0]);
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    if (kUneven && i >= mi_w) break;   // wave-uniform
    if constexpr (kAddBufs == 2) {
      if (i + 1 < MI && !(kUneven && i + 1 >= mi_w)) load_block(i + 1, add[(i + 1) & 1]);
    } else {
      if (i > 0) load_block(i, add[0]);
    }
#pragma unroll
    for (int jp = 0; jp < NI / 2; ++jp) {
      if constexpr (kM16) {   // row 16b + l16, columns jj*32 + 16a + 4 g16 .. +3 (8 consecutive lanes = 8 rows x 16 B: all 32 banks once)
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
              if (16 * b >= slab_rows(i)) continue;
              *reinterpret_cast<f32x4*>(slab + (16 * b + l16) * SLAB_PITCH + (jj * 32 + 16 * a + 4 * g16) * 4) =
                  acc4[2 * i + b < MI2 ? 2 * i + b : 0][2 * (2 * jp + jj) + a];
            }
      } else {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 v = {acc[i][2 * jp + jj][4 * q + 0], acc[i][2 * jp + jj][4 * q + 1], acc[i][2 * jp + jj][4 * q + 2],
                           acc[i][2 * jp + jj][4 * q + 3]};
          *reinterpret_cast<f32x4*>(slab + lrow * SLAB_PITCH + (jj * 32 + 8 * q + 4 * lgrp) * 4) = v;
        }
      }
      __builtin_amdgcn_wave_barrier();  // LDS ops of one wave execute in order; keep the compiler from reordering
      if constexpr (EPI == EPI_RESID_SPLIT) {
        // 8 columns per lane, 8 rows per pass: every global access of the two planes is a full 16-byte piece
        const int r8 = lane >> 3, c8 = (lane & 7) * 8;
        const int nn = n0 + wn * TN + jp * 64 + c8;
        const float4 b0 = *reinterpret_cast<const float4*>(p.bias + nn), b1 = *reinterpret_cast<const float4*>(p.bias + nn + 4);
        f32x4 va[4], vb[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          if (it * 8 >= slab_rows(i)) continue;
          va[it] = *reinterpret_cast<const f32x4*>(slab + (it * 8 + r8) * SLAB_PITCH + c8 * 4);
          vb[it] = *reinterpret_cast<const f32x4*>(slab + (it * 8 + r8) * SLAB_PITCH + c8 * 4 + 16);
        }
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {           // pass pairs: rows r8 and r8 + 8 of a 16-row band share one lo piece
          if (pr * 16 >= slab_rows(i)) continue;
          const u32x4 l = __builtin_bit_cast(u32x4, add[kAddBufs == 2 ? (i & 1) : 0][jp][4 + pr]);
          u32x4 lo4;
          const int m_first = m0 + wm * TM + i * 32 + pr * 16 + r8;
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int it = 2 * pr + hh;
            const int m = m_first + 8 * hh;
            const bool in_range = m < Mrt;
            const u32x4 h = __builtin_bit_cast(u32x4, add[kAddBufs == 2 ? (i & 1) : 0][jp][it]);
            float o[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {             // columns 2e, 2e + 1: bytes 2e, 2e + 1 of this row's 8-byte half of the lo piece
              o[2 * e] = join_f32<T>(h[e] & 0xffffu, sbyte(l[2 * hh + (e >> 1)], (2 * e) & 3));
              o[2 * e + 1] = join_f32<T>(h[e] >> 16, sbyte(l[2 * hh + (e >> 1)], (2 * e + 1) & 3));
            }
            // (residual + bias) + product: the order of the plain-array epilogues
            o[0] = (o[0] + b0.x) + va[it][0]; o[1] = (o[1] + b0.y) + va[it][1]; o[2] = (o[2] + b0.z) + va[it][2];
            o[3] = (o[3] + b0.w) + va[it][3]; o[4] = (o[4] + b1.x) + vb[it][0]; o[5] = (o[5] + b1.y) + vb[it][1];
            o[6] = (o[6] + b1.z) + vb[it][2]; o[7] = (o[7] + b1.w) + vb[it][3];
            const float ssum = row8_sum(((o[0] + o[1]) + (o[2] + o[3])) + ((o[4] + o[5]) + (o[6] + o[7])));
            const float mj = ssum * (1.0f / kLnSlice);
            float q2 = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = o[e] - mj; q2 = fmaf(d, d, q2); }
            const float m2 = row8_sum(q2);
            u32x4 ho;
            unsigned lb[8];
            using TO = std::conditional_t<std::is_same_v<T, bf16_t>, f16_t, bf16_t>;   // the other 16-bit type
            if (p.planes_other) {   // wave-uniform
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                unsigned ha, hb;
                split_f32<TO>(o[2 * e], ha, lb[2 * e]);
                split_f32<TO>(o[2 * e + 1], hb, lb[2 * e + 1]);
                ho[e] = ha | (hb << 16);
              }
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                unsigned ha, hb;
                split_f32<T>(o[2 * e], ha, lb[2 * e]);
                split_f32<T>(o[2 * e + 1], hb, lb[2 * e + 1]);
                ho[e] = ha | (hb << 16);
              }
            }
            lo4[2 * hh] = lb[0] | (lb[1] << 8) | (lb[2] << 16) | (lb[3] << 24);
            lo4[2 * hh + 1] = lb[4] | (lb[5] << 8) | (lb[6] << 16) | (lb[7] << 24);
            if (in_range) {
              store16(reinterpret_cast<unsigned short*>(p.xb_out) + (size_t)m * p.ldc + nn, ho);
              if ((lane & 7) == 0)
                *reinterpret_cast<float2*>(p.st_out + ((size_t)m * (p.N / kLnSlice) + (n0 + wn * TN + jp * 64) / kLnSlice) * 2) =
                    make_float2(ssum, m2);
            }
          }
          // (a band's second row may lie past the live rows while its first does not: its byte half then goes to the plane's padding)
          if (m_first < Mrt)
            store16(reinterpret_cast<unsigned char*>(p.lo_io) + (size_t)(m_first >> 4) * 16 * p.ldc + (size_t)((nn >> 3)) * 128 + r8 * 16, lo4);
        }
        __builtin_amdgcn_wave_barrier();
        continue;
      }
      const int n = n0 + wn * TN + jp * 64 + rd_col;
      f32x4 v[8];
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        if (it * 4 >= slab_rows(i)) continue;
        v[it] = *reinterpret_cast<const f32x4*>(slab + (it * 4 + rd_row) * SLAB_PITCH + rd_col * 4);
      }
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        if (it * 4 >= slab_rows(i)) continue;
        int m = m0 + wm * TM + i * 32 + it * 4 + rd_row;
        const bool in_range = m < Mrt;
        if constexpr (EPI == EPI_RESID_EMIT) {
          // the updated residual row piece (4 columns per lane, 16 lanes = one 64-column slice of one row): fp32 in
          // place, its 16-bit copy for the next GEMM's A operand, and the slice's LayerNorm partials {sum, centred M2}
          // (EPI_RESID_SPLIT has its own block above)
          const float4 a4 = add[kAddBufs == 2 ? (i & 1) : 0][jp][it];
          const float o0 = a4.x + v[it][0], o1 = a4.y + v[it][1], o2 = a4.z + v[it][2], o3 = a4.w + v[it][3];
          const float ssum = row16_sum((o0 + o1) + (o2 + o3));
          const float mj = ssum * (1.0f / kLnSlice);
          const float d0 = o0 - mj, d1 = o1 - mj, d2 = o2 - mj, d3 = o3 - mj;
          const float m2 = row16_sum((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
          if (in_range) {
            store4(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n, o0, o1, o2, o3);
            store4(reinterpret_cast<OutT*>(p.xb_out) + (size_t)m * p.ldc + n, o0, o1, o2, o3);
            if ((lane & 15) == 0)
              *reinterpret_cast<float2*>(p.st_out + ((size_t)m * (p.N / kLnSlice) + (n0 + wn * TN + jp * 64) / kLnSlice) * 2) =
                  make_float2(ssum, m2);
          }
        } else {
          if (in_range) EpilogueOp<T, EPI>::store(p, m, n, v[it][0], v[it][1], v[it][2], v[it][3], add[kAddBufs == 2 ? (i & 1) : 0][jp][it]);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (trace) {
    __builtin_amdgcn_s_waitcnt(0);  // stores issued and acknowledged before the stamp
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid == 0) {
      trace[3] = __builtin_amdgcn_s_memtime();
      trace[7] = (trace_real0 & 0xffffffffull) | ((__builtin_amdgcn_s_memrealtime() - trace_real0) << 32);
    }
  }
}

// One thread per output element; the on-device checker for the MFMA kernels
// (tests: variant -2), never used on the product path.
template <typename T, int EPI>
__global__ void gemm_nt_naive_kernel(const GemmParams p) {
  const int n4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int m = blockIdx.y;
  if (n4 >= p.N || m >= p.M) return;
  const T* a = reinterpret_cast<const T*>(p.A) + (size_t)m * p.lda;
  float v[4];
  for (int e = 0; e < 4; ++e) {
    const T* w = reinterpret_cast<const T*>(p.W) + (size_t)(n4 + e) * p.ldw;
    float s = 0.f;
    for (int k = 0; k < p.K; ++k) s = fmaf(to_f32(a[k]), to_f32(w[k]), s);
    v[e] = s;
  }
  EpilogueOp<T, EPI>::store(p, m, n4, v[0], v[1], v[2], v[3], EpilogueOp<T, EPI>::load(p, m, n4));
}

// ---- host side ----------------------------------------------------------------
struct GemmVariant {
  const char* name;
  int bm, bn, threads;
};
int gemm_num_variants();
int gemm_num_cus();   // compute units of the current device, rounded down to a multiple of 8 (gemm.hip)
const GemmVariant& gemm_variant(int v);
// dtype: 0 fp32, 1 bf16, 2 f16.  variant -1 = auto, -2 = naive.  Returns hipError_t as int; *kernel_name (optional)
// receives a static string naming the kernel that ran.
// (variant -1: the tile with the smallest rounds x tile-time on this shape, gemm.hip)
int gemm_launch(int dtype, int epi, int variant, const GemmParams& p, hipStream_t stream, const char** kernel_name);
int gemm_default_variant(int dtype, int M, int N, int K);
bool gemm_variant_is_built(int dtype, int variant);
// small-M kernel of the 16-bit engines (gemm_skinny.hip): 32 x 64 output tile per workgroup, K split over its waves
bool gemm_skinny_supports(int epi, int M, int N, int K);
int gemm_launch_skinny(int dtype, int epi, const GemmParams& p, hipStream_t stream, const char** kernel_name);
// tests / A-B runs (process-wide hooks behind plipmi_test.h, not product knobs); false = value out of range, nothing changed
bool gemm_force_tile(int variant);            // -1 the cost model chooses, -2 the naive checker, >= 0 that tile for every launch
bool gemm_remap_tile(int from, int to);       // the cost model's choice `from` runs as tile `to` (-1: as itself again)
void gemm_reset_overrides();
// the patch GEMM with its A operand gathered from fp32 pixels while it is staged (ADDR 2: im2col on load, no unfold pass)
bool gemm_gather_supports(int dtype, int B, int image, int patch, int N);
int gemm_launch_gather(int dtype, const GemmParams& p, hipStream_t stream, const char** kernel_name);

}  // namespace plipmi
