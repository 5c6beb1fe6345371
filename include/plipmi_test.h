/*
 * plipmi_test.h -- kernel-level test entries and A/B hooks of libplipmi.so.
 *
 * NOT part of the product interface (include/plipmi.h, the drop-in boundary INTEGRATION.md maps to the reference's call
 * sites): these exports exist so that tests/ can compare single kernels with fp64 references through the same C ABI, and so
 * that tools/ can A/B tile choices on one box.  The product path (plip_amd/, bench.py's timed region) never calls them.
 * The plipmi_test_* setters are PROCESS-WIDE state (every handle, every later launch): test hooks only.
 */
#ifndef PLIPMI_TEST_H
#define PLIPMI_TEST_H

#include "plipmi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Parity tests only: run `tower` on `input` (pixels or ids; mask = NULL) through its
 * first `layer` blocks and copy the fp32 residual stream [B,S,D] to `out`
 * (layer 0 = embeddings, after pre_layrnorm for the vision tower -- HF hidden_states[layer]). */
int plipmi_debug_hidden(plipmi_handle h, int tower, int layer, const void* input, int B, float* out, void* stream);

/* Kernel-level entry for unit tests and micro-benchmarks of the GEMM that carries
 * >98 % of the path's FLOPs:  C = epilogue(A[M,K] * W[N,K]^T).
 *   dtype    PLIPMI_F32 | PLIPMI_BF16 | PLIPMI_F16 (A, W and non-fp32 outputs are that type; 16-bit types as raw uint16)
 *   epilogue 0: C(dtype) = acc + bias        1: C(dtype) = quickgelu(acc + bias)
 *            2: C(f32) += acc + bias         3: C(f32)   = alpha * acc
 *   variant  -1 = the engine's own choice, >= 0 = a specific tile configuration
 *            (plipmi_gemm_variant_name lists them; NULL past the end);
 *            variant -2 = the naive one-thread-per-output checker kernel;
 *            variant -3 = the small-M split-K kernel (16-bit types; epilogues 0..2; N % 64 == 0, K % 256 == 0). */
int plipmi_gemm_nt(int dtype, int epilogue, int variant, int M, int N, int K, const void* A, const void* W,
                   const float* bias, float alpha, void* C, void* stream);
const char* plipmi_gemm_variant_name(int variant);
/* 1 if this build of the library carries `variant` for `dtype`, else 0 */
int plipmi_gemm_variant_built(int dtype, int variant);
/* TEST / A-B HOOKS, process-wide, never called by the product path (the library reads no environment variables either).  Each
 * returns PLIPMI_OK, or PLIPMI_ERR_INVALID for a value outside its range (nothing changes then).  A hook changes what LATER launches
 * are; hipGraphs a handle captured earlier are dropped and re-captured at its next small-batch call.
 *   plipmi_test_force_gemm_tile(v)     every GEMM on tile v (0 .. n-1, plipmi_gemm_variant_name), -2 the naive checker kernel,
 *                                      -1 the engine's own cost model again
 *   plipmi_test_remap_gemm_tile(a, b)  where the cost model chooses tile a, run tile b (A/B runs of the whole step); b = -1 clears
 *   plipmi_test_fused_qkv_attention(m) the text tower's q/k/v projection + attention: 0 two kernels, 1 the product rule (fused where it
 *                                      applies and the batch fills the chip; default), 2 fused wherever it applies
 *   plipmi_test_patch_gather(on)       fp32 pixels: 0 unfold pass + plain patch GEMM, 1 im2col on load where it applies (default)
 *   plipmi_test_reset_hooks()          all of the above back to the product behaviour */
int plipmi_test_force_gemm_tile(int variant);
int plipmi_test_remap_gemm_tile(int from, int to);
int plipmi_test_fused_qkv_attention(int mode);
int plipmi_test_patch_gather(int on);
void plipmi_test_reset_hooks(void);
/* Test hook: the residual-stream planes {hi [rows, D] uint16, lo = the 8-bit remainder plane in its blocked layout,
 * ((rows + 15) / 16) * 16 * D bytes (plip_amd/csrc/common.h lo_plane_off)} from `from_dtype`'s split format to `to_dtype`'s
 * (PLIPMI_BF16 / PLIPMI_F16), in place -- what a text tower with plipmi_config.text_f16_layers does between its f16 and its
 * bf16 blocks on the small-M path.  The value is joined and split again: the new hi is the value rounded to the new operand type,
 * the remainder is rounded once more (2^-16 relative for bf16, 2^-19 for f16).  D % 8 == 0. */
int plipmi_recode_planes(void* hi, void* lo, size_t rows, int D, int from_dtype, int to_dtype, void* stream);
/* same call with explicit leading dimensions (in elements) for A [M,K] and W [N,K]: rows may be padded */
int plipmi_gemm_nt_ld(int dtype, int epilogue, int variant, int M, int N, int K, const void* A, int lda, const void* W,
                      int ldw, const float* bias, float alpha, void* C, void* stream);
/* same call with an in-kernel timeline: trace = device buffer of 8 x uint64 per workgroup
 * {start, prologue done, main loop done, epilogue done (s_memtime ticks: shader cycles, one counter per XCD), tile id,
 *  HW_ID|XCC_ID<<32, k tiles, start (low 32 bits) | lifetime << 32 in s_memrealtime ticks (100 MHz, device-wide)} */
int plipmi_gemm_nt_traced(int dtype, int epilogue, int variant, int M, int N, int K, const void* A, const void* W,
                          const float* bias, float alpha, void* C, uint64_t* trace, void* stream);

/* Kernel-level entry for the LayerNorm-folded epilogues of the 16-bit engines (gemm.h EPI_BIAS_LN / EPI_QGELU_LN /
 * EPI_RESID_EMIT / EPI_RESID_SPLIT; dtype PLIPMI_BF16 | PLIPMI_F16, A, W as raw uint16; "h16" below = that type):
 *   mode 0: C(h16) = rstd[m] * A.W^T + bias[n]       rstd from `stats` [M, ns, 2] fp32 = per-64-column partials
 *   mode 1: C(h16) = quickgelu(that)                 {sum, centred M2} of the LayerNorm input rows (D = 64 * ns), eps as
 *                                                     given; W is expected to carry LayerNorm's gain with CENTRED rows
 *                                                     (sum_k W[n,k] = 0), which is what subtracts the row mean
 *   mode 2: C(f32) += A.W^T + bias;  xb_out(h16)[M,N] = C;  st_out [M, N/64, 2] = partials of the updated rows
 *   mode 3: the same update on a residual kept as two planes, xb_out = hi (uint16 [M, N]: the value rounded to h16) and
 *           C = lo (8-bit remainders in the blocked layout of plip_amd/csrc/common.h lo_plane_off, ((M + 15) / 16) * 16 * N bytes;
 *           bf16: bits(x) ~ (hi << 16) + (lo << 8), f16: x ~ hi + lo * 2^(E(hi) - 18)): the stream at 16 / 19 significand bits,
 *           both planes read and written in place; st_out as in mode 2.  This is the form the engine runs (6 bytes per element
 *           of epilogue traffic; the hi plane is the next GEMM's A operand)
 *   mode 4: mode 3 that READS the planes in dtype's split format and WRITES them in the other 16-bit type's (the last f16
 *           block of a bf16 text tower with plipmi_config.text_f16_layers hands the stream over without a re-coding pass) */
int plipmi_gemm_nt_ln(int dtype, int mode, int variant, int M, int N, int K, const void* A, const void* W, const float* bias,
                      const float* stats, int ns, float eps, void* C, void* xb_out, float* st_out, void* stream);

/* Kernel-level entry for the text tower's fused kernel (csrc/qkv_attention.hip): LayerNorm-folded q/k/v projection (mode 0 of
 * plipmi_gemm_nt_ln with N = 3 * 64 H) with the attention in its epilogue, out [B*S, 64 H] -- bit-identical to
 * plipmi_gemm_nt_ln(mode 0) followed by plipmi_attention(impl 1).  dtype PLIPMI_BF16 | PLIPMI_F16, 65 <= S <= 80, ns = H.
 * (plipmi_test_fused_qkv_attention(0) makes the ENGINE run the two kernels instead, 1 the product rule, 2 the fused one at every batch.) */
int plipmi_qkv_attention(int dtype, const void* A, const void* W, const float* c2, const float* stats, int ns, float eps, void* out,
                         int B, int S, int H, int causal, const int64_t* key_mask,
                         uint64_t* trace /* NULL, or 8 x uint64 per workgroup: {start, prologue done, K loop done, Q/K/V images
                                            written, end (s_memtime), tile id, 0, start | lifetime << 32 (s_memrealtime)} */,
                         void* stream);

/* Kernel-level entry for the attention kernels: out[B*S, H*64] = softmax(q k^T + masks) v over the fused
 * activation qkv [B*S, 3*H*64] (q | k | v, 1/sqrt(64) already folded into q).
 *   impl 0 = exact-fp32 VALU kernel (any dtype), impl 1 = MFMA kernels (bf16 / f16; single pass for S <= 128, chunked
 *   online softmax beyond). */
int plipmi_attention(int dtype, int impl, const void* qkv, void* out, int B, int S, int H, int causal,
                     const int64_t* key_mask, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PLIPMI_TEST_H */
