// kernels.h -- launchers of the HBM-bound kernels around the GEMMs (definitions in kernels.hip / attention.hip).
#pragma once
#include "common.h"

namespace plipmi {

// dtype codes used by the launchers: 0 = fp32, 1 = bf16, 2 = f16 (same as PLIPMI_F32 / PLIPMI_BF16 / PLIPMI_F16)

// y[r,:] = LayerNorm(x[r*x_row_stride : +D]) * g + b ; y is fp32 or bf16, contiguous [rows, D].
// One wavefront per row, statistics in fp32 (two-pass, values held in registers).
hipError_t launch_layernorm(const float* x, size_t x_row_stride, const float* g, const float* b, void* y, int y_dtype,
                            int rows, int D, float eps, hipStream_t s);

// LayerNorm folded into the GEMMs (16-bit engines, gemm.h EPI_*_LN): the one LayerNorm pass a tower keeps -- fp32 rows in,
// the normalised rows out as the split residual stream (common.h split_f32<H>: hi = the 16-bit operand plane [rows, D],
// lo = the 8-bit remainder plane in its blocked layout, lo_plane_bytes(rows, D) bytes) plus their statistics partials st [rows, D/64, 2]
// (D % 64 == 0).  dtype (1 = bf16, 2 = f16) selects H.
hipError_t launch_layernorm_emit(const float* x, const float* g, const float* b, void* hi, void* lo, float* st, int rows, int D,
                                 float eps, int dtype, hipStream_t s);
// the two planes [rows, D] back to plain fp32 (D % 8 == 0)
hipError_t launch_join_planes(const void* hi, const void* lo, float* x, size_t rows, int D, int dtype, hipStream_t s);
// {hi, lo} planes [rows, D] from one 16-bit operand type's split format to the other's (1 bf16, 2 f16), in place (one rounding of the remainder)
hipError_t launch_recode_planes(void* hi, void* lo, size_t rows, int D, int from_dtype, int to_dtype, hipStream_t s);
// weight folding at plipmi_create: Wf[n,:] = H(pre * (W[n,:] * g - mean_k(W[n,:] * g))), c2[n] = pre * (W[n,:].b + bias[n])
hipError_t launch_fold_ln(const float* W, const float* bias, const float* g, const float* b, void* Wf, float* c2, int rows,
                          int K, float pre, int dtype, hipStream_t s);
// token + position embedding with the same by-products (text tower's first block).  bad_id: see launch_text_embed
hipError_t launch_text_embed_emit(const int64_t* ids, const float* tok, const float* pos, void* hi, void* lo, float* st, int B,
                                  int S, int D, int vocab, int* bad_id, int dtype, hipStream_t s);

// pixels fp32 [B,3,H,W] -> patch rows [B*np, Kpad] (dtype), column (c,u,v), zero padded to Kpad.
hipError_t launch_unfold_patches(const float* pixels, void* out, int out_dtype, int B, int image, int patch, int Kpad,
                                 hipStream_t s);

// Same unfold for raw tiles: uint8 [B,H,W,3] (HWC, what PIL / np.asarray give) with the CLIP normalisation
// (u8/255 - mean[c]) / std[c] of reproducibility/embedders/transform.py:45-52 fused in.
hipError_t launch_unfold_patches_u8(const uint8_t* tiles, void* out, int out_dtype, int B, int image, int patch,
                                    int Kpad, hipStream_t s);

// x[b,0,:] = class_embedding + pos[0,:]   (token rows 1.. are written by the patch GEMM epilogue)
hipError_t launch_cls_rows(const float* cls, const float* pos, float* x, int B, int tokens, int D, hipStream_t s);

// x[b,s,:] = tok[ids[b,s],:] + pos[s,:].  An id outside [0,vocab) is clamped for the lookup and raises *bad_id (a flag
// in host-visible memory; nullptr = no report): the reference's lookup raises there (plip.py:68)
hipError_t launch_text_embed(const int64_t* ids, const float* tok, const float* pos, float* x, int B, int S, int D,
                             int vocab, int* bad_id, hipStream_t s);

// Pooled head: row = CLS (ids == nullptr) or the EOS row of each caption, then
// LayerNorm -> bias-free projection (Wt is the projection TRANSPOSED: [D, P]) -> optional L2 normalise.
hipError_t launch_pool_head(const float* x, int S, int D, const int64_t* ids, int eos_id, const float* ln_w,
                            const float* ln_b, float eps, const float* Wt, int P, float* out, int B, int normalize,
                            hipStream_t s);

// Pooled row (CLS, or the EOS row of each caption) -> LayerNorm -> fp32 [B, D]; the projection itself then
// runs on the fp32 MFMA GEMM (used when P % 128 == 0, i.e. every real CLIP/PLIP head).
hipError_t launch_pool_layernorm(const float* x, int S, int D, const int64_t* ids, int eos_id, const float* ln_w,
                                 const float* ln_b, float eps, float* out, int B, hipStream_t s);

// pooled row of every sample (CLS when ids == nullptr, else the caption's EOS row): attention output (bf16) and residual
// row (hi/lo planes -> fp32) copied to compact [B, D] buffers -- the inputs of the last block's pooled-row-only
// out_proj / fc1 / fc2
// cu != nullptr: packed rows, the pooled row of sample b is its last one, cu[b+1]-1
hipError_t launch_pool_gather(const void* att, const void* hi, const void* lo, int S, int D, const int64_t* ids, int eos_id,
                              void* attp, float* xp, int B, int dtype, hipStream_t s, const int* cu = nullptr);

hipError_t launch_l2_normalize(float* x, int N, int D, hipStream_t s);
hipError_t launch_occupy(unsigned long long wall_clock_ticks, hipStream_t s);   // one workgroup, no memory traffic (plipmi_streams_overlap)
// C[M,N] = A[M,K] . W[N,K]^T, exact fp32 MFMA, split-K over the four waves of a 32x32-tile workgroup (N, K % 32 == 0)
// (also the MFMA form of the logits: C = scale * A . W^T; exchanging A and W gives the bit-exact transpose)
hipError_t launch_head_gemm(const float* A, const float* W, float* C, int M, int N, int K, hipStream_t s, float scale = 1.0f);
// out[i] = first arg-max of row i of x [N, M]
hipError_t launch_row_argmax(const float* x, int N, int M, int32_t* out, hipStream_t s);

// logits_per_image[i,j] = scale*<img_i,txt_j>; optional transpose output and per-row first arg-max.
hipError_t launch_logits(const float* img, int Ni, const float* txt, int Nt, int D, float scale, float* lpi,
                         float* lpt, int32_t* argmax, hipStream_t s);

hipError_t launch_topk(const float* scores, int N, int M, int k, int64_t* idx, hipStream_t s);

// streaming top-k: fold the panel scores[rows, 0:ncols] (row stride ld, global column offset col0) into the running
// per-row lists vals/idx [rows, k] (descending; ties: lower index first); init before the first panel, finish after the last
constexpr int kTopkMaxK = 1024;
hipError_t launch_topk_init(float* vals, int64_t* idx, size_t n, hipStream_t s);
hipError_t launch_topk_merge(const float* scores, size_t ld, int rows, int ncols, int64_t col0, int k, float* vals,
                             int64_t* idx, hipStream_t s);
hipError_t launch_topk_finish(int64_t* idx, size_t n, hipStream_t s);

// Pillow-exact 8-bit bicubic resize + centre crop of uint8 HWC images (tables from plip_amd/preprocess.py):
// src [B,H,W,3] -> tmp [B,R,n,3] (rows r0..r0+R of the source, horizontally resampled) -> dst [B,n,n,3]
hipError_t launch_resize_crop_u8(const uint8_t* src, int B, int H, int W, int n, const int* xb, const int* xk, int xks,
                                 int left, const int* yb, const int* yk, int yks, int top, int r0, int R,
                                 uint8_t* tmp, uint8_t* dst, hipStream_t s);

// dst[r, 0:cols] = (T)(scale * src[r, 0:cols]), dst[r, cols:dst_ld] = 0   (weight packing)
hipError_t launch_convert(const float* src, void* dst, int dst_dtype, int rows, int cols, int dst_ld, float scale,
                          hipStream_t s);
// dst[c, r] = src[r, c]   fp32 (projection weights -> [D, P])
hipError_t launch_transpose(const float* src, float* dst, int rows, int cols, hipStream_t s);
hipError_t launch_scale_copy(const float* src, float* dst, int n, float scale, hipStream_t s);

// Multi-head attention over the fused qkv buffer [B*S, 3*D] (q | k | v, head h at columns h*64..), the 1/sqrt(64)
// scale already folded into q.  out [B*S, D].  causal: key j <= query i.  key_mask: int64 [B,S] or nullptr.
//   impl 0 = exact fp32 VALU kernel (any dtype), 1 = MFMA kernel (dtype bf16 or f16)
//   cu (impl 1, S <= 128): packed rows -- sample b owns rows cu[b] .. cu[b+1]-1 of qkv / out (its first cu[b+1]-cu[b]
//   positions; the rest of its S positions do not exist); key_mask keeps its [B, S] layout
hipError_t launch_attention(const void* qkv, void* out, int dtype, int B, int S, int H, int causal,
                            const int64_t* key_mask, int impl, hipStream_t s, const int* cu = nullptr);

// The text tower's LayerNorm-folded q/k/v projection with the attention in its epilogue (qkv_attention.hip): one launch, no
// `qkv` tensor in memory.  A = the residual stream's operand plane [B*S, D], W / c2 = the folded q | k | v weights [3D, D] and
// biases, stats = the rows' LayerNorm partials [B*S, D/64, 2]; out = attention output [B*S, D], bit-identical to
// gemm.h EPI_BIAS_LN followed by launch_attention(impl 1).  16-bit dtypes, 65 .. 80 tokens, D = 64 H.
bool qkv_attention_supports(int dtype, int B, int S, int H, int D);
bool qkv_attention_pays(int B, int H, int num_cus);   // enough workgroups to fill the chip: below, the two kernels are faster
hipError_t launch_qkv_attention(int dtype, const void* A, const void* W, const float* c2, const float* stats, float ln_inv_d,
                                float ln_eps, void* out, int B, int S, int H, int causal, const int64_t* key_mask, hipStream_t s,
                                unsigned long long* trace = nullptr /* test hook: in-kernel timeline, 8 x u64 per workgroup */);

// Packed captions (text tower, opt-in): a causal tower's pooled output depends on rows 0 .. EOS only, so the rows past a
// caption's EOS token need not exist.  One workgroup: len[b] = eos_position(ids[b]) + 1, cu = exclusive prefix sums
// [B+1], rowmap[r] = (b << 8) | t for packed row r (S <= 256), *m_dev = cu[B] = live rows.
hipError_t launch_text_pack(const int64_t* ids, int B, int S, int eos_id, int* cu, int* rowmap, int* m_dev, hipStream_t s);
// token + position embedding of the packed rows (split planes + statistics partials, as launch_text_embed_emit)
hipError_t launch_text_embed_emit_packed(const int64_t* ids, const float* tok, const float* pos, void* hi, void* lo, float* st,
                                         const int* rowmap, const int* m_dev, int max_rows, int S, int D, int vocab,
                                         int* bad_id, int dtype, hipStream_t s);

}  // namespace plipmi
