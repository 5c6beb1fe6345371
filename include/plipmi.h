/*
 * plipmi.h -- C ABI of libplipmi.so, the MI355X (gfx950) PLIP embedding engine.
 *
 * This is the drop-in boundary for the one hot path of PathologyFoundation/plip:
 * the batched encode_images / encode_text forward and the image x text
 * cosine-similarity logits.  The reference has no FFI layer of its own -- the
 * seam is "the object stored in self.model" (reference plip.py:18) -- so every
 * entry point below names the reference call it replaces:
 *
 *   plipmi_create         <- CLIPModel.from_pretrained(...).to(device)        plip.py:26,18
 *                            clip.load(arch) + load_state_dict(...)           reproducibility/embedders/factory.py:21-25
 *   plipmi_encode_image   <- self.model.get_image_features(**batch)           plip.py:50
 *                            self.model.encode_image(images)                  reproducibility/embedders/plip.py:48
 *   plipmi_encode_image_u8<- self.preprocess(images=...) + get_image_features           plip.py:32-35,50
 *   plipmi_resize_crop_u8 <- Resize(n_px, bicubic) + CenterCrop(n_px)                  reproducibility/embedders/transform.py:45-48
 *   plipmi_encode_text    <- self.model.get_text_features(**batch)            plip.py:68
 *                            self.model.encode_text(clip.tokenize(...))       reproducibility/embedders/plip.py:65-66
 *   plipmi_l2_normalize   <- x / np.linalg.norm(x, axis=-1, keepdims=True)    plip.py:75, embedders/plip.py:53,73
 *   plipmi_logits         <- CLIPModel.forward -> logits_per_image/_text      README.md:45-50 (HF modeling_clip.py:810-817)
 *                            image_embeddings.dot(text_embeddings.T); argmax  reproducibility/evaluation/zero_shot/zero_shot.py:12-13
 *   plipmi_topk           <- cosine_sim.argsort()[:, -k:][:, ::-1]            plip.py:78-87, evaluation/retrieval/retrieval.py:13-18
 *   plipmi_similarity_topk<- the same two call sites, fused with the dot product (no [N,N] matrix)
 *
 * Conventions
 *   - plain C, no torch / HIP types in the signatures: device buffers are raw
 *     `void*` device addresses (tensor.data_ptr()), the stream is a
 *     hipStream_t passed as `void*` (NULL = the legacy default stream).
 *   - every call only ENQUEUES work on the given stream and returns; the
 *     caller synchronises.  Inputs are never modified; outputs are caller-owned.
 *   - one handle per GPU / process, not thread-safe per handle.
 *   - return value: 0 = OK, anything else = error; the message is available
 *     from plipmi_last_error() (thread local).
 *   - fp32 tensors are row-major C-contiguous; pixel input is NCHW.
 */
#ifndef PLIPMI_H
#define PLIPMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLIPMI_VERSION 412 /* 0.4.1: `pass_batch` appended to the config struct -- a 0.4.0 caller's shorter struct means 0 = automatic;
                            * 0.4.0: plipmi_config starts with `struct_size` (the struct can grow at its tail without breaking
                            * callers compiled against an older header); test / A-B hooks moved to plipmi_test.h
                            * (0.3.1: `text_f16_layers`, PLIPMI_ERR_TOKEN_ID; 0.3.0: `flags`, `graph_batch`, PLIPMI_F16) */

/* arithmetic the towers' GEMMs and attention run in (accumulation, LayerNorm,
 * softmax statistics, residual stream, projections and logits are always fp32):
 *   PLIPMI_F32   exact fp32 MFMA (v_mfma_f32_32x32x2_f32)
 *   PLIPMI_BF16  bf16 MFMA operands (8 significand bits)  -- BASELINE.json configs[2] as named
 *   PLIPMI_F16   IEEE half MFMA operands (11 significand bits, +-65504; outputs saturate), same matrix-core rate as
 *                bf16.  The reference's own GPU path runs its CLIP in this type (OpenAI clip.load on "cuda" ->
 *                convert_weights: reproducibility/embedders/factory.py:21, scripts/extract_embedding.py:94-97).
 * (The experimental fp8-weights mode of earlier versions, value 2, is gone: 2.7e-3 against the 1e-3 cosine bar.) */
enum { PLIPMI_F32 = 0, PLIPMI_BF16 = 1, PLIPMI_F16 = 2 };

/* plipmi_config.flags -- every switch of a handle's behaviour is an argument of plipmi_create (or a per-handle setter
 * below); the library reads no environment variables for them */
enum {
  PLIPMI_FLAG_SEPARATE_LAYERNORM = 1,  /* 16-bit engines: run the 2 x L block LayerNorms as their own kernels instead
                                        * of folding them into the q/k/v and fc1 GEMMs (A/B measurements) */
  PLIPMI_FLAG_DENSE_LAST_BLOCK = 2,    /* compute the last block's out_proj / fc1 / fc2 on every token, as the reference
                                        * does, instead of on the pooled row of each sample only */
  PLIPMI_FLAG_PACK_CAPTIONS = 4,       /* start with caption packing on (plipmi_set_text_packing) */
  PLIPMI_FLAG_VALU_ATTENTION = 8,      /* exact-fp32 VALU attention kernel instead of the MFMA kernels (A/B measurements) */
  PLIPMI_FLAG_TEXT_TOWER_F16 = 16      /* bf16 engine only: the TEXT tower's operands (weights, activations, its plane of
                                        * the residual stream) are IEEE half instead of bf16.  The text side carries
                                        * 3.2x the image side's embedding error in bf16 (operand rounding of the weights,
                                        * coherent over a caption's tokens) and 40 % of the step's time:
                                        * half precision there buys most of the f16 engine's accuracy for a third of
                                        * its cost.  The image tower stays bf16. */
};

/* towers, for plipmi_debug_hidden */
enum { PLIPMI_VISION = 0, PLIPMI_TEXT = 1 };

/* status codes */
enum {
  PLIPMI_OK = 0,
  PLIPMI_ERR_INVALID = 1,   /* bad argument / unsupported shape */
  PLIPMI_ERR_HIP = 2,       /* a HIP runtime call failed */
  PLIPMI_ERR_NODEVICE = 3,  /* no gfx950 device visible */
  PLIPMI_ERR_NOMEM = 4,
  PLIPMI_ERR_TOKEN_ID = 5   /* an EARLIER plipmi_encode_text on the handle saw a token id outside the vocabulary (below) */
};

typedef struct plipmi_engine* plipmi_handle;

/* Architecture = the numbers in HF CLIPConfig (configuration_clip.py:47-64,97-109).
 * `struct_size` = sizeof(plipmi_config) AS THE CALLER COMPILED IT: plipmi_create reads that many bytes and takes every later
 * member as 0 (= its default), so members appended in later versions do not turn into garbage for older callers; a size
 * smaller than the 0.4.0 struct (through `max_batch`) or larger than the library's own is rejected. */
typedef struct plipmi_config {
  int32_t struct_size;     /* sizeof(plipmi_config) */
  int32_t image_size;      /* 224 */
  int32_t patch_size;      /* 32  */
  int32_t v_width;         /* 768 */
  int32_t v_layers;        /* 12  */
  int32_t v_heads;         /* 12  (head_dim must be 64) */
  int32_t v_mlp;           /* 3072 */
  int32_t vocab_size;      /* 49408 */
  int32_t context_length;  /* 77 */
  int32_t t_width;         /* 512 */
  int32_t t_layers;        /* 12 */
  int32_t t_heads;         /* 8 */
  int32_t t_mlp;           /* 2048 */
  int32_t projection_dim;  /* 512 */
  float   layer_norm_eps;  /* 1e-5 */
  int32_t compute_dtype;   /* PLIPMI_F32 | PLIPMI_BF16 | PLIPMI_F16 */
  int32_t max_batch;       /* images (and captions) per encode call the workspace is sized for */
  int32_t flags;           /* PLIPMI_FLAG_* bits, 0 = the product defaults */
  int32_t graph_batch;     /* small-batch hipGraph replay: 0 = default (batches of <= min(32, max_batch) samples),
                            * > 0 = that many, < 0 = never (plipmi_set_graph_batch changes it later) */
  int32_t text_f16_layers; /* bf16 engine: this many LEADING blocks of the text tower run on IEEE-half (f16) MFMA operands, the
                            * rest of the tower and the whole image tower on bf16.  The bf16 engine's embedding error is mostly
                            * operand rounding in the text tower's first blocks; 0 = a pure bf16 engine,
                            * t_layers = PLIPMI_FLAG_TEXT_TOWER_F16.  The residual stream crosses the switch as the fp32 value the last f16 block's epilogue
                            * computed, re-split for the bf16 blocks (operand plane + 8-bit remainder). */
  int32_t pass_batch;      /* An encode call of B samples runs as ceil(B / pass_batch) back-to-back passes of equal size on the
                            * caller's stream once B >= 2 * pass_batch, so that one pass's per-block activations stay resident in the
                            * 256 MiB Infinity Cache however large the caller's batch is (the reference's `batch_size` is the
                            * caller's, plip.py:31,55).  Same bits either way: a row's embedding does not depend on the batch it
                            * travels in.  0 = automatic (the largest multiple of 32 samples whose per-block working set fits,
                            * 256 for the 16-bit ViT-B/32 engines; no splitting where fewer than 256 samples fit -- the fp32 engine, ViT-L/14 --:
                            * smaller passes lose more in the GEMMs than the cache returns), < 0 = never split. */
} plipmi_config;

/* One pre-LN transformer block, HF CLIPEncoderLayer naming; all DEVICE pointers
 * to fp32, Linear weights stored [out, in] exactly as in the state dict. */
typedef struct plipmi_layer_weights {
  const float *ln1_w, *ln1_b;
  const float *q_w, *q_b, *k_w, *k_b, *v_w, *v_b, *o_w, *o_b;
  const float *ln2_w, *ln2_b;
  const float *fc1_w, *fc1_b, *fc2_w, *fc2_b;
} plipmi_layer_weights;

/* The whole checkpoint (HF CLIPModel state-dict tensors).  `v_layers` / `t_layers`
 * are HOST arrays of per-layer structs whose members are device pointers.  The
 * engine re-packs everything into its own buffers during plipmi_create, so the
 * caller may free these tensors as soon as the create stream has drained. */
typedef struct plipmi_weights {
  const float* v_class_embedding;   /* [Dv]            vision_model.embeddings.class_embedding */
  const float* v_patch_weight;      /* [Dv,3,P,P]      ...patch_embedding.weight (no bias)      */
  const float* v_pos_embedding;     /* [Np+1,Dv]       ...position_embedding.weight             */
  const float *v_pre_ln_w, *v_pre_ln_b;    /* vision_model.pre_layrnorm */
  const float *v_post_ln_w, *v_post_ln_b;  /* vision_model.post_layernorm */
  const float* visual_projection;   /* [P,Dv]  no bias */
  const plipmi_layer_weights* v_layers;
  const float* t_token_embedding;   /* [V,Dt] */
  const float* t_pos_embedding;     /* [ctx,Dt] */
  const float *t_final_ln_w, *t_final_ln_b;
  const float* text_projection;     /* [P,Dt]  no bias */
  const plipmi_layer_weights* t_layers;
} plipmi_weights;

/* ---- lifetime ----------------------------------------------------------- */
int  plipmi_create(const plipmi_config* cfg, const plipmi_weights* w, void* stream, plipmi_handle* out);
void plipmi_destroy(plipmi_handle h);
/* A second handle on the SAME packed weights with a workspace (activations, staging buffers, captured graphs) of its own: the
 * configuration and the setters' state are copied as they are at the call, the weight memory is shared and lives until the last
 * of the handles that use it is destroyed (any order).  A handle runs one batch per tower at a time -- its workspace is the batch's
 * activations --, so a caller that walks a corpus through ONE tower (the loops of plip.py:41-52 and :64-71, the image side of
 * reproducibility/evaluation/zero_shot/zero_shot.py) gives consecutive batches to two handles on two HIP streams: the launch
 * boundaries, prologues / epilogues and the pooled tail of one batch then run under the GEMMs of the next, as the two towers of a
 * pair do for each other -- configs[3]'s shard 99.1 -> 107.5 k img/s (tools/exp/r06_two_batches.py); same bits per row.  Costs the
 * workspace again (ViT-B/32, max_batch 256: 0.56 GB), no second copy of the weights and no packing time. */
int  plipmi_clone(plipmi_handle src, plipmi_handle* out);
int  plipmi_version(void);
const char* plipmi_last_error(void);
/* name of the device the engine runs on ("gfx950:..."), for logs */
const char* plipmi_device_name(plipmi_handle h);

/* ---- the hot path ----------------------------------------------------------
 * pixels  : fp32 [B,3,H,W] NCHW, already CLIP-normalised (what CLIPProcessor /
 *           reproducibility/embedders/transform.py:45-52 produce)
 * out     : fp32 [B, projection_dim]; un-normalised when normalize == 0 (what
 *           PLIP.encode_images returns, plip.py:53), L2-normalised rows when 1
 *           (what CLIPEmbedder returns, embedders/plip.py:53).
 * B may be anything in [0, max_batch]. */
int plipmi_encode_image(plipmi_handle h, const float* pixels, int B, float* out, int normalize, void* stream);

/* Same, from raw tiles: uint8 [B,H,W,3] (HWC RGB, already image_size x image_size).  The CLIP normalisation
 * (u8/255 - mean)/std of reproducibility/embedders/transform.py:45-52 / HF CLIPImageProcessor is fused into the
 * patch GEMM's operand load (large batches: im2col on load from the HWC bytes, one fma per pixel; small ones: a fused unfold pass --
 * the same bits), so a tile crosses PCIe and HBM as 150 KB instead of 602 KB of fp32. */
int plipmi_encode_image_u8(plipmi_handle h, const uint8_t* tiles, int B, float* out, int normalize, void* stream);

/* ids            : int64 [B, context_length] token ids
 * attention_mask : int64 [B, context_length] (1 = token, 0 = padding) or NULL;
 *                  combined with the causal mask like HF create_causal_mask
 * eos_token_id   : pooled row = first position whose id == eos_token_id (0 if
 *                  none); eos_token_id < 0 or == 2 selects "first arg-max of the
 *                  ids" (legacy HF configs and OpenAI-clip), modeling_clip.py:561-581 */
int plipmi_encode_text(plipmi_handle h, const int64_t* ids, const int64_t* attention_mask, int B,
                       int eos_token_id, float* out, int normalize, void* stream);

/* Token ids outside [0, vocab_size).  The reference's embedding lookup raises on them (plip.py:68 -> nn.Embedding; on a
 * GPU as a device-side assert that surfaces at the next synchronisation).  plipmi_encode_text only enqueues work, so the
 * ids -- device memory -- are checked BY the embedding kernel: it clamps the lookup (no wild read) and raises a flag in
 * host-visible memory.  The flag is reported, once, as PLIPMI_ERR_TOKEN_ID (= the embeddings of an earlier encode_text
 * are invalid) by this function -- call it after synchronising the stream -- and by the next plipmi_encode_text on the
 * handle, which then enqueues nothing: a caller that feeds a long id matrix in chunks may therefore see the error from a
 * LATER chunk's call, with the earlier chunks' (valid and invalid) rows already written.  The image-side entry points and
 * the heads never report it: a bad caption does not fail an unrelated plipmi_encode_image. */
int plipmi_check_async(plipmi_handle h);

/* Small-batch launch amortisation.  An encode call of at most `max_batch` samples (default min(32, cfg.max_batch), see
 * plipmi_config.graph_batch; 0 = never) replays a captured hipGraph of its ~170 kernel launches instead of issuing
 * them one by one: the first call of a shape (tower, batch, normalise, pooling rule, mask present) runs eagerly, the
 * second captures on the caller's stream, later ones replay.  Inputs / outputs of a replay travel through handle-owned
 * staging buffers (two device-to-device copies per call), results are bit-identical to the eager path.  The reference's
 * zero_shot_classification runs both towers at batch 8 (plip.py:90-91), where the step is launch-bound. */
int plipmi_set_graph_batch(plipmi_handle h, int max_batch);
/* plipmi_config.pass_batch as the handle resolved it (0 = encode calls are never split): a host layer that drives BOTH towers of a
 * batch on two streams (plip_amd.Engine.encode_pair) cuts the batch itself, so that the towers of one pass finish together before
 * the next pass starts -- the rhythm of back-to-back calls of pass_batch samples, which is what the split is meant to reproduce. */
int plipmi_get_pass_batch(plipmi_handle h);
/* Do two HIP streams run their kernels SIDE BY SIDE?  HIP maps streams onto a few hardware queues, and which queue a new stream
 * lands on depends on how many streams the process created before; two streams on one queue execute in order however independent
 * their work is -- a caller that puts the two towers of a pair on such streams gets the one-stream step.  Runs a
 * workgroup that only waits (no memory traffic) for 250 us on each stream at once, three times, and writes
 * *ratio = shortest elapsed time / 250 us: about 1.0-1.2 when the streams overlap, about 2 when they share a queue.  Synchronises
 * both streams; not for use under stream capture.  plip_amd.Engine.encode_pair picks its second stream with it, once. */
int plipmi_streams_overlap(plipmi_handle h, void* stream_a, void* stream_b, float* ratio);

/* Latency path (16-bit engines; OFF by default).  The big GEMM tiles walk K serially whatever M is -- fc2 at batch 8 is 48
 * dependent K tiles for 24 workgroups -- so after plipmi_set_latency_batch(h, n) an encode call of at most n samples runs every
 * GEMM of its tower on the split-K small-M kernel instead (csrc/gemm_skinny.hip): pair latency 1.16 -> 0.79 ms at batch 1,
 * 1.32 -> 1.09 ms at batch 8 (it loses from batch 16 on).  Same arithmetic, another fp32 summation order -- which moves the
 * roundings of the 16-bit activations: embeddings of the two regimes differ by up to 6e-4 (both inside the parity bar).
 * INSIDE a regime a row's embedding does not depend on the batch it arrives in, bit for bit; with the path off (n = 0, the
 * default) that holds for every batch size. */
int plipmi_set_latency_batch(plipmi_handle h, int max_batch);

/* Caption packing (16-bit engines, off by default; PLIPMI_FLAG_PACK_CAPTIONS starts with it on).  CLIPTextTransformer is causal and
 * pools the EOS row only (modeling_clip.py:543-581), so the positions behind a caption's EOS token -- the tokenizer's
 * padding to 77 -- cannot influence text_embeds; the reference computes them anyway.  With packing on, plipmi_encode_text
 * lays the captions' live rows (0 .. EOS) end to end and runs every kernel of the text tower on those rows only: lengths,
 * row offsets and the live-row count are derived from `ids` ON THE DEVICE (no host synchronisation; the path stays
 * graph-capturable), the GEMMs size their grids for the padded worst case and retire dead tiles at once, attention takes
 * per-caption lengths.  text_embeds are BIT-IDENTICAL to the padded computation.  A caption of L tokens then costs L/77 of
 * a padded one.  The default (off) executes every padded position, which is what the bench's headline line measures. */
int plipmi_set_text_packing(plipmi_handle h, int on);

/* in-place row-wise x / sqrt(sum x^2), no epsilon (modeling_clip.py:57-65) */
int plipmi_l2_normalize(plipmi_handle h, float* x, int N, int D, void* stream);

/* logits_per_image[i,j] = scale * <img_i, txt_j>  ([Ni,Nt], required)
 * logits_per_text       = its transpose            ([Nt,Ni], optional, may be NULL)
 * argmax_per_image[i]   = first arg-max_j          (int32 [Ni], optional, may be NULL)
 * img [Ni,D], txt [Nt,D] fp32; pass scale = exp(logit_scale) for CLIPModel.forward,
 * 1.0 for the plain dot product of the zero-shot / retrieval heads. */
int plipmi_logits(plipmi_handle h, const float* img, int Ni, const float* txt, int Nt, int D, float scale,
                  float* logits_per_image, float* logits_per_text, int32_t* argmax_per_image, void* stream);

/* top-k columns of each row of scores [N,M], descending (ties: lower index first);
 * idx int64 [N,k].  Replaces argsort()[:, -k:][:, ::-1] (plip.py:84). */
int plipmi_topk(plipmi_handle h, const float* scores, int N, int M, int k, int64_t* idx, void* stream);

/* Resize (Pillow-exact 8-bit bicubic) + centre crop on the GPU, the step in front of plipmi_encode_image_u8:
 * replaces the host-side `_transform` Resize/CenterCrop (reproducibility/embedders/transform.py:45-48) and the
 * CLIPImageProcessor resize + center_crop behind `self.preprocess(images=...)` (plip.py:32-35) for batches of
 * equally sized uint8 HWC images.  src [B,H,W,3] -> dst [B,n_px,n_px,3], bit-identical to
 * Image.resize((nw,nh), BICUBIC).crop(...).  The caller supplies Pillow's fixed-point tables restricted to the crop
 * window (plip_amd/preprocess.py:resize_crop_plan): x/ybounds int32 [n_px,2] = (first input index, taps),
 * x/ycoef int32 [n_px,ksize]; a NULL pair means that axis already has the right size (crop only, at left/top).
 * tmp [B,nrows,n_px,3] holds source rows row0..row0+nrows after the horizontal pass. */
int plipmi_resize_crop_u8(plipmi_handle h, const uint8_t* src, int B, int H, int W, int n_px, const int32_t* xbounds,
                          const int32_t* xcoef, int xksize, int left, const int32_t* ybounds, const int32_t* ycoef,
                          int yksize, int top, int row0, int nrows, uint8_t* tmp, uint8_t* dst, void* stream);

/* Fused similarity + top-k: for every row q of keys [Nq,D] the k rows j of space [Ns,D] with the largest <q, space_j>,
 * descending (ties: lower j first), WITHOUT materialising the [Nq,Ns] score matrix: scores are produced in
 * [<=4096, <=8192] fp32 panels (MFMA fp32 GEMM) and folded into per-row running lists.  idx int64 [Nq,k] (required),
 * vals fp32 [Nq,k] (optional, may be NULL).  1 <= k <= min(Ns, 1024), D % 32 == 0.  Scratch (<= ~150 MiB) is
 * allocated inside the handle on first use.  Replaces the per-query `t.dot(image_embeddings.T).argsort()[-50:][::-1]`
 * loop of reproducibility/evaluation/retrieval/retrieval.py:13-18 and cosine_sim.argsort()[:, -k:][:, ::-1] of
 * plip.py:78-87 for corpora whose [N,N] matrix would not fit. */
int plipmi_similarity_topk(plipmi_handle h, const float* keys, int Nq, const float* space, int Ns, int D, int k,
                           int64_t* idx, float* vals, void* stream);

/* ---- measurement ------------------------------------------------------------
 * (kernel-level test entries and A/B hooks live in plipmi_test.h: same library, not part of the product interface) */
/* Per-kernel timing with HIP events recorded on the launch stream.  While
 * enabled every kernel launch of the handle is bracketed by two events; the
 * totals are read back (this call synchronises) as rows of `plipmi_kernel_stat`. */
typedef struct plipmi_kernel_stat {
  char   name[96];     /* kernel symbol family, e.g. "gemm_nt<bf16,128x128,bias_qgelu>" */
  int64_t calls;
  double total_ms;     /* sum of event-measured durations */
  double flops;        /* algorithmic FLOPs of those calls (GEMM/attention), 0 for byte-bound kernels */
  double bytes;        /* algorithmic bytes moved by those calls */
} plipmi_kernel_stat;
int plipmi_profile_enable(plipmi_handle h, int on);   /* on=1 starts (and clears), on=0 stops */
int plipmi_profile_read(plipmi_handle h, plipmi_kernel_stat* rows, int max_rows, int* n_rows);

#ifdef __cplusplus
}
#endif
#endif /* PLIPMI_H */
