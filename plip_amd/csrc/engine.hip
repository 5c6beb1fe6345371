// engine.hip -- the C ABI of libplipmi.so (include/plipmi.h): weight packing, the two
// tower drivers, logits / top-k heads and the event-based per-kernel profile.
//
// Data layout in HBM (per handle)
//   weights    one slab: per layer  Wqkv [3D,D] (q rows pre-scaled by 1/8 = 64^-1/2, exact),
//              Wo [D,D], W1 [F,D], W2 [D,F] in the compute dtype, K contiguous (= the HF
//              [out,in] layout, so no transposes); biases / LayerNorm / embeddings fp32;
//              projection matrices transposed to [D,P] fp32 for the pooled head.
//   workspace  per tower, sized for max_batch: residual stream x fp32 [M,D]; h [M,D],
//              qkv [M,3D], attn [M,D], mlp [M,F] in the compute dtype (M = B * tokens).
//              The two towers have separate workspaces so they can run on two streams.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <map>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/plipmi.h"
#include "../../include/plipmi_test.h"
#include "gemm.h"
#include "kernels.h"

#ifndef PLIPMI_DEFAULT_ATTENTION
#define PLIPMI_DEFAULT_ATTENTION 1
#endif
// Batches of at most this many samples take the latency path (split-K small-M GEMMs) -- 0: nobody does unless the caller asks
// (plipmi_set_latency_batch).  It is faster up to batch 8 (pair latency 1.16 -> 0.79 ms at batch 1, 1.32 -> 1.09 ms at 8;
// slower from 16 on: profiles/r04_small_batch_latency.txt), but its other summation order moves the 16-bit activations'
// roundings: embeddings of the two regimes differ by up to 6e-4, and by default a row's embedding is the same bits whatever
// batch it arrives in -- which the host loops' coalescing (plip_amd/plip.py) and the caches' consumers rely on.
static constexpr int kLatencyBatch = 0;

using namespace plipmi;

static thread_local char g_err[512] = "";
// TEST / A-B hooks (plipmi_test.h, process-wide; the product path never writes them).
// g_fuse_qkv_attention (plipmi_test_fused_qkv_attention): 0 = the text tower's q/k/v projection and its attention run as two kernels
// even where the fused kernel (qkv_attention.hip) applies, 1 = the product rule (fused where it applies AND the batch fills the chip),
// 2 = fused wherever it applies, small batches too.
static int g_fuse_qkv_attention = 1;
// g_patch_gather (plipmi_test_patch_gather): 0 = pixels always go through the unfold pass + the plain patch GEMM, 1 (default) = the
// patch GEMM gathers them itself where gemm_gather_supports() says so.
static int g_patch_gather = 1;
// Captured hipGraphs hold the launches of the hook settings they were captured under: every hook change bumps this epoch, and a handle
// whose graphs are older drops them before its next small-batch call.
static unsigned g_hook_epoch = 0;

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return fail(PLIPMI_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

namespace {

struct LayerW {
  void *wqkv, *wo, *w1, *w2;
  float *bqkv, *bo, *b1, *b2, *ln1w, *ln1b, *ln2w, *ln2b;
  // LayerNorm-folded engine: wqkv / w1 hold the centred W * g (LayerNorm gain and mean subtraction folded in),
  // bqkv / b1 hold c2 = W b_ln + bias
};
struct Tower {
  int D = 0, F = 0, L = 0, H = 0, S = 0;
  int dtype = 0;   // operand type of this tower's GEMMs / attention (the engine's, or f16 for the text tower under PLIPMI_FLAG_TEXT_TOWER_F16)
  // plipmi_config.text_f16_layers: the first lead_f16 blocks of a bf16 text tower run on IEEE-half operands (the blocks where
  // bf16's operand rounding costs the embeddings most -- DESIGN.md section 2.1), the rest on the tower's type.  `cur` = the
  // operand type of the block being enqueued (what run_gemm / attention launch with); `planes` = the split format the
  // residual planes {h, lo} currently hold: a block whose type differs re-codes them first (recode_planes).
  int lead_f16 = 0;
  int cur = 0, planes = 0;
  // latency path (set per forward): the batch is small (<= plipmi_engine::latency_batch samples), so every GEMM of the tower
  // takes the split-K small-M kernel instead of walking K serially on a handful of big tiles
  bool small = false;
  int layer_dtype(int l) const { return l < lead_f16 ? PLIPMI_F16 : dtype; }
  std::vector<LayerW> layers;
  // workspace
  float* x = nullptr;
  void *h = nullptr, *qkv = nullptr, *att = nullptr, *mlp = nullptr;
  // LayerNorm-folded engine: the residual stream lives as two planes (common.h split_f32): h = its operand-type plane
  // (the A operand of the q/k/v and fc1 GEMMs), lo = the 8-bit remainder plane (blocked layout); x only holds the
  // embedding rows before the first LayerNorm and a joined copy where something needs plain fp32.  st = the rows'
  // statistics partials [M, D/64, 2]
  float* st = nullptr;
  void* lo = nullptr;
  // packed captions (text tower, plipmi_set_text_packing): rows past a caption's EOS are not computed; everything about the
  // packing lives on the device -- cu [B+1] row offsets, rowmap [B*S] packed row -> (sample << 8 | position), mdev = live rows
  int *cu = nullptr, *rowmap = nullptr, *mdev = nullptr;
  bool packed = false;   // set for the duration of one packed forward
  // last block, pooled rows only (one row per sample): residual row, attention output, bf16 copy, MLP hidden, partials
  float* xp = nullptr; void *attp = nullptr, *hp = nullptr, *mlpp = nullptr; float* stp = nullptr;
};
struct LnArgs {         // the LayerNorm side of a folded GEMM (gemm.h EPI_*_LN / EPI_RESID_EMIT)
  const float* stats = nullptr; int ns = 0; float inv_d = 0.f, eps = 0.f;   // consumer
  void* xb_out = nullptr; float* st_out = nullptr; void* lo_io = nullptr;     // producer (lo_io: EPI_RESID_SPLIT's lo plane)
  int planes_other = 0;                                                         // producer: write the planes in the other 16-bit format
};
// One captured tower forward (hipGraph) per (tower, input kind, batch, normalise, pooling rule, mask?): the ~170 launches
// of a small-batch encode are replayed with ONE host call instead of being issued one by one (launch-bound at the
// reference's own batch size of 8, plip.py:90-91).  Inputs / outputs of a captured forward live in handle-owned staging
// buffers, because the caller's pointers change from call to call and graph nodes hold theirs fixed.
struct GraphEntry {
  int seen = 0;                 // calls so far: the first runs eagerly (it also sets the kernels' attributes)
  hipGraphExec_t exec = nullptr;
};
struct ProfRec {
  const char* name;
  hipEvent_t t0, t1;
  double flops, bytes;
};

}  // namespace

struct DeviceSlab {            // one hipMalloc, freed with its last owner (a handle and the clones that share its weights)
  char* p = nullptr;
  ~DeviceSlab() { if (p) hipFree(p); }
};

struct plipmi_engine {
  plipmi_config cfg;
  int dtype = 0;
  size_t esz = 4;
  // 16-bit engines: the 2 x L LayerNorms of the blocks are folded into the GEMMs around them (no LayerNorm pass, no
  // normalised activations in memory); PLIPMI_FLAG_SEPARATE_LAYERNORM restores the separate LayerNorm kernels for A/B runs
  bool ln_fold = false;
  // The last block's out_proj / fc1 / fc2 (and both of its residual adds) only ever reach the output through the row that
  // is pooled afterwards (CLS / EOS): the encode paths run them on that one row per sample (PLIPMI_FLAG_DENSE_LAST_BLOCK
  // computes all rows, as plipmi_debug_hidden always does).  LayerNorm-folded engines only.
  bool pooled_last = false;
  // Packed captions (plipmi_set_text_packing; PLIPMI_FLAG_PACK_CAPTIONS): the text tower computes rows 0 .. EOS of each caption
  // only -- causal attention and EOS pooling mean the rows behind EOS cannot reach the embedding.  Off by default: the
  // default engine executes every padded position, like the reference does.
  bool text_pack = false;
  // small-batch hipGraph replay (plipmi_config.graph_batch, plipmi_set_graph_batch): batches of at most this many samples
  int graph_batch = 0;
  // latency path: batches of at most this many samples run their GEMMs on the split-K small-M kernel (16-bit engines).  The
  // reference drives zero_shot_classification / retrieval at batch 8 (plip.py:90-91,112).  Results of the two regimes differ
  // by fp32 summation order only; inside a regime a row's embedding does not depend on the batch it arrives in.
  int latency_batch = 0;
  // plipmi_config.pass_batch resolved: an encode call of B >= 2 * pass_batch samples runs as equal back-to-back passes of at most
  // pass_batch (0 = never split).  A pass's per-block activations (qkv, att, mlp, the residual planes) then stay in the 256 MiB Infinity
  // Cache whatever the caller's batch: at bs = 512 / 1024 in ONE pass the ViT-B/32 step ran 8 % / 3 % under the bs = 256 rate
  // (BENCH_r05 batch_scaling).  Same bits: a row's embedding does not depend on the batch it travels in.
  int pass_batch = 0;
  std::map<std::tuple<int, int, int, int, int>, GraphEntry> graphs;
  unsigned graphs_epoch = 0;  // g_hook_epoch the captured graphs belong to
  void* g_vin = nullptr;      // staged image input (fp32 pixels or uint8 tiles) [graph_batch_cap, 3, H, W] x 4 B
  int64_t* g_tin = nullptr;   // staged ids   [graph_batch_cap, ctx]
  int64_t* g_tmask = nullptr; // staged attention mask
  float *g_vout = nullptr, *g_tout = nullptr;   // staged embeddings [graph_batch_cap, P]
  int graph_batch_cap = 0;
  hipStream_t cap_stream = nullptr;   // captures run here: the legacy default stream cannot capture, and the caller's
                                      // stream never enters capture mode (other threads may be enqueueing on it)
  int np = 0, kpad = 0;
  Tower vis, txt;
  void* patch_w = nullptr;  // [Dv, kpad]
  void* patches = nullptr;  // [max_batch*np, kpad]
  float *cls = nullptr, *vpos = nullptr, *pre_w = nullptr, *pre_b = nullptr, *post_w = nullptr, *post_b = nullptr,
        *vproj_t = nullptr;
  float *tok = nullptr, *tpos = nullptr, *fin_w = nullptr, *fin_b = nullptr, *tproj_t = nullptr;
  float *vproj = nullptr, *tproj = nullptr;        // [P, D] fp32, the HF layout (NT GEMM operand)
  float *vpooled = nullptr, *tpooled = nullptr;    // [max_batch, D] fp32 LayerNorm'd pooled rows
  char* slab = nullptr;       // this handle's own allocation: [weights | workspace] (plipmi_create) or [workspace] (plipmi_clone)
  size_t slab_bytes = 0;
  std::shared_ptr<DeviceSlab> own, weights;   // `weights` keeps the allocation the weight pointers point into alive (a clone: its source's)
  int attn_impl = 0, attn_impl_vis = 0, attn_impl_txt = 0;
  // raised by the embedding kernels when a token id lies outside the vocabulary (host-visible memory; reported by the next
  // call on the handle and by plipmi_check_async -- the reference's lookup raises, plip.py:68)
  int* bad_id = nullptr;
  bool half() const { return dtype != PLIPMI_F32; }
  char devname[128] = "";
  // plipmi_similarity_topk scratch (allocated on first use, grown on demand)
  char* sim_ws = nullptr;
  size_t sim_ws_bytes = 0;
  // profiling
  bool prof = false;
  std::vector<ProfRec> recs;
  std::vector<hipEvent_t> pool;
};

namespace {

struct Scope {  // brackets one kernel launch with events while profiling is on
  plipmi_engine* e;
  hipStream_t s;
  bool on;
  ProfRec r;
  Scope(plipmi_engine* e_, hipStream_t s_, const char* name, double flops, double bytes) : e(e_), s(s_), on(e_->prof) {
    if (!on) return;
    r.name = name; r.flops = flops; r.bytes = bytes;
    r.t0 = take(); r.t1 = take();
    hipEventRecord(r.t0, s);
  }
  void rename(const char* name) { r.name = name; }
  ~Scope() {
    if (!on) return;
    hipEventRecord(r.t1, s);
    e->recs.push_back(r);
  }
  hipEvent_t take() {
    if (!e->pool.empty()) { hipEvent_t ev = e->pool.back(); e->pool.pop_back(); return ev; }
    hipEvent_t ev; hipEventCreate(&ev); return ev;
  }
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Carver {  // two-pass slab carving: pass 1 sizes, pass 2 hands out pointers
  char* base = nullptr;
  size_t off = 0;
  template <typename T> T* take(size_t count, size_t elem) {
    off = align_up(off, 256);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * elem;
    return p;
  }
};

// The slab of a handle made by plipmi_create is [packed weights | workspace]; a handle made by plipmi_clone shares its source's weights
// (carve_weights is not run for it: it keeps the copied pointers) and carves a workspace of its own.
void carve_weights(plipmi_engine* e, Carver& c) {
  const plipmi_config& g = e->cfg;
  const size_t es = e->esz;
  e->patch_w = c.take<void>((size_t)g.v_width * e->kpad, es);
  e->cls = c.take<float>(g.v_width, 4);
  e->vpos = c.take<float>((size_t)(e->np + 1) * g.v_width, 4);
  e->pre_w = c.take<float>(g.v_width, 4);  e->pre_b = c.take<float>(g.v_width, 4);
  e->post_w = c.take<float>(g.v_width, 4); e->post_b = c.take<float>(g.v_width, 4);
  e->vproj_t = c.take<float>((size_t)g.v_width * g.projection_dim, 4);
  e->tok = c.take<float>((size_t)g.vocab_size * g.t_width, 4);
  e->tpos = c.take<float>((size_t)g.context_length * g.t_width, 4);
  e->fin_w = c.take<float>(g.t_width, 4);  e->fin_b = c.take<float>(g.t_width, 4);
  e->tproj_t = c.take<float>((size_t)g.t_width * g.projection_dim, 4);
  e->vproj = c.take<float>((size_t)g.v_width * g.projection_dim, 4);
  e->tproj = c.take<float>((size_t)g.t_width * g.projection_dim, 4);
  for (Tower* t : {&e->vis, &e->txt}) {
    const size_t D = t->D, F = t->F;
    t->layers.resize(t->L);
    for (LayerW& w : t->layers) {
      w.wo = c.take<void>(D * D, es); w.w2 = c.take<void>(D * F, es);
      w.wqkv = c.take<void>(3 * D * D, es); w.w1 = c.take<void>(F * D, es);
      w.bqkv = c.take<float>(3 * D, 4); w.bo = c.take<float>(D, 4); w.b1 = c.take<float>(F, 4); w.b2 = c.take<float>(D, 4);
      w.ln1w = c.take<float>(D, 4); w.ln1b = c.take<float>(D, 4); w.ln2w = c.take<float>(D, 4); w.ln2b = c.take<float>(D, 4);
    }
  }
}

void carve_workspace(plipmi_engine* e, Carver& c) {
  const plipmi_config& g = e->cfg;
  const size_t es = e->esz;
  const size_t B = (size_t)g.max_batch;
  e->vpooled = c.take<float>(B * g.v_width, 4);
  e->tpooled = c.take<float>(B * g.t_width, 4);
  for (Tower* t : {&e->vis, &e->txt}) {
    const size_t D = t->D, F = t->F;
    const size_t M = B * t->S;
    t->x = c.take<float>(M * D, 4);
    t->h = c.take<void>(M * D, es);
    if (e->ln_fold) { t->st = c.take<float>(M * (D / kLnSlice) * 2, 4); t->lo = c.take<void>(lo_plane_bytes(M, D), 1); }
    if (e->ln_fold && t == &e->txt) {
      t->cu = c.take<int>(B + 1, 4); t->rowmap = c.take<int>(M, 4); t->mdev = c.take<int>(1, 4);
    }
    if (e->pooled_last) {
      t->xp = c.take<float>(B * D, 4); t->attp = c.take<void>(B * D, es); t->hp = c.take<void>(B * D, es);
      t->mlpp = c.take<void>(B * F, es); t->stp = c.take<float>(B * (D / kLnSlice) * 2, 4);
    }
    t->qkv = c.take<void>(M * 3 * D, es);
    t->att = c.take<void>(M * D, es);
    t->mlp = c.take<void>(M * F, es);
  }
  e->patches = c.take<void>(B * e->np * e->kpad, es);
  const size_t gb = (size_t)e->graph_batch_cap;
  if (gb) {
    e->g_vin = c.take<void>(gb * 3 * g.image_size * g.image_size, 4);
    e->g_tin = c.take<int64_t>(gb * g.context_length, 8);
    e->g_tmask = c.take<int64_t>(gb * g.context_length, 8);
    e->g_vout = c.take<float>(gb * g.projection_dim, 4);
    e->g_tout = c.take<float>(gb * g.projection_dim, 4);
  }
}

int pack_tower(plipmi_engine* e, Tower& t, const plipmi_layer_weights* src, hipStream_t s) {
  const int D = t.D, F = t.F;
  const float qscale = 0.125f;  // head_dim 64 -> 64^-0.5, a power of two: folding it into Wq/bq is exact
  for (int l = 0; l < t.L; ++l) {
    const int dt = t.layer_dtype(l);
    const plipmi_layer_weights& w = src[l];
    LayerW& d = t.layers[l];
    if (e->ln_fold) {
      // W' = (W * g, rows centred) in the operand type (q rows also x 1/8), c2 = W b_ln + bias -> the bias slot
      char* wq = reinterpret_cast<char*>(d.wqkv);
      HIP_TRY(launch_fold_ln(w.q_w, w.q_b, w.ln1_w, w.ln1_b, wq, d.bqkv, D, D, qscale, dt, s));
      HIP_TRY(launch_fold_ln(w.k_w, w.k_b, w.ln1_w, w.ln1_b, wq + (size_t)D * D * e->esz, d.bqkv + D, D, D, 1.f, dt, s));
      HIP_TRY(launch_fold_ln(w.v_w, w.v_b, w.ln1_w, w.ln1_b, wq + (size_t)2 * D * D * e->esz, d.bqkv + 2 * D, D, D, 1.f, dt, s));
      HIP_TRY(launch_fold_ln(w.fc1_w, w.fc1_b, w.ln2_w, w.ln2_b, d.w1, d.b1, F, D, 1.f, dt, s));
    } else {
      char* wq = reinterpret_cast<char*>(d.wqkv);
      HIP_TRY(launch_convert(w.q_w, wq, dt, D, D, D, qscale, s));
      HIP_TRY(launch_convert(w.k_w, wq + (size_t)D * D * e->esz, dt, D, D, D, 1.f, s));
      HIP_TRY(launch_convert(w.v_w, wq + (size_t)2 * D * D * e->esz, dt, D, D, D, 1.f, s));
      HIP_TRY(launch_convert(w.fc1_w, d.w1, dt, F, D, D, 1.f, s));
    }
    if (!e->ln_fold) {
      HIP_TRY(launch_scale_copy(w.q_b, d.bqkv, D, qscale, s));
      HIP_TRY(launch_scale_copy(w.k_b, d.bqkv + D, D, 1.f, s));
      HIP_TRY(launch_scale_copy(w.v_b, d.bqkv + 2 * D, D, 1.f, s));
      HIP_TRY(launch_scale_copy(w.fc1_b, d.b1, F, 1.f, s));
    }
    HIP_TRY(launch_convert(w.o_w, d.wo, dt, D, D, D, 1.f, s));
    HIP_TRY(launch_convert(w.fc2_w, d.w2, dt, D, F, F, 1.f, s));
    HIP_TRY(launch_scale_copy(w.o_b, d.bo, D, 1.f, s));
    HIP_TRY(launch_scale_copy(w.fc2_b, d.b2, D, 1.f, s));
    HIP_TRY(launch_scale_copy(w.ln1_w, d.ln1w, D, 1.f, s));
    HIP_TRY(launch_scale_copy(w.ln1_b, d.ln1b, D, 1.f, s));
    HIP_TRY(launch_scale_copy(w.ln2_w, d.ln2w, D, 1.f, s));
    HIP_TRY(launch_scale_copy(w.ln2_b, d.ln2b, D, 1.f, s));
  }
  return PLIPMI_OK;
}

// "kernel name|role": the profile keeps the launches of one kernel symbol apart by what they compute (out-proj and fc2
// share a symbol but not a roofline: one is HBM-bound, the other MFMA-bound); bench.py merges them back per symbol.
const char* name_with_role(const char* name, const char* role) {
  static thread_local std::map<std::pair<const char*, const char*>, std::string> cache;
  std::string& v = cache[std::make_pair(name, role)];
  if (v.empty()) v = std::string(name) + "|" + role;
  return v.c_str();
}

int run_gemm(plipmi_engine* e, const Tower& t, int epi, const void* A, const void* W, void* C, const float* bias, int M, int N, int K,
             int ldc, int np, hipStream_t s, const char* role, const LnArgs* ln = nullptr, const int* m_dev = nullptr) {
  GemmParams p;
  p.A = A; p.W = W; p.C = C; p.bias = bias; p.m_dev = m_dev;
  p.M = M; p.N = N; p.K = K; p.lda = K; p.ldw = K; p.ldc = ldc; p.alpha = 1.f; p.np = np;
  if (ln) {
    p.ln_stats = ln->stats; p.ln_ns = ln->ns; p.ln_inv_d = ln->inv_d; p.ln_eps = ln->eps;
    p.xb_out = ln->xb_out; p.st_out = ln->st_out; p.lo_io = ln->lo_io; p.planes_other = ln->planes_other;
  }
  bool skinny = role[0] == '~';     // '~role': pooled-row GEMM of the last block -> the small-M split-K kernel when it fits
  if (skinny) ++role;
  skinny = skinny || (t.small && !m_dev);   // latency path: the whole tower of a small batch (packed rows keep the big kernels)
  const char* name = "gemm_nt";
  // algorithmic bytes: operands once, output once (bf16 outputs 2 B, residual read + written -- as one fp32 array or as the 16 + 8-bit
  // 16-bit planes --, + bf16 copy when EPI_RESID_EMIT writes one)
  const double out_bytes = epi_is_colwise(epi) ? (double)M * N * e->esz
                           : (double)M * N * (epi == EPI_RESID_SPLIT ? 6.0 : epi_is_resid(epi) ? 8.0 : 4.0) + (epi == EPI_RESID_EMIT ? (double)M * N * 2.0 : 0.0);
  Scope sc(e, s, name, 2.0 * M * N * (double)K, ((double)M * K + (double)N * K) * e->esz + out_bytes);
  const int rc = (skinny && e->half() && gemm_skinny_supports(epi, M, N, K))
                     ? gemm_launch_skinny(t.cur, epi, p, s, &name)
                     : gemm_launch(t.cur, epi, -1, p, s, &name);
  if (e->prof) sc.rename(name_with_role(name, role));
  if (rc != 0) return fail(PLIPMI_ERR_HIP, "gemm launch (%s, M=%d N=%d K=%d) failed: %s", name, M, N, K,
                           hipGetErrorString((hipError_t)rc));
  return PLIPMI_OK;
}

#define RUN(expr) do { int rc_ = (expr); if (rc_ != PLIPMI_OK) return rc_; } while (0)

// Block l is about to be enqueued: launch with its operand type and, on a LayerNorm-folded engine, make the residual planes
// speak it (hi IS the block's A operand).  On the big-tile path the predecessor's fc2 epilogue already wrote the planes in this block's
// format (GemmParams.planes_other); otherwise (the small-M path) they are re-coded in place: joined in the old code, split in the new
// one -- one more rounding of the 8-bit remainder (common.h split_f32), the new hi = the value correctly rounded to the new operand type.
int enter_block(plipmi_engine* e, Tower& t, int l, int M, hipStream_t s) {
  t.cur = t.layer_dtype(l);
  if (e->ln_fold && t.planes != t.cur) {
    Scope sc(e, s, "recode_planes", 0, (double)M * t.D * 6);
    HIP_TRY(launch_recode_planes(t.h, t.lo, (size_t)M, t.D, t.planes, t.cur, s));
    t.planes = t.cur;
  }
  return PLIPMI_OK;
}

// LayerNorm-folded q/k/v projection + attention of one block: ONE kernel where the sequence fits the fused tile (77-token
// captions: qkv_attention.hip, the `qkv` activation never reaches memory), else the GEMM and the attention kernel.
// Either way t.att holds the attention output afterwards, the same bits.
int run_qkv_attention(plipmi_engine* e, Tower& t, const LayerW& w, int B, int causal, const int64_t* key_mask, hipStream_t s,
                      const LnArgs& use) {
  const int M = B * t.S, D = t.D;
  const int impl = (&t == &e->vis) ? e->attn_impl_vis : e->attn_impl_txt;
  const int* cu = t.packed ? t.cu : nullptr;
  const int* md = t.packed ? t.mdev : nullptr;
  const double att_flops = 4.0 * B * t.H * (double)t.S * t.S * 64;
  if (g_fuse_qkv_attention && impl == 1 && !t.packed && !t.small && qkv_attention_supports(t.cur, B, t.S, t.H, D) &&
      (g_fuse_qkv_attention == 2 || qkv_attention_pays(B, t.H, gemm_num_cus()))) {
    Scope sc(e, s, "qkv_attention", 2.0 * M * 3.0 * D * (double)D + att_flops, ((double)M * D * 2 + 3.0 * D * D) * e->esz);
    HIP_TRY(launch_qkv_attention(t.cur, t.h, w.wqkv, w.bqkv, use.stats, use.inv_d, use.eps, t.att, B, t.S, t.H, causal, key_mask, s));
    return PLIPMI_OK;
  }
  RUN(run_gemm(e, t, EPI_BIAS_LN, t.h, w.wqkv, t.qkv, w.bqkv, M, 3 * D, D, 3 * D, 0, s, "qkv", &use, md));
  Scope sc(e, s, impl ? "attention_mfma" : "attention_valu", att_flops, (double)M * 4 * D * e->esz);
  HIP_TRY(launch_attention(t.qkv, t.att, t.cur, B, t.S, t.H, causal, key_mask, impl, s, cu));
  return PLIPMI_OK;
}

// n_layers pre-LN residual blocks over the tower's residual stream x (CLIPEncoderLayer, modeling_clip.py:362-383)
int run_layers(plipmi_engine* e, Tower& t, int B, int n_layers, int causal, const int64_t* key_mask, hipStream_t s,
               bool more_follow = false) {
  const int M = B * t.S, D = t.D, F = t.F;
  const float eps = e->cfg.layer_norm_eps;
  const int impl = (&t == &e->vis) ? e->attn_impl_vis : e->attn_impl_txt;
  const int* cu = t.packed ? t.cu : nullptr;          // packed captions: row offsets / live-row count on the device
  const int* md = t.packed ? t.mdev : nullptr;
  auto attention = [&]() -> int {
    Scope sc(e, s, impl ? "attention_mfma" : "attention_valu", 4.0 * B * t.H * (double)t.S * t.S * 64, (double)M * 4 * D * e->esz);
    HIP_TRY(launch_attention(t.qkv, t.att, t.cur, B, t.S, t.H, causal, key_mask, impl, s, cu));
    return PLIPMI_OK;
  };
  if (e->ln_fold) {
    // LayerNorm never runs as a pass: the residual stream x = {t.h, t.lo} (operand-type plane + 8-bit remainder plane) and
    // t.st = the rows' statistics partials come from x's producer (embedding kernel, or the residual GEMM's epilogue); the
    // consuming GEMMs read the bf16 plane as their A operand, carry LayerNorm's gain, centring and bias in their weights and
    // apply rstd in their epilogues.  HF order (modeling_clip.py:370-381) is unchanged:
    // x += out_proj(attn(LN1(x))); x += fc2(quick_gelu(fc1(LN2(x)))).
    LnArgs use;  use.stats = t.st; use.ns = D / kLnSlice; use.inv_d = 1.0f / (float)D; use.eps = eps;
    LnArgs emit; emit.xb_out = t.h; emit.st_out = t.st; emit.lo_io = t.lo;
    for (int l = 0; l < n_layers; ++l) {
      const LayerW& w = t.layers[l];
      RUN(enter_block(e, t, l, M, s));
      RUN(run_qkv_attention(e, t, w, B, causal, key_mask, s, use));
      RUN(run_gemm(e, t, EPI_RESID_SPLIT, t.att, w.wo, nullptr, w.bo, M, D, D, D, 0, s, "out_proj", &emit, md));
      RUN(run_gemm(e, t, EPI_QGELU_LN, t.h, w.w1, t.mlp, w.b1, M, F, D, F, 0, s, "fc1", &use, md));
      // a block whose successor runs on the other 16-bit operand type (the last f16 block of a mixed text tower) writes its
      // planes in the successor's format from fc2's epilogue -- no re-coding pass over the stream (the tiled kernels only:
      // the small-M kernel of the latency path keeps the separate pass, enter_block)
      const int next_dt = l + 1 < t.L ? t.layer_dtype(l + 1) : t.cur;
      LnArgs emit2 = emit;
      emit2.planes_other = (next_dt != t.cur && !(t.small && !md)) ? 1 : 0;
      RUN(run_gemm(e, t, EPI_RESID_SPLIT, t.mlp, w.w2, nullptr, w.b2, M, D, F, D, 0, s, "fc2", &emit2, md));
      if (emit2.planes_other) t.planes = next_dt;
    }
    if (!more_follow) {
      if (t.packed) return fail(PLIPMI_ERR_INVALID, "packed rows have no every-token form");   // a consumer of plain fp32 rows follows (the every-token head, plipmi_debug_hidden)
      Scope sc(e, s, "join_planes", 0, (double)M * D * 7);
      HIP_TRY(launch_join_planes(t.h, t.lo, t.x, (size_t)M, D, t.planes, s));
    }
    return PLIPMI_OK;
  }
  for (int l = 0; l < n_layers; ++l) {
    const LayerW& w = t.layers[l];
    RUN(enter_block(e, t, l, M, s));
    { Scope sc(e, s, "layernorm", 0, (double)M * D * (4 + e->esz));
      HIP_TRY(launch_layernorm(t.x, D, w.ln1w, w.ln1b, t.h, t.cur, M, D, eps, s)); }
    RUN(run_gemm(e, t, EPI_BIAS, t.h, w.wqkv, t.qkv, w.bqkv, M, 3 * D, D, 3 * D, 0, s, "qkv"));
    RUN(attention());
    RUN(run_gemm(e, t, EPI_BIAS_RESID, t.att, w.wo, t.x, w.bo, M, D, D, D, 0, s, "out_proj"));
    { Scope sc(e, s, "layernorm", 0, (double)M * D * (4 + e->esz));
      HIP_TRY(launch_layernorm(t.x, D, w.ln2w, w.ln2b, t.h, t.cur, M, D, eps, s)); }
    RUN(run_gemm(e, t, EPI_BIAS_QGELU, t.h, w.w1, t.mlp, w.b1, M, F, D, F, 0, s, "fc1"));
    RUN(run_gemm(e, t, EPI_BIAS_RESID, t.mlp, w.w2, t.x, w.b2, M, D, F, D, 0, s, "fc2"));
  }
  return PLIPMI_OK;
}

// The last block on the pooled rows only.  CLIPModel.get_image_features / get_text_features (modeling_clip.py:683-753) hand
// back the projection of ONE row per sample -- CLS after post_layernorm (:650), the EOS row after final_layer_norm
// (:559-581) -- so of the last block's work only q/k/v + attention need every token (keys and values); its out_proj, both
// residual adds, LayerNorm 2, fc1 and fc2 are row-wise and reach the output through that one row.  The reference computes
// them for all 50 / 77 tokens because CLIPModel also returns last_hidden_state, which this path does not.  Results are
// those of the full computation on the pooled rows (same arithmetic, row by row); plipmi_debug_hidden runs the full block.
int run_last_block_pooled(plipmi_engine* e, Tower& t, int B, int causal, const int64_t* key_mask, const int64_t* ids,
                          int eos_id, hipStream_t s) {
  const int M = B * t.S, D = t.D, F = t.F;
  const LayerW& w = t.layers[t.L - 1];
  LnArgs use; use.stats = t.st; use.ns = D / kLnSlice; use.inv_d = 1.0f / (float)D; use.eps = e->cfg.layer_norm_eps;
  const int* cu = t.packed ? t.cu : nullptr;
  RUN(enter_block(e, t, t.L - 1, M, s));
  RUN(run_qkv_attention(e, t, w, B, causal, key_mask, s, use));
  { Scope sc(e, s, "pool_gather", 0, (double)B * D * (2 * e->esz + 8));
    HIP_TRY(launch_pool_gather(t.att, t.h, t.lo, t.S, D, ids, eos_id, t.attp, t.xp, B, t.cur, s, cu)); }
  LnArgs emit; emit.xb_out = t.hp; emit.st_out = t.stp;
  RUN(run_gemm(e, t, EPI_RESID_EMIT, t.attp, w.wo, t.xp, w.bo, B, D, D, D, 0, s, "~out_proj_pooled", &emit));
  use.stats = t.stp;
  RUN(run_gemm(e, t, EPI_QGELU_LN, t.hp, w.w1, t.mlpp, w.b1, B, F, D, F, 0, s, "~fc1_pooled", &use));
  RUN(run_gemm(e, t, EPI_BIAS_RESID, t.mlpp, w.w2, t.xp, w.b2, B, D, F, D, 0, s, "~fc2_pooled"));
  return PLIPMI_OK;
}

// CLIPVisionEmbeddings + pre_layrnorm (modeling_clip.py:202-218,642): x = LN(cat(cls, conv(pixels)) + pos)
int vision_embed(plipmi_engine* e, const float* pixels, const uint8_t* tiles_u8, int B, hipStream_t s) {
  const plipmi_config& g = e->cfg;
  Tower& t = e->vis;
  t.cur = t.planes = t.layer_dtype(0);
  // fp32 pixels, 16-bit engine, 16- / 32-pixel patches: the patch GEMM reads the pixels itself (im2col on load -- four pixels per lane
  // into registers, rounded to the operand type, written to its A stage), no unfold pass and no `patches` round trip.  Same operand
  // bits as the unfold kernel's, hence the same embedding rows.
  // uint8 tiles (round 6): the same gather on the HWC bytes, CLIP normalisation as one fma per pixel -- the rows the unfold_u8 pass +
  // plain patch GEMM produce, bit for bit.
  const bool gather = g_patch_gather && e->half() && !t.small && e->kpad == 3 * g.patch_size * g.patch_size &&
                      gemm_gather_supports(t.dtype, B, g.image_size, g.patch_size, t.D);
  if (gather) {
    { Scope sc(e, s, "cls_rows", 0, (double)B * t.D * 4);
      HIP_TRY(launch_cls_rows(e->cls, e->vpos, t.x, B, t.S, t.D, s)); }
    GemmParams p;
    p.A = nullptr; p.W = e->patch_w; p.C = t.x; p.bias = e->vpos;
    p.M = B * e->np; p.N = t.D; p.K = e->kpad; p.lda = e->kpad; p.ldw = e->kpad; p.ldc = t.D; p.alpha = 1.f; p.np = e->np;
    p.pix = pixels; p.tiles = tiles_u8; p.img_hw = g.image_size; p.patch_log2 = g.patch_size == 32 ? 5 : 4;
    const char* name = "gemm_nt";
    Scope sc(e, s, name, 2.0 * p.M * p.N * (double)p.K, (double)B * 3 * g.image_size * g.image_size * (tiles_u8 ? 1 : 4) + (double)p.N * p.K * e->esz + (double)p.M * p.N * 4);
    const int rc = gemm_launch_gather(t.dtype, p, s, &name);
    if (e->prof) sc.rename(name_with_role(name, "patch_embed"));
    if (rc != 0) return fail(PLIPMI_ERR_HIP, "patch GEMM (im2col on load) failed: %s", hipGetErrorString((hipError_t)rc));
  } else if (tiles_u8) {
    Scope sc(e, s, "unfold_patches_u8", 0, (double)B * 3 * g.image_size * g.image_size + (double)B * e->np * e->kpad * e->esz);
    HIP_TRY(launch_unfold_patches_u8(tiles_u8, e->patches, t.dtype, B, g.image_size, g.patch_size, e->kpad, s));
  } else {
    Scope sc(e, s, "unfold_patches", 0, (double)B * 3 * g.image_size * g.image_size * 4 + (double)B * e->np * e->kpad * e->esz);
    HIP_TRY(launch_unfold_patches(pixels, e->patches, t.dtype, B, g.image_size, g.patch_size, e->kpad, s)); }
  if (!gather) {
  { Scope sc(e, s, "cls_rows", 0, (double)B * t.D * 4);
    HIP_TRY(launch_cls_rows(e->cls, e->vpos, t.x, B, t.S, t.D, s)); }
  RUN(run_gemm(e, t, EPI_PATCH, e->patches, e->patch_w, t.x, e->vpos, B * e->np, t.D, e->kpad, t.D, e->np, s, "patch_embed"));
  }
  if (e->ln_fold) {   // the tower's one LayerNorm pass: fp32 embedding rows in, the split residual stream + row statistics out
    Scope sc(e, s, "layernorm", 0, (double)B * t.S * t.D * 8.2);
    HIP_TRY(launch_layernorm_emit(t.x, e->pre_w, e->pre_b, t.h, t.lo, t.st, B * t.S, t.D, g.layer_norm_eps, t.dtype, s));
    return PLIPMI_OK;
  }
  { Scope sc(e, s, "layernorm", 0, (double)B * t.S * t.D * 8);
    HIP_TRY(launch_layernorm(t.x, t.D, e->pre_w, e->pre_b, t.x, 0, B * t.S, t.D, g.layer_norm_eps, s)); }
  return PLIPMI_OK;
}

int text_embed(plipmi_engine* e, const int64_t* ids, int B, hipStream_t s, int eos_id = -1) {
  Tower& t = e->txt;
  t.cur = t.planes = t.layer_dtype(0);     // the embedding kernel emits the planes in the first block's format
  if (t.packed) {
    { Scope sc(e, s, "text_pack", 0, (double)B * t.S * 12);
      HIP_TRY(launch_text_pack(ids, B, t.S, eos_id, t.cu, t.rowmap, t.mdev, s)); }
    Scope sc(e, s, "text_embed", 0, (double)B * t.S * t.D * 8.2);
    HIP_TRY(launch_text_embed_emit_packed(ids, e->tok, e->tpos, t.h, t.lo, t.st, t.rowmap, t.mdev, B * t.S, t.S, t.D,
                                          e->cfg.vocab_size, e->bad_id, t.planes, s));
    return PLIPMI_OK;
  }
  Scope sc(e, s, "text_embed", 0, (double)B * t.S * t.D * (e->ln_fold ? 8.2 : 8.0));
  if (e->ln_fold) HIP_TRY(launch_text_embed_emit(ids, e->tok, e->tpos, t.h, t.lo, t.st, B, t.S, t.D, e->cfg.vocab_size, e->bad_id, t.planes, s));
  else HIP_TRY(launch_text_embed(ids, e->tok, e->tpos, t.x, B, t.S, t.D, e->cfg.vocab_size, e->bad_id, s));
  return PLIPMI_OK;
}

// pooled row -> LayerNorm -> bias-free projection (-> L2 normalise).  Widths that are multiples of 32 (every
// config plipmi_create accepts today) run the projection on the split-K exact-fp32 MFMA head kernel; the fused
// one-block-per-sample kernel covers anything else.
int run_head(plipmi_engine* e, Tower& t, const float* x, int S, const int64_t* ids, int eos_id, const float* ln_w,
             const float* ln_b, const float* W, const float* Wt, float* pooled, float* out, int B, int normalize, hipStream_t s) {
  const int P = e->cfg.projection_dim, D = t.D;
  if (P % 32 == 0 && D % 32 == 0) {
    { Scope sc(e, s, "pool_layernorm", 0, (double)B * D * 8);
      HIP_TRY(launch_pool_layernorm(x, S, D, ids, eos_id, ln_w, ln_b, e->cfg.layer_norm_eps, pooled, B, s)); }
    { Scope sc(e, s, "head_gemm", 2.0 * B * P * (double)D, ((double)B * D + (double)P * D + (double)B * P) * 4);
      HIP_TRY(launch_head_gemm(pooled, W, out, B, P, D, s)); }
    if (normalize) { Scope sc(e, s, "l2_normalize", 0, (double)B * P * 8); HIP_TRY(launch_l2_normalize(out, B, P, s)); }
    return PLIPMI_OK;
  }
  Scope sc(e, s, "pool_head", 2.0 * B * D * P, (double)D * P * 4);
  HIP_TRY(launch_pool_head(x, S, D, ids, eos_id, ln_w, ln_b, e->cfg.layer_norm_eps, Wt, P, out, B, normalize, s));
  return PLIPMI_OK;
}

// a token id outside the vocabulary seen by an EARLIER encode_text (the flag is written by the device, so it is known only
// once that work has run): report once, then clear
int check_async(plipmi_engine* e) {
  if (e->bad_id && *reinterpret_cast<volatile int*>(e->bad_id) != 0) {
    *reinterpret_cast<volatile int*>(e->bad_id) = 0;
    return fail(PLIPMI_ERR_TOKEN_ID, "an earlier plipmi_encode_text on this handle was given a token id outside [0, %d) "
                "(the reference's embedding lookup raises there, plip.py:68); its embeddings are invalid", e->cfg.vocab_size);
  }
  return PLIPMI_OK;
}
// (the sticky token-id flag is reported by plipmi_encode_text and plipmi_check_async only: a bad caption must not fail an
//  unrelated encode_image, and whether it did used to depend on whether the embedding kernel had already run)
int check_batch(plipmi_engine* e, int B) {
  if (!e) return fail(PLIPMI_ERR_INVALID, "null handle");
  if (B < 0 || B > e->cfg.max_batch)
    return fail(PLIPMI_ERR_INVALID, "batch %d outside [0, max_batch=%d]", B, e->cfg.max_batch);
  return PLIPMI_OK;
}

// the three tower forwards, on whatever pointers they are given (caller's, or the staging buffers under capture)
int image_forward(plipmi_handle h, const float* pixels, const uint8_t* tiles, int B, float* out, int normalize,
                         hipStream_t s) {
  h->vis.small = h->half() && B <= h->latency_batch;
  RUN(vision_embed(h, pixels, tiles, B, s));
  if (h->pooled_last) {
    RUN(run_layers(h, h->vis, B, h->vis.L - 1, 0, nullptr, s, /*more_follow=*/true));
    RUN(run_last_block_pooled(h, h->vis, B, 0, nullptr, nullptr, -1, s));
    return run_head(h, h->vis, h->vis.xp, 1, nullptr, -1, h->post_w, h->post_b, h->vproj, h->vproj_t, h->vpooled, out, B, normalize, s);
  }
  RUN(run_layers(h, h->vis, B, h->vis.L, 0, nullptr, s));
  return run_head(h, h->vis, h->vis.x, h->vis.S, nullptr, -1, h->post_w, h->post_b, h->vproj, h->vproj_t, h->vpooled, out, B, normalize, s);
}
int text_forward(plipmi_handle h, const int64_t* ids, const int64_t* mask, int B, int eos_token_id, float* out,
                        int normalize, hipStream_t s) {
  // Packed captions (opt-in): the tower is causal and only the EOS row is pooled, so rows past EOS cannot reach the output;
  // they are left out of every kernel.  Needs the pooled last block (no every-token consumer) and the bf16 MFMA attention.
  struct Unpack { Tower& t; ~Unpack() { t.packed = false; } } unpack{h->txt};
  h->txt.packed = h->text_pack && h->ln_fold && h->pooled_last && h->attn_impl_txt == 1 && h->txt.S <= 128;
  h->txt.small = h->half() && B <= h->latency_batch;
  RUN(text_embed(h, ids, B, s, eos_token_id));
  if (h->pooled_last) {
    RUN(run_layers(h, h->txt, B, h->txt.L - 1, 1, mask, s, /*more_follow=*/true));
    RUN(run_last_block_pooled(h, h->txt, B, 1, mask, ids, eos_token_id, s));
    return run_head(h, h->txt, h->txt.xp, 1, nullptr, -1, h->fin_w, h->fin_b, h->tproj, h->tproj_t, h->tpooled, out, B, normalize, s);
  }
  RUN(run_layers(h, h->txt, B, h->txt.L, 1, mask, s));
  return run_head(h, h->txt, h->txt.x, h->txt.S, ids, eos_token_id, h->fin_w, h->fin_b, h->tproj, h->tproj_t, h->tpooled, out, B, normalize, s);
}

// Small batches: replay a captured graph of the same launches.  kind 0 = fp32 pixels, 1 = uint8 tiles, 2 = text.
// Call 1 of a shape runs eagerly (and leaves every kernel's attributes set), call 2 captures, later calls replay.
template <typename Fwd>
int graph_or_eager(plipmi_handle h, int kind, int B, int normalize, int eos, int has_mask, hipStream_t s,
                          const void* in, size_t in_bytes, void* in_stage, const int64_t* mask, size_t mask_bytes,
                          float* out, float* out_stage, Fwd&& forward /* (in, mask, out, stream) -> rc */) {
  const bool eligible = h->graph_batch > 0 && B <= h->graph_batch && !h->prof;
  if (!eligible) return forward(in, mask, out, s);
  if (h->graphs_epoch != g_hook_epoch) {   // a test hook changed what the launches are since these graphs were captured
    for (auto& kv : h->graphs) if (kv.second.exec) hipGraphExecDestroy(kv.second.exec);
    h->graphs.clear();
    h->graphs_epoch = g_hook_epoch;
  }
  GraphEntry& ge = h->graphs[std::make_tuple(kind, B, normalize, eos, has_mask)];
  if (ge.seen++ == 0) return forward(in, mask, out, s);
  HIP_TRY(hipMemcpyAsync(in_stage, in, in_bytes, hipMemcpyDeviceToDevice, s));
  if (has_mask) HIP_TRY(hipMemcpyAsync(h->g_tmask, mask, mask_bytes, hipMemcpyDeviceToDevice, s));
  if (!ge.exec) {
    hipGraph_t graph = nullptr;
    if (!h->cap_stream) HIP_TRY(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal));
    const int rc = forward(in_stage, has_mask ? h->g_tmask : nullptr, out_stage, h->cap_stream);
    const hipError_t ee = hipStreamEndCapture(h->cap_stream, &graph);   // always end the capture: the stream must leave capture mode
    if (rc != PLIPMI_OK) { if (graph) hipGraphDestroy(graph); return rc; }
    if (ee != hipSuccess) return fail(PLIPMI_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ee));
    const hipError_t ie = hipGraphInstantiate(&ge.exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (ie != hipSuccess) { ge.exec = nullptr; return fail(PLIPMI_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ie)); }
  }
  HIP_TRY(hipGraphLaunch(ge.exec, s));
  HIP_TRY(hipMemcpyAsync(out, out_stage, (size_t)B * h->cfg.projection_dim * 4, hipMemcpyDeviceToDevice, s));
  return PLIPMI_OK;
}

// plipmi_config.pass_batch: how many equal passes an encode call of B samples runs as (1 = the whole batch at once), and the
// rows of pass i -- the first B % n passes take one sample more
int passes_of(const plipmi_engine* h, int B) {
  return (h->pass_batch > 0 && B >= 2 * h->pass_batch) ? (B + h->pass_batch - 1) / h->pass_batch : 1;
}
int pass_rows(int B, int n, int i) { return B / n + (i < B % n ? 1 : 0); }

}  // namespace

extern "C" {

int plipmi_version(void) { return PLIPMI_VERSION; }
const char* plipmi_last_error(void) { return g_err; }
const char* plipmi_device_name(plipmi_handle h) { return h ? h->devname : ""; }

int plipmi_create(const plipmi_config* cfg, const plipmi_weights* w, void* stream, plipmi_handle* out) {
  if (!cfg || !w || !out) return fail(PLIPMI_ERR_INVALID, "null argument");
  *out = nullptr;
  // The caller's struct may be OLDER (shorter) than this library's: read what it has, later members are 0 = their defaults
  // (a member appended at the tail used to be read as garbage from a caller compiled against the previous header).
  const size_t kMinSize = offsetof(plipmi_config, max_batch) + sizeof(int32_t);   // the members every version has had
  if (cfg->struct_size < (int)kMinSize || cfg->struct_size > (int)sizeof(plipmi_config))
    return fail(PLIPMI_ERR_INVALID, "plipmi_config.struct_size = %d: expected sizeof(plipmi_config) of the caller's header, %zu .. %zu "
                "(this library: version %d)", cfg->struct_size, kMinSize, sizeof(plipmi_config), PLIPMI_VERSION);
  plipmi_config g_copy;
  memset(&g_copy, 0, sizeof(g_copy));
  memcpy(&g_copy, cfg, (size_t)cfg->struct_size);
  g_copy.struct_size = (int32_t)sizeof(plipmi_config);
  const plipmi_config& g = g_copy;
  if (g.compute_dtype != PLIPMI_F32 && g.compute_dtype != PLIPMI_BF16 && g.compute_dtype != PLIPMI_F16)
    return fail(PLIPMI_ERR_INVALID, "compute_dtype must be PLIPMI_F32, PLIPMI_BF16 or PLIPMI_F16");
  if (g.flags & ~(PLIPMI_FLAG_SEPARATE_LAYERNORM | PLIPMI_FLAG_DENSE_LAST_BLOCK | PLIPMI_FLAG_PACK_CAPTIONS | PLIPMI_FLAG_VALU_ATTENTION |
                  PLIPMI_FLAG_TEXT_TOWER_F16))
    return fail(PLIPMI_ERR_INVALID, "unknown bits in plipmi_config.flags (0x%x)", (unsigned)g.flags);
  if ((g.flags & PLIPMI_FLAG_TEXT_TOWER_F16) && g.compute_dtype != PLIPMI_BF16)
    return fail(PLIPMI_ERR_INVALID, "PLIPMI_FLAG_TEXT_TOWER_F16 is a mode of the bf16 engine (compute_dtype PLIPMI_BF16)");
  if (g.text_f16_layers < 0 || g.text_f16_layers > g.t_layers || (g.text_f16_layers > 0 && g.compute_dtype != PLIPMI_BF16))
    return fail(PLIPMI_ERR_INVALID, "text_f16_layers = %d: 0 .. t_layers leading text blocks, bf16 engine only", g.text_f16_layers);
  if (g.v_heads <= 0 || g.t_heads <= 0 || g.v_width != g.v_heads * 64 || g.t_width != g.t_heads * 64)
    return fail(PLIPMI_ERR_INVALID, "head_dim must be 64 (v_width=%d/%d heads, t_width=%d/%d heads)", g.v_width,
                g.v_heads, g.t_width, g.t_heads);
  if (g.patch_size <= 0 || g.image_size % g.patch_size) return fail(PLIPMI_ERR_INVALID, "image_size %% patch_size != 0");
  if (g.v_width % 128 || g.v_mlp % 128 || g.t_width % 128 || g.t_mlp % 128)
    return fail(PLIPMI_ERR_INVALID, "widths and MLP sizes must be multiples of 128 (GEMM tile)");
  if (g.v_width > 2048 || g.t_width > 2048 || g.projection_dim > 1024 || g.projection_dim <= 0)
    return fail(PLIPMI_ERR_INVALID, "width > 2048 or projection_dim > 1024 not supported");
  if (g.max_batch <= 0 || g.v_layers <= 0 || g.t_layers <= 0 || g.context_length <= 0 || g.vocab_size <= 0)
    return fail(PLIPMI_ERR_INVALID, "non-positive size in config");
  const int tokens = (g.image_size / g.patch_size) * (g.image_size / g.patch_size) + 1;
  if (tokens > 1024 || g.context_length > 1024) return fail(PLIPMI_ERR_INVALID, "more than 1024 tokens per sequence");

  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess)
    return fail(PLIPMI_ERR_NODEVICE, "no HIP device visible (libplipmi needs an MI355X / gfx950 GPU)");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(PLIPMI_ERR_NODEVICE, "device %d is %s; libplipmi is built for gfx950 only", dev, prop.gcnArchName);

  plipmi_engine* e = new plipmi_engine();
  e->cfg = g;
  e->dtype = g.compute_dtype;
  e->esz = e->half() ? 2 : 4;
  e->vis.dtype = e->dtype;
  e->txt.dtype = (g.flags & PLIPMI_FLAG_TEXT_TOWER_F16) ? PLIPMI_F16 : e->dtype;
  e->txt.lead_f16 = e->txt.dtype == PLIPMI_BF16 ? g.text_f16_layers : 0;
  e->vis.cur = e->vis.planes = e->vis.dtype;
  e->txt.cur = e->txt.planes = e->txt.layer_dtype(0);
  e->ln_fold = e->half() && !(g.flags & PLIPMI_FLAG_SEPARATE_LAYERNORM);
  e->pooled_last = e->ln_fold && !(g.flags & PLIPMI_FLAG_DENSE_LAST_BLOCK);
  e->text_pack = e->pooled_last && (g.flags & PLIPMI_FLAG_PACK_CAPTIONS);
  e->latency_batch = e->half() ? kLatencyBatch : 0;
  {
    // per-sample bytes of one block's activations in the larger tower: q/k/v (3D) + attention output (D) + the two residual planes
    // (2D) + the MLP hidden (F), 16-bit each (the fp32 engine: twice that and no planes -- the same rule errs on the safe side)
    auto per_sample = [&](int S, int D, int F) { return (double)S * (3.0 * D + D + 2.0 * D + F) * (double)(g.compute_dtype == PLIPMI_F32 ? 4 : 2); };
    const double ps = std::max(per_sample(tokens, g.v_width, g.v_mlp), per_sample(g.context_length, g.t_width, g.t_mlp));
    const int fit = (int)(208e6 / ps) / 32 * 32;       // 208 MB: the cache minus a tower's block weights and the other tower's share
    // passes under 256 samples cost more in the GEMMs (tile quantisation, the 128x128 tile below 12 800 rows) than the cache returns:
    // the fp32 ViT-B/32 engine (fit = 128) ran bs = 256 as two passes at 10.1 k img/s instead of one at 12.5 k
    e->pass_batch = g.pass_batch > 0 ? g.pass_batch : (g.pass_batch == 0 && fit >= 256) ? fit : 0;
  }
  e->graph_batch_cap = std::min(g.max_batch, 32);
  e->graph_batch = g.graph_batch < 0 ? 0 : g.graph_batch == 0 ? e->graph_batch_cap : std::min(g.graph_batch, e->graph_batch_cap);
  e->np = tokens - 1;
  e->kpad = (int)align_up((size_t)3 * g.patch_size * g.patch_size, 64);
  snprintf(e->devname, sizeof(e->devname), "%s:%s", prop.gcnArchName, prop.name);
  e->vis.D = g.v_width; e->vis.F = g.v_mlp; e->vis.L = g.v_layers; e->vis.H = g.v_heads; e->vis.S = tokens;
  e->txt.D = g.t_width; e->txt.F = g.t_mlp; e->txt.L = g.t_layers; e->txt.H = g.t_heads; e->txt.S = g.context_length;
  // attention kernel: exact-fp32 VALU kernel for the fp32 engine, MFMA kernels for the 16-bit engines
  // (PLIPMI_FLAG_VALU_ATTENTION forces the VALU kernel for A/B runs)
  e->attn_impl = (g.flags & PLIPMI_FLAG_VALU_ATTENTION) ? 0 : PLIPMI_DEFAULT_ATTENTION;
  if (!e->half()) e->attn_impl = 0;
  e->attn_impl_txt = e->attn_impl ? 1 : 0;  // S <= 128: single-pass MFMA kernel, longer: chunked online softmax
  e->attn_impl_vis = e->attn_impl ? 1 : 0;

  Carver sizing;
  carve_weights(e, sizing);
  carve_workspace(e, sizing);
  e->slab_bytes = align_up(sizing.off, 256);
  hipError_t me = hipMalloc(reinterpret_cast<void**>(&e->slab), e->slab_bytes);
  if (me != hipSuccess) {
    const size_t need = e->slab_bytes;
    delete e;
    return fail(PLIPMI_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", need, hipGetErrorString(me));
  }
  e->own = std::make_shared<DeviceSlab>();
  e->own->p = e->slab;
  e->weights = e->own;
  Carver placing;
  placing.base = e->slab;
  carve_weights(e, placing);
  carve_workspace(e, placing);
  if (hipHostMalloc(reinterpret_cast<void**>(&e->bad_id), sizeof(int), hipHostMallocMapped) != hipSuccess) e->bad_id = nullptr;
  else *e->bad_id = 0;

  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  int rc = PLIPMI_OK;
  auto body = [&]() -> int {
    const int Dv = g.v_width, Dt = g.t_width, P = g.projection_dim;
    HIP_TRY(launch_convert(w->v_patch_weight, e->patch_w, e->vis.dtype, Dv, 3 * g.patch_size * g.patch_size, e->kpad, 1.f, s));
    HIP_TRY(launch_scale_copy(w->v_class_embedding, e->cls, Dv, 1.f, s));
    HIP_TRY(launch_scale_copy(w->v_pos_embedding, e->vpos, tokens * Dv, 1.f, s));
    HIP_TRY(launch_scale_copy(w->v_pre_ln_w, e->pre_w, Dv, 1.f, s));
    HIP_TRY(launch_scale_copy(w->v_pre_ln_b, e->pre_b, Dv, 1.f, s));
    HIP_TRY(launch_scale_copy(w->v_post_ln_w, e->post_w, Dv, 1.f, s));
    HIP_TRY(launch_scale_copy(w->v_post_ln_b, e->post_b, Dv, 1.f, s));
    HIP_TRY(launch_transpose(w->visual_projection, e->vproj_t, P, Dv, s));
    HIP_TRY(launch_scale_copy(w->visual_projection, e->vproj, P * Dv, 1.f, s));
    HIP_TRY(launch_scale_copy(w->text_projection, e->tproj, P * Dt, 1.f, s));
    HIP_TRY(hipMemcpyAsync(e->tok, w->t_token_embedding, (size_t)g.vocab_size * Dt * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(launch_scale_copy(w->t_pos_embedding, e->tpos, g.context_length * Dt, 1.f, s));
    HIP_TRY(launch_scale_copy(w->t_final_ln_w, e->fin_w, Dt, 1.f, s));
    HIP_TRY(launch_scale_copy(w->t_final_ln_b, e->fin_b, Dt, 1.f, s));
    HIP_TRY(launch_transpose(w->text_projection, e->tproj_t, P, Dt, s));
    RUN(pack_tower(e, e->vis, w->v_layers, s));
    RUN(pack_tower(e, e->txt, w->t_layers, s));
    return PLIPMI_OK;
  };
  rc = body();
  if (rc != PLIPMI_OK) {
    if (e->bad_id) hipHostFree(e->bad_id);
    delete e;                  // frees the slab with its last owner
    return rc;
  }
  *out = e;
  return PLIPMI_OK;
}

int plipmi_clone(plipmi_handle src, plipmi_handle* out) {
  if (!src || !out) return fail(PLIPMI_ERR_INVALID, "null argument");
  *out = nullptr;
  plipmi_engine* e = new plipmi_engine(*src);      // configuration, setters' state and every weight pointer
  e->graphs.clear();                               // per-handle state starts empty
  e->cap_stream = nullptr;
  e->sim_ws = nullptr; e->sim_ws_bytes = 0;
  e->prof = false; e->recs.clear(); e->pool.clear();
  e->bad_id = nullptr;
  e->own.reset();
  e->slab = nullptr;
  e->vis.small = e->txt.small = false; e->vis.packed = e->txt.packed = false;
  e->vis.cur = e->vis.planes = e->vis.dtype;
  e->txt.cur = e->txt.planes = e->txt.layer_dtype(0);
  Carver sizing;
  carve_workspace(e, sizing);
  e->slab_bytes = align_up(sizing.off, 256);
  hipError_t me = hipMalloc(reinterpret_cast<void**>(&e->slab), e->slab_bytes);
  if (me != hipSuccess) {
    const size_t need = e->slab_bytes;
    delete e;
    return fail(PLIPMI_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", need, hipGetErrorString(me));
  }
  e->own = std::make_shared<DeviceSlab>();
  e->own->p = e->slab;
  Carver placing;
  placing.base = e->slab;
  carve_workspace(e, placing);
  if (hipHostMalloc(reinterpret_cast<void**>(&e->bad_id), sizeof(int), hipHostMallocMapped) != hipSuccess) e->bad_id = nullptr;
  else *e->bad_id = 0;
  *out = e;
  return PLIPMI_OK;
}

void plipmi_destroy(plipmi_handle h) {
  if (!h) return;
  for (ProfRec& r : h->recs) { hipEventDestroy(r.t0); hipEventDestroy(r.t1); }
  for (hipEvent_t ev : h->pool) hipEventDestroy(ev);
  for (auto& kv : h->graphs) if (kv.second.exec) hipGraphExecDestroy(kv.second.exec);
  if (h->cap_stream) hipStreamDestroy(h->cap_stream);
  if (h->sim_ws) hipFree(h->sim_ws);
  if (h->bad_id) hipHostFree(h->bad_id);
  delete h;
}

int plipmi_encode_image(plipmi_handle h, const float* pixels, int B, float* out, int normalize, void* stream) {
  RUN(check_batch(h, B));
  if (B == 0) return PLIPMI_OK;
  if (!pixels || !out) return fail(PLIPMI_ERR_INVALID, "null pixels/out");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t n = (size_t)B * 3 * h->cfg.image_size * h->cfg.image_size;
  if (const int np_ = passes_of(h, B); np_ > 1) {
    const size_t per = (size_t)3 * h->cfg.image_size * h->cfg.image_size;
    for (int b0 = 0, i = 0; i < np_; ++i) {
      const int nb = pass_rows(B, np_, i);
      RUN(image_forward(h, pixels + (size_t)b0 * per, nullptr, nb, out + (size_t)b0 * h->cfg.projection_dim, normalize, s));
      b0 += nb;
    }
    return PLIPMI_OK;
  }
  return graph_or_eager(h, 0, B, normalize != 0, 0, 0, s, pixels, n * 4, h->g_vin, nullptr, 0, out, h->g_vout,
                        [&](const void* in, const int64_t*, float* o, hipStream_t st) {
                          return image_forward(h, reinterpret_cast<const float*>(in), nullptr, B, o, normalize, st); });
}

int plipmi_encode_image_u8(plipmi_handle h, const uint8_t* tiles, int B, float* out, int normalize, void* stream) {
  RUN(check_batch(h, B));
  if (B == 0) return PLIPMI_OK;
  if (!tiles || !out) return fail(PLIPMI_ERR_INVALID, "null tiles/out");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t n = (size_t)B * 3 * h->cfg.image_size * h->cfg.image_size;
  if (const int np_ = passes_of(h, B); np_ > 1) {
    const size_t per = (size_t)3 * h->cfg.image_size * h->cfg.image_size;
    for (int b0 = 0, i = 0; i < np_; ++i) {
      const int nb = pass_rows(B, np_, i);
      RUN(image_forward(h, nullptr, tiles + (size_t)b0 * per, nb, out + (size_t)b0 * h->cfg.projection_dim, normalize, s));
      b0 += nb;
    }
    return PLIPMI_OK;
  }
  return graph_or_eager(h, 1, B, normalize != 0, 0, 0, s, tiles, n, h->g_vin, nullptr, 0, out, h->g_vout,
                        [&](const void* in, const int64_t*, float* o, hipStream_t st) {
                          return image_forward(h, nullptr, reinterpret_cast<const uint8_t*>(in), B, o, normalize, st); });
}

int plipmi_encode_text(plipmi_handle h, const int64_t* ids, const int64_t* attention_mask, int B, int eos_token_id,
                       float* out, int normalize, void* stream) {
  RUN(check_batch(h, B));
  RUN(check_async(h));
  if (B == 0) return PLIPMI_OK;
  if (!ids || !out) return fail(PLIPMI_ERR_INVALID, "null ids/out");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const size_t n = (size_t)B * h->cfg.context_length * 8;
  if (const int np_ = passes_of(h, B); np_ > 1) {
    const size_t per = (size_t)h->cfg.context_length;
    for (int b0 = 0, i = 0; i < np_; ++i) {
      const int nb = pass_rows(B, np_, i);
      RUN(text_forward(h, ids + (size_t)b0 * per, attention_mask ? attention_mask + (size_t)b0 * per : nullptr, nb, eos_token_id,
                       out + (size_t)b0 * h->cfg.projection_dim, normalize, s));
      b0 += nb;
    }
    return PLIPMI_OK;
  }
  return graph_or_eager(h, 2, B, normalize != 0, eos_token_id, attention_mask != nullptr, s, ids, n, h->g_tin,
                        attention_mask, n, out, h->g_tout,
                        [&](const void* in, const int64_t* m, float* o, hipStream_t st) {
                          return text_forward(h, reinterpret_cast<const int64_t*>(in), m, B, eos_token_id, o, normalize, st); });
}

int plipmi_set_graph_batch(plipmi_handle h, int max_batch) {
  if (!h) return fail(PLIPMI_ERR_INVALID, "null handle");
  h->graph_batch = std::max(0, std::min(max_batch, h->graph_batch_cap));
  return PLIPMI_OK;
}

int plipmi_get_pass_batch(plipmi_handle h) { return h ? h->pass_batch : 0; }

int plipmi_streams_overlap(plipmi_handle h, void* stream_a, void* stream_b, float* ratio) {
  if (!h || !ratio) return fail(PLIPMI_ERR_INVALID, "bad argument");
  hipStream_t sa = reinterpret_cast<hipStream_t>(stream_a), sb = reinterpret_cast<hipStream_t>(stream_b);
  int dev = 0, khz = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) khz = 100000;
  const double us = 250.0;
  const unsigned long long ticks = (unsigned long long)(us * 1e-3 * khz);
  HIP_TRY(launch_occupy(ticks / 8, sa));            // code object loaded, queues awake
  HIP_TRY(launch_occupy(ticks / 8, sb));
  double best = 1e30;
  for (int rep = 0; rep < 3; ++rep) {               // the shortest of three: a host hiccup only ever lengthens a window
    HIP_TRY(hipStreamSynchronize(sa));
    HIP_TRY(hipStreamSynchronize(sb));
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    HIP_TRY(launch_occupy(ticks, sa));
    HIP_TRY(launch_occupy(ticks, sb));
    HIP_TRY(hipStreamSynchronize(sa));
    HIP_TRY(hipStreamSynchronize(sb));
    clock_gettime(CLOCK_MONOTONIC, &t1);
    best = std::min(best, (t1.tv_sec - t0.tv_sec) * 1e6 + (t1.tv_nsec - t0.tv_nsec) * 1e-3);
  }
  *ratio = (float)(best / us);
  return PLIPMI_OK;
}

int plipmi_set_latency_batch(plipmi_handle h, int max_batch) {
  if (!h) return fail(PLIPMI_ERR_INVALID, "null handle");
  const int v = h->half() ? std::max(0, max_batch) : 0;
  if (v != h->latency_batch) {       // captured forwards hold the other regime's launches
    for (auto& kv : h->graphs) if (kv.second.exec) hipGraphExecDestroy(kv.second.exec);
    h->graphs.clear();
  }
  h->latency_batch = v;
  return PLIPMI_OK;
}

int plipmi_set_text_packing(plipmi_handle h, int on) {
  if (!h) return fail(PLIPMI_ERR_INVALID, "null handle");
  if (on && !h->pooled_last)
    return fail(PLIPMI_ERR_INVALID, "caption packing needs a 16-bit engine's pooled last block (compute_dtype bf16 / f16, LayerNorm folding on, last block not dense)");
  if ((on != 0) != h->text_pack) {   // captured text forwards hold the other form's launches
    for (auto& kv : h->graphs) if (kv.second.exec) hipGraphExecDestroy(kv.second.exec);
    h->graphs.clear();
  }
  h->text_pack = on != 0;
  return PLIPMI_OK;
}

int plipmi_debug_hidden(plipmi_handle h, int tower, int layer, const void* input, int B, float* out, void* stream) {
  RUN(check_batch(h, B));
  if (B == 0) return PLIPMI_OK;
  if (!input || !out) return fail(PLIPMI_ERR_INVALID, "null input/out");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Tower& t = tower == PLIPMI_VISION ? h->vis : h->txt;
  if (tower != PLIPMI_VISION && tower != PLIPMI_TEXT) return fail(PLIPMI_ERR_INVALID, "tower must be 0 or 1");
  if (layer < 0 || layer > t.L) return fail(PLIPMI_ERR_INVALID, "layer %d outside [0,%d]", layer, t.L);
  t.small = h->half() && B <= h->latency_batch;
  if (tower == PLIPMI_VISION) RUN(vision_embed(h, reinterpret_cast<const float*>(input), nullptr, B, s));
  else RUN(text_embed(h, reinterpret_cast<const int64_t*>(input), B, s));
  RUN(run_layers(h, t, B, layer, tower == PLIPMI_TEXT, nullptr, s));
  HIP_TRY(hipMemcpyAsync(out, t.x, (size_t)B * t.S * t.D * 4, hipMemcpyDeviceToDevice, s));
  return PLIPMI_OK;
}

int plipmi_l2_normalize(plipmi_handle h, float* x, int N, int D, void* stream) {
  if (!h || !x || N < 0 || D <= 0) return fail(PLIPMI_ERR_INVALID, "bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Scope sc(h, s, "l2_normalize", 0, (double)N * D * 8);
  HIP_TRY(launch_l2_normalize(x, N, D, s));
  return PLIPMI_OK;
}

int plipmi_logits(plipmi_handle h, const float* img, int Ni, const float* txt, int Nt, int D, float scale,
                  float* logits_per_image, float* logits_per_text, int32_t* argmax_per_image, void* stream) {
  if (!h || !img || !txt || !logits_per_image || Ni < 0 || Nt < 0 || D <= 0) return fail(PLIPMI_ERR_INVALID, "bad argument");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (Ni == 0 || Nt == 0) return PLIPMI_OK;
  // Exact-fp32 MFMA whenever the shape tiles (the bs=256 logits of CLIPModel.forward do): one 32x32 tile per workgroup,
  // K split over its four waves.  logits_per_text is the same kernel with the operands exchanged -- products commute
  // and the k order is identical, so it is bit-for-bit the transpose.  Other shapes (e.g. 10 class prompts) take the
  // scalar-FMA kernel.
  if (D % 32 == 0 && Nt % 32 == 0 && (!logits_per_text || Ni % 32 == 0) && (size_t)Ni * Nt <= (1u << 22)) {
    { Scope sc(h, s, "logits_mfma", 2.0 * Ni * (double)Nt * D, ((double)Ni + Nt) * D * 4 + (double)Ni * Nt * 4);
      HIP_TRY(launch_head_gemm(img, txt, logits_per_image, Ni, Nt, D, s, scale)); }
    if (logits_per_text) {
      Scope sc(h, s, "logits_mfma", 2.0 * Ni * (double)Nt * D, ((double)Ni + Nt) * D * 4 + (double)Ni * Nt * 4);
      HIP_TRY(launch_head_gemm(txt, img, logits_per_text, Nt, Ni, D, s, scale)); }
    if (argmax_per_image) { Scope sc(h, s, "row_argmax", 0, (double)Ni * Nt * 4); HIP_TRY(launch_row_argmax(logits_per_image, Ni, Nt, argmax_per_image, s)); }
    return PLIPMI_OK;
  }
  Scope sc(h, s, "logits", 2.0 * Ni * (double)Nt * D, ((double)Ni + Nt) * D * 4 + (double)Ni * Nt * 4);
  HIP_TRY(launch_logits(img, Ni, txt, Nt, D, scale, logits_per_image, logits_per_text, argmax_per_image, s));
  return PLIPMI_OK;
}

int plipmi_topk(plipmi_handle h, const float* scores, int N, int M, int k, int64_t* idx, void* stream) {
  if (!h || !scores || !idx || N < 0 || M <= 0 || k <= 0 || k > M) return fail(PLIPMI_ERR_INVALID, "bad argument (need 0 < k <= M)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Scope sc(h, s, "topk", 0, (double)N * M * 4 * k);
  HIP_TRY(launch_topk(scores, N, M, k, idx, s));
  return PLIPMI_OK;
}

int plipmi_resize_crop_u8(plipmi_handle h, const uint8_t* src, int B, int H, int W, int n_px, const int32_t* xbounds,
                          const int32_t* xcoef, int xksize, int left, const int32_t* ybounds, const int32_t* ycoef,
                          int yksize, int top, int row0, int nrows, uint8_t* tmp, uint8_t* dst, void* stream) {
  if (!h || B < 0 || H <= 0 || W <= 0 || n_px <= 0) return fail(PLIPMI_ERR_INVALID, "bad argument");
  if (B == 0) return PLIPMI_OK;
  if (!src || !tmp || !dst) return fail(PLIPMI_ERR_INVALID, "null src/tmp/dst");
  if ((xbounds == nullptr) != (xcoef == nullptr) || (ybounds == nullptr) != (ycoef == nullptr))
    return fail(PLIPMI_ERR_INVALID, "bounds and coefficients come in pairs");
  if (row0 < 0 || nrows <= 0 || row0 + nrows > H) return fail(PLIPMI_ERR_INVALID, "rows [%d, %d) outside the %d-row image", row0, row0 + nrows, H);
  if (!xbounds && (left < 0 || left + n_px > W)) return fail(PLIPMI_ERR_INVALID, "crop columns outside the image");
  if (!ybounds && (top < row0 || top + n_px > row0 + nrows)) return fail(PLIPMI_ERR_INVALID, "crop rows outside the staged rows");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Scope sc(h, s, "resize_crop_u8", 0, (double)B * ((double)nrows * W * 3 + 2.0 * nrows * n_px * 3 + (double)n_px * n_px * 3));
  HIP_TRY(launch_resize_crop_u8(src, B, H, W, n_px, xbounds, xcoef, xksize, left, ybounds, ycoef, yksize, top, row0, nrows,
                                tmp, dst, s));
  return PLIPMI_OK;
}

int plipmi_similarity_topk(plipmi_handle h, const float* keys, int Nq, const float* space, int Ns, int D, int k,
                           int64_t* idx, float* vals, void* stream) {
  if (!h || Nq < 0 || Ns <= 0 || D <= 0) return fail(PLIPMI_ERR_INVALID, "bad argument");
  if (Nq > 0 && (!keys || !space || !idx)) return fail(PLIPMI_ERR_INVALID, "null keys/space/idx");
  if (k <= 0 || k > Ns || k > kTopkMaxK)
    return fail(PLIPMI_ERR_INVALID, "need 0 < k <= min(Ns, %d), got k=%d Ns=%d", kTopkMaxK, k, Ns);
  if (D % 32) return fail(PLIPMI_ERR_INVALID, "embedding width %d must be a multiple of 32", D);
  if (Nq == 0) return PLIPMI_OK;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // panel geometry: [QB queries] x [PB space vectors] of fp32 scores at a time (<= 128 MiB), never [Nq, Ns]
  const int PB = (int)std::min<size_t>(8192, align_up((size_t)Ns, 256));
  const int QB = std::min(Nq, 4096);
  const int tail = Ns % PB;  // the last panel is staged zero-padded so the GEMM's N stays a whole number of tiles
  const size_t sc_bytes = align_up((size_t)QB * PB * 4, 256);
  const size_t tl_bytes = tail ? align_up((size_t)align_up((size_t)tail, 256) * D * 4, 256) : 0;
  const size_t vl_bytes = vals ? 0 : align_up((size_t)QB * k * 4, 256);
  const size_t need = sc_bytes + tl_bytes + vl_bytes;
  if (need > h->sim_ws_bytes) {
    if (h->sim_ws) { HIP_TRY(hipStreamSynchronize(s)); HIP_TRY(hipFree(h->sim_ws)); h->sim_ws = nullptr; h->sim_ws_bytes = 0; }
    hipError_t me = hipMalloc(reinterpret_cast<void**>(&h->sim_ws), need);
    if (me != hipSuccess) { h->sim_ws = nullptr; return fail(PLIPMI_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", need, hipGetErrorString(me)); }
    h->sim_ws_bytes = need;
  }
  float* scores = reinterpret_cast<float*>(h->sim_ws);
  float* tail_w = reinterpret_cast<float*>(h->sim_ws + sc_bytes);
  float* own_vals = reinterpret_cast<float*>(h->sim_ws + sc_bytes + tl_bytes);
  const int tail_n = (int)align_up((size_t)tail, 256);
  if (tail) {
    HIP_TRY(hipMemsetAsync(tail_w, 0, (size_t)tail_n * D * 4, s));
    HIP_TRY(hipMemcpyAsync(tail_w, space + (size_t)(Ns - tail) * D, (size_t)tail * D * 4, hipMemcpyDeviceToDevice, s));
  }
  for (int q0 = 0; q0 < Nq; q0 += QB) {
    const int rows = std::min(QB, Nq - q0);
    float* v = vals ? vals + (size_t)q0 * k : own_vals;
    int64_t* ix = idx + (size_t)q0 * k;
    HIP_TRY(launch_topk_init(v, ix, (size_t)rows * k, s));
    for (int p0 = 0; p0 < Ns; p0 += PB) {
      const int cols = std::min(PB, Ns - p0);
      const bool is_tail = cols < PB;
      GemmParams p;
      p.A = keys + (size_t)q0 * D; p.W = is_tail ? tail_w : space + (size_t)p0 * D; p.C = scores; p.bias = nullptr;
      p.M = rows; p.N = is_tail ? tail_n : PB; p.K = D; p.lda = D; p.ldw = D; p.ldc = PB; p.alpha = 1.f; p.np = 1;
      const char* name = "gemm_nt";
      { Scope sc(h, s, name, 2.0 * rows * (double)p.N * D, ((double)rows * D + (double)p.N * D + (double)rows * p.N) * 4);
        const int rc = gemm_launch(PLIPMI_F32, EPI_SCALE, p.N % 256 == 0 && rows > 128 ? -1 : 1, p, s, &name);
        sc.rename(name);
        if (rc != 0) return fail(PLIPMI_ERR_HIP, "similarity gemm failed: %s", hipGetErrorString((hipError_t)rc)); }
      { Scope sc(h, s, "topk_merge", 0, (double)rows * cols * 4);
        HIP_TRY(launch_topk_merge(scores, (size_t)PB, rows, cols, (int64_t)p0, k, v, ix, s)); }
    }
    HIP_TRY(launch_topk_finish(ix, (size_t)rows * k, s));
  }
  return PLIPMI_OK;
}

int plipmi_gemm_nt(int dtype, int epilogue, int variant, int M, int N, int K, const void* A, const void* W,
                   const float* bias, float alpha, void* C, void* stream) {
  return plipmi_gemm_nt_traced(dtype, epilogue, variant, M, N, K, A, W, bias, alpha, C, nullptr, stream);
}

int plipmi_gemm_nt_traced(int dtype, int epilogue, int variant, int M, int N, int K, const void* A, const void* W,
                          const float* bias, float alpha, void* C, uint64_t* trace, void* stream) {
  if (dtype != PLIPMI_F32 && dtype != PLIPMI_BF16 && dtype != PLIPMI_F16) return fail(PLIPMI_ERR_INVALID, "bad dtype");
  if (epilogue < 0 || epilogue > EPI_SCALE) return fail(PLIPMI_ERR_INVALID, "epilogue must be 0..3");
  if (M < 0 || N <= 0 || K <= 0 || !A || !W || !C) return fail(PLIPMI_ERR_INVALID, "bad shape / null pointer");
  if (epilogue != EPI_SCALE && !bias) return fail(PLIPMI_ERR_INVALID, "bias required for this epilogue");
  GemmParams p;
  p.A = A; p.W = W; p.C = C; p.bias = bias; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldw = K; p.ldc = N;
  p.alpha = alpha; p.np = 1;
  p.trace = reinterpret_cast<unsigned long long*>(trace);
  const int rc = (variant == -3 && dtype != PLIPMI_F32)     // -3: the small-M split-K kernel (gemm_skinny.hip)
                     ? gemm_launch_skinny(dtype, epilogue, p, reinterpret_cast<hipStream_t>(stream), nullptr)
                     : gemm_launch(dtype, epilogue, variant, p, reinterpret_cast<hipStream_t>(stream), nullptr);
  if (rc != 0) return fail(PLIPMI_ERR_HIP, "gemm launch failed (variant %d, M=%d N=%d K=%d): %s", variant, M, N, K,
                           hipGetErrorString((hipError_t)rc));
  return PLIPMI_OK;
}

int plipmi_gemm_nt_ln(int dtype, int mode, int variant, int M, int N, int K, const void* A, const void* W, const float* bias,
                      const float* stats, int ns, float eps, void* C, void* xb_out, float* st_out, void* stream) {
  if (dtype != PLIPMI_BF16 && dtype != PLIPMI_F16) return fail(PLIPMI_ERR_INVALID, "LayerNorm-folded epilogues are 16-bit-engine forms");
  if (mode < 0 || mode > 4 || M < 0 || N <= 0 || K <= 0 || !A || !W || !C || !bias) return fail(PLIPMI_ERR_INVALID, "bad argument");
  if (mode < 2 && (!stats || ns <= 0)) return fail(PLIPMI_ERR_INVALID, "mode 0/1 need the row statistics");
  if (mode >= 2 && (!xb_out || !st_out || N % kLnSlice)) return fail(PLIPMI_ERR_INVALID, "mode 2/3 need xb_out, st_out and N % 64 == 0");
  if (mode >= 3 && variant == -3) return fail(PLIPMI_ERR_INVALID, "the small-M kernel has no split-plane epilogue");
  GemmParams p;
  p.A = A; p.W = W; p.C = C; p.bias = bias; p.M = M; p.N = N; p.K = K; p.lda = K; p.ldw = K; p.ldc = N; p.alpha = 1.f; p.np = 1;
  p.ln_stats = stats; p.ln_ns = ns; p.ln_inv_d = ns > 0 ? 1.0f / (float)(ns * kLnSlice) : 0.f; p.ln_eps = eps;
  p.xb_out = xb_out; p.st_out = st_out;
  if (mode >= 3) { p.lo_io = C; p.C = nullptr; p.planes_other = mode == 4; }
  const int epi = mode == 0 ? EPI_BIAS_LN : mode == 1 ? EPI_QGELU_LN : mode == 2 ? EPI_RESID_EMIT : EPI_RESID_SPLIT;
  const int rc = variant == -3 ? gemm_launch_skinny(dtype, epi, p, reinterpret_cast<hipStream_t>(stream), nullptr)
                               : gemm_launch(dtype, epi, variant, p, reinterpret_cast<hipStream_t>(stream), nullptr);
  if (rc != 0) return fail(PLIPMI_ERR_HIP, "gemm launch failed (LN mode %d, variant %d, M=%d N=%d K=%d): %s", mode, variant, M, N, K,
                           hipGetErrorString((hipError_t)rc));
  return PLIPMI_OK;
}

int plipmi_attention(int dtype, int impl, const void* qkv, void* out, int B, int S, int H, int causal,
                     const int64_t* key_mask, void* stream) {
  if ((dtype != PLIPMI_F32 && dtype != PLIPMI_BF16 && dtype != PLIPMI_F16) || !qkv || !out || B < 0 || S <= 0 || H <= 0)
    return fail(PLIPMI_ERR_INVALID, "bad argument");
  hipError_t e = launch_attention(qkv, out, dtype, B, S, H, causal, key_mask, impl, reinterpret_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(PLIPMI_ERR_HIP, "attention launch (impl %d, S=%d) failed: %s", impl, S, hipGetErrorString(e));
  return PLIPMI_OK;
}

int plipmi_gemm_nt_ld(int dtype, int epilogue, int variant, int M, int N, int K, const void* A, int lda, const void* W,
                      int ldw, const float* bias, float alpha, void* C, void* stream) {
  if (dtype != PLIPMI_F32 && dtype != PLIPMI_BF16 && dtype != PLIPMI_F16) return fail(PLIPMI_ERR_INVALID, "bad dtype");
  if (epilogue < 0 || epilogue > EPI_SCALE) return fail(PLIPMI_ERR_INVALID, "epilogue must be 0..3");
  const int per16 = dtype == PLIPMI_F32 ? 4 : 8;
  if (M < 0 || N <= 0 || K <= 0 || !A || !W || !C || lda < K || ldw < K || lda % per16 || ldw % per16)
    return fail(PLIPMI_ERR_INVALID, "bad shape / leading dimension (must be >= K and a multiple of 16 bytes)");
  if (epilogue != EPI_SCALE && !bias) return fail(PLIPMI_ERR_INVALID, "bias required for this epilogue");
  GemmParams p;
  p.A = A; p.W = W; p.C = C; p.bias = bias; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldw = ldw; p.ldc = N;
  p.alpha = alpha; p.np = 1;
  const int rc = gemm_launch(dtype, epilogue, variant, p, reinterpret_cast<hipStream_t>(stream), nullptr);
  if (rc != 0) return fail(PLIPMI_ERR_HIP, "gemm launch failed (variant %d, M=%d N=%d K=%d): %s", variant, M, N, K,
                           hipGetErrorString((hipError_t)rc));
  return PLIPMI_OK;
}

int plipmi_test_force_gemm_tile(int variant) {
  if (!gemm_force_tile(variant)) return fail(PLIPMI_ERR_INVALID, "tile %d: -1 (cost model), -2 (naive checker) or 0 .. %d", variant, gemm_num_variants() - 1);
  ++g_hook_epoch;
  return PLIPMI_OK;
}
int plipmi_test_remap_gemm_tile(int from, int to) {
  if (!gemm_remap_tile(from, to)) return fail(PLIPMI_ERR_INVALID, "remap %d -> %d: tiles are 0 .. %d (to = -1 clears)", from, to, gemm_num_variants() - 1);
  ++g_hook_epoch;
  return PLIPMI_OK;
}
int plipmi_test_fused_qkv_attention(int mode) {
  if (mode < 0 || mode > 2) return fail(PLIPMI_ERR_INVALID, "fused q/k/v + attention mode %d: 0 (two kernels), 1 (product rule), 2 (fused wherever it applies)", mode);
  g_fuse_qkv_attention = mode;
  ++g_hook_epoch;
  return PLIPMI_OK;
}
int plipmi_test_patch_gather(int on) {
  if (on != 0 && on != 1) return fail(PLIPMI_ERR_INVALID, "patch gather %d: 0 (unfold pass) or 1 (im2col on load where it applies)", on);
  g_patch_gather = on;
  ++g_hook_epoch;
  return PLIPMI_OK;
}
void plipmi_test_reset_hooks(void) {
  g_fuse_qkv_attention = 1;
  g_patch_gather = 1;
  gemm_reset_overrides();
  ++g_hook_epoch;
}
int plipmi_qkv_attention(int dtype, const void* A, const void* W, const float* c2, const float* stats, int ns, float eps, void* out,
                         int B, int S, int H, int causal, const int64_t* key_mask, uint64_t* trace, void* stream) {
  if (!A || !W || !c2 || !stats || !out || ns <= 0 || ns * kLnSlice != H * 64) return fail(PLIPMI_ERR_INVALID, "bad argument");
  if (!qkv_attention_supports(dtype, B, S, H, H * 64))
    return fail(PLIPMI_ERR_INVALID, "the fused q/k/v + attention kernel takes 16-bit operands, 65 .. 80 tokens, widths of 64 H (a multiple of 128)");
  HIP_TRY(launch_qkv_attention(dtype, A, W, c2, stats, 1.0f / (float)(ns * kLnSlice), eps, out, B, S, H, causal, key_mask,
                               reinterpret_cast<hipStream_t>(stream), reinterpret_cast<unsigned long long*>(trace)));
  return PLIPMI_OK;
}
int plipmi_check_async(plipmi_handle h) {
  if (!h) return fail(PLIPMI_ERR_INVALID, "null handle");
  return check_async(h);
}
int plipmi_recode_planes(void* hi, void* lo, size_t rows, int D, int from_dtype, int to_dtype, void* stream) {
  if (!hi || !lo || D <= 0 || D % 8) return fail(PLIPMI_ERR_INVALID, "null planes / width not a multiple of 8");
  HIP_TRY(launch_recode_planes(hi, lo, rows, D, from_dtype, to_dtype, reinterpret_cast<hipStream_t>(stream)));
  return PLIPMI_OK;
}

int plipmi_gemm_variant_built(int dtype, int variant) {
  if (dtype != PLIPMI_F32 && dtype != PLIPMI_BF16 && dtype != PLIPMI_F16) return 0;
  return gemm_variant_is_built(dtype, variant) ? 1 : 0;
}

const char* plipmi_gemm_variant_name(int variant) {
  if (variant < 0 || variant >= gemm_num_variants()) return nullptr;
  return gemm_variant(variant).name;
}

int plipmi_profile_enable(plipmi_handle h, int on) {
  if (!h) return fail(PLIPMI_ERR_INVALID, "null handle");
  if (on) {
    for (ProfRec& r : h->recs) { h->pool.push_back(r.t0); h->pool.push_back(r.t1); }
    h->recs.clear();
  }
  h->prof = on != 0;
  return PLIPMI_OK;
}

int plipmi_profile_read(plipmi_handle h, plipmi_kernel_stat* rows, int max_rows, int* n_rows) {
  if (!h || !rows || !n_rows || max_rows <= 0) return fail(PLIPMI_ERR_INVALID, "bad argument");
  HIP_TRY(hipDeviceSynchronize());
  int n = 0;
  for (ProfRec& r : h->recs) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, r.t0, r.t1));
    int k = 0;
    for (; k < n; ++k)
      if (strncmp(rows[k].name, r.name, sizeof(rows[k].name) - 1) == 0) break;
    if (k == n) {
      if (n == max_rows) continue;
      memset(&rows[n], 0, sizeof(rows[n]));
      strncpy(rows[n].name, r.name, sizeof(rows[n].name) - 1);
      ++n;
    }
    rows[k].calls += 1;
    rows[k].total_ms += ms;
    rows[k].flops += r.flops;
    rows[k].bytes += r.bytes;
  }
  *n_rows = n;
  return PLIPMI_OK;
}

}  // extern "C"
