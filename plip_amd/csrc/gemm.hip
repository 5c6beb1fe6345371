// gemm.hip -- variant registry and dispatch for the NT GEMM (kernels live in gemm.h).
#include <stdlib.h>

#include <algorithm>
#include "gemm_inst.h"

namespace plipmi {

int gemm_num_cus();

static const GemmVariant kVariants[kNumVariants] = {
    {"128x128_w2x2_regstage", 128, 128, 256, false}, {"128x128_w2x2_glds", 128, 128, 256, true},
    {"256x128_w4x2_regstage", 256, 128, 512, false}, {"256x128_w4x2_glds", 256, 128, 512, true},
    {"256x256_w4x2_regstage", 256, 256, 512, false}, {"256x256_w4x2_glds", 256, 256, 512, true},
    {"256x256_w4x2_glds_fragpipe", 256, 256, 512, true}, {"256x256_w4x2_glds_fragpipe_prio", 256, 256, 512, true},
    {"128x128_w2x2_glds_fragpipe", 128, 128, 256, true}, {"256x128_w4x2_glds_fragpipe", 256, 128, 512, true},
    {"256x256_w4x2_persist", 256, 256, 512, true}, {"128x128_w2x2_persist", 128, 128, 256, true},
    {"256x128_w4x2_persist", 256, 128, 512, true},
    {"256x256_w4x2_glds_fragpipe_l2pf2", 256, 256, 512, true}, {"128x128_w2x2_glds_fragpipe_l2pf2", 128, 128, 256, true},
    {"256x256_w4x2_glds_fragpipe_l2pf3", 256, 256, 512, true},
    {"256x256_w4x2_glds_spreadfill", 256, 256, 512, true}, {"128x128_w2x2_glds_spreadfill", 128, 128, 256, true},
    {"256x128_w4x2_glds_3stage", 256, 128, 512, true}, {"128x256_w2x4_glds_3stage", 128, 256, 512, true},
    {"256x256_w2x4_8phase", 256, 256, 512, true},
    {"256x128_w4x2_3stage_xprefetch", 256, 128, 512, true}, {"128x256_w2x4_3stage_xprefetch", 128, 256, 512, true},
    {"192x256_w2x4_glds_spreadfill", 192, 256, 512, true}, {"192x256_w2x4_glds_fragpipe", 192, 256, 512, true},
    {"320x256_w2x4_glds_spreadfill", 320, 256, 512, true}, {"320x256_w2x4_glds", 320, 256, 512, true},
    {"256x256_w2x2_1wave_fragpipe", 256, 256, 256, true}, {"256x256_w2x2_1wave_spreadfill", 256, 256, 256, true},
    {"256x256_w4x2_glds_fill2", 256, 256, 512, true}, {"320x256_w2x4_glds_fill2", 320, 256, 512, true},
    {"192x256_w2x4_glds_fill2", 192, 256, 512, true},
    {"256x256_w4x2_glds_fill3", 256, 256, 512, true}, {"320x256_w2x4_glds_fill3", 320, 256, 512, true},
    {"192x256_w2x4_glds_fill3", 192, 256, 512, true},
    {"256x256_w4x2_bufdma_fill3", 256, 256, 512, true}, {"320x256_w2x4_bufdma_fill3", 320, 256, 512, true},
    {"192x256_w2x4_bufdma_fill3", 192, 256, 512, true},
    {"256x256_w4x2_bufdma_fragpipe", 256, 256, 512, true}, {"320x256_w2x4_bufdma", 320, 256, 512, true},
    {"192x256_w2x4_bufdma_fragpipe", 192, 256, 512, true}, {"128x128_w2x2_bufdma_fragpipe", 128, 128, 256, true},
    {"256x256_w4x2_bufdma_fill2", 256, 256, 512, true},
};

int gemm_num_cus() {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
    cus = cus / 8 * 8;
  }
  return cus;
}

int gemm_num_variants() { return kNumVariants; }
const GemmVariant& gemm_variant(int v) { return kVariants[v]; }

// Process-wide override of the tile choice: a TEST / A-B hook only (PLIPMI_GEMM_VARIANT, plipmi_set_gemm_variant).
// The product path never writes it; the tile POLICY is a per-call argument owned by the handle.
static int g_override = -100;  // -100 = not yet read
void gemm_set_default_override(int variant) { g_override = variant; }
bool gemm_variant_is_built(int dtype, int variant) {
  return dtype == 1 ? gemm_built_bf16(variant) : gemm_built_f32(variant);
}

int gemm_default_variant(int dtype, int M, int N, int K, int epi, int policy) {
  if (g_override == -100) {
    const char* e = getenv("PLIPMI_GEMM_VARIANT");
    g_override = e ? atoi(e) : -1;
  }
  if (g_override >= 0 || g_override == -2) {
    if (g_override >= 0 && (N % kVariants[g_override].bn != 0 || !gemm_variant_is_built(dtype, g_override))) return 1;
    return g_override;
  }
  // Tile choice = wave quantisation.  One 256-wide tile family runs one workgroup per CU (LDS-bound), so a GEMM
  // takes ceil(tiles / CUs) rounds of roughly tile-area-proportional time (measured, profiles/
  // r01_gemm_variants_tflops.txt: per-output cost is within 5 % across 192x256 / 256x256 / 320x256); the 128x128
  // tile runs two workgroups per CU at ~1.2x the per-output cost, its last partial round cheaper.  Pick the
  // candidate with the smallest rounds x tile-time.  At bs=256 this selects 256x256 for the QKV projections,
  // 320x256 for fc1, 192x256 for fc2 / out-proj / patch embedding -- the measured best in all nine shapes.
  (void)K;
  if (M <= 1024) return 1;
  // Policy 1 = the two towers are co-scheduled on two streams: CUs a partial round would leave idle are taken by
  // the other tower's kernels, so quantisation stops mattering and the tile with the fewest L2->LDS bytes per FLOP
  // wins (in-process A/B, profiles/r01_gemm_policy_ab.txt: 5.69 ms/step vs 6.04 ms with the cost model).
  if (policy == 1 && dtype == 1 && N % 256 == 0) return 36;
  // Policy 2: as 1, but the fp32 residual epilogues take the 192x256 tile, whose register budget lets it request the
  // residual rows one block ahead (gemm.h kRowOperand).
  if (policy == 2 && dtype == 1 && N % 256 == 0) return (epi_is_resid(epi) || epi == EPI_PATCH) ? 37 : 36;
  // Policy 3: as 2, with the 256x256 fill2 tile for the bias-only (QKV) epilogue
  if (policy == 3 && dtype == 1 && N % 256 == 0)
    return (epi_is_resid(epi) || epi == EPI_PATCH) ? 37 : ((epi == EPI_BIAS || epi == EPI_BIAS_LN) ? 42 : 36);
  const int cus = gemm_num_cus();
  struct Cand { int variant, bm, bn, per_cu; double rel; };
  const Cand cands_bf16[] = {{42, 256, 256, 1, 1.00}, {36, 320, 256, 1, 1.00}, {37, 192, 256, 1, 1.05}, {41, 128, 128, 2, 1.21}};
  const Cand cands_f32[] = {{38, 256, 256, 1, 1.00}, {39, 320, 256, 1, 1.00}, {40, 192, 256, 1, 1.05}, {41, 128, 128, 2, 1.21}};
  const Cand* cands = dtype == 1 ? cands_bf16 : cands_f32;
  int best = 41;
  double best_cost = 1e300;
  for (int i = 0; i < 4; ++i) {
    const Cand& c = cands[i];
    if (N % c.bn) continue;
    const long tiles = (long)((M + c.bm - 1) / c.bm) * (N / c.bn);
    const long slots = (long)cus * c.per_cu;
    const double t_round = (double)c.bm * c.bn * c.per_cu * c.rel;   // time for a CU to finish its resident tiles
    const long full = tiles / slots, rem = tiles % slots;
    double cost = full * t_round;
    if (rem) cost += (c.per_cu == 2 && rem <= cus) ? 0.6 * t_round : t_round;  // lone workgroups run faster
    if (cost < best_cost) { best_cost = cost; best = c.variant; }
  }
  return best;
}

// EXPERIMENTAL fp8 test hook (gemm_fp8.hip): variants 0..3, bias / bias_qgelu, N % 256 == 0, K % 128 == 0
int gemm_launch_fp8(int epi, int variant, const GemmParams& p, hipStream_t stream) {
  if (p.M <= 0) return 0;
  if (p.N % 256 != 0 || p.K % 128 != 0) return (int)hipErrorInvalidValue;
  if ((size_t)p.M * p.lda >= (1ull << 32) || (size_t)p.N * p.ldw >= (1ull << 32)) return (int)hipErrorInvalidValue;
  GemmLaunchFn fn = gemm_get_fp8(variant < 0 ? 0 : variant, epi);
  if (!fn) return (int)hipErrorInvalidValue;
  GemmParams pr = p;
  const int nbn = p.N / 256;
  const double a_bytes = (double)p.M * p.K, w_bytes = (double)p.N * p.K;
  int best_xn = 1;
  double best = 1e300;
  for (int xn = 1; xn <= 8; xn *= 2) {
    if (nbn % xn) continue;
    double cost = a_bytes * xn + w_bytes * (8.0 / xn);
    if (w_bytes / xn > 2.5e6) cost += 4.0 * w_bytes * 8.0;
    if (cost < best) { best = cost; best_xn = xn; }
  }
  pr.gw = nbn / best_xn;
  return fn(pr, stream);
}

static const char* kEpiNames[EPI_COUNT] = {"bias", "bias_qgelu", "bias_resid", "scale", "patch", "ln_bias", "ln_qgelu",
                                           "resid_emit", "resid_split"};

int gemm_launch(int dtype, int epi, int variant, const GemmParams& p, hipStream_t stream, const char** kernel_name,
                int policy) {
  if (variant == -1) variant = gemm_default_variant(dtype, p.M, p.N, p.K, epi, policy);
  // the buffer-addressed kernels (35..42) carry 32-bit byte offsets: operands or outputs of 4 GiB and more take the
  // 64-bit-address tile (variant 1, global_load_lds with per-lane 64-bit addresses)
  if (variant >= 35 && variant <= 42) {
    const size_t es = dtype == 1 ? 2 : 4;
    const size_t out_rows = epi == EPI_PATCH ? (size_t)p.M + p.M / (p.np > 0 ? p.np : 1) + 1 : (size_t)p.M;
    const size_t span = std::max(std::max((size_t)p.M * p.lda * es, (size_t)p.N * p.ldw * es), out_rows * p.ldc * 4);
    if (span >= (1ull << 32)) variant = 1;
  }
  if (p.M <= 0) return 0;
  // a device-side row count is read by gemm_nt_kernel only (not by the archived persistent / 8-phase forms or the naive kernel)
  if (p.m_dev && (variant < 0 || (variant >= 10 && variant <= 12) || variant == 20)) return (int)hipErrorInvalidValue;
  const int bk = dtype == 1 ? 64 : 32;
  if (variant >= 0) {
    if (variant >= kNumVariants) return (int)hipErrorInvalidValue;
    if (p.N % kVariants[variant].bn != 0 || p.K % bk != 0) return (int)hipErrorInvalidValue;
  } else if (p.N % 4 != 0) {
    return (int)hipErrorInvalidValue;
  }
  GemmLaunchFn fn = dtype == 1 ? gemm_get_bf16(variant, epi) : gemm_get_f32(variant, epi);
  if (!fn) return (int)hipErrorInvalidValue;
  GemmParams pr = p;
  if (variant >= 0) {
    // column-group raster: minimise A*xn + W*(8/xn) fabric bytes subject to an XCD's W share fitting its L2
    const double esz = dtype == 1 ? 2.0 : 4.0;
    const int nbn = p.N / kVariants[variant].bn;
    const double a_bytes = (double)p.M * p.K * esz, w_bytes = (double)p.N * p.K * esz;
    static int force = -2;
    if (force == -2) { const char* e = getenv("PLIPMI_GEMM_XN"); force = e ? atoi(e) : -1; }
    int best_xn = 1;
    double best = 1e300;
    for (int xn = 1; xn <= 8; xn *= 2) {
      if (nbn % xn) continue;
      const double share = w_bytes / xn;
      double cost = a_bytes * xn + w_bytes * (8.0 / xn);
      if (share > 2.5e6) cost += 4.0 * w_bytes * 8.0;  // W share thrashes the L2: every M row re-fetches it
      if (cost < best) { best = cost; best_xn = xn; }
    }
    if (force > 0 && nbn % force == 0) best_xn = force;
    pr.gw = nbn / best_xn;
  }
  if (p.trace) {  // timeline runs may ablate parts of the kernel (never on the product path: trace is null there)
    static int ablate = -1;
    if (ablate < 0) { const char* e = getenv("PLIPMI_GEMM_ABLATE"); ablate = e ? atoi(e) : 0; }
    if (ablate) { pr.ablate = ablate; return fn(pr, stream); }
  }
  if (kernel_name) {
    // static table of names: "gemm_nt<dtype,tile,epi>"
    static char names[2][kNumVariants + 1][EPI_COUNT][64];
    char* nm = names[dtype][variant < 0 ? kNumVariants : variant][epi];
    if (!nm[0])
      snprintf(nm, 64, "gemm_nt<%s,%s,%s>", dtype == 1 ? "bf16" : "f32", variant < 0 ? "naive" : kVariants[variant].name,
               kEpiNames[epi]);
    *kernel_name = nm;
  }
  return fn(pr, stream);
}

}  // namespace plipmi
