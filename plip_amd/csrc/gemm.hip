// gemm.hip -- variant registry and dispatch for the NT GEMM (kernels live in gemm.h).
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include "gemm_inst.h"

namespace plipmi {

int gemm_num_cus();

static const GemmVariant kVariants[kNumVariants] = {
    {"128x128_w2x2_glds64", 128, 128, 256}, {"128x128_w2x2_bufdma", 128, 128, 256}, {"256x256_w4x2_bufdma", 256, 256, 512},
    {"320x256_w2x4_bufdma", 320, 256, 512}, {"192x256_w2x4_bufdma", 192, 256, 512}, {"160x256_w2x4_bufdma", 160, 256, 512},
    {"160x256_w2x4_ring3", 160, 256, 512},
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

// Process-wide overrides for TESTS and A/B measurements (plipmi_test.h: plipmi_test_force_gemm_tile / _remap_gemm_tile): the product
// path never writes them, and nothing here is read from the environment.
static int g_override = -1;         // -1 = the cost model below chooses, -2 = the naive checker kernel, >= 0 = that tile
static int g_remap[kNumVariants];   // A/B runs: the cost model's choice a runs as g_remap[a] - 1 (0 = itself)
bool gemm_force_tile(int variant) {
  if (variant < -2 || variant >= kNumVariants) return false;
  g_override = variant;
  return true;
}
bool gemm_remap_tile(int from, int to) {
  if (from < 0 || from >= kNumVariants || to < -1 || to >= kNumVariants) return false;
  g_remap[from] = to + 1;
  return true;
}
void gemm_reset_overrides() {
  g_override = -1;
  for (int& r : g_remap) r = 0;
}
bool gemm_variant_is_built(int dtype, int variant) {
  return dtype == 1 ? gemm_built_bf16(variant) : dtype == 2 ? gemm_built_f16(variant) : gemm_built_f32(variant);
}

int gemm_default_variant(int dtype, int M, int N, int K) {
  if (g_override >= 0 || g_override == -2) {
    if (g_override >= 0 && (N % kVariants[g_override].bn != 0 || !gemm_variant_is_built(dtype, g_override))) return 0;
    return g_override;
  }
  // Tile choice = wave quantisation.  The 256-wide tiles run one workgroup per CU (LDS-bound), so a GEMM takes
  // ceil(tiles / CUs) rounds of roughly tile-area-proportional time; `rel` is a tile's measured cost per output against
  // 256x256 / 320x256 (K loop cycles per MFMA, prologue + epilogue share: profiles/r03_gemm_tiles.txt); the 128x128 tile runs
  // two workgroups per CU, its last partial round cheaper.  Smallest rounds x tile-time wins.  At bs=256 that is 256x256 for
  // q/k/v, 320x256 for fc1 and 160x256 on its three-stage ring (240 / 248 tiles = one round on 256 CUs) for out-proj / fc2 /
  // the patch GEMM -- the measured best on all eight shapes, kernel by kernel and in the step, on one stream and on two
  // (profiles/r03_gemm_tiles.txt).
  (void)K;
  if (M <= 1024) return 0;
  const int cus = gemm_num_cus();
  struct Cand { int variant, bm, bn, per_cu; double rel; };
  // (6 = 160x256 on three LDS stages: as 5 with operands in the Infinity Cache, 6-9 % faster with operands from HBM)
  // (3 at 0.99: with the 16x16x32 loops a round-count tie between 256x256 and 320x256 -- ViT-L/14's fc1 -- goes to the larger tile,
  //  +1.7 % on the ViT-L/14@336 step; no ViT-B/32 choice changes)
  const Cand cands[] = {{2, 256, 256, 1, 1.00}, {3, 320, 256, 1, 0.99}, {4, 192, 256, 1, 1.05}, {6, 160, 256, 1, 1.08},
                        {5, 160, 256, 1, 1.10}, {1, 128, 128, 2, 1.21}};
  int best = 1;
  double best_cost = 1e300;
  for (const Cand& c : cands) {
    if (N % c.bn) continue;
    const long tiles = (long)((M + c.bm - 1) / c.bm) * (N / c.bn);
    const long slots = (long)cus * c.per_cu;
    const double t_round = (double)c.bm * c.bn * c.per_cu * c.rel;   // time for a CU to finish its resident tiles
    const long full = tiles / slots, rem = tiles % slots;
    double cost = full * t_round;
    if (rem) cost += (c.per_cu == 2 && rem <= cus) ? 0.6 * t_round : t_round;  // lone workgroups run faster
    if (cost < best_cost) { best_cost = cost; best = c.variant; }
  }
  if (g_remap[best] > 0 && N % kVariants[g_remap[best] - 1].bn == 0) best = g_remap[best] - 1;
  return best;
}

static const char* kEpiNames[EPI_COUNT] = {"bias", "bias_qgelu", "bias_resid", "scale", "patch", "ln_bias", "ln_qgelu",
                                           "resid_emit", "resid_split"};

int gemm_launch(int dtype, int epi, int variant, const GemmParams& p, hipStream_t stream, const char** kernel_name) {
  if (variant == -1) variant = gemm_default_variant(dtype, p.M, p.N, p.K);
  // the buffer-addressed kernels (1..) carry 32-bit byte offsets: operands or outputs of 4 GiB and more take the
  // 64-bit-address tile (variant 0, global_load_lds with per-lane 64-bit addresses)
  if (variant >= 1) {
    const size_t es = dtype == 0 ? 4 : 2;
    const size_t out_rows = epi == EPI_PATCH ? (size_t)p.M + p.M / (p.np > 0 ? p.np : 1) + 1 : (size_t)p.M;
    const size_t span = std::max(std::max((size_t)p.M * p.lda * es, (size_t)p.N * p.ldw * es), out_rows * p.ldc * 4);
    if (span >= (1ull << 32)) variant = 0;
  }
  if (p.M <= 0) return 0;
  // a device-side row count is read by gemm_nt_kernel only (not by the naive kernel)
  if (p.m_dev && variant < 0) return (int)hipErrorInvalidValue;
  const int bk = dtype == 0 ? 32 : 64;
  if (variant >= 0) {
    if (variant >= kNumVariants) return (int)hipErrorInvalidValue;
    if (p.N % kVariants[variant].bn != 0 || p.K % bk != 0) return (int)hipErrorInvalidValue;
  } else if (p.N % 4 != 0) {
    return (int)hipErrorInvalidValue;
  }
  GemmLaunchFn fn = dtype == 1 ? gemm_get_bf16(variant, epi) : dtype == 2 ? gemm_get_f16(variant, epi) : gemm_get_f32(variant, epi);
  if (!fn) return (int)hipErrorInvalidValue;
  GemmParams pr = p;
  if (variant >= 0) {
    // column-group raster: minimise A*xn + W*(8/xn) fabric bytes subject to an XCD's W share fitting its L2
    const double esz = dtype == 0 ? 4.0 : 2.0;
    const int nbn = p.N / kVariants[variant].bn;
    const double a_bytes = (double)p.M * p.K * esz, w_bytes = (double)p.N * p.K * esz;
    int best_xn = 1;
    double best = 1e300;
    for (int xn = 1; xn <= 8; xn *= 2) {
      if (nbn % xn) continue;
      const double share = w_bytes / xn;
      double cost = a_bytes * xn + w_bytes * (8.0 / xn);
      if (share > 2.5e6) cost += 4.0 * w_bytes * 8.0;  // W share thrashes the L2: every M row re-fetches it
      if (cost < best) { best = cost; best_xn = xn; }
    }
    pr.gw = nbn / best_xn;
  }
  if (kernel_name) {
    // static table of names: "gemm_nt<dtype,tile,epi>"
    static char names[3][kNumVariants + 1][EPI_COUNT][64];
    char* nm = names[dtype][variant < 0 ? kNumVariants : variant][epi];
    if (!nm[0])
      snprintf(nm, 64, "gemm_nt<%s,%s,%s>", dtype == 1 ? "bf16" : dtype == 2 ? "f16" : "f32", variant < 0 ? "naive" : kVariants[variant].name,
               kEpiNames[epi]);
    *kernel_name = nm;
  }
  return fn(pr, stream);
}

// The patch GEMM with im2col on load (gemm.h ADDR 2): C rows = patches, A gathered from the fp32 NCHW pixels.  Applies to 16-bit
// engines, patch sides 16 / 32 (whole 64-column K tiles of 4 / 2 patch rows), widths of whole 256-column tiles, and batches the cost
// model gives the ring tile anyway; everything else keeps the unfold pass + the plain patch GEMM.
bool gemm_gather_supports(int dtype, int B, int image, int patch, int N) {
  if (dtype != 1 && dtype != 2) return false;
  if (patch != 16 && patch != 32) return false;
  if (image % patch || N % 256) return false;
  const int g = image / patch, M = B * g * g, K = 3 * patch * patch;
  if (K / 64 < 2 || (size_t)B * 3 * image * image * 4 >= (1ull << 32)) return false;
  // the ring tile addresses C and W through 32-bit buffer offsets as well: the output rows (patch rows + one CLS row per image + 1,
  // as gemm_launch counts them) and the weight span must stay below 4 GiB too -- for 16-pixel patches the fp32 output of an image
  // (197 x 768 x 4 B) is LARGER than its pixels; beyond, the unfold pass + gemm_launch's 64-bit tile take over
  if (((size_t)M + B + 1) * N * 4 >= (1ull << 32) || (size_t)N * K * 2 >= (1ull << 32)) return false;
  return gemm_default_variant(dtype, M, N, K) == 6;
}
int gemm_launch_gather(int dtype, const GemmParams& p, hipStream_t stream, const char** kernel_name) {
  if (p.M <= 0) return 0;
  if ((!p.pix && !p.tiles) || (dtype != 1 && dtype != 2) || p.N % 256 || p.K % 64 || p.K / 64 < 2) return (int)hipErrorInvalidValue;
  GemmParams pr = p;
  pr.gw = 0;    // N-major sweep: the three or four column tiles of a row panel read the same pixels back to back
  const bool u8 = p.tiles != nullptr;
  if (kernel_name)
    *kernel_name = u8 ? (dtype == 1 ? "gemm_nt<bf16,160x256_w2x4_ring3,patch_gather_u8>" : "gemm_nt<f16,160x256_w2x4_ring3,patch_gather_u8>")
                      : (dtype == 1 ? "gemm_nt<bf16,160x256_w2x4_ring3,patch_gather>" : "gemm_nt<f16,160x256_w2x4_ring3,patch_gather>");
  GemmLaunchFn fn = u8 ? (dtype == 1 ? gemm_get_gather_u8_bf16() : gemm_get_gather_u8_f16())
                       : (dtype == 1 ? gemm_get_gather_bf16() : gemm_get_gather_f16());
  return fn(pr, stream);
}

}  // namespace plipmi
