// gemm_inst.h -- per-dtype instantiation table of gemm_nt_kernel (included by gemm_f32.hip / gemm_bf16.hip
// so the two halves compile in parallel).
#pragma once
#include "gemm.h"
#ifdef PLIPMI_ALL_VARIANTS
#include "gemm_persist.h"
#include "gemm_8phase.h"
#endif

namespace plipmi {

typedef int (*GemmLaunchFn)(const GemmParams&, hipStream_t);

template <typename T, int BM, int BN, int WM, int WN, int EPI, bool GLDS, int SCHED = 0, int L2PF = 0, int NSTAGE = 2, int ADDR = 0>
int launch_tiled(const GemmParams& p, hipStream_t stream) {
  constexpr int NT = WM * WN * 64;
  // + rstd per tile row for the LayerNorm-folded epilogues
  constexpr int LDS = NSTAGE * (BM + BN) * 128 + (epi_is_ln(EPI) ? BM * 4 : 0);
  auto kern = gemm_nt_kernel<T, BM, BN, WM, WN, EPI, GLDS, SCHED, L2PF, NSTAGE, ADDR>;
  static bool attr_set = false;  // one handle per process; set once per instantiation
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int nblk = ((p.M + BM - 1) / BM) * (p.N / BN);
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(NT), LDS, stream, p);
  return (int)hipGetLastError();
}

int gemm_num_cus();  // gemm.hip

#ifdef PLIPMI_ALL_VARIANTS
// persistent form: one resident workgroup per CU slot walks a strip of tiles
template <typename T, int BM, int BN, int WM, int WN, int EPI, int SCHED = 1>
int launch_persist(const GemmParams& p, hipStream_t stream) {
  constexpr int NT = WM * WN * 64;
  constexpr int LDS = 2 * (BM + BN) * 128;
  auto kern = gemm_nt_persist_kernel<T, BM, BN, WM, WN, EPI, SCHED>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int ntiles = ((p.M + BM - 1) / BM) * (p.N / BN);
  const int per_cu = (160 * 1024) / LDS;                       // resident workgroups per CU (LDS-bound)
  int grid = gemm_num_cus() * (per_cu < 1 ? 1 : per_cu);
  const int need = (ntiles + 7) / 8 * 8;                        // grid must be a multiple of 8 (XCD strips)
  if (grid > need) grid = need;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), LDS, stream, p);
  return (int)hipGetLastError();
}

template <typename T, int EPI>
int launch_8phase(const GemmParams& p, hipStream_t stream) {
  constexpr int LDS = 2 * 512 * 128;
  auto kern = gemm_nt_8phase_kernel<T, EPI>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int nblk = ((p.M + 255) / 256) * (p.N / 256);
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), LDS, stream, p);
  return (int)hipGetLastError();
}

#endif  // PLIPMI_ALL_VARIANTS

template <typename T, int EPI>
int launch_naive(const GemmParams& p, hipStream_t stream) {
  dim3 grid((p.N / 4 + 63) / 64, p.M);
  hipLaunchKernelGGL((gemm_nt_naive_kernel<T, EPI>), grid, dim3(64), 0, stream, p);
  return (int)hipGetLastError();
}

constexpr int kNumVariants = 43;

// Which tile variants a build carries.  The default build holds the PRODUCT set only -- per dtype the three
// one-workgroup-per-CU buffer-DMA tiles the policy picks from, the 128x128 two-per-CU tile for narrow / small
// problems, and the 64-bit-address 128x128 tile for operands of 4 GiB and more -- which is what parity tests and
// build() pay for.  -DPLIPMI_ALL_VARIANTS restores the 43-entry schedule archaeology of round 1
// (profiles/r01_gemm_variants_tflops.txt) for A/B work; variant numbers are the same in both builds.
#ifdef PLIPMI_ALL_VARIANTS
constexpr bool kAllVariants = true;
#else
constexpr bool kAllVariants = false;
#endif
template <typename T>
constexpr bool gemm_variant_built(int v) {
  if (kAllVariants) return v >= -2 && v < kNumVariants && v != -1;
  if (v == -2 || v == 1 || v == 41) return true;
  return sizeof(T) == 2 ? (v == 36 || v == 37 || v == 42) : (v == 38 || v == 39 || v == 40);
}

// table[variant][epilogue]
template <typename T>
struct GemmTable {
  template <int EPI>
  static GemmLaunchFn pick(int variant) {
    constexpr bool kB = sizeof(T) == 2, kF = sizeof(T) == 4;
    // LayerNorm-folded epilogues: bf16 product tiles only
    constexpr bool kLn = epi_is_ln(EPI) || epi_emits_stats(EPI);
    if constexpr (kLn && !kB) {
      return nullptr;
    } else {
    switch (variant) {
      case 1: return launch_tiled<T, 128, 128, 2, 2, EPI, true>;
      case 36: if constexpr (kB || kAllVariants) return launch_tiled<T, 320, 256, 2, 4, EPI, true, 6, 0, 2, 1>; else return nullptr;
      case 37: if constexpr (kB || kAllVariants) return launch_tiled<T, 192, 256, 2, 4, EPI, true, 6, 0, 2, 1>; else return nullptr;
      case 42: if constexpr (kB || kAllVariants) return launch_tiled<T, 256, 256, 4, 2, EPI, true, 5, 0, 2, 1>; else return nullptr;
      case 38: if constexpr ((kF || kAllVariants) && !kLn) return launch_tiled<T, 256, 256, 4, 2, EPI, true, 1, 0, 2, 1>; else return nullptr;
      case 39: if constexpr ((kF || kAllVariants) && !kLn) return launch_tiled<T, 320, 256, 2, 4, EPI, true, 0, 0, 2, 1>; else return nullptr;
      case 40: if constexpr ((kF || kAllVariants) && !kLn) return launch_tiled<T, 192, 256, 2, 4, EPI, true, 1, 0, 2, 1>; else return nullptr;
      case 41: return launch_tiled<T, 128, 128, 2, 2, EPI, true, 1, 0, 2, 1>;
      case -2: if constexpr (!kLn) return launch_naive<T, EPI>; else return nullptr;
      default: break;
    }
#ifdef PLIPMI_ALL_VARIANTS
    if constexpr (!kLn) {
      switch (variant) {
        case 0: return launch_tiled<T, 128, 128, 2, 2, EPI, false>;
        case 2: return launch_tiled<T, 256, 128, 4, 2, EPI, false>;
        case 3: return launch_tiled<T, 256, 128, 4, 2, EPI, true>;
        case 4: return launch_tiled<T, 256, 256, 4, 2, EPI, false>;
        case 5: return launch_tiled<T, 256, 256, 4, 2, EPI, true>;
        case 6: return launch_tiled<T, 256, 256, 4, 2, EPI, true, 1>;
        case 7: return launch_tiled<T, 256, 256, 4, 2, EPI, true, 2>;
        case 8: return launch_tiled<T, 128, 128, 2, 2, EPI, true, 1>;
        case 9: return launch_tiled<T, 256, 128, 4, 2, EPI, true, 1>;
        case 10: return launch_persist<T, 256, 256, 4, 2, EPI, 1>;
        case 11: return launch_persist<T, 128, 128, 2, 2, EPI, 1>;
        case 12: return launch_persist<T, 256, 128, 4, 2, EPI, 1>;
        case 13: return launch_tiled<T, 256, 256, 4, 2, EPI, true, 1, 2>;
        case 14: return launch_tiled<T, 128, 128, 2, 2, EPI, true, 1, 2>;
        case 15: return launch_tiled<T, 256, 256, 4, 2, EPI, true, 1, 3>;
        case 16: return launch_tiled<T, 256, 256, 4, 2, EPI, true, 3>;
        case 17: return launch_tiled<T, 128, 128, 2, 2, EPI, true, 3>;
        case 18: return launch_tiled<T, 256, 128, 4, 2, EPI, true, 1, 0, 3>;
        case 19: return launch_tiled<T, 128, 256, 2, 4, EPI, true, 1, 0, 3>;
        case 20: return launch_8phase<T, EPI>;
        case 21: return launch_tiled<T, 256, 128, 4, 2, EPI, true, 4, 0, 3>;
        case 22: return launch_tiled<T, 128, 256, 2, 4, EPI, true, 4, 0, 3>;
        case 23: return launch_tiled<T, 192, 256, 2, 4, EPI, true, 3>;
        case 24: return launch_tiled<T, 192, 256, 2, 4, EPI, true, 1>;
        case 25: return launch_tiled<T, 320, 256, 2, 4, EPI, true, 3>;
        case 26: return launch_tiled<T, 320, 256, 2, 4, EPI, true, 0>;
        case 27: return launch_tiled<T, 256, 256, 2, 2, EPI, true, 1>;
        case 28: return launch_tiled<T, 256, 256, 2, 2, EPI, true, 3>;
        case 29: return launch_tiled<T, 256, 256, 4, 2, EPI, true, 5>;
        case 30: return launch_tiled<T, 320, 256, 2, 4, EPI, true, 5>;
        case 31: return launch_tiled<T, 192, 256, 2, 4, EPI, true, 5>;
        case 32: return launch_tiled<T, 256, 256, 4, 2, EPI, true, 6>;
        case 33: return launch_tiled<T, 320, 256, 2, 4, EPI, true, 6>;
        case 34: return launch_tiled<T, 192, 256, 2, 4, EPI, true, 6>;
        case 35: return launch_tiled<T, 256, 256, 4, 2, EPI, true, 6, 0, 2, 1>;
        default: break;
      }
    }
#endif
    return nullptr;
    }
  }
  static GemmLaunchFn get(int variant, int epi) {
    switch (epi) {
      case EPI_BIAS: return pick<EPI_BIAS>(variant);
      case EPI_BIAS_QGELU: return pick<EPI_BIAS_QGELU>(variant);
      case EPI_BIAS_RESID: return pick<EPI_BIAS_RESID>(variant);
      case EPI_SCALE: return pick<EPI_SCALE>(variant);
      case EPI_PATCH: return pick<EPI_PATCH>(variant);
      case EPI_BIAS_LN: return pick<EPI_BIAS_LN>(variant);
      case EPI_QGELU_LN: return pick<EPI_QGELU_LN>(variant);
      case EPI_RESID_EMIT: return pick<EPI_RESID_EMIT>(variant);
      case EPI_RESID_SPLIT: return pick<EPI_RESID_SPLIT>(variant);
      default: return nullptr;
    }
  }
};

GemmLaunchFn gemm_get_f32(int variant, int epi);
bool gemm_built_f32(int variant);
bool gemm_built_bf16(int variant);
GemmLaunchFn gemm_get_bf16(int variant, int epi);
GemmLaunchFn gemm_get_fp8(int variant, int epi);   // experimental: variants 0..3, bias / bias_qgelu only

}  // namespace plipmi
