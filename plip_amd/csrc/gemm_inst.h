// gemm_inst.h -- per-dtype instantiation table of gemm_nt_kernel (included by gemm_f32.hip / gemm_bf16.hip
// so the two halves compile in parallel).
#pragma once
#include "gemm.h"
#include "gemm_persist.h"
#include "gemm_8phase.h"

namespace plipmi {

typedef int (*GemmLaunchFn)(const GemmParams&, hipStream_t);

template <typename T, int BM, int BN, int WM, int WN, int EPI, bool GLDS, int SCHED = 0, int L2PF = 0, int NSTAGE = 2, int ADDR = 0>
int launch_tiled(const GemmParams& p, hipStream_t stream) {
  constexpr int NT = WM * WN * 64;
  constexpr int LDS = NSTAGE * (BM + BN) * 128;
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

template <typename T, int EPI>
int launch_naive(const GemmParams& p, hipStream_t stream) {
  dim3 grid((p.N / 4 + 63) / 64, p.M);
  hipLaunchKernelGGL((gemm_nt_naive_kernel<T, EPI>), grid, dim3(64), 0, stream, p);
  return (int)hipGetLastError();
}

constexpr int kNumVariants = 43;

// table[variant][epilogue]
template <typename T>
struct GemmTable {
  template <int EPI>
  static GemmLaunchFn pick(int variant) {
    switch (variant) {
      case 0: return launch_tiled<T, 128, 128, 2, 2, EPI, false>;
      case 1: return launch_tiled<T, 128, 128, 2, 2, EPI, true>;
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
      case 36: return launch_tiled<T, 320, 256, 2, 4, EPI, true, 6, 0, 2, 1>;
      case 37: return launch_tiled<T, 192, 256, 2, 4, EPI, true, 6, 0, 2, 1>;
      case 38: return launch_tiled<T, 256, 256, 4, 2, EPI, true, 1, 0, 2, 1>;
      case 39: return launch_tiled<T, 320, 256, 2, 4, EPI, true, 0, 0, 2, 1>;
      case 40: return launch_tiled<T, 192, 256, 2, 4, EPI, true, 1, 0, 2, 1>;
      case 41: return launch_tiled<T, 128, 128, 2, 2, EPI, true, 1, 0, 2, 1>;
      case 42: return launch_tiled<T, 256, 256, 4, 2, EPI, true, 5, 0, 2, 1>;
      case -2: return launch_naive<T, EPI>;
      default: return nullptr;
    }
  }
  static GemmLaunchFn get(int variant, int epi) {
    switch (epi) {
      case EPI_BIAS: return pick<EPI_BIAS>(variant);
      case EPI_BIAS_QGELU: return pick<EPI_BIAS_QGELU>(variant);
      case EPI_BIAS_RESID: return pick<EPI_BIAS_RESID>(variant);
      case EPI_SCALE: return pick<EPI_SCALE>(variant);
      case EPI_PATCH: return pick<EPI_PATCH>(variant);
      default: return nullptr;
    }
  }
};

GemmLaunchFn gemm_get_f32(int variant, int epi);
GemmLaunchFn gemm_get_bf16(int variant, int epi);
GemmLaunchFn gemm_get_fp8(int variant, int epi);   // experimental: variants 0..3, bias / bias_qgelu only

}  // namespace plipmi
