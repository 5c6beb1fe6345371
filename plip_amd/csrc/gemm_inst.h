// gemm_inst.h -- per-dtype instantiation table of gemm_nt_kernel (included by gemm_f32.hip / gemm_bf16.hip /
// gemm_f16.hip so the three compile in parallel).
#pragma once
#include <mutex>

#include "gemm.h"

namespace plipmi {

typedef int (*GemmLaunchFn)(const GemmParams&, hipStream_t);

template <typename T, int BM, int BN, int WM, int WN, int EPI, int SCHED = 0, int ADDR = 0, int NSTAGE = 2>
int launch_tiled(const GemmParams& p, hipStream_t stream) {
  constexpr int NT = WM * WN * 64;
  // + rstd per tile row for the LayerNorm-folded epilogues
  constexpr int LDS = NSTAGE * (BM + BN) * 128 + (epi_is_ln(EPI) ? BM * 4 : 0);
  auto kern = gemm_nt_kernel<T, BM, BN, WM, WN, EPI, SCHED, ADDR, NSTAGE>;
  // once per instantiation and process, also when two handles are created from two threads (a handle itself is not thread-safe)
  static std::once_flag attr_once;
  static hipError_t attr_rc = hipSuccess;
  std::call_once(attr_once, [&]() {
    attr_rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
  });
  if (attr_rc != hipSuccess) return (int)attr_rc;
  const int nblk = ((p.M + BM - 1) / BM) * (p.N / BN);
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(NT), LDS, stream, p);
  return (int)hipGetLastError();
}

int gemm_num_cus();  // gemm.hip

// The patch GEMM with im2col on load (gemm.h ADDR 2): the ring tile's EPI_PATCH kernel whose A operand is gathered from the fp32
// pixels (GemmParams.pix); 16-bit types only.
GemmLaunchFn gemm_get_gather_bf16();
GemmLaunchFn gemm_get_gather_f16();
GemmLaunchFn gemm_get_gather_u8_bf16();   // ... from native uint8 HWC tiles (GemmParams.tiles), CLIP normalisation fused
GemmLaunchFn gemm_get_gather_u8_f16();

template <typename T, int EPI>
int launch_naive(const GemmParams& p, hipStream_t stream) {
  dim3 grid((p.N / 4 + 63) / 64, p.M);
  hipLaunchKernelGGL((gemm_nt_naive_kernel<T, EPI>), grid, dim3(64), 0, stream, p);
  return (int)hipGetLastError();
}

// The tile variants, each built for every dtype; numbers are what plipmi_gemm_nt(variant=...) and
// plipmi_test_force_gemm_tile take, names come from gemm.hip.
//   0  128x128, 2x2 waves, two workgroups per CU, 64-bit lane addresses (operands of 4 GiB and more; small problems)
//   1  128x128, 2x2 waves, buffer-form LDS-DMA
//   2  256x256, 4x2 waves; 16-bit engines: 16x16x32 MFMAs, hand-placed K steps, barrier in front of the tile's last groups
//   3  320x256, 2x4 waves; 16-bit engines: as 2
//   4  192x256, 2x4 waves
//   5  160x256, 2x4 waves with 3 + 2 row blocks per wave row: 240 / 248 tiles on the bs=256 residual GEMMs (256 CUs)
//   6  160x256 on a ring of three LDS stages (two K tiles of lookahead, barrier in front of the last K step's MFMAs);
//      16-bit engines: 16x16x32 MFMAs, hand-placed K steps
constexpr int kNumVariants = 7;

template <typename T>
constexpr bool gemm_variant_built(int v) { return v == -2 || (v >= 0 && v < kNumVariants); }

// table[variant][epilogue]
template <typename T>
struct GemmTable {
  template <int EPI>
  static GemmLaunchFn pick(int variant) {
    constexpr bool kH = sizeof(T) == 2;
    // LayerNorm-folded epilogues: 16-bit engines only
    constexpr bool kLn = epi_is_ln(EPI) || epi_emits_stats(EPI);
    if constexpr (kLn && !kH) {
      return nullptr;
    } else {
      switch (variant) {
        case 0: return launch_tiled<T, 128, 128, 2, 2, EPI, 0, 0>;
        case 1: return launch_tiled<T, 128, 128, 2, 2, EPI, 1, 1>;
        case 2: return launch_tiled<T, 256, 256, 4, 2, EPI, kH ? 8 : 1, 1>;
        case 3: return launch_tiled<T, 320, 256, 2, 4, EPI, kH ? 8 : 0, 1>;
        case 4: return launch_tiled<T, 192, 256, 2, 4, EPI, kH ? 6 : 1, 1>;
        case 5: return launch_tiled<T, 160, 256, 2, 4, EPI, kH ? 6 : 1, 1>;
        case 6: return launch_tiled<T, 160, 256, 2, 4, EPI, kH ? 7 : 1, 1, 3>;
        case -2: if constexpr (!kLn) return launch_naive<T, EPI>; else return nullptr;
        default: return nullptr;
      }
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
GemmLaunchFn gemm_get_bf16(int variant, int epi);
GemmLaunchFn gemm_get_f16(int variant, int epi);
bool gemm_built_f32(int variant);
bool gemm_built_bf16(int variant);
bool gemm_built_f16(int variant);

}  // namespace plipmi
