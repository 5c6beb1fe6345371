// gemm_fp8.hip -- EXPERIMENTAL fp8 (OCP e4m3fn) instantiations of gemm_nt_kernel: the `configs[4]` headroom probe.
// A [M,K] and W [N,K] are fp8, the output is bf16 (bias / bias+QuickGELU epilogues), accumulation fp32 through
// v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales.  Reached only through the plipmi_gemm_nt test hook
// (dtype 2); the engine does not use it yet.
#include "gemm_inst.h"

namespace plipmi {

template <int EPI>
static GemmLaunchFn pick_fp8(int variant) {
  switch (variant) {
    case 0: return launch_tiled<fp8_t, 256, 256, 4, 2, EPI, true, 6, 0, 2, 1>;
    case 1: return launch_tiled<fp8_t, 192, 256, 2, 4, EPI, true, 6, 0, 2, 1>;
    case 2: return launch_tiled<fp8_t, 320, 256, 2, 4, EPI, true, 6, 0, 2, 1>;
    case 3: return launch_tiled<fp8_t, 256, 256, 4, 2, EPI, true, 3, 0, 2, 1>;
    default: return nullptr;
  }
}

GemmLaunchFn gemm_get_fp8(int variant, int epi) {
  switch (epi) {
    case EPI_BIAS: return pick_fp8<EPI_BIAS>(variant);
    case EPI_BIAS_QGELU: return pick_fp8<EPI_BIAS_QGELU>(variant);
    default: return nullptr;
  }
}

}  // namespace plipmi
