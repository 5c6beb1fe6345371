// f16 instantiations (v_mfma_f32_16x16x32_f16 / v_mfma_f32_32x32x16_f16, fp32 accumulate) of the NT GEMM.
#include "gemm_inst.h"
namespace plipmi {
GemmLaunchFn gemm_get_f16(int variant, int epi) { return GemmTable<f16_t>::get(variant, epi); }
bool gemm_built_f16(int variant) { return gemm_variant_built<f16_t>(variant); }
GemmLaunchFn gemm_get_gather_f16() { return launch_tiled<f16_t, 160, 256, 2, 4, EPI_PATCH, 7, 2, 3>; }
GemmLaunchFn gemm_get_gather_u8_f16() { return launch_tiled<f16_t, 160, 256, 2, 4, EPI_PATCH, 7, 3, 3>; }
}  // namespace plipmi
