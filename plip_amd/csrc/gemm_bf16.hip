// bf16 instantiations (v_mfma_f32_16x16x32_bf16 / v_mfma_f32_32x32x16_bf16, fp32 accumulate) of the NT GEMM.
#include "gemm_inst.h"
namespace plipmi {
GemmLaunchFn gemm_get_bf16(int variant, int epi) { return GemmTable<bf16_t>::get(variant, epi); }
bool gemm_built_bf16(int variant) { return gemm_variant_built<bf16_t>(variant); }
GemmLaunchFn gemm_get_gather_bf16() { return launch_tiled<bf16_t, 160, 256, 2, 4, EPI_PATCH, 7, 2, 3>; }
GemmLaunchFn gemm_get_gather_u8_bf16() { return launch_tiled<bf16_t, 160, 256, 2, 4, EPI_PATCH, 7, 3, 3>; }
}  // namespace plipmi
