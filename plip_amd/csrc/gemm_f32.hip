// fp32 instantiations (v_mfma_f32_32x32x2_f32, exact fp32) of the NT GEMM.
#include "gemm_inst.h"
namespace plipmi {
GemmLaunchFn gemm_get_f32(int variant, int epi) { return GemmTable<float>::get(variant, epi); }
bool gemm_built_f32(int variant) { return gemm_variant_built<float>(variant); }
}  // namespace plipmi
