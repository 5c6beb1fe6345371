// bf16 instantiations (v_mfma_f32_32x32x16_bf16, fp32 accumulate) of the NT GEMM.
#include "gemm_inst.h"
namespace plipmi {
GemmLaunchFn gemm_get_bf16(int variant, int epi) { return GemmTable<bf16_t>::get(variant, epi); }
}  // namespace plipmi
