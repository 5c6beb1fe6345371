// attention_mfma.hip -- bf16 MFMA attention (placeholder until the kernel lands; the engine uses the
// exact VALU kernel and this entry reports "not supported" so nothing can silently fall back).
#include "kernels.h"
namespace plipmi {
hipError_t launch_attention_mfma(const void*, void*, int, int, int, int, const int64_t*, hipStream_t) {
  return hipErrorNotSupported;
}
}  // namespace plipmi
