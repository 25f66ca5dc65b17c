// Matrix-core paged decode attention for gfx950 (bf16 / f16).
#include "bsgl_device.h"
#include "bsgl_launch.h"

namespace bsgl {

bool decode_mfma_supported(const DecodeArgs& a, const char** why) {
  (void)a;
  *why = "not built yet";
  return false;
}

hipError_t launch_decode_mfma(const DecodeArgs& a, hipStream_t stream) {
  (void)a;
  (void)stream;
  return hipErrorNotSupported;
}

}  // namespace bsgl
