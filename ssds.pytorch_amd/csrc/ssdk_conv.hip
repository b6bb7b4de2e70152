// ssdk_conv.hip -- fused conv + folded-BN + activation on MFMA (placeholder until the implicit-GEMM
// kernel lands; the entry points exist so that the C-ABI is complete and fail loudly).
#include "ssdk_common.h"

extern "C" size_t ssdk_conv_workspace_bytes(int, int, int, int, int, int, int, int) { return 0; }

extern "C" int ssdk_conv_bn_act(const void*, const void*, const float*, const float*, int, int, int, int,
                                int, int, int, int, int, int, void*, void*, size_t, void*) {
  ssdk::set_error("conv_bn_act: not built yet");
  return SSDK_E_BADARG;
}
