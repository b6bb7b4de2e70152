// ssdk_debug.hip -- SSDK_LDS_POISON=1: fill the LDS of every CU with NaN patterns in front of each kernel of the
// plan executor and of the decode stage.  A kernel that reads LDS it did not write is deterministic as long as the
// previous tenant of its CU is always the same kernel -- and silently wrong next to any other work (two streams,
// another process).  With the poison such a read turns into NaNs in the output, which the parity tests catch.
// Debug aid only: one extra launch per op.
#include "ssdk_common.h"

namespace ssdk {

__global__ __launch_bounds__(256) void lds_poison_kernel(u32 words) {
  extern __shared__ u32 lds_words[];
  for (u32 i = threadIdx.x; i < words; i += 256) lds_words[i] = 0x7fc07fc0u;  // NaN as fp32, 2 x bf16 and 2 x fp16
  __syncthreads();
  if (lds_words[(threadIdx.x * 37u) % words] == 1u) lds_words[0] = 2u;  // keep the stores alive
}

bool lds_poison_enabled() {
  static const bool on = [] {
    const char* e = getenv("SSDK_LDS_POISON");
    return e && atoi(e) != 0;
  }();
  return on;
}

void lds_poison(hipStream_t stream) {
  if (!lds_poison_enabled()) return;
  static bool attr = false;
  constexpr int kBytes = 160 * 1024;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lds_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kBytes);
    attr = true;
  }
  // one workgroup owns a whole CU's LDS; 4 rounds over the 256 CUs so that every CU is visited
  hipLaunchKernelGGL(lds_poison_kernel, dim3(1024), dim3(256), kBytes, stream, (u32)(kBytes / 4));
  (void)hipGetLastError();
}

__global__ __launch_bounds__(256) void lds_probe_kernel(u32 words, unsigned* poisoned) {
  extern __shared__ u32 lds_words[];
  u32 n = 0;
  for (u32 i = threadIdx.x; i < words; i += 256) n += lds_words[i] == 0x7fc07fc0u ? 1u : 0u;  // reads, never writes
  if (n) atomicAdd(poisoned, n);
}

}  // namespace ssdk

// (debug) poisons the LDS (whatever SSDK_LDS_POISON says), then counts the poisoned words a kernel that never wrote
// its 64 KiB of LDS can see: *count > 0 proves that LDS contents survive from one kernel to the next on this stack.
extern "C" int ssdk_debug_lds_probe(unsigned* count, void* stream) {
  using namespace ssdk;
  hipStream_t st = (hipStream_t)stream;
  constexpr int kBytes = 160 * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(lds_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kBytes);
  hipLaunchKernelGGL(lds_poison_kernel, dim3(1024), dim3(256), kBytes, st, (u32)(kBytes / 4));
  int rc = check_launch("lds_poison_kernel");
  if (rc) return rc;
  hipLaunchKernelGGL(lds_probe_kernel, dim3(512), dim3(256), 64 * 1024, st, (u32)(64 * 1024 / 4), count);
  return check_launch("lds_probe_kernel");
}
