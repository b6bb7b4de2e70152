// ssdk_api.cpp -- host-only part of libssdk.so: error text, device facts, anchor generation.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "ssdk_common.h"

namespace ssdk {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static thread_local const char* g_last_kernel = "";

int check_launch(const char* what) {
  g_last_kernel = what;
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return SSDK_E_LAUNCH;
  }
  return SSDK_OK;
}

}  // namespace ssdk

extern "C" int ssdk_version(void) { return SSDK_VERSION; }
extern "C" size_t ssdk_struct_size(int which) {
  switch (which) {
    case SSDK_SIZEOF_LEVEL: return sizeof(ssdk_level);
    case SSDK_SIZEOF_CONV_DESC: return sizeof(ssdk_conv_desc);
    case SSDK_SIZEOF_MBCONV_DESC: return sizeof(ssdk_mbconv_desc);
    case SSDK_SIZEOF_FUSE_DESC: return sizeof(ssdk_fuse_desc);
    case SSDK_SIZEOF_STEM_DESC: return sizeof(ssdk_stem_desc);
    case SSDK_SIZEOF_POOL_DESC: return sizeof(ssdk_pool_desc);
    case SSDK_SIZEOF_XPAIR_DESC: return sizeof(ssdk_xpair_desc);
    case SSDK_SIZEOF_OP: return sizeof(ssdk_op);
    default: return 0;
  }
}
extern "C" int ssdk_abi_check(int header_version, size_t sizeof_op) {
  if (header_version / 10 != SSDK_VERSION / 10 || sizeof_op != sizeof(ssdk_op)) {
    ssdk::set_error("ssdk_abi_check: caller was compiled for ABI %d with sizeof(ssdk_op) = %zu, this library is ABI %d with %zu",
                    header_version, sizeof_op, SSDK_VERSION, sizeof(ssdk_op));
    return SSDK_E_BADARG;
  }
  return SSDK_OK;
}
extern "C" const char* ssdk_last_error(void) { return ssdk::g_err; }
extern "C" const char* ssdk_last_kernel(void) { return ssdk::g_last_kernel; }

extern "C" int ssdk_device_info(int* cu_count, int* clock_khz, size_t* hbm_bytes, char* arch, int arch_len) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    ssdk::set_error("no HIP device");
    (void)hipGetLastError();
    return SSDK_E_NODEVICE;
  }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    ssdk::set_error("hipGetDeviceProperties failed");
    (void)hipGetLastError();
    return SSDK_E_NODEVICE;
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (clock_khz) *clock_khz = prop.clockRate;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, (size_t)arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  return SSDK_OK;
}

// box.py:46-58.  torch.round is round-half-to-even == nearbyintf in the default rounding mode; every
// operation is a separately rounded fp32 operation like the tensor ops of the reference.
extern "C" int ssdk_generate_anchors(int stride, const float* ratios, int nr, const float* scales, int ns,
                                     float* out) {
  if (!ratios || !scales || !out || nr < 1 || ns < 1 || stride < 1) {
    ssdk::set_error("generate_anchors: bad argument (stride=%d nr=%d ns=%d)", stride, nr, ns);
    return SSDK_E_BADARG;
  }
  const volatile float wh = (float)stride;
  int k = 0;
  for (int s = 0; s < ns; ++s) {      // scale-major  (box.py:49-50)
    for (int r = 0; r < nr; ++r) {    // ratio-minor  (box.py:51)
      const volatile float ratio = ratios[r];
      const volatile float scale = scales[s];
      volatile float area = wh * wh;
      volatile float q = area / ratio;
      volatile float ws = nearbyintf(sqrtf(q));      // box.py:54
      volatile float wr = ws * ratio;
      volatile float hs = nearbyintf(wr);            // box.py:55
      volatile float dws = ws * scale, dhs = hs * scale;
      volatile float x1 = wh - dws, y1 = wh - dhs;   // box.py:56
      volatile float x2 = wh + dws, y2 = wh + dhs;   // box.py:57
      volatile float hx2 = 0.5f * x2, hy2 = 0.5f * y2;
      out[k * 4 + 0] = 0.5f * x1;
      out[k * 4 + 1] = 0.5f * y1;
      out[k * 4 + 2] = hx2 - 1.0f;
      out[k * 4 + 3] = hy2 - 1.0f;
      ++k;
    }
  }
  return SSDK_OK;
}
