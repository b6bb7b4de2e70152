// Stand-alone reproducer for the claim of DESIGN.md (round 1, section 8.4): "a v_cmp ... v_cndmask select read a
// stale quarter of VCC (lanes 48-63) when three HIP queues competed for CUs held by persistent workgroups".
//   hipcc --offload-arch=gfx950 -O3 tools/repro/vcc_select_repro.hip -o /tmp/vcc_repro && /tmp/vcc_repro [launches]
// Two queues are kept full of long-running "filler" workgroups (one per CU each, VALU + LDS busy loops); on a third
// queue the victim kernel runs the exact instruction pair of the round-1 clamp (v_cmp_ngt_f32 vcc, 0, v ; s_nop 0 ;
// v_cndmask_b32 v, 0, v, vcc) on known data, thousands of launches, every lane of every launch checked on the host.
// Prints the number of wrong lanes (expected: 0 unless the hardware / driver fault is real).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

__global__ __launch_bounds__(512) void filler(float* sink, int iters) {
  __shared__ float lds[4096];
  float a = threadIdx.x * 0.001f, b = 1.0001f;
  for (int i = 0; i < iters; ++i) {
    a = fmaf(a, b, 0.5f);
    lds[(threadIdx.x * 7 + i) & 4095] = a;
    a += lds[(threadIdx.x * 13 + i * 3) & 4095] * 1e-9f;
    if ((i & 255) == 0) __syncthreads();
  }
  if (a == 123.456f) sink[0] = a;  // never true: keeps the loop alive
}

// NaN-propagating clamp at zero, the round-1 code path: keep x unless NOT(0 > x) fails, i.e. result = (0 > x) ? 0 : x
// with NaN kept -- written as the literal instruction pair the compiler had emitted
__global__ __launch_bounds__(256) void victim(const float* in, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = in[i];
  asm volatile("v_cmp_ngt_f32 vcc, 0, %0\n\ts_nop 0\n\tv_cndmask_b32 %0, 0, %0, vcc" : "+v"(v) : : "vcc");
  out[i] = v;
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 20000;
  const int n = 64 * 1024;  // 256 workgroups of 256 lanes: one per CU
  std::vector<float> h(n), want(n), got(n);
  for (int i = 0; i < n; ++i) {
    const int k = i % 7;
    h[i] = k == 0 ? -1.5f - i : (k == 1 ? NAN : (k == 2 ? 0.0f : (k == 3 ? -0.0f : 0.25f * i + 1.0f)));
    want[i] = (0.0f > h[i]) ? 0.0f : h[i];  // (NaN and +-0 are kept: NOT(0 > x))
  }
  float *din, *dout, *sink;
  CK(hipMalloc((void**)&din, n * 4));
  CK(hipMalloc((void**)&dout, n * 4));
  CK(hipMalloc((void**)&sink, 16));
  CK(hipMemcpy(din, h.data(), n * 4, hipMemcpyHostToDevice));
  hipStream_t f1, f2, vs;
  CK(hipStreamCreateWithFlags(&f1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&f2, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&vs, hipStreamNonBlocking));
  long long wrong = 0, wrong_hi = 0;
  const int per_round = 50;
  for (int done = 0; done < launches; done += per_round) {
    // keep both filler queues busy for the whole round (one 512-thread workgroup per CU and queue, ~0.3 ms each)
    for (int k = 0; k < 6; ++k) {
      hipLaunchKernelGGL(filler, dim3(256), dim3(512), 0, f1, sink, 40000);
      hipLaunchKernelGGL(filler, dim3(256), dim3(512), 0, f2, sink, 40000);
    }
    for (int k = 0; k < per_round; ++k) {
      CK(hipMemsetAsync(dout, 0xff, n * 4, vs));
      hipLaunchKernelGGL(victim, dim3(n / 256), dim3(256), 0, vs, din, dout, n);
      CK(hipMemcpyAsync(got.data(), dout, n * 4, hipMemcpyDeviceToHost, vs));
      CK(hipStreamSynchronize(vs));
      for (int i = 0; i < n; ++i) {
        const bool same = (std::isnan(want[i]) && std::isnan(got[i])) || (want[i] == got[i] && std::signbit(want[i]) == std::signbit(got[i]));
        if (!same) {
          ++wrong;
          if ((i & 63) >= 48) ++wrong_hi;
        }
      }
    }
    CK(hipDeviceSynchronize());
  }
  printf("{\"victim_launches\": %d, \"lanes_checked\": %lld, \"wrong_lanes\": %lld, \"wrong_in_lanes_48_63\": %lld}\n", launches,
         (long long)launches * n, wrong, wrong_hi);
  return wrong ? 1 : 0;
}
