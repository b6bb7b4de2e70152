// Micro-benchmark: how fast can ONE workgroup per CU (4 waves) stream an L2-resident weight matrix into registers,
// (a) fragment-shaped: a wave instruction = 16 rows x 64 bytes (row stride = K*2 bytes), as conv_smallmap_kernel reads W
// (b) packed: a wave instruction = 1 KiB contiguous (fragment-major pre-packed image)
// Prints bytes per clock per CU for both.  Build: hipcc --offload-arch=gfx950 -O3 -o wstream wstream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int INFLIGHT>
__global__ __launch_bounds__(256) void stream(const unsigned short* w, int K, int steps, unsigned* sink, int rows_per_wg) {
  const unsigned tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
  const unsigned short* base;
  if (MODE == 0) base = w + (size_t)((blockIdx.x % rows_per_wg) * 64 + wave * 16 + fr) * K + fg * 8;  // + st*32
  else base = w + (size_t)((blockIdx.x % rows_per_wg) * 4 + wave) * 16 * K + lane * 8;               // + st*512
  u32x4 r[INFLIGHT];
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < INFLIGHT; ++i) r[i] = *reinterpret_cast<const u32x4*>(base + (size_t)i * (MODE == 0 ? 32 : 512));
  for (int st = INFLIGHT; st < steps; st += INFLIGHT) {
#pragma unroll
    for (int i = 0; i < INFLIGHT; ++i) {
      acc ^= r[i];
      r[i] = *reinterpret_cast<const u32x4*>(base + (size_t)(st + i) * (MODE == 0 ? 32 : 512));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int i = 0; i < INFLIGHT; ++i) acc ^= r[i];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}
template <int MODE, int INFLIGHT>
static void run(const unsigned short* w, int K, unsigned* sink, int wgs, int rows_per_wg, const char* name) {
  const int steps = K / 32;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((stream<MODE, INFLIGHT>), dim3(wgs), dim3(256), 0, 0, w, K, steps, sink, rows_per_wg);
  hipEventRecord(e0);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((stream<MODE, INFLIGHT>), dim3(wgs), dim3(256), 0, 0, w, K, steps, sink, rows_per_wg);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / reps, bytes_wg = 64.0 * K * 2;
  printf("%-28s wgs=%4d K=%6d inflight=%2d  %7.1f us  %6.1f B/clk/CU (2.4 GHz, %d WG/CU)  total %.2f TB/s\n", name, wgs, K, INFLIGHT, us,
         bytes_wg * ((wgs + 255) / 256) / (us * 2400.0), (wgs + 255) / 256, bytes_wg * wgs / us / 1e6);
}
int main() {
  const int K = 4608, ROWS = 512;  // 512 x 4608 bf16 = 4.7 MB (the 8x8 head level)
  unsigned short* w; unsigned* sink;
  hipMalloc(&w, (size_t)ROWS * K * 2 + 65536); hipMemset(w, 1, (size_t)ROWS * K * 2 + 65536);
  hipMalloc(&sink, 4);
  for (int wgs : {256, 512, 1024}) {
    run<0, 18>(w, K, sink, wgs, 8, "fragment 16x64B");
    run<1, 18>(w, K, sink, wgs, 8, "packed 1KiB");
    run<0, 8>(w, K, sink, wgs, 8, "fragment 16x64B");
    run<1, 8>(w, K, sink, wgs, 8, "packed 1KiB");
    run<0, 32>(w, K, sink, wgs, 8, "fragment 16x64B");
    run<1, 32>(w, K, sink, wgs, 8, "packed 1KiB");
  }
  return 0;
}
