// v_permlane16_swap_b32 on gfx950: which 16-lane rows of the two operands trade places?
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/swap16.hip -o tools/micro/swap16 ; prints the row a lane's two results came from
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
  const unsigned l = threadIdx.x;
  unsigned a = 0x100u + l, b = 0x200u + l;  // a: tag 1, b: tag 2, low byte = source lane
  v2u r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[l] = r.x;
  out[64 + l] = r.y;
}
int main() {
  unsigned* d;
  unsigned h[128];
  hipMalloc(&d, sizeof h);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int r = 0; r < 4; ++r)
    printf("row %d: first result = operand %u row %u (lane %u -> %u) | second result = operand %u row %u\n", r, h[16 * r] >> 8,
           (h[16 * r] & 0xff) >> 4, 16 * r, h[16 * r] & 0xff, h[64 + 16 * r] >> 8, (h[64 + 16 * r] & 0xff) >> 4);
  return 0;
}
