// Prints what v_permlane32_swap / v_permlane16_swap do to lane ids (diagnostic).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
  unsigned x = threadIdx.x;
  auto a = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  out[threadIdx.x] = a[0];
  out[64 + threadIdx.x] = a[1];
  auto b = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  out[128 + threadIdx.x] = b[0];
  out[192 + threadIdx.x] = b[1];
  unsigned y = 1000 + threadIdx.x;
  auto c = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  out[256 + threadIdx.x] = c[0];
  out[320 + threadIdx.x] = c[1];
}
int main() {
  unsigned* d; unsigned h[384];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"p32 r0", "p32 r1", "p16 r0", "p16 r1", "p32xy r0", "p32xy r1"};
  for (int r = 0; r < 6; ++r) { printf("%s:", names[r]); for (int i = 0; i < 64; ++i) printf(" %u", h[r * 64 + i]); printf("\n"); }
  return 0;
}
