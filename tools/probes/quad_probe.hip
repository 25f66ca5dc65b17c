// Checks the four-lane-group max / sum helpers of the matrix-core decode kernel (diagnostic).
#include <hip/hip_runtime.h>
#include <stdio.h>
// half exchanges written in assembly: both registers are read-write operands
// (s_nop 1 = the two wait states a VALU write needs before v_permlane*_swap reads it)
__device__ __forceinline__ void swap32(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap16(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float quad_group_max(float x) {
  float a = x, b = x;
  swap32(a, b);
  a = fmaxf(a, b);
  b = a;
  swap16(a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float quad_group_sum(float x) {
  float a = x, b = x;
  swap32(a, b);
  a = a + b;
  b = a;
  swap16(a, b);
  return a + b;
}
__global__ void k(const float* in, float* out) {
  float x = in[threadIdx.x];
  out[threadIdx.x] = quad_group_max(x);
  out[64 + threadIdx.x] = quad_group_sum(x);
}
int main() {
  float h[64], r[128]; float *d, *o;
  for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 64) + 100.f * (i >> 4);
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) {
    float mx = fmaxf(fmaxf(h[i & 15], h[(i & 15) + 16]), fmaxf(h[(i & 15) + 32], h[(i & 15) + 48]));
    float sm = h[i & 15] + h[(i & 15) + 16] + h[(i & 15) + 32] + h[(i & 15) + 48];
    if (r[i] != mx || r[64 + i] != sm) { if (bad < 8) printf("lane %d: max %g (want %g) sum %g (want %g)\n", i, r[i], mx, r[64 + i], sm); ++bad; }
  }
  printf("quad probe: %d bad lanes\n", bad);
  return 0;
}
