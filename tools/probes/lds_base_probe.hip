// What HW_REG_LDS_ALLOC says for the two 68 KiB workgroups a CU holds at once (raw register per workgroup).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned* o) {
  __shared__ char s[69632];
  s[threadIdx.x] = (char)threadIdx.x;
  unsigned r;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_LDS_ALLOC)" : "=s"(r));
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 5000) {}  // 50 us: every slot of the chip is taken meanwhile
  __syncthreads();
  if (threadIdx.x == 0) o[blockIdx.x] = r + (s[17] == 99 ? 1u << 31 : 0u);
}
int main() {
  const int n = 2048;
  unsigned* d;
  if (hipMalloc(&d, n * 4) != hipSuccess) return 1;
  hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d);
  std::vector<unsigned> h(n);
  if (hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  std::map<unsigned, int> hist;
  for (unsigned v : h) hist[v]++;
  for (auto& kv : hist) printf("LDS_ALLOC 0x%08x (base field 0x%02x, size field 0x%03x): %d workgroups\n", kv.first, kv.first & 0xff, (kv.first >> 12) & 0x1ff, kv.second);
  return 0;
}
