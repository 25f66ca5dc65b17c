// What do the two waves of a SIMD share?  One workgroup of 8 waves per CU (wave w and wave w + 4
// sit on the same SIMD), each wave runs one of three loops of the attention step's instruction mix
// per iteration:
//   M  16 x v_mfma_f32_32x32x16_bf16 (4 independent accumulators), nothing else
//   V  the softmax arithmetic of one 32-token step: 16 x (v_fma_f32, v_exp_f32), 8 x v_cvt_pk_bf16_f32,
//      16 x v_add_f32, in four independent chains
//   B  both in one wave: every product followed by its share of the vector work (1 fma + 1 exp + 1 add,
//      every second one a conversion)
// and reports s_memtime cycles per iteration (and the kernel's wall time) for:
//   M alone (waves 4 - 7 idle), V alone, B alone, M beside M, V beside V, B beside B (two waves per SIMD, same role),
//   M beside V (waves 0 - 3 products, waves 4 - 7 vector work: the roles of an alternating-segment attention loop).
// Build and run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/probes/coexec_probe.hip -o /tmp/coexec && /tmp/coexec
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;

__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }

template <int ROLE>  // 0 idle, 1 M, 2 V, 3 B, 4 D (V as ONE dependent chain per element: fma -> exp -> add on one sum)
__device__ __forceinline__ void run(int iters, float seed, float* sink, uint64_t* cycles) {
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = seed * (float)(i + r);
  uint4 araw = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u), braw = araw;
  asm volatile("" : "+v"(araw.x), "+v"(braw.x));
  const bf16x8_t a = __builtin_bit_cast(bf16x8_t, araw), b = __builtin_bit_cast(bf16x8_t, braw);
  float s[16], l[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < 16; ++r) s[r] = seed * 0.01f * (float)r;
  uint32_t pw[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const float c0 = seed * 1e-3f, c1 = -seed;
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (ROLE == 1) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (ROLE == 2) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float p = fexp2(fmaf(s[k], c0, c1));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(l[k & 3]) : "v"(p));
        s[k] = p;
        if (k & 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pw[k >> 1]) : "v"(s[k - 1]), "v"(p));
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (ROLE == 4) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        float t;
        asm volatile("v_fma_f32 %0, %2, %3, %4\n\tv_exp_f32 %0, %0\n\tv_add_f32 %1, %1, %0" : "=&v"(t), "+v"(l[0]) : "v"(s[k]), "v"(c0), "v"(c1));
        s[k] = t;
        if (k & 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pw[k >> 1]) : "v"(s[k - 1]), "v"(t));
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (ROLE == 3) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k & 3], 0, 0, 0);
        const float p = fexp2(fmaf(s[k], c0, c1));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(l[k & 3]) : "v"(p));
        s[k] = p;
        if (k & 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pw[k >> 1]) : "v"(s[k - 1]), "v"(p));
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float out = l[0] + l[1] + l[2] + l[3];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) out += acc[i][r];
#pragma unroll
  for (int r = 0; r < 16; ++r) out += s[r];
#pragma unroll
  for (int i = 0; i < 8; ++i) out += (float)pw[i];
  if (out == 12345.678f) *sink = out;  // keeps everything alive
  if ((threadIdx.x & 63) == 0) *cycles = t1 - t0;
}

__global__ __launch_bounds__(512) void probe(int role_lo, int role_hi, int iters, float seed, float* sink, uint64_t* cycles) {
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? role_lo : role_hi;
  uint64_t* c = cycles + (blockIdx.x * 8 + wave);
  switch (__builtin_amdgcn_readfirstlane(role)) {
    case 1: run<1>(iters, seed, sink, c); break;
    case 2: run<2>(iters, seed, sink, c); break;
    case 3: run<3>(iters, seed, sink, c); break;
    case 4: run<4>(iters, seed, sink, c); break;
    default: if ((threadIdx.x & 63) == 0) *c = 0; break;
  }
  if (role_lo < 0 && (threadIdx.x & 63) == 0) {  // which SIMD is this wave on?
    uint32_t hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    *c = hw;
  }
}

int main() {
  const int n_wg = 256, iters = 2000;
  uint64_t* cyc;
  float* sink;
  hipMalloc(&cyc, n_wg * 8 * sizeof(uint64_t));
  hipMalloc(&sink, 4);
  uint64_t* host = (uint64_t*)malloc(n_wg * 8 * sizeof(uint64_t));
  hipLaunchKernelGGL(probe, dim3(n_wg), dim3(512), 0, 0, -1, -1, 1, 1.0f, sink, cyc);
  hipDeviceSynchronize();
  hipMemcpy(host, cyc, n_wg * 8 * sizeof(uint64_t), hipMemcpyDeviceToHost);
  for (int g = 0; g < 2; ++g) {
    printf("workgroup %d: SIMD of waves 0..7:", g);
    for (int w = 0; w < 8; ++w) printf(" %d", (int)((host[g * 8 + w] >> 4) & 3));
    printf("   (CU %d)\n", (int)((host[g * 8] >> 8) & 15));
  }
  const char* names[] = {"idle", "M (16 products)", "V (softmax arithmetic)", "B (both in one wave)", "D (V, dependent chains)"};
  const int cases[][2] = {{1, 0}, {2, 0}, {3, 0}, {4, 0}, {1, 1}, {2, 2}, {3, 3}, {1, 2}, {2, 1}, {1, 4}, {4, 1}, {3, 2}};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (auto& cs : cases) {
    hipLaunchKernelGGL(probe, dim3(n_wg), dim3(512), 0, 0, cs[0], cs[1], iters, 1.0f, sink, cyc);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe, dim3(n_wg), dim3(512), 0, 0, cs[0], cs[1], iters, 1.0f, sink, cyc);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(host, cyc, n_wg * 8 * sizeof(uint64_t), hipMemcpyDeviceToHost);
    double lo = 0, hi = 0;
    for (int g = 0; g < n_wg; ++g)
      for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi) += (double)host[g * 8 + w];
    lo /= n_wg * 4.0 * iters;
    hi /= n_wg * 4.0 * iters;
    printf("waves 0-3: %-24s waves 4-7: %-24s cycles per iteration: %7.1f | %7.1f   kernel %.3f ms = %.1f ns per iteration\n", names[cs[0]], names[cs[1]], lo, hi, ms, ms * 1e6 / iters);
  }
  return 0;
}
