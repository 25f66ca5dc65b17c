// What does one gap between two 32x32x16 matrix products cost with a given set of filler instructions,
// one wave per SIMD?  Tuning aid, not part of the library: prints cycles per product for a list of gap contents.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-honor-nans [-mllvm -amdgpu-mfma-vgpr-form] tools/ubench_gap.hip -o ubench_gap
// (with the option the products' accumulators live in vector registers; run the binary on the GPU box)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define SB() __builtin_amdgcn_sched_barrier(0)
#define PIN(x) asm volatile("" : "+v"(x))

__device__ __forceinline__ f32x16 mm(const uint4& a, const uint4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gap_kernel(float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(1024))) char smem[65536];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((float*)smem)[i] = 0.001f * (i & 255);
  __syncthreads();
  uint4 a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = make_uint4(0x3c003c00u + lane + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
    b[i] = make_uint4(0x3c003c00u + 2 * lane + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
  }
  f32x16 acc[4], sn, sc;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  for (int r = 0; r < 16; ++r) { sn[r] = 0.f; sc[r] = 0.01f * (lane + r); }
  float p[16], e[16], l = 0.f, mx = 0.f;
  unsigned pw[8];
  for (int r = 0; r < 16; ++r) { p[r] = 0.f; e[r] = 0.f; }
  for (int r = 0; r < 8; ++r) pw[r] = 0;
  const float scale = 0.127f, negm = -0.3f;
  const char* vbase = smem + lane * 16;  // (conflict-free: consecutive lanes, consecutive 16 bytes)
  const char* tbase = smem + lane * 8;
  uint4 vf[4] = {};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    SB();
    if constexpr (MODE == 0) {  // chained products, bare
#pragma unroll
      for (int k = 0; k < 8; ++k) { sn = mm(a[k], b[k], sn); SB(); }
    } else if constexpr (MODE == 1) {  // chained + {fma, exp, fma, exp, cvt}: each exp reads the fma ahead of it
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        sn = mm(a[k], b[k], sn);
        p[2 * k] = __builtin_amdgcn_exp2f(fmaf(sc[2 * k], scale, negm)); PIN(p[2 * k]);
        p[2 * k + 1] = __builtin_amdgcn_exp2f(fmaf(sc[2 * k + 1], scale, negm)); PIN(p[2 * k + 1]);
        if (k >= 1) { pw[k - 1] = pack2(p[2 * k - 2], p[2 * k - 1]); PIN(pw[k - 1]); }
        SB();
      }
    } else if constexpr (MODE == 2) {  // chained + {exp, exp, fma, fma, cvt}: arguments made a gap ahead
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        sn = mm(a[k], b[k], sn);
        p[2 * k] = __builtin_amdgcn_exp2f(e[2 * k]); PIN(p[2 * k]);
        p[2 * k + 1] = __builtin_amdgcn_exp2f(e[2 * k + 1]); PIN(p[2 * k + 1]);
        e[(2 * k + 2) & 15] = fmaf(sc[(2 * k + 2) & 15], scale, negm); PIN(e[(2 * k + 2) & 15]);
        e[(2 * k + 3) & 15] = fmaf(sc[(2 * k + 3) & 15], scale, negm); PIN(e[(2 * k + 3) & 15]);
        if (k >= 1) { pw[k - 1] = pack2(p[2 * k - 2], p[2 * k - 1]); PIN(pw[k - 1]); }
        SB();
      }
    } else if constexpr (MODE == 3) {  // chained + {fma, exp, cvt}: one exponential per gap
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        sn = mm(a[k], b[k], sn);
        p[k] = __builtin_amdgcn_exp2f(e[k]); PIN(p[k]);
        e[(k + 1) & 15] = fmaf(sc[(k + 1) & 15], scale, negm); PIN(e[(k + 1) & 15]);
        if (k & 1) { pw[k >> 1] = pack2(p[k - 1], p[k]); PIN(pw[k >> 1]); }
        SB();
      }
    } else if constexpr (MODE == 4) {  // four accumulators + {2 add}
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        acc[k & 3] = mm(a[k], b[k], acc[k & 3]);
        asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(l) : "v"(p[2 * k]), "v"(p[2 * k + 1]));
        SB();
      }
    } else if constexpr (MODE == 5 || MODE == 6) {  // four accumulators + {2 add, 2 transposing reads used 3 (5) or 7 (6) gaps later}
      constexpr int AH = MODE == 5 ? 3 : 7;
      uint4 vq[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) vq[k] = vf[k & 3];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        acc[k & 3] = mm(a[k], vq[k], acc[k & 3]);
        if (k + AH < 8 || MODE == 6) {
          const int kk = (k + AH) & 7;
          const s16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tbase + kk * 2048));
          const s16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tbase + kk * 2048 + 1024));
          const uint2 va2 = __builtin_bit_cast(uint2, va), vb2 = __builtin_bit_cast(uint2, vb);
          if (MODE == 5) vq[kk] = make_uint4(va2.x, va2.y, vb2.x, vb2.y);
          else vf[kk & 3] = make_uint4(va2.x, va2.y, vb2.x, vb2.y);
        }
        asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(l) : "v"(p[2 * k]), "v"(p[2 * k + 1]));
        SB();
      }
      if (MODE == 5) {
#pragma unroll
        for (int k = 0; k < 4; ++k) vf[k] = vq[k + 4];
      }
    } else if constexpr (MODE == 7) {  // chained + {3 max3} reading the scores of the block before
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        sn = mm(a[k], b[k], sn);
        mx = fmaxf(fmaxf(sc[2 * k], sc[2 * k + 1]), mx); PIN(mx);
        mx = fmaxf(fmaxf(sc[(2 * k + 5) & 15], sc[(2 * k + 7) & 15]), mx); PIN(mx);
        SB();
      }
    } else if constexpr (MODE == 8) {  // chained + one plain 16-byte LDS read per gap, used a block later
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        sn = mm(a[k], b[k], sn);
        vf[k & 3] = *(const uint4*)(vbase + k * 1024);
        SB();
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { PIN(vf[k].x); l += __builtin_bit_cast(float, vf[k].x); }
    } else if constexpr (MODE == 9) {  // mode 1 twice, the exponentials reading the other round's products
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        sn = mm(a[k], b[k], k == 0 ? f32x16{} : sn);
        p[2 * k] = __builtin_amdgcn_exp2f(fmaf(sc[2 * k], scale, negm)); PIN(p[2 * k]);
        p[2 * k + 1] = __builtin_amdgcn_exp2f(fmaf(sc[2 * k + 1], scale, negm)); PIN(p[2 * k + 1]);
        if (k >= 1) { pw[k - 1] = pack2(p[2 * k - 2], p[2 * k - 1]); PIN(pw[k - 1]); }
        SB();
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        sc = mm(a[k], b[k], k == 0 ? f32x16{} : sc);
        p[2 * k] = __builtin_amdgcn_exp2f(fmaf(sn[2 * k], scale, negm)); PIN(p[2 * k]);
        p[2 * k + 1] = __builtin_amdgcn_exp2f(fmaf(sn[2 * k + 1], scale, negm)); PIN(p[2 * k + 1]);
        if (k >= 1) { pw[k - 1] = pack2(p[2 * k - 2], p[2 * k - 1]); PIN(pw[k - 1]); }
        SB();
      }
    } else if constexpr (MODE == 10) {  // a whole half-step: score block (mode 1) then PV block with the packed words and LDS reads
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        sn = mm(a[k], b[k], k == 0 ? f32x16{} : sn);
        p[2 * k] = __builtin_amdgcn_exp2f(fmaf(sc[2 * k], scale, negm)); PIN(p[2 * k]);
        p[2 * k + 1] = __builtin_amdgcn_exp2f(fmaf(sc[2 * k + 1], scale, negm)); PIN(p[2 * k + 1]);
        if (k >= 1) { pw[k - 1] = pack2(p[2 * k - 2], p[2 * k - 1]); PIN(pw[k - 1]); }
        if (k >= 5) {
          const int kk = k - 5;
          const s16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tbase + kk * 2048));
          const s16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tbase + kk * 2048 + 1024));
          const uint2 va2 = __builtin_bit_cast(uint2, va), vb2 = __builtin_bit_cast(uint2, vb);
          vf[kk] = make_uint4(va2.x, va2.y, vb2.x, vb2.y);
        }
        SB();
      }
      pw[7] = pack2(p[14], p[15]);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        acc[k & 3] = mm(make_uint4(pw[4 * (k >> 2)], pw[4 * (k >> 2) + 1], pw[4 * (k >> 2) + 2], pw[4 * (k >> 2) + 3]), vf[k & 3], acc[k & 3]);
        if (k + 3 < 8) {
          const int kk = k + 3;
          const s16x4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tbase + kk * 2048));
          const s16x4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(tbase + kk * 2048 + 1024));
          const uint2 va2 = __builtin_bit_cast(uint2, va), vb2 = __builtin_bit_cast(uint2, vb);
          vf[kk & 3] = make_uint4(va2.x, va2.y, vb2.x, vb2.y);
        }
        a[k] = *(const uint4*)(vbase + k * 1024);  // (K of the next score block)
        asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(l) : "v"(p[2 * k]), "v"(p[2 * k + 1]));
        SB();
      }
      { f32x16 t = sc; sc = sn; sn = t; }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = l + mx;
  for (int r = 0; r < 16; ++r) s += sn[r] + acc[0][r] + acc[1][r] + acc[2][r] + acc[3][r] + p[r] + e[r] + sc[r];
  for (int r = 0; r < 8; ++r) s += (float)pw[r];
  for (int r = 0; r < 4; ++r) s += __builtin_bit_cast(float, vf[r].x);
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE> void run(const char* what, float* out, unsigned long long* cyc, int iters, double per_iter = 8.0) {
  const int n_wg = 256;
  hipLaunchKernelGGL(gap_kernel<MODE>, dim3(n_wg), dim3(256), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL(gap_kernel<MODE>, dim3(n_wg), dim3(256), 0, 0, out, cyc, iters);
  if (hipDeviceSynchronize() != hipSuccess) { printf("mode %d failed\n", MODE); exit(1); }
  std::vector<unsigned long long> h(n_wg * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (auto v : h) sum += (double)v;
  printf("mode %d  %6.1f cycles per product   %s\n", MODE, sum / h.size() / iters / per_iter, what);
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4);
  hipMalloc(&cyc, 256 * 4 * 8);
  const int iters = 2000;
  run<0>("chained, bare", out, cyc, iters);
  run<1>("chained + fma,exp,fma,exp,cvt (exp reads the fma ahead of it)", out, cyc, iters);
  run<2>("chained + exp,exp,fma,fma,cvt (arguments a gap ahead)", out, cyc, iters);
  run<3>("chained + exp,fma,cvt/2", out, cyc, iters);
  run<4>("4 accumulators + 2 add", out, cyc, iters);
  run<5>("4 accumulators + 2 add + 2 ds_read_tr used 3 gaps later", out, cyc, iters);
  run<6>("4 accumulators + 2 add + 2 ds_read_tr used 7 gaps later", out, cyc, iters);
  run<7>("chained + 2 max3", out, cyc, iters);
  run<8>("chained + ds_read_b128", out, cyc, iters);
  run<9>("mode 1 twice, exponentials read the other round's products", out, cyc, iters, 16.0);
  run<10>("half-step: score block + PV block with V and K reads (16 products; + 32 register moves per round)", out, cyc, iters, 16.0);
  return 0;
}
