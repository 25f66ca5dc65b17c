// Device-side helpers shared by the gfx950 attention kernels: element traits
// for the three KV dtypes, 16-byte row pieces, DPP row reductions and the
// (max, sum, value) merge of online-softmax partial states.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace bsgl {

struct bf16_tag {};
struct f16_tag {};
struct f32_tag {};

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2_t;

__device__ __forceinline__ float as_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t as_u32(float f) { return __builtin_bit_cast(uint32_t, f); }

// A lane always moves one 16-byte piece of a row: VEC elements of type T.
template <typename Tag> struct Elem;

template <> struct Elem<bf16_tag> {
  static constexpr int BYTES = 2;
  static constexpr int VEC = 8;
  __device__ static __forceinline__ void unpack(const uint4& r, float (&f)[8]) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = as_f32(w[i] << 16);
      f[2 * i + 1] = as_f32(w[i] & 0xffff0000u);
    }
  }
  // acc += sum_i a[i] * b[i] over the 8 packed elements, fp32 accumulation
  __device__ static __forceinline__ float dot(const uint4& a, const uint4& b, float acc) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.x), __builtin_bit_cast(bf16x2_t, b.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.y), __builtin_bit_cast(bf16x2_t, b.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.z), __builtin_bit_cast(bf16x2_t, b.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a.w), __builtin_bit_cast(bf16x2_t, b.w), acc, false);
    return acc;
  }
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) {
    // round-to-nearest-even, NaN stays NaN; one v_cvt_pk_bf16_f32
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    const f32x2_t f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2_t));
  }
  __device__ static __forceinline__ uint4 pack(const float (&f)[8]) {
    return make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
  }
};

template <> struct Elem<f16_tag> {
  static constexpr int BYTES = 2;
  static constexpr int VEC = 8;
  __device__ static __forceinline__ void unpack(const uint4& r, float (&f)[8]) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f16x2_t h = __builtin_bit_cast(f16x2_t, w[i]);
      f[2 * i] = (float)h.x;
      f[2 * i + 1] = (float)h.y;
    }
  }
  __device__ static __forceinline__ float dot(const uint4& a, const uint4& b, float acc) {
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a.x), __builtin_bit_cast(f16x2_t, b.x), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a.y), __builtin_bit_cast(f16x2_t, b.y), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a.z), __builtin_bit_cast(f16x2_t, b.z), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2_t, a.w), __builtin_bit_cast(f16x2_t, b.w), acc, false);
    return acc;
  }
  __device__ static __forceinline__ uint32_t pack2(float lo, float hi) {
    f16x2_t h;
    h.x = (_Float16)lo;
    h.y = (_Float16)hi;
    return __builtin_bit_cast(uint32_t, h);
  }
  __device__ static __forceinline__ uint4 pack(const float (&f)[8]) {
    return make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
  }
};

template <> struct Elem<f32_tag> {
  static constexpr int BYTES = 4;
  static constexpr int VEC = 4;
  __device__ static __forceinline__ void unpack(const uint4& r, float (&f)[4]) {
    f[0] = as_f32(r.x); f[1] = as_f32(r.y); f[2] = as_f32(r.z); f[3] = as_f32(r.w);
  }
  __device__ static __forceinline__ float dot(const uint4& a, const uint4& b, float acc) {
    acc = fmaf(as_f32(a.x), as_f32(b.x), acc);
    acc = fmaf(as_f32(a.y), as_f32(b.y), acc);
    acc = fmaf(as_f32(a.z), as_f32(b.z), acc);
    acc = fmaf(as_f32(a.w), as_f32(b.w), acc);
    return acc;
  }
  __device__ static __forceinline__ uint4 pack(const float (&f)[4]) {
    return make_uint4(as_u32(f[0]), as_u32(f[1]), as_u32(f[2]), as_u32(f[3]));
  }
};

// ---- cross-lane sums --------------------------------------------------------
// DPP controls (gfx9 encoding): quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E,
// row_half_mirror = 0x141 (lane i <-> 7-i of each 8), row_mirror = 0x140
// (lane i <-> 15-i of each 16).  After the four steps every lane of a 16-lane
// DPP row holds the row's total.
template <int CTRL> __device__ __forceinline__ float dpp_f32(float x) {
  return __builtin_amdgcn_update_dpp(0.f, x, CTRL, 0xf, 0xf, false);
}

// all-reduce (sum) over aligned groups of LANES consecutive lanes
template <int LANES> __device__ __forceinline__ float group_sum(float x) {
  static_assert(LANES == 1 || LANES == 2 || LANES == 4 || LANES == 8 || LANES == 16 || LANES == 32 || LANES == 64, "");
  if constexpr (LANES >= 2) x += dpp_f32<0xB1>(x);
  if constexpr (LANES >= 4) x += dpp_f32<0x4E>(x);
  if constexpr (LANES >= 8) x += dpp_f32<0x141>(x);
  if constexpr (LANES >= 16) x += dpp_f32<0x140>(x);
  if constexpr (LANES >= 32) x += __shfl_xor(x, 16);
  if constexpr (LANES >= 64) x += __shfl_xor(x, 32);
  return x;
}

template <int LANES> __device__ __forceinline__ float group_max(float x) {
  if constexpr (LANES >= 2) x = fmaxf(x, dpp_f32<0xB1>(x));
  if constexpr (LANES >= 4) x = fmaxf(x, dpp_f32<0x4E>(x));
  if constexpr (LANES >= 8) x = fmaxf(x, dpp_f32<0x141>(x));
  if constexpr (LANES >= 16) x = fmaxf(x, dpp_f32<0x140>(x));
  if constexpr (LANES >= 32) x = fmaxf(x, __shfl_xor(x, 16));
  if constexpr (LANES >= 64) x = fmaxf(x, __shfl_xor(x, 32));
  return x;
}

// Inclusive scan over the 64 lanes of a wave on the vector ALU (no LDS traffic): three
// shifted adds give windows of 4, row_shr:4 / row_shr:8 complete each 16-lane row,
// row_bcast:15 / row_bcast:31 carry the totals of rows 0 and 2 / of lanes 0-31 upward.
// DPP controls: row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143; a
// lane switched off by the row / bank mask, or shifted in from outside, contributes
// the identity.  Lane 63 ends up with the wave's total.
struct ScanAdd {
  static constexpr int kIdentity = 0;
  __device__ static __forceinline__ int op(int a, int b) { return a + b; }
};
struct ScanMax {
  static constexpr int kIdentity = 0;  // for non-negative inputs
  __device__ static __forceinline__ int op(int a, int b) { return a > b ? a : b; }
};
template <typename Op, int CTRL, int ROW_MASK, int BANK_MASK> __device__ __forceinline__ int scan_step(int x) {
  return Op::op(x, __builtin_amdgcn_update_dpp(Op::kIdentity, x, CTRL, ROW_MASK, BANK_MASK, false));
}
template <typename Op> __device__ __forceinline__ int wave_inclusive_scan(int x) {
  const int s1 = __builtin_amdgcn_update_dpp(Op::kIdentity, x, 0x111, 0xf, 0xf, false);
  const int s2 = __builtin_amdgcn_update_dpp(Op::kIdentity, x, 0x112, 0xf, 0xf, false);
  const int s3 = __builtin_amdgcn_update_dpp(Op::kIdentity, x, 0x113, 0xf, 0xf, false);
  int y = Op::op(Op::op(x, s1), Op::op(s2, s3));
  y = scan_step<Op, 0x114, 0xf, 0xe>(y);
  y = scan_step<Op, 0x118, 0xf, 0xc>(y);
  y = scan_step<Op, 0x142, 0xa, 0xf>(y);
  y = scan_step<Op, 0x143, 0xc, 0xf>(y);
  return y;
}

constexpr float kNegInf = -__builtin_huge_valf();
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// exp2 of a difference that may be (-inf) - (finite) but never inf - inf
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

}  // namespace bsgl
