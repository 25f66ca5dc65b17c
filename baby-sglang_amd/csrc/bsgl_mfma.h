// Pieces shared by the matrix-core attention kernels (decode and extend) on gfx950:
// MFMA wrappers per dtype, the swizzled LDS image of a 16-token K/V tile, the
// LDS-DMA issue / wait helpers and the four-lane-group reductions.
#pragma once

#include "bsgl_device.h"

namespace bsgl {
namespace mfma {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;

constexpr int kTile = 16;        // tokens per tile (= MFMA M of the score product)
constexpr int kFinePiece = 256;  // shortest piece that DecodeArgs::balance_target_hi may cut a request into
constexpr int kSlotBlock = 512;  // page-table entries staged in LDS at a time

// qk / pv32: 16x16x32 products (8 k-values per lane); pv: the 16x16x16 form (4 per lane).
template <typename Tag> struct Mma;
template <> struct Mma<bf16_tag> {
  __device__ static __forceinline__ f32x4 pv32(const uint4& p, const uint4& v, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, p), __builtin_bit_cast(bf16x8_t, v), c, 0, 0, 0);
  }
  __device__ static __forceinline__ f32x4 qk(const uint4& k, const uint4& q, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, k), __builtin_bit_cast(bf16x8_t, q), c, 0, 0, 0);
  }
  __device__ static __forceinline__ f32x4 pv(const uint2& p, const s16x4_t& v, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_t, p), v, c, 0, 0, 0);
  }
};
template <> struct Mma<f16_tag> {
  __device__ static __forceinline__ f32x4 pv32(const uint4& p, const uint4& v, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, p), __builtin_bit_cast(f16x8_t, v), c, 0, 0, 0);
  }
  __device__ static __forceinline__ f32x4 qk(const uint4& k, const uint4& q, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, k), __builtin_bit_cast(f16x8_t, q), c, 0, 0, 0);
  }
  __device__ static __forceinline__ f32x4 pv(const uint2& p, const s16x4_t& v, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(f16x4_t, p), __builtin_bit_cast(f16x4_t, v), c, 0, 0, 0);
  }
};

// Probabilities as the A operand of the PV product.  f16 keeps 11 significant bits,
// enough for the 1e-3 bar.  bf16 keeps 8, which would put an error of 2^-9 * max|V|
// on every output whatever its size, so bf16 probabilities go in as TWO operands,
// p = hi + lo with lo = bf16(p - hi), and the PV product is issued for both: the
// matrix cores are idle most of the time in these memory-bound kernels.
template <typename Tag> struct PFrag;
template <> struct PFrag<f16_tag> {
  static constexpr int N = 1;
  uint2 f[1];
  __device__ __forceinline__ explicit PFrag(const float (&p)[4]) {
    f[0] = make_uint2(Elem<f16_tag>::pack2(p[0], p[1]), Elem<f16_tag>::pack2(p[2], p[3]));
  }
};
template <> struct PFrag<bf16_tag> {
  static constexpr int N = 2;
  uint2 f[2];
  __device__ __forceinline__ explicit PFrag(const float (&p)[4]) {
    using E = Elem<bf16_tag>;
    f[0] = make_uint2(E::pack2(p[0], p[1]), E::pack2(p[2], p[3]));
    const float r0 = p[0] - as_f32(f[0].x << 16), r1 = p[1] - as_f32(f[0].x & 0xffff0000u);
    const float r2 = p[2] - as_f32(f[0].y << 16), r3 = p[3] - as_f32(f[0].y & 0xffff0000u);
    f[1] = make_uint2(E::pack2(r0, r1), E::pack2(r2, r3));
  }
};

__device__ __forceinline__ constexpr int rec_of_token(int t) { return (t & 3) | ((t & 4) << 1) | ((t & 8) >> 1); }
__device__ __forceinline__ constexpr int xr_of_rec(int r) { return ((r & 3) << 2) | ((r >> 2) & 3); }

// One LDS-DMA instruction: 64 lanes x 16 B from per-lane global addresses to
// LDS bytes [lds_dst, lds_dst + 1024), lane-linear.  M0 carries the LDS address
// and is restored, the compiler does not see the LDS write or count the load:
// completion is awaited with dma_wait<N>() below.
template <bool NT>
__device__ __forceinline__ void dma_1k(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  if constexpr (NT) {  // non-temporal: K/V bytes are read once per step
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off nt\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
  } else {
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(gsrc), "s"(lds_dst)
        : "memory");
  }
}

// The 4-byte form: 64 lanes x 4 B from per-lane global addresses to LDS bytes
// [lds_dst, lds_dst + 256).  No register receives anything, so nothing the compiler does with
// registers can touch data that is still on its way (an asm load INTO a register is taken to be
// complete where the statement ends, and a copy of that register may be made before the data lands).
__device__ __forceinline__ void dma_256(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dword %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_dst)
      : "memory");
}

// max / sum over the four lanes {l, l^16, l^32, l^48} (the lane groups that share a
// q head), every lane receiving the result: two half exchanges on the vector ALU.
// v_permlane32_swap a, b: lanes 32-63 of a trade places with lanes 0-31 of b;
// v_permlane16_swap a, b: the odd 16-lane rows of a with the even rows of b.
// Written in assembly with both registers as read-write operands: with
// __builtin_amdgcn_permlane{16,32}_swap hipcc (ROCm 7.2) folds the two results of
// one call into the same register in expressions such as max(r[0], r[1]).
// s_nop 1 = the two wait states a VALU write needs before the swap reads it.
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

// wait until at most N of this wave's vector-memory operations are outstanding
template <int N> __device__ __forceinline__ void dma_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace mfma
}  // namespace bsgl
