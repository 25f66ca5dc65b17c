// The steps either side of the attention call, for gfx950: RMSNorm (+ residual add) and
// rotary embedding of q / k fused with the QKV split and the KV-cache write.
//
// Both are a few hundred KB per decode step: launch-latency work, so the point is one
// launch instead of the reference's dozen elementwise torch ops, and arithmetic that
// rounds exactly where the reference rounds, so that results agree with it bit for bit
// except where a transcendental (cos, sin) or a sum order differs in the last place.
//
// rmsnorm_kernel (reference: qwen2.py:125-141 RMSNorm.forward):
//   h = x + residual (rounded to the input dtype), residual_out = h
//   out = weight * dtype(float(h) * rsqrt(mean(float(h)^2) + eps))      (product rounded to dtype)
//
// rope_kv_kernel (reference: qwen2.py:246-268 RotaryEmbedding, :285-290 QKV split,
// memory_pool.py:152-171 set_kv_buffer):
//   freqs = float(position) * inv_freq;  cos, sin = dtype(cos(freqs)), dtype(sin(freqs))
//   x1, x2 = halves of a head row;  out = [dtype(dtype(x1 cos) - dtype(x2 sin)), dtype(dtype(x1 sin) + dtype(x2 cos))]
//   q heads -> q_out;  k heads -> k_slab[out_cache_loc[t]] (and k_out);  v heads copied likewise.
#include "bsgl_device.h"
#include "bsgl_launch.h"

namespace bsgl {

namespace {

template <typename Tag> __device__ __forceinline__ float rnd(float x);
template <> __device__ __forceinline__ float rnd<bf16_tag>(float x) { return as_f32(Elem<bf16_tag>::pack2(x, 0.f) << 16); }
template <> __device__ __forceinline__ float rnd<f16_tag>(float x) { return (float)(_Float16)x; }
template <> __device__ __forceinline__ float rnd<f32_tag>(float x) { return x; }

// ---------------------------------------------------------------------------------- RMSNorm
// One workgroup per row; a thread keeps up to CHUNKS 16-byte chunks of the row in registers
// between the two passes (sum of squares, scale), so the row is read once.  CHUNKS = 4 serves rows
// of up to 8192 16-bit (4096 fp32) elements, every model of the target family; 16 serves wider rows.
constexpr int kNormThreads = 256;
constexpr int kNormChunks = 4;
constexpr int kNormChunksWide = 16;

template <typename Tag, int CHUNKS>
__global__ __launch_bounds__(kNormThreads) void rmsnorm_kernel(NormArgs a) {
#pragma clang fp contract(off)  // round where the reference rounds: no fused multiply-add
  using E = Elem<Tag>;
  constexpr int V = E::VEC;
  const int row = blockIdx.x;
  const int tid = threadIdx.x;
  const int n_chunks = a.hidden / V;
  const char* x_row = (const char*)a.x + (int64_t)row * a.x_row_stride * E::BYTES;
  const char* r_row = a.residual ? (const char*)a.residual + (int64_t)row * a.hidden * E::BYTES : nullptr;
  char* ro_row = a.residual_out ? (char*)a.residual_out + (int64_t)row * a.hidden * E::BYTES : nullptr;
  char* o_row = (char*)a.out + (int64_t)row * a.out_row_stride * E::BYTES;

  float h[CHUNKS][V];
  float ss = 0.f;
#pragma unroll
  for (int k = 0; k < CHUNKS; ++k) {
    const int c = tid + k * kNormThreads;
    if (c < n_chunks) {
      E::unpack(*(const uint4*)(x_row + (int64_t)c * 16), h[k]);
      if (r_row != nullptr) {
        float r[V];
        E::unpack(*(const uint4*)(r_row + (int64_t)c * 16), r);
#pragma unroll
        for (int i = 0; i < V; ++i) h[k][i] = rnd<Tag>(h[k][i] + r[i]);
        if (ro_row != nullptr) *(uint4*)(ro_row + (int64_t)c * 16) = E::pack(h[k]);
      }
#pragma unroll
      for (int i = 0; i < V; ++i) ss += h[k][i] * h[k][i];
    }
  }
  __shared__ float s_part[kNormThreads / 64];
  ss = group_sum<64>(ss);
  if ((tid & 63) == 0) s_part[tid >> 6] = ss;
  __syncthreads();
  float total = 0.f;
#pragma unroll
  for (int w = 0; w < kNormThreads / 64; ++w) total += s_part[w];
  const float rs = 1.0f / sqrtf(total / (float)a.hidden + a.eps);
#pragma unroll
  for (int k = 0; k < CHUNKS; ++k) {
    const int c = tid + k * kNormThreads;
    if (c < n_chunks) {
      float w[V], o[V];
      E::unpack(*(const uint4*)((const char*)a.weight + (int64_t)c * 16), w);
#pragma unroll
      for (int i = 0; i < V; ++i) o[i] = w[i] * rnd<Tag>(h[k][i] * rs);  // packed (= rounded) below
      *(uint4*)(o_row + (int64_t)c * 16) = E::pack(o);
    }
  }
}

// ------------------------------------------------------------------------------- RoPE + KV write
// One workgroup per token.  cos / sin of the token's D/2 frequencies are computed once into LDS
// (already rounded to the dtype, as the reference casts them before multiplying), then every
// thread moves (head, chunk) items: a 16-byte chunk of the first half of a head row and the
// matching chunk of the second half.
constexpr int kRopeThreads = 256;

template <typename Tag>
__global__ __launch_bounds__(kRopeThreads) void rope_kv_kernel(RopeArgs a) {
#pragma clang fp contract(off)  // x1 cos - x2 sin is two rounded products and a difference in the reference
  using E = Elem<Tag>;
  constexpr int V = E::VEC;
  __shared__ float s_cos[128], s_sin[128];  // head_dim <= 256
  const int t = blockIdx.x;
  const int tid = threadIdx.x;
  const int half = a.head_dim / 2;
  const float pos = (float)a.positions[t];
  for (int j = tid; j < half; j += kRopeThreads) {
    const float f = pos * a.inv_freq[j];
    s_cos[j] = rnd<Tag>(cosf(f));
    s_sin[j] = rnd<Tag>(sinf(f));
  }
  __syncthreads();

  const int cph = half / V;  // chunk pairs per head
  const int n_heads = a.n_q_heads + 2 * a.n_kv_heads;
  int64_t slot = -1;
  if (a.out_cache_loc != nullptr) {
    slot = a.out_cache_loc[t];
    if (slot < 0 || slot >= a.n_slots) slot = -1;
  }
  const char* src_row = (const char*)a.qkv + (int64_t)t * a.qkv_row_stride * E::BYTES;
  for (int item = tid; item < n_heads * cph; item += kRopeThreads) {
    const int head = item / cph, c = item % cph;
    const char* src = src_row + ((int64_t)head * a.head_dim + c * V) * E::BYTES;
    const uint4 lo_raw = *(const uint4*)src;
    const uint4 hi_raw = *(const uint4*)(src + (int64_t)half * E::BYTES);
    const bool is_q = head < a.n_q_heads;
    const bool is_v = head >= a.n_q_heads + a.n_kv_heads;
    uint4 lo_out = lo_raw, hi_out = hi_raw;
    if (!is_v) {
      float x1[V], x2[V], o1[V], o2[V];
      E::unpack(lo_raw, x1);
      E::unpack(hi_raw, x2);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const float cs = s_cos[c * V + i], sn = s_sin[c * V + i];
        o1[i] = rnd<Tag>(x1[i] * cs) - rnd<Tag>(x2[i] * sn);  // packed (= rounded) below
        o2[i] = rnd<Tag>(x1[i] * sn) + rnd<Tag>(x2[i] * cs);
      }
      lo_out = E::pack(o1);
      hi_out = E::pack(o2);
    }
    if (is_q) {
      char* dst = (char*)a.q_out + ((int64_t)t * a.q_out_row_stride + (int64_t)head * a.head_dim + c * V) * E::BYTES;
      *(uint4*)dst = lo_out;
      *(uint4*)(dst + (int64_t)half * E::BYTES) = hi_out;
      continue;
    }
    const int kvh = is_v ? head - a.n_q_heads - a.n_kv_heads : head - a.n_q_heads;
    const int64_t col = ((int64_t)kvh * a.head_dim + c * V) * E::BYTES;
    char* rows = (char*)(is_v ? a.v_out : a.k_out);
    if (rows != nullptr) {
      char* dst = rows + (int64_t)t * a.kv_out_row_stride * E::BYTES + col;
      *(uint4*)dst = lo_out;
      *(uint4*)(dst + (int64_t)half * E::BYTES) = hi_out;
    }
    if (slot >= 0) {
      char* dst = (char*)(is_v ? a.v_slab : a.k_slab) + slot * a.slab_row_stride * E::BYTES + col;
      *(uint4*)dst = lo_out;
      *(uint4*)(dst + (int64_t)half * E::BYTES) = hi_out;
    }
  }
}

}  // namespace

hipError_t launch_rmsnorm(const NormArgs& a, hipStream_t stream) {
  if (a.rows == 0) return hipSuccess;
  const dim3 grid(a.rows), block(kNormThreads);
  const bool wide = a.hidden > kNormThreads * kNormChunks * (a.dtype == kF32 ? 4 : 8);
  switch (a.dtype) {
    case kBF16:
      if (wide) hipLaunchKernelGGL((rmsnorm_kernel<bf16_tag, kNormChunksWide>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((rmsnorm_kernel<bf16_tag, kNormChunks>), grid, block, 0, stream, a);
      break;
    case kF16:
      if (wide) hipLaunchKernelGGL((rmsnorm_kernel<f16_tag, kNormChunksWide>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((rmsnorm_kernel<f16_tag, kNormChunks>), grid, block, 0, stream, a);
      break;
    default:
      if (wide) hipLaunchKernelGGL((rmsnorm_kernel<f32_tag, kNormChunksWide>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((rmsnorm_kernel<f32_tag, kNormChunks>), grid, block, 0, stream, a);
      break;
  }
  return hipGetLastError();
}

int rmsnorm_max_hidden(int dtype) { return kNormThreads * kNormChunksWide * (dtype == kF32 ? 4 : 8); }

hipError_t launch_rope_kv(const RopeArgs& a, hipStream_t stream) {
  if (a.n_tok == 0) return hipSuccess;
  const dim3 grid(a.n_tok), block(kRopeThreads);
  switch (a.dtype) {
    case kBF16: hipLaunchKernelGGL(rope_kv_kernel<bf16_tag>, grid, block, 0, stream, a); break;
    case kF16: hipLaunchKernelGGL(rope_kv_kernel<f16_tag>, grid, block, 0, stream, a); break;
    default: hipLaunchKernelGGL(rope_kv_kernel<f32_tag>, grid, block, 0, stream, a); break;
  }
  return hipGetLastError();
}

}  // namespace bsgl
