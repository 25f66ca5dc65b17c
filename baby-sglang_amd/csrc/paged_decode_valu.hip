// Generic paged decode attention for gfx950: any KV dtype (bf16 / f16 / f32),
// head_dim 64 or 128, up to 8 query heads per kv head.  Follows the reference's
// _decode_attention (attn_backend.py:112-149) but reads every K/V byte once:
// the q heads of a GQA group share the K/V row in registers instead of a
// repeat_interleave copy, and the page-table gather is fused into the loads.
//
// Work decomposition
//   grid  = (rows, n_kv_heads / HPW, n_splits)       block = 4 waves of 64
//   A lane owns one 16-byte piece of a (token, kv head) row.  LPR lanes cover a
//   row of one head; HPW adjacent kv heads of the same token sit next to each
//   other in the slab, so one wave load instruction reads 64 * 16 B = 1 KiB of
//   whole, contiguous slab rows.  Each wave keeps U such loads of K and of V in
//   flight.  Scores are reduced over the LPR lanes with DPP row operations;
//   softmax is the online (running max / running sum) form in base 2.
//   Per-wave states are merged through LDS, per-split states through a second
//   small kernel (merge_states_kernel).
#include "bsgl_device.h"
#include "bsgl_launch.h"

namespace bsgl {

namespace {

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;
constexpr int kSlotTile = 1024;  // page-table entries staged in LDS at a time

template <typename Tag, int D, int G, int U>
__global__ __launch_bounds__(kThreads) void decode_valu_kernel(DecodeArgs a, int hpw) {
  using E = Elem<Tag>;
  constexpr int VEC = E::VEC;
  constexpr int LPR = D / VEC;    // lanes per (token, head) row
  constexpr int RPW = 64 / LPR;   // rows per wave load instruction
  constexpr int RED = VEC + 2;    // floats per (lane, q head) in the wave merge

  __shared__ int32_t s_slots[kSlotTile];
  __shared__ float s_red[2][64][G][RED];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int row = blockIdx.x;
  const int split = blockIdx.z;

  const int c = lane % LPR;
  const int r = lane / LPR;
  const int head_local = r % hpw;
  const int tok_sub = r / hpw;
  const int tpi = RPW / hpw;  // tokens per wave load instruction
  const int kvh = blockIdx.y * hpw + head_local;

  // ---- this row's token range ------------------------------------------------
  int64_t req = a.req_idx[row];
  req = req < 0 ? 0 : (req >= a.kv.table_rows ? a.kv.table_rows - 1 : req);
  int64_t seq64 = a.seq_lens[row];
  const int seq = (int)(seq64 < 0 ? 0 : (seq64 > a.kv.table_cols ? a.kv.table_cols : seq64));
  const int chunk = (seq + a.n_splits - 1) / a.n_splits;
  const int start = split * chunk;
  const int end = min(seq, start + chunk);
  const int32_t* table_row = a.kv.table + req * a.kv.table_row_stride;
  const uint32_t max_slot = (uint32_t)(a.kv.n_slots - 1);

  // ---- q pieces (packed, as stored) -------------------------------------------
  uint4 qraw[G];
  {
    const char* qrow = (const char*)a.q + (int64_t)row * a.q_row_stride * E::BYTES;
#pragma unroll
    for (int g = 0; g < G; ++g)
      qraw[g] = *(const uint4*)(qrow + ((int64_t)(kvh * G + g) * D + c * VEC) * E::BYTES);
  }
  const float scale_log2e = a.scale * kLog2e;
  const int64_t row_bytes = a.kv.slab_row_stride * E::BYTES;
  const int64_t piece_off = ((int64_t)kvh * D + c * VEC) * E::BYTES;
  const char* kbase = (const char*)a.kv.k_slab + piece_off;
  const char* vbase = (const char*)a.kv.v_slab + piece_off;

  float m[G], l[G], o[G][VEC];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    m[g] = kNegInf;
    l[g] = 0.f;
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[g][i] = 0.f;
  }

  for (int blk = start; blk < end; blk += kSlotTile) {
    const int n_blk = min(kSlotTile, end - blk);
    __syncthreads();
    for (int i = tid; i < n_blk; i += kThreads) {
      uint32_t s = (uint32_t)table_row[blk + i];
      s_slots[i] = (int32_t)(s > max_slot ? max_slot : s);
    }
    __syncthreads();

    const int tok_per_step = U * tpi;
    for (int base = wave * tok_per_step; base < n_blk; base += kWaves * tok_per_step) {
      uint4 kr[U], vr[U];
      bool valid[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = base + u * tpi + tok_sub;
        valid[u] = idx < n_blk;
        // tokens past the end re-read the last valid one (finite data, weight 0)
        const int64_t off = (int64_t)s_slots[valid[u] ? idx : n_blk - 1] * row_bytes;
        kr[u] = *(const uint4*)(kbase + off);
        vr[u] = *(const uint4*)(vbase + off);
      }
      float p[U][G];
      float mt[G];
      bool grow = false;
#pragma unroll
      for (int g = 0; g < G; ++g) mt[g] = kNegInf;
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          float s = group_sum<LPR>(E::dot(qraw[g], kr[u], 0.f)) * scale_log2e;
          s = valid[u] ? s : kNegInf;
          p[u][g] = s;
          mt[g] = fmaxf(mt[g], s);
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) grow |= mt[g] > m[g];
      if (__any(grow)) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float mn = fmaxf(m[g], mt[g]);
          const float alpha = (mn == kNegInf) ? 1.f : fast_exp2(m[g] - mn);
          m[g] = mn;
          l[g] *= alpha;
#pragma unroll
          for (int i = 0; i < VEC; ++i) o[g][i] *= alpha;
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float vf[VEC];
        E::unpack(vr[u], vf);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          // m[g] is finite whenever a valid score exists; invalid scores are -inf -> 0
          const float pe = (m[g] == kNegInf) ? 0.f : fast_exp2(p[u][g] - m[g]);
          l[g] += pe;
#pragma unroll
          for (int i = 0; i < VEC; ++i) o[g][i] = fmaf(pe, vf[i], o[g][i]);
        }
      }
    }
  }

  // ---- merge lane groups that hold different tokens of the same row piece -----
  for (int off = LPR * hpw; off < 64; off <<= 1) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float mo = __shfl_xor(m[g], off);
      const float lo = __shfl_xor(l[g], off);
      const float mn = fmaxf(m[g], mo);
      const float fa = (m[g] == kNegInf) ? 0.f : fast_exp2(m[g] - mn);
      const float fb = (mo == kNegInf) ? 0.f : fast_exp2(mo - mn);
      l[g] = l[g] * fa + lo * fb;
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[g][i] = o[g][i] * fa + __shfl_xor(o[g][i], off) * fb;
      m[g] = mn;
    }
  }

  // ---- merge the 4 waves through LDS: (2,3) -> (0,1), then 1 -> 0 --------------
  const bool owner = lane < LPR * hpw;
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    const int senders_from = round == 0 ? 2 : 1;
    const int n_send = round == 0 ? 2 : 1;
    __syncthreads();
    if (owner && wave >= senders_from && wave < senders_from + n_send) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) s_red[wave - senders_from][lane][g][i] = o[g][i];
        s_red[wave - senders_from][lane][g][VEC] = m[g];
        s_red[wave - senders_from][lane][g][VEC + 1] = l[g];
      }
    }
    __syncthreads();
    if (owner && wave < n_send) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float mo = s_red[wave][lane][g][VEC];
        const float lo = s_red[wave][lane][g][VEC + 1];
        const float mn = fmaxf(m[g], mo);
        const float fa = (m[g] == kNegInf) ? 0.f : fast_exp2(m[g] - mn);
        const float fb = (mo == kNegInf) ? 0.f : fast_exp2(mo - mn);
        l[g] = l[g] * fa + lo * fb;
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[g][i] = o[g][i] * fa + s_red[wave][lane][g][i] * fb;
        m[g] = mn;
      }
    }
  }

  // ---- write: final output (1 split, same dtype) or a partial state -------------
  const bool direct = a.n_splits == 1 && a.out_dtype == a.dtype;
  if (wave == 0 && owner) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int hq = kvh * G + g;
      const float inv = l[g] > 0.f ? 1.f / l[g] : 0.f;
      float of[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) of[i] = o[g][i] * inv;
      const float lse = l[g] > 0.f ? (m[g] + __builtin_amdgcn_logf(l[g])) * kLn2 : kNegInf;
      if (direct) {
        char* orow = (char*)a.out + (int64_t)row * a.out_row_stride * E::BYTES;
        *(uint4*)(orow + ((int64_t)hq * D + c * VEC) * E::BYTES) = E::pack(of);
        if (a.lse != nullptr && c == 0) a.lse[(int64_t)row * a.n_q_heads + hq] = lse;
      } else {
        const int64_t prow = (int64_t)split * a.rows * a.n_q_heads + (int64_t)row * a.n_q_heads + hq;
        float* po = a.ws_o + prow * D + c * VEC;
#pragma unroll
        for (int i = 0; i < VEC; i += 4) *(float4*)(po + i) = make_float4(of[i], of[i + 1], of[i + 2], of[i + 3]);
        if (c == 0) a.ws_lse[prow] = lse;
      }
    }
  }
}

// out[row, head, :] = sum_p w_p o_p / sum_p w_p,  w_p = exp(lse_p - max lse).
// One wave per (row, head): lane p first holds part p's weight (n_parts <= 64),
// then every lane accumulates its D / 64 output elements over the parts.
template <typename Tag, int D>
__global__ __launch_bounds__(256) void merge_states_kernel(MergeArgs a) {
  using E = Elem<Tag>;
  constexpr int EPL = D / 64;  // elements per lane
  const int lane = threadIdx.x & 63;
  const int64_t n_rh = (int64_t)a.rows * a.n_heads;
  const int64_t rh = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (rh >= n_rh) return;
  const float my_lse = lane < a.n_parts ? a.parts_lse[(int64_t)lane * n_rh + rh] : kNegInf;
  const float mx = group_max<64>(my_lse);
  const float my_w = (my_lse == kNegInf) ? 0.f : fast_exp2((my_lse - mx) * kLog2e);
  const float wsum = group_sum<64>(my_w);
  // (even and odd parts are summed apart and added at the end: the order of the decode kernel's
  // own merge, whose wave halves read the parts of either parity, so that the two agree bit for bit)
  float acc2[2][EPL];
#pragma unroll
  for (int i = 0; i < EPL; ++i) acc2[0][i] = acc2[1][i] = 0.f;
  const float* po = a.parts_o + rh * D + lane * EPL;
  const int64_t part_stride = n_rh * D;
  for (int p0 = 0; p0 < a.n_parts; p0 += 8) {
    // 8 parts per round so their loads are in flight together; parts past the end
    // re-read the last one with weight 0
    float w[8];
    float2 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int p = min(p0 + j, a.n_parts - 1);
      w[j] = p0 + j < a.n_parts ? __shfl(my_w, p) : 0.f;
      if constexpr (EPL == 2) v[j] = *(const float2*)(po + p * part_stride);
      else v[j] = make_float2(po[p * part_stride], 0.f);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      acc2[j & 1][0] = fmaf(w[j], v[j].x, acc2[j & 1][0]);
      if constexpr (EPL == 2) acc2[j & 1][1] = fmaf(w[j], v[j].y, acc2[j & 1][1]);
    }
  }
  float acc[EPL];
#pragma unroll
  for (int i = 0; i < EPL; ++i) acc[i] = acc2[0][i] + acc2[1][i];
  const float inv = wsum > 0.f ? 1.f / wsum : 0.f;
  const int64_t row = rh / a.n_heads;
  const int head = (int)(rh % a.n_heads);
  char* dst = (char*)a.out + (row * a.out_row_stride + (int64_t)head * D + lane * EPL) * E::BYTES;
  if constexpr (E::BYTES == 4) {
    if constexpr (EPL == 2) *(float2*)dst = make_float2(acc[0] * inv, acc[1] * inv);
    else *(float*)dst = acc[0] * inv;
  } else {
    if constexpr (EPL == 2) *(uint32_t*)dst = E::pack2(acc[0] * inv, acc[1] * inv);
    else *(uint16_t*)dst = (uint16_t)E::pack2(acc[0] * inv, 0.f);
  }
  if (a.lse_out != nullptr && lane == 0)
    a.lse_out[rh] = wsum > 0.f ? mx + __builtin_amdgcn_logf(wsum) * kLn2 : kNegInf;
}

template <typename Tag, int D>
constexpr int rows_per_wave_inst() {
  return 64 / (D / Elem<Tag>::VEC);
}

template <typename Tag, int D, int G>
hipError_t launch_valu_inst(const DecodeArgs& a, hipStream_t stream) {
  constexpr int RPW = rows_per_wave_inst<Tag, D>();
  // heads per workgroup: the largest power of two <= RPW that divides n_kv_heads
  int hpw = 1;
  while (hpw * 2 <= RPW && a.n_kv_heads % (hpw * 2) == 0) hpw *= 2;
  dim3 grid(a.rows, a.n_kv_heads / hpw, a.n_splits);
  constexpr int U = 4;
  hipLaunchKernelGGL((decode_valu_kernel<Tag, D, G, U>), grid, dim3(kThreads), 0, stream, a, hpw);
  return hipGetLastError();
}

template <typename Tag, int D>
hipError_t launch_valu_g(const DecodeArgs& a, hipStream_t stream) {
  switch (a.n_q_heads / a.n_kv_heads) {
    case 1: return launch_valu_inst<Tag, D, 1>(a, stream);
    case 2: return launch_valu_inst<Tag, D, 2>(a, stream);
    case 3: return launch_valu_inst<Tag, D, 3>(a, stream);
    case 4: return launch_valu_inst<Tag, D, 4>(a, stream);
    case 5: return launch_valu_inst<Tag, D, 5>(a, stream);
    case 6: return launch_valu_inst<Tag, D, 6>(a, stream);
    case 7: return launch_valu_inst<Tag, D, 7>(a, stream);
    case 8: return launch_valu_inst<Tag, D, 8>(a, stream);
    default: return hipErrorInvalidValue;
  }
}

template <typename Tag>
hipError_t launch_valu_d(const DecodeArgs& a, hipStream_t stream) {
  return a.head_dim == 128 ? launch_valu_g<Tag, 128>(a, stream) : launch_valu_g<Tag, 64>(a, stream);
}

template <typename Tag>
hipError_t launch_merge_d(const MergeArgs& a, hipStream_t stream) {
  const int64_t n_rh = (int64_t)a.rows * a.n_heads;
  const dim3 grid((unsigned)((n_rh + 3) / 4));
  if (a.head_dim == 128)
    hipLaunchKernelGGL((merge_states_kernel<Tag, 128>), grid, dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL((merge_states_kernel<Tag, 64>), grid, dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace

bool decode_valu_supported(const DecodeArgs& a, const char** why) {
  if (a.head_dim != 64 && a.head_dim != 128) {
    *why = "head_dim must be 64 or 128";
    return false;
  }
  if (a.n_kv_heads <= 0 || a.n_q_heads % a.n_kv_heads != 0) {
    *why = "n_q_heads must be a multiple of n_kv_heads";
    return false;
  }
  const int g = a.n_q_heads / a.n_kv_heads;
  if (g < 1 || g > 8) {
    *why = "query heads per kv head must be in 1..8";
    return false;
  }
  return true;
}

hipError_t launch_decode_valu(const DecodeArgs& a, hipStream_t stream) {
  if (a.rows == 0) return hipSuccess;
  hipError_t e;
  switch (a.dtype) {
    case kBF16: e = launch_valu_d<bf16_tag>(a, stream); break;
    case kF16: e = launch_valu_d<f16_tag>(a, stream); break;
    case kF32: e = launch_valu_d<f32_tag>(a, stream); break;
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess || decode_direct_out(a)) return e;
  MergeArgs mg;
  mg.out = a.out;
  mg.out_row_stride = a.out_row_stride;
  mg.lse_out = a.lse;
  mg.parts_o = a.ws_o;
  mg.parts_lse = a.ws_lse;
  mg.n_parts = a.n_splits;
  mg.rows = a.rows;
  mg.n_heads = a.n_q_heads;
  mg.head_dim = a.head_dim;
  mg.dtype = a.out_dtype;
  return launch_merge(mg, stream);
}

hipError_t launch_merge(const MergeArgs& a, hipStream_t stream) {
  if (a.rows == 0 || a.n_heads == 0) return hipSuccess;
  switch (a.dtype) {
    case kBF16: return launch_merge_d<bf16_tag>(a, stream);
    case kF16: return launch_merge_d<f16_tag>(a, stream);
    case kF32: return launch_merge_d<f32_tag>(a, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace bsgl
