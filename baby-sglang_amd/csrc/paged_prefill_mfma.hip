// Matrix-core paged extend (prefill) attention for gfx950 (bf16 / f16, head_dim 128).
//
// Follows the reference's _prefill_attention (attn_backend.py:34-110): request i owns
// q rows [cu_ext[i], cu_ext[i+1]); row r of it sees page-table positions
// 0 .. prefix_lens[i] + r of its own sequence (the additive -inf mask of
// attn_backend.py:74-92 as a comparison), K/V gathered through the page table.
//
// One workgroup = one wave = (16 consecutive extend positions of one request) x
// (one kv head) x (up to RT q heads of that kv head's GQA group; a group larger
// than 4 heads is split over two workgroups so accumulators and Q fragments stay
// in registers): the score tiles of a K/V tile reuse the same K fragments
// (ds_read_b128) and the same transposed V fragments (ds_read_b64_tr_b16), so a
// K/V byte is fetched once per 16 positions for RT q heads instead of once per head.  The streaming side
// is the decode kernel's: page-table entries -> 32-bit row offsets in LDS, LDS-DMA
// gather of 16-token tiles into a private ring behind a counted s_waitcnt, online
// softmax in base 2 on the accumulator layout (a lane owns one query position and 4
// tokens), P fed to the PV product straight from the score accumulator.
//
// Work list without a host read: prefill_cu_kernel writes the running sums of
// extended_lens and of ceil(extended_lens / 16); workgroup b finds its request by
// binary search and exits if b is past the last block.
#include "bsgl_device.h"
#include "bsgl_launch.h"
#include "bsgl_mfma.h"

namespace bsgl {

namespace {

using namespace mfma;

template <typename Tag, int RT, int NSTAGE>
__global__ __launch_bounds__(64) void prefill_mfma_kernel(PrefillArgs a) {
  using E = Elem<Tag>;
  constexpr int D = 128;
  constexpr int REC = 512;  // K row | V row of one kv head
  constexpr int TILE_BYTES = kTile * REC;
  constexpr int DMA_PER_TILE = TILE_BYTES / 1024;  // 8, two tokens each
  constexpr int RING = NSTAGE * TILE_BYTES;
  constexpr int kOPad = D + 4;
  static_assert(RING >= 16 * kOPad * 4, "output staging reuses the ring");

  __shared__ __attribute__((aligned(1024))) char smem[RING + kSlotBlock * 4];
  int32_t* s_slots = (int32_t*)(smem + RING);
  const uint32_t lds_ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int lane = threadIdx.x;
  const int lq = lane & 15;  // query position inside the block (column of S^T, row of O)
  const int lg = lane >> 4;
  const int kvh = blockIdx.y;
  const int G = a.n_q_heads / a.n_kv_heads;
  const int rt0 = blockIdx.z * RT;           // first q head of the group handled here
  const int n_rt = min(RT, G - rt0);         // heads past it compute on zero queries and are not written

  // ---- which (request, block of 16 positions) is this? ----------------------------------
  const int64_t blk_id = blockIdx.x;
  if (blk_id >= a.cu_blk[a.batch]) return;
  int lo = 0, hi = a.batch;  // last request with cu_blk[lo] <= blk_id
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.cu_blk[mid] <= blk_id) lo = mid; else hi = mid;
  }
  const int i_req = lo;
  const int q0 = (int)(blk_id - a.cu_blk[i_req]) * 16;
  const int64_t ext64 = a.extended_lens[i_req];
  const int n_rows = (int)min((int64_t)16, ext64 - q0);  // >= 1 by construction of cu_blk
  const int64_t q_row0 = a.cu_ext[i_req] + q0;
  int64_t req = a.req_idx[i_req];
  req = req < 0 ? 0 : (req >= a.kv.table_rows ? a.kv.table_rows - 1 : req);
  const int64_t seq64 = a.seq_lens[i_req];
  const int seq = (int)(seq64 < 0 ? 0 : (seq64 > a.kv.table_cols ? a.kv.table_cols : seq64));
  const int64_t pre64 = a.prefix_lens[i_req];
  const int prefix = (int)(pre64 < 0 ? 0 : (pre64 > seq ? seq : pre64));
  // row r of the block sees tokens 0 .. prefix + q0 + r; padding columns copy the last real row
  const int kv_len = min(seq, prefix + q0 + n_rows);
  const int my_last = min(prefix + q0 + min(lq, n_rows - 1), kv_len - 1);
  const int32_t* table_row = a.kv.table + req * a.kv.table_row_stride;
  const uint32_t max_slot = (uint32_t)(a.kv.n_slots - 1);
  const uint32_t row_units = (uint32_t)(a.kv.slab_row_stride * 2 / 16);

  // ---- Q^T fragments: lane (position lq, group lg) holds Q[q_row0 + lq][head][32 s + 8 lg ..]
  uint4 qf[RT][4];
  {
    const char* qrow = (const char*)a.q + (q_row0 + lq) * a.q_row_stride * 2;
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        qf[rt][s] = make_uint4(0, 0, 0, 0);
        if (lq < n_rows && rt < n_rt) qf[rt][s] = *(const uint4*)(qrow + (((int64_t)kvh * G + rt0 + rt) * D + 32 * s + 8 * lg) * 2);
      }
  }
  const float scale_log2e = a.scale * kLog2e;

  // ---- DMA source pieces: two tokens per instruction, [K 256 B | V 256 B] per token -----
  const int piece = lane & 31;
  const bool piece_is_v = piece >= 16;
  const int lane_tok = lane >> 5;
  const uint64_t lane_src = (uint64_t)(piece_is_v ? a.kv.v_slab : a.kv.k_slab) + (uint64_t)kvh * (D * 2) +
                            (uint64_t)((((piece & 15)) ^ (lane_tok << 2)) << 4);

  const int k_rec = rec_of_token(lq);
  int k_off[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) k_off[s] = k_rec * REC + (((4 * s + lg) ^ xr_of_rec(k_rec)) << 4);
  const int v_rec = rec_of_token(4 * lg + ((lane >> 2) & 3));
  int v_off[8];
#pragma unroll
  for (int n = 0; n < 8; ++n)
    v_off[n] = v_rec * REC + 256 + (((2 * n + ((lane >> 1) & 1)) ^ xr_of_rec(v_rec)) << 4) + ((lane & 1) << 3);

  float m_run[RT], l_run[RT];
  f32x4 o_acc[RT][8];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    m_run[rt] = kNegInf;
    l_run[rt] = 0.f;
#pragma unroll
    for (int n = 0; n < 8; ++n) o_acc[rt][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  for (int blk = 0; blk < kv_len; blk += kSlotBlock) {
    const int n_blk = min(kSlotBlock, kv_len - blk);
    const int n_tiles = (n_blk + kTile - 1) / kTile;
    for (int i = lane; i < n_blk; i += 64) {
      uint32_t s = (uint32_t)table_row[blk + i];
      s = s > max_slot ? max_slot : s;
      s_slots[i] = (int32_t)(s * row_units);
    }
    // retire every compiler-visible load before the tile loop (see the decode kernel)
    __builtin_amdgcn_s_waitcnt(0x0F70);

    auto issue_tile = [&](int tile) {
      const uint32_t stage = lds_ring + (uint32_t)(tile % NSTAGE) * TILE_BYTES;
      const uint32_t tok_off = (uint32_t)s_slots[min(tile * kTile + (lane & 15), n_blk - 1)];
#pragma unroll
      for (int i = 0; i < DMA_PER_TILE; ++i) {
        const int t0 = 2 * i;
        const uint32_t off0 = __builtin_amdgcn_readlane(tok_off, t0);
        const uint32_t off1 = __builtin_amdgcn_readlane(tok_off, t0 + 1);
        uint64_t src = ((uint64_t)(lane < 32 ? off0 : off1) << 4) + lane_src;
        src ^= (uint64_t)(xr_of_rec(rec_of_token(t0)) << 4);
        dma_1k<false>((const void*)src, __builtin_amdgcn_readfirstlane(stage + rec_of_token(t0) * REC));
      }
    };

#pragma unroll
    for (int t = 0; t < NSTAGE - 1; ++t)
      if (t < n_tiles) issue_tile(t);

    for (int tile = 0; tile < n_tiles; ++tile) {
      const int ahead = n_tiles - 1 - tile;
      if (ahead >= NSTAGE - 1) {
        issue_tile(tile + NSTAGE - 1);
        dma_wait<(NSTAGE - 1) * DMA_PER_TILE>();
      } else if (NSTAGE > 3 && ahead == 2) {
        dma_wait<2 * DMA_PER_TILE>();
      } else if (NSTAGE > 2 && ahead == 1) {
        dma_wait<1 * DMA_PER_TILE>();
      } else {
        dma_wait<0>();
      }
      const char* st = smem + (tile % NSTAGE) * TILE_BYTES;
      const int tok0 = blk + tile * kTile + 4 * lg;  // first of this lane's 4 tokens

      uint4 kf[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) kf[s] = *(const uint4*)(st + k_off[s]);

      uint2 pf[RT][PFrag<Tag>::N];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) sc = Mma<Tag>::qk(kf[s], qf[rt][s], sc);
        float sv[4];
        float tmax = kNegInf;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sv[r] = (tok0 + r <= my_last) ? sc[r] * scale_log2e : kNegInf;  // causal + end of sequence
          tmax = fmaxf(tmax, sv[r]);
        }
        tmax = quad_group_max(tmax);
        if (__any(tmax > m_run[rt])) {
          const float m_new = fmaxf(m_run[rt], tmax);
          // a column with nothing visible yet keeps m = -inf: its factor is 1, not exp2(nan)
          const float alpha = m_new == kNegInf ? 1.f : fast_exp2(m_run[rt] - m_new);
          m_run[rt] = m_new;
          l_run[rt] *= alpha;
          float ar[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, 4 * lg + r);
#pragma unroll
          for (int n = 0; n < 8; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) o_acc[rt][n][r] *= ar[r];
        }
        float pr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pr[r] = m_run[rt] == kNegInf ? 0.f : fast_exp2(sv[r] - m_run[rt]);
          l_run[rt] += pr[r];
        }
        const PFrag<Tag> frag(pr);
#pragma unroll
        for (int k = 0; k < PFrag<Tag>::N; ++k) pf[rt][k] = frag.f[k];
      }
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        const s16x4_t vf = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(st + v_off[n]));
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
          for (int k = 0; k < PFrag<Tag>::N; ++k) o_acc[rt][n] = Mma<Tag>::pv(pf[rt][k], vf, o_acc[rt][n]);
      }
    }
  }

  // ---- normalise and write: rows through LDS so that a lane stores 16 contiguous bytes ----
  float* o_lds = (float*)smem;
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    const float l = quad_group_sum(l_run[rt]);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    float ir[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ir[r] = __shfl(inv, 4 * lg + r);
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) o_lds[(4 * lg + r) * kOPad + 16 * n + lq] = o_acc[rt][n][r] * ir[r];
    const int64_t head = (int64_t)kvh * G + rt0 + rt;
    for (int idx = lane; idx < (rt < n_rt ? n_rows : 0) * (D / 8); idx += 64) {
      const int r = idx / (D / 8), c8 = idx % (D / 8);
      const float* src = o_lds + r * kOPad + c8 * 8;
      const uint4 packed = make_uint4(E::pack2(src[0], src[1]), E::pack2(src[2], src[3]),
                                      E::pack2(src[4], src[5]), E::pack2(src[6], src[7]));
      *(uint4*)((char*)a.out + ((q_row0 + r) * a.out_row_stride + head * D + c8 * 8) * 2) = packed;
    }
  }
}

template <typename Tag, int RT>
hipError_t launch_rt(const PrefillArgs& a, hipStream_t stream) {
  const int group = a.n_q_heads / a.n_kv_heads;
  dim3 grid((unsigned)a.max_blocks, a.n_kv_heads, (group + RT - 1) / RT);
  hipLaunchKernelGGL((prefill_mfma_kernel<Tag, RT, 2>), grid, dim3(64), 0, stream, a);
  return hipGetLastError();
}

template <typename Tag>
hipError_t launch_tag(const PrefillArgs& a, hipStream_t stream) {
  // q heads per workgroup: the whole group up to 4, otherwise two halves
  switch (a.n_q_heads / a.n_kv_heads) {
    case 1: return launch_rt<Tag, 1>(a, stream);
    case 2: return launch_rt<Tag, 2>(a, stream);
    case 3: case 5: case 6: return launch_rt<Tag, 3>(a, stream);
    case 4: case 7: case 8: return launch_rt<Tag, 4>(a, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

bool prefill_mfma_supported(const PrefillArgs& a, const char** why) {
  if (a.dtype != kBF16 && a.dtype != kF16) {
    *why = "matrix-core extend needs bf16 or f16";
    return false;
  }
  if (a.head_dim != 128) {
    *why = "matrix-core extend needs head_dim 128";
    return false;
  }
  if (a.n_kv_heads <= 0 || a.n_q_heads % a.n_kv_heads != 0 || a.n_q_heads / a.n_kv_heads > 8) {
    *why = "matrix-core extend needs 1..8 query heads per kv head";
    return false;
  }
  const int64_t row_bytes = a.kv.slab_row_stride * 2;
  if (((uintptr_t)a.kv.k_slab | (uintptr_t)a.kv.v_slab) % 256 != 0 || row_bytes % 256 != 0) {
    *why = "matrix-core extend needs 256-byte aligned slab rows";
    return false;
  }
  if (a.kv.n_slots * (row_bytes / 16) >= 0xffffffffll) {
    *why = "slab larger than 64 GiB";
    return false;
  }
  return true;
}

hipError_t launch_prefill_mfma(const PrefillArgs& a, hipStream_t stream) {
  if (a.total_q_tokens == 0 || a.batch == 0) return hipSuccess;
  return a.dtype == kBF16 ? launch_tag<bf16_tag>(a, stream) : launch_tag<f16_tag>(a, stream);
}

}  // namespace bsgl
