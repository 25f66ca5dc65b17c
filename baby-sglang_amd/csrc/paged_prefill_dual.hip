// Matrix-core paged extend attention, ONE wave per SIMD with two row halves interleaved
// (bf16 / f16, head_dim 128, at least two q heads per kv head, blocks of 128 positions).
//
// Same contract, work list, K/V streaming and LDS image as paged_prefill_mfma32.hip (reference:
// _prefill_attention, attn_backend.py:34-110).  There a SIMD holds two waves of 32 rows each, from two
// workgroups, and each wave's step is a chain of its own: what one wave waits for, the other can only
// partly fill.  Here a SIMD holds ONE wave with the whole register file, which owns 64 positions of one
// q head as two halves A and B of 32, and the halves' chains are interleaved in program order so that
// every consumer stands a block of eight products behind its producer.  Per step s of 32 tokens:
//
//     block 1   scores of A for step s + 1   beside   exponentials of B for step s
//     block 2   PV products of B for step s           (largest of A's fresh scores in the gaps)
//     block 3   scores of B for step s + 1   beside   exponentials of A for step s + 1
//     block 4   PV products of A for step s + 1       (largest of B's fresh scores; K of step s + 2)
//
// The K fragments of a step are read from LDS once for both halves, and a workgroup (4 waves: 128
// positions x 2 q heads) brings a K/V byte in once for twice the products of the two-waves-per-SIMD
// form.  The workgroup has the CU's LDS to itself: the ring is NSTAGE steps deep, a step's requests go
// out NSTAGE - 1 steps ahead, and the wait at the head of step s is for step s + 2, so that the K
// fragments of the next step can be fetched before the next barrier.
#include "bsgl_device.h"
#include "bsgl_launch.h"
#include "bsgl_mfma.h"

#include <type_traits>

// Timing cuts (bit mask, tools/build_variant.sh sets it; RESULTS ARE WRONG with any of them, they exist to
// price a phase): 4 = no K/V request inside the step loop, 8 = no exponentials / row sums / maxima,
// 16 = no LDS read of K / V fragments in the loop, 32 = no wait and no barrier at the head of a step.
#ifndef BSGL_CUT
#define BSGL_CUT 0
#endif
#ifndef BSGL_DUAL_PIN
#define BSGL_DUAL_PIN 0
#endif
#ifndef BSGL_DUAL_STAGES
#define BSGL_DUAL_STAGES 8
#endif

namespace bsgl {

namespace {

using namespace mfma;

typedef __attribute__((ext_vector_type(16))) float f32x16;

template <typename Tag> struct MmaD;
template <> struct MmaD<bf16_tag> {
  __device__ static __forceinline__ f32x16 mm(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct MmaD<f16_tag> {
  __device__ static __forceinline__ f32x16 mm(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

// row of the 32x32 accumulator that register r of a lane in half-wave `hi` holds
__device__ __forceinline__ constexpr int acc_row_d(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// at most min(steps, K) steps' requests (REQ each) stay in flight
template <int K, int REQ> __device__ __forceinline__ void wait_leave(int steps) {
  if constexpr (K <= 0) {
    dma_wait<0>();
  } else {
    if (steps >= K) dma_wait<K * REQ>();
    else wait_leave<K - 1, REQ>(steps);
  }
}

template <int V> using ic = std::integral_constant<int, V>;

// f(ic<I>{}) for I = 0 .. N - 1, written out (constant indices into register arrays, `if constexpr` per index)
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(ic<I>{});
    static_for<I + 1, N>(f);
  }
}

template <typename Tag, bool PROF, int NSTAGE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void prefill_dual_kernel(PrefillArgs a) {
  using E = Elem<Tag>;
  constexpr int D = 128;
  constexpr int REC = 512;  // K row | V row of one kv head
  constexpr int WAVES = 4;
  constexpr int HPW = 2;     // q heads per workgroup
  constexpr int ROWS = 128;  // positions per workgroup
  constexpr int TILE_BYTES = kTile * REC;
  constexpr int STEP = 2 * kTile;
  constexpr int STEP_BYTES = 2 * TILE_BYTES;
  constexpr int DPW = (STEP_BYTES / 1024) / WAVES;  // K/V requests per wave and step: 4
  constexpr int AHEAD = NSTAGE - 1;                 // a step's requests go out this many steps ahead of it
  constexpr int LEAVE = NSTAGE - 4;                 // steps that stay in flight over the wait at a step's head
  constexpr int RING = NSTAGE * STEP_BYTES;
  constexpr int kOPad = D + 4;
  constexpr int STAGE_OUT = 32 * kOPad * 4;  // output staging of one wave (one half at a time)
  constexpr int BUF = RING > WAVES * STAGE_OUT ? RING : WAVES * STAGE_OUT;
  constexpr int SLOTS = kSlotBlock;
  constexpr int NTHR = 64 * WAVES;
  constexpr int EPT = SLOTS / NTHR;
  static_assert(DPW == 4 && SLOTS == EPT * NTHR && SLOTS / STEP == 16, "as the request and entry code below assumes");
  static_assert(NSTAGE >= 5 && NSTAGE <= 8 && AHEAD + 1 < SLOTS / STEP, "ring depth the loop below is unrolled for");

  __shared__ __attribute__((aligned(1024))) char smem[BUF + 2 * SLOTS * 4];
  int32_t* s_slots = (int32_t*)(smem + BUF);
  const uint32_t lds_ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x;
  const uint64_t t_entry = PROF ? __builtin_readcyclecounter() : 0;
  const uint64_t rt_entry = PROF ? __builtin_amdgcn_s_memrealtime() : 0;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lp = lane & 31;  // column of S^T (position inside a half), row of O
  const int hi = lane >> 5;
  const int G = a.n_q_heads / a.n_kv_heads;

  // ---- which (request, block of 128 positions, kv head, pair of q heads): the launch order of
  // paged_prefill_mfma.hip (index = 8 * slot + xcd) ----
  const int n_z = (G + HPW - 1) / HPW;
  const int per_group = a.grp_blocks * n_z;
  const unsigned slot = blockIdx.x >> 3;
  const int64_t g = (int64_t)(slot / (unsigned)per_group) * 8 + (blockIdx.x & 7);
  if (g >= a.cu_blk[a.batch] * a.n_kv_heads) return;
  const int within = (int)(slot % (unsigned)per_group);
  const int kvh = (int)(g % a.n_kv_heads);
  const int64_t pg = g / a.n_kv_heads;
  const int rt0 = (within % n_z) * HPW;
  const int n_rt = min(HPW, G - rt0);
  int i_req = 0;
  for (int base = 1; base <= a.batch; base += 64) {
    const int j = base + lane;
    const bool le = j <= a.batch && a.cu_blk[j] <= pg;
    i_req += __popcll(__ballot(le));
  }
  i_req = min(i_req, a.batch - 1);
  // rows this request may own: never past the last row of q / out, whatever extended_lens says
  const int64_t q_first = a.cu_ext[i_req];
  const int64_t ext_raw = max((int64_t)0, min(a.extended_lens[i_req], (int64_t)a.total_q_tokens - q_first));
  const int n_blocks = (int)((ext_raw + ROWS - 1) / ROWS);
  const int bg0 = (int)(pg - a.cu_blk[i_req]) * a.grp_blocks;
  const int nb = min(a.grp_blocks, n_blocks - bg0);
  const int bi = within / n_z;  // block of the group, counted from the request's end
  if (bi >= nb) return;
  // The last pair of an odd GQA group holds one q head: its workgroups take TWO blocks of that head each
  // (waves 2, 3 the block before the one of waves 0, 1), and every second one of them retires.
  const bool two_blocks = n_rt == 1;
  if (two_blocks && (bi & 1)) return;
  const int q0_wg = (bg0 + nb - 1 - bi) * ROWS;
  const int rows_wg = (int)min((int64_t)ROWS, ext_raw - q0_wg);
  const int hw = wave >> 1;  // this wave's q head among the workgroup's two (or its block of the two)
  const int my_bi = bi + (two_blocks ? hw : 0);
  const bool head_ok = two_blocks ? my_bi < nb : hw < n_rt;  // no such head / block: the wave only helps to load
  const int q0_blk = (bg0 + nb - 1 - (head_ok ? my_bi : bi)) * ROWS;
  const int rows_blk = (int)min((int64_t)ROWS, ext_raw - q0_blk);
  const int pos0 = 64 * (wave & 1);  // this wave's 64 positions inside the block; half h: + 32 h
  int64_t req = a.req_idx[i_req];
  req = req < 0 ? 0 : (req >= a.kv.table_rows ? a.kv.table_rows - 1 : req);
  const int64_t seq64 = a.seq_lens[i_req];
  const int seq = (int)(seq64 < 0 ? 0 : (seq64 > a.kv.table_cols ? a.kv.table_cols : seq64));
  const int64_t pre64 = a.prefix_lens[i_req];
  const int prefix = (int)(pre64 < 0 ? 0 : (pre64 > seq ? seq : pre64));
  const int kv_len = min(seq, prefix + q0_wg + rows_wg);
  const int32_t* table_row = a.kv.table + req * a.kv.table_row_stride;
  const uint32_t max_slot = (uint32_t)(a.kv.n_slots - 1);
  const uint32_t row_units = (uint32_t)(a.kv.slab_row_stride * 2 / 16);
  const int head = kvh * G + rt0 + (two_blocks ? 0 : hw);

  // per half: rows it really has, what they may see
  int n_rows[2], my_last[2], full_until[2];
  int64_t q_row0[2];
  int len_any = 0;  // tokens any row of this wave may see (wave-uniform)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int q0 = q0_blk + pos0 + 32 * h;
    n_rows[h] = head_ok ? max(0, min(32, rows_blk - pos0 - 32 * h)) : 0;
    q_row0[h] = q_first + q0;
    if (n_rows[h] > 0) len_any = max(len_any, min(kv_len, prefix + q0 + n_rows[h]));
    my_last[h] = min(prefix + q0 + min(lp, max(n_rows[h], 1) - 1), kv_len - 1);
    full_until[h] = min(prefix + q0, kv_len - 1);
  }

  // ---- Q^T fragments: lane (position lp, half-wave hi) holds Q[row][head][16 kb + 8 hi ..] ----
  uint4 qf[2][8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const char* qrow = (const char*)a.q + (q_row0[h] + lp) * a.q_row_stride * 2;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      qf[h][kb] = make_uint4(0, 0, 0, 0);
      if (lp < n_rows[h]) qf[h][kb] = *(const uint4*)(qrow + ((int64_t)head * D + 16 * kb + 8 * hi) * 2);
    }
  }
  const float scale_log2e = a.scale * kLog2e;

  // ---- DMA source pieces: two tokens per instruction, [K 256 B | V 256 B] per token ----
  const int piece = lane & 31;
  const bool piece_is_v = piece >= 16;
  const int lane_tok = lane >> 5;
  const uint64_t lane_src = (uint64_t)(piece_is_v ? a.kv.v_slab : a.kv.k_slab) + (uint64_t)kvh * (D * 2) +
                            (uint64_t)((((piece & 15)) ^ (lane_tok << 2)) << 4);
  // K as the A operand: lane (token lp of the step, half-wave hi) reads K[token][16 kb + 8 hi ..]
  const int k_rec = rec_of_token(lp & 15);
  int k_off[8];
#pragma unroll
  for (int kb = 0; kb < 8; ++kb)
    k_off[kb] = (lp >> 4) * TILE_BYTES + k_rec * REC + (((2 * kb + hi) ^ xr_of_rec(k_rec)) << 4);
  // V as the B operand through transposing reads: see paged_prefill_mfma32.hip
  int v_off[2][4];
  {
    const int i16 = lane & 15, dh = (lane >> 4) & 1;
#pragma unroll
    for (int jh = 0; jh < 2; ++jh) {
      const int rec = rec_of_token(8 * jh + 4 * hi + (i16 >> 2));
#pragma unroll
      for (int db = 0; db < 4; ++db)
        v_off[jh][db] = rec * REC + 256 + (((4 * db + 2 * dh + ((i16 & 3) >> 1)) ^ xr_of_rec(rec)) << 4) + ((i16 & 1) << 3);
    }
  }

  uint64_t prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  constexpr float kSlack = 6.f;
  const float slack_raw = kSlack / scale_log2e;
  // per half: reference score of the lane's position, -(that) * scale * log2(e), sum of the lane's probabilities
  float m_run[2] = {kNegInf, kNegInf}, neg_mc[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
  f32x16 o_acc[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) o_acc[h][db][r] = 0.f;

  // ---- page-table entries: SLOTS at a time into the two halves of s_slots ----
  const int n_steps = (kv_len + STEP - 1) / STEP;
  auto entry_ptr = [&](int e) __attribute__((always_inline)) { return table_row + max(min(e, kv_len - 1), 0); };  // entries past the end repeat the last one
  auto entry_off = [&](uint32_t s) __attribute__((always_inline)) { return (int32_t)((s > max_slot ? max_slot : s) * row_units); };
  {
    uint32_t e[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) e[k] = (uint32_t)*entry_ptr(tid + k * NTHR);
#pragma unroll
    for (int k = 0; k < EPT; ++k) s_slots[tid + k * NTHR] = entry_off(e[k]);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // retire every compiler-visible load before the step loop
  __syncthreads();
  const uint32_t lds_slots = lds_ring + BUF;

  // a step is loaded by all waves together: wave w issues instructions 4 w .. 4 w + 3; instruction i
  // brings tokens 2i, 2i+1 of the step (tile i / 8).  The row offsets of a step are read from s_slots one
  // request early, the addresses are made ahead of the wait and the barrier of the step that issues them.
  uint32_t off[DPW];
  auto read_offsets = [&](int step) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < DPW; ++r)
      off[r] = (uint32_t)s_slots[((step * STEP) & (2 * SLOTS - 1)) + 2 * (wave * DPW + r) + lane_tok];
  };
  uint64_t src4[DPW];
  auto make_src = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < DPW; ++r) {
      const int i = wave * DPW + r;
      const uint32_t xr = (uint32_t)xr_of_rec(rec_of_token(2 * (i & 7)));
      src4[r] = (((uint64_t)off[r] << 4) + lane_src) ^ (uint64_t)(xr << 4);
    }
  };
  auto issue_step = [&](int step) __attribute__((always_inline)) {
    uint32_t dst[DPW];
#pragma unroll
    for (int r = 0; r < DPW; ++r) {
      const int i = wave * DPW + r;
      dst[r] = __builtin_amdgcn_readfirstlane(lds_ring + (uint32_t)(step % NSTAGE) * STEP_BYTES + (uint32_t)(i >> 3) * TILE_BYTES +
                                              (uint32_t)rec_of_token(2 * (i & 7)) * REC);
    }
    uint32_t keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
        "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, off\n\t"
        "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src4[0]), "v"(src4[1]), "v"(src4[2]), "v"(src4[3]), "s"(dst[0]), "s"(dst[1]), "s"(dst[2]), "s"(dst[3])
        : "memory");
  };

#pragma unroll
  for (int s = 0; s < AHEAD; ++s)
    if (s < n_steps) { read_offsets(s); make_src(); issue_step(s); }
  read_offsets(AHEAD);

  f32x16 sc[2];       // raw scores of the half's coming step
  uint32_t pw[2][8];  // the half's probabilities of the step whose PV products come next, two to a word
  float pf[2][16];    // ... and as floats until their row sums are taken
  uint4 kf[8];        // K fragments of the step whose scores come next
  uint4 vf[4];        // V fragments travel three products ahead of their use
  uint4 kf_cut[4] = {};  // (BSGL_CUT & 16: stand-ins for fragments)
  if constexpr ((BSGL_CUT & 16) != 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(kf_cut[i].x), "+v"(kf_cut[i].y), "+v"(kf_cut[i].z), "+v"(kf_cut[i].w));
  }
  auto max16 = [&](const f32x16& v) __attribute__((always_inline)) {
    float m = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
    return fmaxf(m, fmaxf(fmaxf(fmaxf(v[8], v[9]), fmaxf(v[10], v[11])), fmaxf(fmaxf(v[12], v[13]), fmaxf(v[14], v[15]))));
  };
  auto v_frag = [&](const char* st, int i) __attribute__((always_inline)) {  // V fragment of PV product i: tile i >> 2, value columns 32 (i & 3) ..
    if constexpr ((BSGL_CUT & 16) != 0) return kf_cut[i & 3];
    const s16x4_t va = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4_t*)(st + (i >> 2) * TILE_BYTES + v_off[0][i & 3]));
    const s16x4_t vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4_t*)(st + (i >> 2) * TILE_BYTES + v_off[1][i & 3]));
    const uint2 va2 = __builtin_bit_cast(uint2, va), vb2 = __builtin_bit_cast(uint2, vb);
    return make_uint4(va2.x, va2.y, vb2.x, vb2.y);
  };

  // One K/V request of this wave (request r of the step whose addresses are in src4) into LDS address dst.
  auto issue_one = [&](int r, uint32_t dst) __attribute__((always_inline)) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src4[r]), "s"(dst) : "memory");
  };
  auto dst_of = [&](int stage, int r) __attribute__((always_inline)) {
    const int i = wave * DPW + r;
    return __builtin_amdgcn_readfirstlane(lds_ring + (uint32_t)stage * STEP_BYTES + (uint32_t)(i >> 3) * TILE_BYTES +
                                          (uint32_t)rec_of_token(2 * (i & 7)) * REC);
  };

  // A half's scores of `step` against its reference: mask them where the step crosses the half's diagonal
  // (rare), and -- given the largest of the lane's 16 -- rescale where a lane stands far above its reference
  // (lazy running maximum, rare after the first steps).
  auto mask_scores = [&](auto hc, int step) __attribute__((always_inline)) {
    constexpr int h = decltype(hc)::value;
    if (__builtin_expect(step * STEP + STEP - 1 > full_until[h], 0)) {  // wave-uniform
      int t0m = step * STEP + 4 * hi;
      asm volatile("" : "+v"(t0m));
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[h][r] = (t0m + acc_row_d(r, 0) <= my_last[h]) ? sc[h][r] : kNegInf;
    }
  };
  auto check_max = [&](auto hc, float lmax) __attribute__((always_inline)) {
    constexpr int h = decltype(hc)::value;
    if (__builtin_expect(__any(lmax > m_run[h] + slack_raw), 0)) {
      float x = lmax, y = lmax;
      swap32(x, y);  // the partner lane holds the position's other 16 tokens
      const float m_new = fmaxf(m_run[h], fmaxf(x, y));
      // a position with nothing visible yet keeps m = -inf: its factor is 1, not exp2(nan)
      const float alpha = m_new == kNegInf ? 1.f : fast_exp2((m_run[h] - m_new) * scale_log2e);
      m_run[h] = m_new;
      neg_mc[h] = -(m_new == kNegInf ? 0.f : m_new) * scale_log2e;
      l_run[h] *= alpha;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float ar = __shfl(alpha, acc_row_d(r, 0) + 4 * hi);
#pragma unroll
        for (int db = 0; db < 4; ++db) o_acc[h][db][r] *= ar;
      }
    }
  };
  constexpr int NE_PV = 4;  // exponentials of a half-step that ride in the PV block ahead of its score block's partner

  // Scores of half hq from the K fragments in kf.  In the gaps: the remaining exponentials of half he (none:
  // he < 0; its first NE_PV were made in the PV block before) and their packing into the A operand of its
  // PV products; the first three V fragments of that PV block (stage vst) are asked for on the way.
  auto block_qe = [&](auto hqc, auto hec, auto vc, auto kloc, auto dmac, const char* vst, const char* kst, bool issue_ok) __attribute__((always_inline)) {
    constexpr bool with_v = decltype(vc)::value != 0;
    constexpr int k_lo = decltype(kloc)::value;  // K fragments 0 .. k_lo - 1 of the NEXT step (stage kst) behind their last use
    constexpr int dma = decltype(dmac)::value;   // -1, or 64 * ring stage + 8 * request + gap: one K/V request of this wave
    constexpr int hq = decltype(hqc)::value;
    constexpr int he = decltype(hec)::value;
    constexpr int hx = he < 0 ? 0 : he;
    f32x16 sn;
#pragma unroll
    for (int r = 0; r < 16; ++r) sn[r] = 0.f;
    // (every value is pinned where it is made -- an empty asm statement that takes and returns it: with
    // branches inside the step, the compiler otherwise sinks the arithmetic to the block of its first use)
    auto ex = [&](int el) __attribute__((always_inline)) {
      if constexpr ((BSGL_CUT & 8) != 0) pf[hx][el] = sc[hx][el];
      else pf[hx][el] = fast_exp2(fmaf(sc[hx][el], scale_log2e, neg_mc[hx]));
      asm volatile("" : "+v"(pf[hx][el]));
    };
    auto pk = [&](int j) __attribute__((always_inline)) {
      pw[hx][j] = E::pack2(pf[hx][2 * j], pf[hx][2 * j + 1]);
      asm volatile("" : "+v"(pw[hx][j]));
    };
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 8>([&](auto kbc) __attribute__((always_inline)) {
      constexpr int kb = decltype(kbc)::value;
      sn = MmaD<Tag>::mm(kf[kb], qf[hq][kb], sn);
      if constexpr (he >= 0) {
        // elements 4 .. 11 two to a gap, 12 .. 15 one to a gap; a pair is packed a gap after it is complete
        if constexpr (kb < 4) { ex(4 + 2 * kb); ex(5 + 2 * kb); } else { ex(8 + kb); }
        if constexpr (kb >= 1 && kb <= 4) pk(1 + kb);
        if constexpr (kb == 4) pk(0);
        if constexpr (kb == 5) pk(1);
        if constexpr (kb == 6) pk(6);
      }
      if constexpr (with_v && kb >= 5) vf[kb - 5] = v_frag(vst, kb - 5);
      if constexpr (kb < k_lo && (BSGL_CUT & 16) == 0) kf[kb] = *(const uint4*)(kst + k_off[kb]);
      if constexpr (dma >= 0 && (dma & 7) == kb) {
        if (issue_ok) issue_one((dma >> 3) & 7, dst_of(dma >> 6, (dma >> 3) & 7));
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if constexpr (he >= 0) pk(7);
    sc[hq] = sn;
  };
  // PV products of half hp (its probabilities in pw / pf, the step's V in stage st, fragments 0 - 2 on
  // their way) and its row sums.  In the gaps, for the OTHER half ho and its fresh scores of step_o: the
  // mask, the largest score, the check against its reference, and its first NE_PV exponentials; where kst is
  // given the K fragments of the next score block from stage kst; where dma_stage >= 0 this wave's four K/V
  // requests (addresses in src4) into that ring stage.
  auto block_pv = [&](auto hpc, auto hoc, auto kc, auto dmac, const char* st, const char* kst, int step_o, bool issue_ok) __attribute__((always_inline)) {
    constexpr int k_from = decltype(kc)::value;  // K fragments k_from .. 7 of the next score block (stage kst), one per gap; 8: none
    constexpr int dma = decltype(dmac)::value;   // as in block_qe
    constexpr int hp = decltype(hpc)::value;
    constexpr int ho = decltype(hoc)::value;
    float mx0 = kNegInf, mx1 = kNegInf, mx2 = kNegInf;
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, 8>([&](auto ic_) __attribute__((always_inline)) {
      constexpr int i = decltype(ic_)::value;
      constexpr int kk = i >> 2;
      o_acc[hp][i & 3] = MmaD<Tag>::mm(make_uint4(pw[hp][4 * kk], pw[hp][4 * kk + 1], pw[hp][4 * kk + 2], pw[hp][4 * kk + 3]),
                                       vf[i % 4], o_acc[hp][i & 3]);
      if constexpr (i + 3 < 8) vf[(i + 3) % 4] = v_frag(st, i + 3);
      if constexpr (k_from + i < 8 && (BSGL_CUT & 16) == 0) kf[k_from + i] = *(const uint4*)(kst + k_off[k_from + i]);
      if constexpr (dma >= 0 && (dma & 7) == i) {
        if (issue_ok) issue_one((dma >> 3) & 7, dst_of(dma >> 6, (dma >> 3) & 7));
      }
      // (as instructions of their own: the compiler otherwise pairs the sums into packed adds and parks them)
      if constexpr ((BSGL_CUT & 8) == 0) {
      asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(l_run[hp]) : "v"(pf[hp][2 * i]), "v"(pf[hp][2 * i + 1]));
      if constexpr (i == 1) {
        mask_scores(ic<ho>{}, step_o);
        mx0 = fmaxf(fmaxf(sc[ho][0], sc[ho][1]), sc[ho][2]);
        mx1 = fmaxf(fmaxf(sc[ho][3], sc[ho][4]), sc[ho][5]);
        mx2 = fmaxf(fmaxf(sc[ho][6], sc[ho][7]), sc[ho][8]);
        asm volatile("" : "+v"(mx0), "+v"(mx1), "+v"(mx2));
      }
      if constexpr (i == 2) {
        mx0 = fmaxf(fmaxf(sc[ho][9], sc[ho][10]), mx0);
        mx1 = fmaxf(fmaxf(sc[ho][11], sc[ho][12]), mx1);
        mx2 = fmaxf(fmaxf(sc[ho][13], sc[ho][14]), mx2);
        asm volatile("" : "+v"(mx0), "+v"(mx1), "+v"(mx2));
      }
      if constexpr (i == 3) check_max(ic<ho>{}, fmaxf(fmaxf(fmaxf(mx0, mx1), mx2), sc[ho][15]));
      if constexpr (i >= 8 - NE_PV) {
        pf[ho][i - (8 - NE_PV)] = fast_exp2(fmaf(sc[ho][i - (8 - NE_PV)], scale_log2e, neg_mc[ho]));
        asm volatile("" : "+v"(pf[ho][i - (8 - NE_PV)]));
      }
      } else if constexpr (i >= 8 - NE_PV) {
        pf[ho][i - (8 - NE_PV)] = sc[ho][i - (8 - NE_PV)];
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  };

  // ---- fill the pipeline: scores of both halves for step 0, A's softmax of step 0, then A's PV of step 0
  // with the head of B's softmax in its gaps: the state every step of the loop starts from ----
  wait_leave<AHEAD - 2, DPW>(min(n_steps, AHEAD) - 2);  // steps 0 and 1 have landed (this wave's share ...
  __syncthreads();                                       // ... and everybody's)
  const bool any_rows = len_any > 0;  // wave-uniform: a wave without rows only loads and meets the barriers
  if (any_rows) {
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) kf[kb] = *(const uint4*)(smem + k_off[kb]);
    block_qe(ic<0>{}, ic<-1>{}, ic<0>{}, ic<0>{}, ic<-1>{}, smem, smem, false);
    block_qe(ic<1>{}, ic<-1>{}, ic<1>{}, ic<0>{}, ic<-1>{}, smem, smem, false);
    mask_scores(ic<0>{}, 0);
    check_max(ic<0>{}, max16(sc[0]));
#pragma unroll
    for (int r = 0; r < 16; ++r) pf[0][r] = fast_exp2(fmaf(sc[0][r], scale_log2e, neg_mc[0]));
#pragma unroll
    for (int j = 0; j < 8; ++j) pw[0][j] = E::pack2(pf[0][2 * j], pf[0][2 * j + 1]);
    // (K of step 1 on the way; whatever the stage holds if there is no step 1: those scores get masked)
    block_pv(ic<0>{}, ic<1>{}, ic<0>{}, ic<-1>{}, smem, smem + STEP_BYTES, 0, false);
  }

  // The head of a step: this wave's share of step + 2 has landed, the workgroup barrier (everybody's has,
  // and everybody is done with step - 1: the stage it used takes the requests of step + AHEAD, which go out
  // in the gaps of the step's first PV block).
  // The entries of block b + 1 are asked for at the head of step 16 b + 2, have landed by the wait of
  // step 16 b + AHEAD (requests complete in order) and are turned into row offsets there; they are first
  // read one step later, for the requests of step 16 b + 16.
  auto top = [&](int step, bool issue_ok) __attribute__((always_inline)) {
    const int in_blk = step & (SLOTS / STEP - 1);
    const bool more_entries = ((step >> 4) + 1) * SLOTS < kv_len;  // wave-uniform
    const bool has_wait = ((BSGL_CUT & 32) == 0) && step + 2 < n_steps;  // wave-uniform
    if (issue_ok) make_src();
    if (has_wait) {
      if (in_blk >= 3 && in_blk < AHEAD && more_entries) dma_wait<LEAVE * DPW + EPT>();
      else wait_leave<LEAVE, DPW>(n_steps - 3 - step);
      if (in_blk == AHEAD && more_entries) {
        // (this wave's own pieces, landed by the wait above; the barrier below publishes them)
        int32_t* half = s_slots + (((step >> 4) + 1) & 1) * SLOTS;
        uint32_t e[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) e[k] = (uint32_t)half[tid + k * NTHR];
#pragma unroll
        for (int k = 0; k < EPT; ++k) half[tid + k * NTHR] = entry_off(e[k]);
      }
      __syncthreads();
      read_offsets(step + AHEAD + 1);
      if (in_blk == 2 && more_entries) {
        const int e = ((step >> 4) + 1) * SLOTS + tid;
        const uint32_t half = lds_slots + (uint32_t)((((step >> 4) + 1) & 1) * SLOTS * 4) + (uint32_t)wave * 256u;
#pragma unroll
        for (int k = 0; k < EPT; ++k)
          dma_256((const void*)entry_ptr(e + k * NTHR), __builtin_amdgcn_readfirstlane(half + (uint32_t)(k * NTHR * 4)));
      }
    }
  };

  // step `step` of the loop: ring stage STAGE holds its K/V, the next two those of steps + 1 and + 2
  auto do_step = [&](int step, auto stage_c) __attribute__((always_inline)) {
    constexpr int STAGE = decltype(stage_c)::value;
    constexpr int DSTAGE = (STAGE + AHEAD) % NSTAGE;  // (= the stage of step - 1)
    const char* st = smem + STAGE * STEP_BYTES;
    const char* sx = smem + ((STAGE + 1) % NSTAGE) * STEP_BYTES;
    const char* sy = smem + ((STAGE + 2) % NSTAGE) * STEP_BYTES;
    const bool issue_ok = ((BSGL_CUT & 4) == 0) && step + AHEAD < n_steps;  // wave-uniform
    // (PROF: the stamps are read where the step ends -- reading one waits for every LDS read on its way)
    const uint64_t t0 = PROF ? __builtin_readcyclecounter() : 0;
    top(step, issue_ok);
    const uint64_t t3 = PROF ? __builtin_readcyclecounter() : 0;
    uint64_t tb1 = 0, tb2 = 0, tb3 = 0;
    if (step * STEP >= len_any) {  // nothing this wave's rows may see (wave-uniform): it only helps to load
      if (issue_ok) {
#pragma unroll
        for (int r = 0; r < DPW; ++r) issue_one(r, dst_of(DSTAGE, r));
      }
      return;
    }
    constexpr int KLO = 4;  // K fragments of step + 2 fetched in block 3 (the rest in block 4)
    block_qe(ic<0>{}, ic<1>{}, ic<1>{}, ic<0>{}, ic<64 * DSTAGE + 8 * 0 + 4>{}, st, st, issue_ok);  // block 1 (kf: K of step + 1)
    if (PROF) tb1 = __builtin_readcyclecounter();
    block_pv(ic<1>{}, ic<0>{}, ic<8>{}, ic<64 * DSTAGE + 8 * 1 + 6>{}, st, st, step + 1, issue_ok);  // block 2
    if (PROF) tb2 = __builtin_readcyclecounter();
    if (step + 1 < n_steps) {  // wave-uniform
      block_qe(ic<1>{}, ic<0>{}, ic<1>{}, ic<KLO>{}, ic<64 * DSTAGE + 8 * 2 + 4>{}, sx, sy, issue_ok);  // block 3
      if (PROF) tb3 = __builtin_readcyclecounter();
      block_pv(ic<0>{}, ic<1>{}, ic<KLO>{}, ic<64 * DSTAGE + 8 * 3 + 6>{}, sx, sy, step + 1, issue_ok);  // block 4
      if (PROF) {  // (phase sums: head of the step, blocks 1, 2, 3, whole body)
        const uint64_t te = __builtin_readcyclecounter();
        prof[0] += t3 - t0; prof[1] += tb1 - t3; prof[2] += tb2 - tb1; prof[4] += tb3 - tb2; prof[3] += te - t3; prof[5] += 1;
      }
    }
  };
  for (int step = 0; step < n_steps; step += NSTAGE) {
    do_step(step, ic<0>{});
    if (step + 1 < n_steps) do_step(step + 1, ic<1>{});
    if (step + 2 < n_steps) do_step(step + 2, ic<2>{});
    if (step + 3 < n_steps) do_step(step + 3, ic<3>{});
    if (step + 4 < n_steps) do_step(step + 4, ic<4>{});
    if constexpr (NSTAGE > 5) { if (step + 5 < n_steps) do_step(step + 5, ic<5>{}); }
    if constexpr (NSTAGE > 6) { if (step + 6 < n_steps) do_step(step + 6, ic<6>{}); }
    if constexpr (NSTAGE > 7) { if (step + 7 < n_steps) do_step(step + 7, ic<7>{}); }
  }

  const uint64_t t_loop_end = PROF ? __builtin_readcyclecounter() : 0;
  // ---- normalise and write: a half's rows through LDS so that a lane stores 16 contiguous bytes ----
  __syncthreads();  // the ring is free: each wave stages its rows in its own region
  float* o_lds = (float*)(smem + wave * STAGE_OUT);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    float x = l_run[h], y = l_run[h];
    swap32(x, y);
    const float l_pos = x + y;  // both halves of the position's tokens
    const float inv = l_pos > 0.f ? 1.f / l_pos : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float ir = __shfl(inv, acc_row_d(r, 0) + 4 * hi);
#pragma unroll
      for (int db = 0; db < 4; ++db) o_lds[(acc_row_d(r, 0) + 4 * hi) * kOPad + 32 * db + lp] = o_acc[h][db][r] * ir;
    }
    for (int idx = lane; idx < 32 * (D / 8); idx += 64) {
      const int r = idx / (D / 8), c8 = idx % (D / 8);
      if (r >= n_rows[h]) continue;
      const int64_t orow = q_row0[h] + r;
      const float* src = o_lds + r * kOPad + c8 * 8;
      const uint4 packed = make_uint4(E::pack2(src[0], src[1]), E::pack2(src[2], src[3]),
                                      E::pack2(src[4], src[5]), E::pack2(src[6], src[7]));
      *(uint4*)((char*)a.out + (orow * a.out_row_stride + (int64_t)head * D + c8 * 8) * 2) = packed;
    }
  }
  if (PROF && lane == 0) {
    prof[7] = (__builtin_readcyclecounter() - t_entry) & 0xffffffffu;
    prof[6] = (uint64_t)(prof[7] - (t_loop_end - t_entry)) << 32;
    prof[7] |= ((__builtin_amdgcn_s_memrealtime() - rt_entry) & 0xffffffffu) << 32;
    uint64_t* dst = a.prof + 8 * ((uint64_t)blockIdx.x * WAVES + wave);
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = prof[i];
  }
}

}  // namespace

bool prefill_dual_supported(const PrefillArgs& a) {
  return a.blk_rows == 128 && a.fold_group == 0 && a.head_dim == 128 && a.n_kv_heads > 0 && a.n_q_heads / a.n_kv_heads >= 2;
}

// the grid of paged_prefill_mfma32.hip's launch for 128-position blocks and 2 q heads per workgroup
hipError_t launch_prefill_dual(const PrefillArgs& a, hipStream_t stream) {
  const int group = a.n_q_heads / a.n_kv_heads;
  const int n_z = (group + 1) / 2;
  const int64_t local_groups = ((int64_t)a.max_groups * a.n_kv_heads + 7) / 8;
  const int64_t n_wg = 8 * local_groups * a.grp_blocks * n_z;
  if (n_wg > 0x7fffffffll) return hipErrorInvalidValue;
  dim3 grid((unsigned)n_wg, 1, 1);
  auto go = [&](auto kernel) __attribute__((always_inline)) { hipLaunchKernelGGL(kernel, grid, dim3(256), 0, stream, a); };
  const bool prof = a.prof != nullptr;
  constexpr int NS = BSGL_DUAL_STAGES;
  if (a.dtype == kBF16) prof ? go(prefill_dual_kernel<bf16_tag, true, NS>) : go(prefill_dual_kernel<bf16_tag, false, NS>);
  else prof ? go(prefill_dual_kernel<f16_tag, true, NS>) : go(prefill_dual_kernel<f16_tag, false, NS>);
  return hipGetLastError();
}

}  // namespace bsgl
