// Matrix-core paged extend (prefill) attention for gfx950 (bf16 / f16, head_dim 128).
//
// Follows the reference's _prefill_attention (attn_backend.py:34-110): request i owns
// q rows [cu_ext[i], cu_ext[i+1]); row r of it sees page-table positions
// 0 .. prefix_lens[i] + r of its own sequence (the additive -inf mask of
// attn_backend.py:74-92 as a comparison), K/V gathered through the page table.
//
// One workgroup = WAVES waves (4, or 1 for batches of very short extends); a wave owns
// 16 consecutive extend positions of one request, the workgroup 16 * WAVES of them, x
// (one kv head) x (up to RT q heads of that kv head's GQA group; a group larger
// than 4 heads is split over two workgroups so accumulators and Q fragments stay
// in registers): the score tiles of a K/V tile reuse the same K fragments
// (ds_read_b128) and the same transposed V fragments (ds_read_b64_tr_b16), so a
// K/V byte is fetched once per 16 positions for RT q heads instead of once per head.  The streaming side
// is the decode kernel's: page-table entries -> 32-bit row offsets in LDS, LDS-DMA
// gather of 16-token tiles into a ring behind a counted s_waitcnt, online softmax in
// base 2 on the accumulator layout, P fed to the PV product straight from the score
// accumulator.  A step covers TWO tiles (32 tokens): a lane then owns one query
// position and 4 + 4 tokens, which is exactly the 8 k-values per lane of a 16x16x32
// product -- the summation order over tokens is free as long as P and V^T agree on
// it, so the two tiles' probability fragments and transposed V fragments are simply
// concatenated and PV runs at the full-rate 16x16x32 shape.  Probabilities enter the PV
// product once, rounded to the dtype (the decode kernel's two-operand split of bf16
// probabilities would be a third of this kernel's matrix work), and the softmax
// denominator comes out of the matrix cores as well: one more 16x16x32 product of the same
// probability fragment with a fragment of ones accumulates, in the rows' own layout, the
// sum of the ROUNDED probabilities -- so numerator and denominator carry the same rounding
// and the output is an exact convex combination of V rows -- and replaces eight vector adds
// per head and step.  The waves of a
// workgroup share every K/V step: each issues a quarter of its DMA, waits for its
// own share, and one barrier per step both publishes it and retires the stage that
// the step after next will overwrite (3-stage ring).
//
// Work list without a host read: prefill_cu_kernel writes the running sums of
// extended_lens and of each request's block groups; a workgroup finds its request by
// binary search and exits if it is past the last group or block.
#include "bsgl_device.h"
#include "bsgl_launch.h"
#include "bsgl_mfma.h"

#include <type_traits>

namespace bsgl {

namespace {

using namespace mfma;

// Register budget: 512 / waves-per-SIMD.  Two heads per workgroup fit two waves per
// SIMD (256 registers), one head three; four heads need the whole file.  (The
// single-wave kernel is limited by its private ring in LDS, not by registers.)
// PROF: the build with cycle stamps (see paged_prefill_mfma32.hip for why it is a build of its own)
template <typename Tag, int RT, int WAVES, bool PROF>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(WAVES == 1 || RT > 2 ? 1 : RT == 1 ? 3 : 2)))
void prefill_mfma_kernel(PrefillArgs a) {
  using E = Elem<Tag>;
  constexpr int D = 128;
  constexpr int REC = 512;  // K row | V row of one kv head
  constexpr int NSTAGE = 3;
  constexpr int TILE_BYTES = kTile * REC;
  constexpr int STEP = 2 * kTile;                  // tokens per step: two tiles
  constexpr int STEP_BYTES = 2 * TILE_BYTES;
  constexpr int DMA_PER_STEP = STEP_BYTES / 1024;  // 16 instructions, two tokens each
  constexpr int DPW = DMA_PER_STEP / WAVES;        // instructions per wave and step
  constexpr int RING = NSTAGE * STEP_BYTES;
  constexpr int kOPad = D + 4;
  constexpr int STAGE_OUT = 16 * kOPad * 4;        // output staging of one wave
  constexpr int BUF = RING > WAVES * STAGE_OUT ? RING : WAVES * STAGE_OUT;
  constexpr int SLOTS = kSlotBlock;  // page-table entries staged in LDS at a time (4096 was measured: no faster)
  static_assert(DMA_PER_STEP % WAVES == 0 && SLOTS % STEP == 0, "");

  __shared__ __attribute__((aligned(1024))) char smem[BUF + SLOTS * 4];
  int32_t* s_slots = (int32_t*)(smem + BUF);
  const uint32_t lds_ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x;
  const uint64_t t_entry = PROF ? __builtin_readcyclecounter() : 0;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 15;  // query position inside the wave's 16 (column of S^T, row of O)
  const int lg = lane >> 4;
  const int G = a.n_q_heads / a.n_kv_heads;
  // Folded heads (fold_group > 0, set by the host layer with n_kv_heads = 1 and head_dim = 128): two
  // kv heads of 64 values, neighbours in a slab row, streamed as one 128-wide kv head.  q head h
  // belongs to kv head h / fold_group: its query is zero outside that half of the row and only that
  // half of its output row is written.
  const int fold = a.fold_group;

  // ---- which (request, block of 16 * WAVES positions, kv head, q-head slice) is this? -------
  // Workgroups are dealt to the 8 XCDs round-robin by launch index, and each XCD has its own
  // L2.  The workgroups that stream the same K/V rows -- the blocks of one request x one kv
  // head x all q-head slices -- are therefore given launch indices that are EQUAL modulo 8
  // and adjacent in that XCD's sequence: index = 8 * slot + xcd, slot = (local group,
  // block in group, slice).  A group is grp_blocks consecutive blocks of a request (so that
  // one long prompt still spreads over all XCDs); group g of the (request, block group, kv
  // head) list goes to XCD g % 8.  K/V then comes from HBM once per group instead of once
  // per workgroup.  (Only placement depends on the dealing order, never the result.)
  const int n_z = (G + RT - 1) / RT;
  const int per_group = a.grp_blocks * n_z;
  const unsigned slot = blockIdx.x >> 3;
  const int64_t g = (int64_t)(slot / (unsigned)per_group) * 8 + (blockIdx.x & 7);
  if (g >= a.cu_blk[a.batch] * a.n_kv_heads) return;
  const int within = (int)(slot % (unsigned)per_group);
  const int kvh = (int)(g % a.n_kv_heads);
  const int64_t pg = g / a.n_kv_heads;       // (request, block group) index
  const int rt0 = (within % n_z) * RT;       // first q head of the group handled here
  const int n_rt = min(RT, G - rt0);         // heads past it compute on zero queries and are not written
  // the request that owns group pg: the last one with cu_blk[i] <= pg, i.e. the number of running
  // sums cu_blk[1..batch] that do not exceed pg (64 compared per load, all loads independent: a
  // binary search would be a chain of dependent loads at the head of every workgroup)
  int i_req = 0;
  for (int base = 1; base <= a.batch; base += 64) {
    const int j = base + lane;
    const bool le = j <= a.batch && a.cu_blk[j] <= pg;
    i_req += __popcll(__ballot(le));
  }
  i_req = min(i_req, a.batch - 1);
  // rows this request may own: never past the last row of q / out, whatever extended_lens says
  // (inconsistent metadata gives wrong numbers, not an access outside the tensors)
  const int64_t q_first = a.cu_ext[i_req];
  const int64_t ext_raw = max((int64_t)0, min(a.extended_lens[i_req], (int64_t)a.total_q_tokens - q_first));
  const int n_blocks = (int)((ext_raw + 16 * WAVES - 1) / (16 * WAVES));
  const int bg0 = (int)(pg - a.cu_blk[i_req]) * a.grp_blocks;  // first block of the group
  const int nb = min(a.grp_blocks, n_blocks - bg0);            // blocks in it (>= 1)
  if (within / n_z >= nb) return;
  // a request's last blocks see the most tokens: they go first
  const int q0_wg = (bg0 + nb - 1 - within / n_z) * (16 * WAVES);
  const int64_t ext64 = ext_raw;
  const int rows_wg = (int)min((int64_t)(16 * WAVES), ext64 - q0_wg);  // >= 1: the block exists
  const int q0 = q0_wg + 16 * wave;
  const int n_rows = max(0, min(16, rows_wg - 16 * wave));  // 0: this wave only helps to load
  const int64_t q_row0 = q_first + q0;
  int64_t req = a.req_idx[i_req];
  req = req < 0 ? 0 : (req >= a.kv.table_rows ? a.kv.table_rows - 1 : req);
  const int64_t seq64 = a.seq_lens[i_req];
  const int seq = (int)(seq64 < 0 ? 0 : (seq64 > a.kv.table_cols ? a.kv.table_cols : seq64));
  const int64_t pre64 = a.prefix_lens[i_req];
  const int prefix = (int)(pre64 < 0 ? 0 : (pre64 > seq ? seq : pre64));
  // row r of the request sees tokens 0 .. prefix + r; padding columns copy the wave's last real row
  const int kv_len = min(seq, prefix + q0_wg + rows_wg);                 // what the workgroup streams
  const int wave_len = n_rows > 0 ? min(kv_len, prefix + q0 + n_rows) : 0;  // what this wave's rows can see
  const int my_last = min(prefix + q0 + min(lq, max(n_rows, 1) - 1), kv_len - 1);
  const int full_until = min(prefix + q0, kv_len - 1);  // last token EVERY row of this wave sees
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
        if (lq < n_rows && rt < n_rt) {
          if (fold == 0) {
            qf[rt][s] = *(const uint4*)(qrow + (((int64_t)kvh * G + rt0 + rt) * D + 32 * s + 8 * lg) * 2);
          } else if ((32 * s + 8 * lg) / 64 == (rt0 + rt) / fold) {
            // folded heads: the 64 values of q head h stand in the half of the 128-wide row that its kv head owns
            qf[rt][s] = *(const uint4*)(qrow + ((int64_t)(rt0 + rt) * 64 + (32 * s + 8 * lg) % 64) * 2);
          }
        }
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

  // tuning aid: cycles this wave spent waiting for its DMA, at the barrier, issuing DMA, in
  // the score / softmax half and in the PV half of its steps, and the step count
  uint64_t prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float m_run[RT], neg_mc[RT];  // reference score of the lane's row, and -(that) * scale * log2(e)
  constexpr float kSlack = 6.f;
  const float slack_raw = kSlack / scale_log2e;
  f32x4 o_acc[RT][8];
  f32x4 l_acc[RT];  // row sums of the probabilities, rows as in o_acc (the same value in every column)
  const uint32_t one2 = E::pack2(1.f, 1.f);
  const uint4 ones = make_uint4(one2, one2, one2, one2);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    m_run[rt] = kNegInf;
    neg_mc[rt] = 0.f;
    l_acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 8; ++n) o_acc[rt][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  for (int blk = 0; blk < kv_len; blk += SLOTS) {
    const int n_blk = min(SLOTS, kv_len - blk);
    const int n_steps = (n_blk + STEP - 1) / STEP;
    const uint64_t t_stage = PROF ? __builtin_readcyclecounter() : 0;
    if (PROF && blk == 0) prof[6] = (t_stage - t_entry) & 0xffffffffu;  // low half: before the first page-table entry is asked for
    __syncthreads();  // every wave is done with the previous block's offsets and tiles
    for (int i = tid; i < n_blk; i += 64 * WAVES) {
      uint32_t s = (uint32_t)table_row[blk + i];
      s = s > max_slot ? max_slot : s;
      s_slots[i] = (int32_t)(s * row_units);
    }
    // retire every compiler-visible load before the step loop (see the decode kernel)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();

    // a step is loaded by all waves together: wave w issues instructions w*DPW .. w*DPW+DPW-1;
    // instruction i brings tokens 2i, 2i+1 of the step (tile i / 8); tokens past the end of
    // the block re-read its last one and are masked below
    auto issue_step = [&](int step) {
      const uint32_t stage = lds_ring + (uint32_t)(step % NSTAGE) * STEP_BYTES;
      // each lane reads the row offsets of ITS tokens straight from the staged entries: instruction
      // j of this wave moves tokens 2 i and 2 i + 1 (i = wave * DPW + j), the lower half of the
      // lanes the first of them
      uint32_t off[DPW];
#pragma unroll
      for (int j = 0; j < DPW; ++j)
        off[j] = (uint32_t)s_slots[min(step * STEP + 2 * (wave * DPW + j) + lane_tok, n_blk - 1)];
#pragma unroll
      for (int j = 0; j < DPW; ++j) {
        const int i = wave * DPW + j;  // wave-uniform
        const int t0 = 2 * (i & 7);
        const uint32_t rec = (uint32_t)rec_of_token(t0);
        const uint32_t xr = (uint32_t)xr_of_rec((int)rec);
        uint64_t src = ((uint64_t)off[j] << 4) + lane_src;
        src ^= (uint64_t)(xr << 4);
        dma_1k<false>((const void*)src,
                      __builtin_amdgcn_readfirstlane(stage + (uint32_t)(i >> 3) * TILE_BYTES + rec * REC));
      }
    };

    if (PROF) prof[5] += (__builtin_readcyclecounter() - t_stage) << 32;  // high half: staging the block's entries
    issue_step(0);
    if (n_steps > 1) issue_step(1);

    // one step; STAGE = step % NSTAGE as a compile-time value, so that every LDS read of the
    // step addresses its ring stage through the instruction's immediate offset instead of a
    // vector add per read
    auto do_step = [&](int step, auto stage_c) {
      constexpr int STAGE = decltype(stage_c)::value;
      const uint64_t t0 = PROF ? __builtin_readcyclecounter() : 0;
      // this wave's share of the step has landed (one younger step may stay in flight) ...
      if (step + 1 < n_steps) dma_wait<DPW>(); else dma_wait<0>();
      const uint64_t t1 = PROF ? __builtin_readcyclecounter() : 0;
      // ... and so has everybody's; all waves are also done reading step - 1
      __syncthreads();
      const uint64_t t2 = PROF ? __builtin_readcyclecounter() : 0;
      if (step + 2 < n_steps) issue_step(step + 2);  // into the stage step - 1 used
      const uint64_t t3 = PROF ? __builtin_readcyclecounter() : 0;
      if (PROF) { prof[0] += t1 - t0; prof[1] += t2 - t1; prof[2] += t3 - t2; prof[5] += 1; }
      const char* st = smem + STAGE * STEP_BYTES;
      const int tok0 = blk + step * STEP + 4 * lg;  // first of this lane's tokens in tile 0; tile 1: + 16
      if (blk + step * STEP >= wave_len) return;  // nothing this wave's rows may see (wave-uniform)

      uint4 kf[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[u][s] = *(const uint4*)(st + u * TILE_BYTES + k_off[s]);

      uint4 pf[RT];
      const uint64_t t4 = PROF ? __builtin_readcyclecounter() : 0;
      // scores, running max and probabilities of each q head for this step.  The running max
      // is kept in raw score units and the scale enters once, inside the exponent:
      // p = exp2(s * c - m * c) is one fused multiply-add per score.  Only steps that reach
      // past what the wave's first row may see (causal edge or end of sequence) compare.
      const bool step_masked = blk + step * STEP + STEP - 1 > full_until;  // wave-uniform
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) {
        f32x4 sc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          sc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 4; ++s) sc[u] = Mma<Tag>::qk(kf[u][s], qf[rt][s], sc[u]);
        }
        float raw[8];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) raw[4 * u + r] = sc[u][r];
        if (step_masked) {
          // (the comparisons hang on a value the compiler cannot see through, or it makes all
          // eight of them in every step, masked or not)
          int t0m = tok0;
          asm volatile("" : "+v"(t0m));
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) raw[4 * u + r] = (t0m + 16 * u + r <= my_last) ? raw[4 * u + r] : kNegInf;
        }
        // Lazy running maximum.  Softmax needs ONE reference value per row that all four lane
        // groups agree on, not the exact maximum: p = exp2((s - m) c) stays correct under any m as
        // long as it cannot overflow, because the row sums carry the same factor.  So the exact
        // maximum (two cross-lane exchanges) and the rescale are only done when some lane sees a
        // score more than kSlack (in exponent units) above the row's reference; until then
        // probabilities may reach 2^kSlack, which f32 sums and bf16 / f16 operands hold easily.
        const float lmax = fmaxf(fmaxf(fmaxf(raw[0], raw[1]), fmaxf(raw[2], raw[3])),
                                 fmaxf(fmaxf(raw[4], raw[5]), fmaxf(raw[6], raw[7])));
        if (__any(lmax > m_run[rt] + slack_raw)) {
          const float tmax = quad_group_max(lmax);
          const float m_new = fmaxf(m_run[rt], tmax);
          // a column with nothing visible yet keeps m = -inf: its factor is 1, not exp2(nan)
          const float alpha = m_new == kNegInf ? 1.f : fast_exp2((m_run[rt] - m_new) * scale_log2e);
          m_run[rt] = m_new;
          neg_mc[rt] = -(m_new == kNegInf ? 0.f : m_new) * scale_log2e;  // masked scores: exp2(-inf) = 0
          float ar[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, 4 * lg + r);
#pragma unroll
          for (int r = 0; r < 4; ++r) l_acc[rt][r] *= ar[r];
#pragma unroll
          for (int n = 0; n < 8; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) o_acc[rt][n][r] *= ar[r];
        }
        float pa[4], pb[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pa[r] = fast_exp2(fmaf(raw[r], scale_log2e, neg_mc[rt]));
          pb[r] = fast_exp2(fmaf(raw[4 + r], scale_log2e, neg_mc[rt]));
        }
        // k-values 0-3 of the lane: its tokens of tile 0, k-values 4-7: those of tile 1
        pf[rt] = make_uint4(E::pack2(pa[0], pa[1]), E::pack2(pa[2], pa[3]), E::pack2(pb[0], pb[1]), E::pack2(pb[2], pb[3]));
        l_acc[rt] = Mma<Tag>::pv32(pf[rt], ones, l_acc[rt]);
      }
      const uint64_t t5 = PROF ? __builtin_readcyclecounter() : 0;
#pragma unroll
      for (int n = 0; n < 8; ++n) {
        const s16x4_t va = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(st + v_off[n]));
        const s16x4_t vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(st + TILE_BYTES + v_off[n]));
        const uint2 va2 = __builtin_bit_cast(uint2, va), vb2 = __builtin_bit_cast(uint2, vb);
        const uint4 vf = make_uint4(va2.x, va2.y, vb2.x, vb2.y);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) o_acc[rt][n] = Mma<Tag>::pv32(pf[rt], vf, o_acc[rt][n]);
      }
      if (PROF) {
        const uint64_t t6 = __builtin_readcyclecounter();
        prof[3] += t5 - t4;
        prof[4] += t6 - t5;
      }
    };
    static_assert(NSTAGE == 3, "the step loop below is unrolled over the ring");
    for (int step = 0; step < n_steps; step += 3) {
      do_step(step, std::integral_constant<int, 0>{});
      if (step + 1 < n_steps) do_step(step + 1, std::integral_constant<int, 1>{});
      if (step + 2 < n_steps) do_step(step + 2, std::integral_constant<int, 2>{});
    }
  }

  const uint64_t t_loop_end = PROF ? __builtin_readcyclecounter() : 0;
  // ---- normalise and write: rows through LDS so that a lane stores 16 contiguous bytes ----
  __syncthreads();  // the ring is free: each wave stages its rows in its own region
  float* o_lds = (float*)(smem + wave * STAGE_OUT);
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) {
    float ir[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ir[r] = l_acc[rt][r] > 0.f ? 1.f / l_acc[rt][r] : 0.f;
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
      if (fold == 0) {
        *(uint4*)((char*)a.out + ((q_row0 + r) * a.out_row_stride + head * D + c8 * 8) * 2) = packed;
      } else if ((c8 * 8) / 64 == (int)head / fold) {
        *(uint4*)((char*)a.out + ((q_row0 + r) * a.out_row_stride + head * 64 + (c8 * 8) % 64) * 2) = packed;
      }
    }
  }
  if (PROF && lane == 0) {
    prof[7] = __builtin_readcyclecounter() - t_entry;       // the wave's whole life
    prof[6] |= (uint64_t)(prof[7] - (t_loop_end - t_entry)) << 32;  // high half: after the last step (normalise, store)
    uint64_t* dst = a.prof + 8 * ((uint64_t)blockIdx.x * WAVES + wave);
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = prof[i];
  }
}

template <typename Tag, int RT>
hipError_t launch_rt(const PrefillArgs& a, hipStream_t stream) {
  const int group = a.n_q_heads / a.n_kv_heads;
  const int n_z = (group + RT - 1) / RT;
  // 8 XCD sequences of ceil(groups / 8) local groups, grp_blocks * n_z workgroups each
  const int64_t local_groups = ((int64_t)a.max_groups * a.n_kv_heads + 7) / 8;
  const int64_t n_wg = 8 * local_groups * a.grp_blocks * n_z;
  if (n_wg > 0x7fffffffll) return hipErrorInvalidValue;
  dim3 grid((unsigned)n_wg, 1, 1);
  // (8-wave workgroups of 128 positions were measured too: the barrier then costs 860 cycles a
  // step instead of 280 and the kernel is 15 % slower)
  const bool prof = a.prof != nullptr;
  if (a.blk_rows == 64) {
    if (prof) hipLaunchKernelGGL((prefill_mfma_kernel<Tag, RT, 4, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((prefill_mfma_kernel<Tag, RT, 4, false>), grid, dim3(256), 0, stream, a);
  } else {
    if (prof) hipLaunchKernelGGL((prefill_mfma_kernel<Tag, RT, 1, true>), grid, dim3(64), 0, stream, a);
    else hipLaunchKernelGGL((prefill_mfma_kernel<Tag, RT, 1, false>), grid, dim3(64), 0, stream, a);
  }
  return hipGetLastError();
}

template <typename Tag>
hipError_t launch_tag(const PrefillArgs& a, hipStream_t stream) {
  // q heads per workgroup.  Measured on MI355X (7B chunked prefill, 64-row blocks):
  // occupancy beats K/V reuse -- 4 heads per workgroup (one wave per SIMD) reach 306
  // TFLOP/s, 1 head (three waves) 358, 2 heads (two waves) 388, odd groups included.
  const int g = a.n_q_heads / a.n_kv_heads;
  int rt = g == 1 ? 1 : 2;
  if (knob(kKnobPrefillRT) > 0) rt = knob(kKnobPrefillRT);  // tuning override
  // (kKnobPrefill16: tuning override, keeps this kernel where the 32x32x16 form applies)
  if (rt == 2 && prefill_mfma32_supported(a) && knob(kKnobPrefill16) != 1) return launch_prefill_mfma32(a, stream);
  switch (rt) {
    case 1: return launch_rt<Tag, 1>(a, stream);
    case 2: return launch_rt<Tag, 2>(a, stream);
    case 4: return launch_rt<Tag, 4>(a, stream);
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
  if (a.n_kv_heads <= 0 || a.n_q_heads % a.n_kv_heads != 0 || a.n_q_heads / a.n_kv_heads > (a.fold_group > 0 ? 16 : 8)) {
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
