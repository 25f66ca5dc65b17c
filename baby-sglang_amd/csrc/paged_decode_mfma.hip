// Matrix-core paged decode attention for gfx950 (bf16 / f16, head_dim 128).
//
// One workgroup = ONE wave = one (query row, kv-head group, sequence split).
// The wave streams its tokens' K and V rows from the paged slabs into a private
// LDS ring with LDS-DMA loads (global_load_lds_dwordx4: no VGPRs, per-lane
// source address = page-table gather, 1 KiB of whole slab rows per
// instruction), keeps NSTAGE-1 tiles of 16 tokens in flight behind a counted
// s_waitcnt, and consumes each tile with matrix instructions:
//
//   S^T[token][q]  = K[token][:] . Q^T          4 x v_mfma_f32_16x16x32 (A = K rows
//                    (ds_read_b128), B = Q^T held in registers, q heads padded to 16)
//   online softmax   in base 2 on the accumulator layout: a lane owns one q head
//                    and 4 tokens; max / sum across the 4 lane groups
//   O[q][d]       += P[q][token] . V[token][d]   8 x v_mfma_f32_16x16x16 (A = P packed
//                    from the accumulator as is, B = V through ds_read_b64_tr_b16)
//
// so the vector ALU only does the softmax and neither K nor V is ever copied
// per q head (the reference repeats them, attn_backend.py:132-135).  There is
// no barrier anywhere: a wave only reads LDS bytes its own DMA wrote.
//
// LDS image of a tile: 16 token records of HPT * 512 B: [K head 0..HPT-1 | V head
// 0..HPT-1], 256 B per head row.  Token t of the tile sits in record
// rec(t) = t with bits 2 and 3 swapped, and 16-byte chunk c of a row is stored at
// chunk c ^ xr(rec), xr(r) = ((r & 3) << 2) | ((r >> 2) & 3): with that the
// transposed V reads are bank-conflict free and the K row reads 2-way.  The DMA
// destination is linear in the lane, so the swizzle is applied to each lane's
// SOURCE address.
//
// Fused KV write (optional): when the new token's K/V rows are handed in, the
// split-0 workgroup copies them into the slab, and every DMA whose page-table
// slot equals the row's new slot reads the NEW rows instead of the slab, so the
// write needs no earlier kernel and no ordering against the gather.
//
// Balanced splits (with the fused merge): pieces have one batch-wide target length
// -- about sum(seq_lens) / (rows * target) tokens, and at least max(seq_lens) /
// n_splits -- so a long request is cut into more pieces than a short one and every
// workgroup streams about the same bytes.  The grid is FLAT: workgroup b is piece
// b / head_groups of the batch's piece list, which each workgroup rebuilds for
// itself from seq_lens (sum, max, running piece counts: no host read, no extra
// launch).  Consecutive workgroups are therefore all live -- idle workgroups
// scattered through a launch make the hardware's round-robin placement pile the
// live ones onto a subset of the CUs, which costs up to 50 % on ragged batches --
// and only the few past the end of the list leave at once.
//
// Fused merge (when the sequence is split): a workgroup stores its normalised
// partial output and log-sum-exp write-through (sc1, 16 bytes per lane via an
// LDS transpose), drains its stores, then draws a ticket with one agent-scope
// atomic add on the (row, head group) counter.  The workgroup that draws the
// last ticket reads every partial back with sc1 loads (they bypass its L1),
// merges them and writes the final rows, then stores 0 to the counter: counters
// are zero before and after every launch.  (Drawing the ticket early and letting
// the last drawer wait for a "ready" count was measured and is slower: under
// load every trip to memory costs about 2.3 us, and the wait adds trips.)
#include "bsgl_device.h"
#include "bsgl_launch.h"
#include "bsgl_mfma.h"

#include <type_traits>

namespace bsgl {

namespace {

using namespace mfma;

// x / d for 0 <= x < 2^22, 0 < d < 2^22, exact: one reciprocal, one multiply and a step either
// way (the product is within one of the quotient).  The compiler's 32-bit division is three times
// as many instructions, and the piece search below is a chain of them on a workgroup's way to
// its first K/V request.
__device__ __forceinline__ int small_div(int x, int d) {
  int q = (int)((float)x * __builtin_amdgcn_rcpf((float)d));
  const int r = x - q * d;
  q += r >= d ? 1 : 0;
  q -= r < 0 ? 1 : 0;
  return q;
}

// Two waves per SIMD (8 single-wave workgroups per CU, which is also what the LDS rings
// allow): 256 registers per lane, which bounds how many partial rows the merge reads at once.
// W = waves per workgroup.  1: a piece is one wave's.  2 or 4 (launches with too few pieces to
// give every SIMD a wave): the waves of a workgroup take the piece's tiles in turn, each through a
// ring of its own -- a wave's tile rate is set by its 8 LDS-DMA instructions per tile, so W waves
// stream a piece W times as fast -- and wave 0 adds up their (m, l, O) in LDS before the piece's
// result is stored, so the number of parts the split merge sees does not grow.
template <typename Tag, int HPT, int NSTAGE, bool NT, int W = 1, bool TRACE = false>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_waves_per_eu(2)))
void decode_mfma_kernel(DecodeArgs a) {
  using E = Elem<Tag>;
  constexpr int D = 128;
  constexpr int REC = HPT * 512;                // bytes of one token record in LDS
  constexpr int TILE_BYTES = kTile * REC;
  constexpr int DMA_PER_TILE = TILE_BYTES / 1024;
  constexpr int TOK_PER_DMA = 1024 / REC;       // 1 (HPT 2) or 2 (HPT 1)
  constexpr int RING = NSTAGE * TILE_BYTES;
  static_assert(HPT == 1 || HPT == 2, "");
  static_assert((NSTAGE - 1) * DMA_PER_TILE < 64, "vmcnt is a 6-bit counter");

  static_assert(W == 1 || RING >= (HPT * 34) * 64 * 4, "a wave's state is handed over through its ring");
  __shared__ __attribute__((aligned(1024))) char smem_all[W * RING + kSlotBlock * 4];
  int32_t* s_slots = (int32_t*)(smem_all + W * RING);
  const int lane = threadIdx.x & 63;
  const int wave = W == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  char* smem = smem_all + wave * RING;  // this wave's ring
  const uint32_t lds_ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  uint64_t* trace = nullptr;
  uint64_t trace_cyc0 = 0;  // (timeline build) the core-clock count where streaming starts
  uint32_t trace_xcc = 0;
  if (TRACE && a.trace != nullptr && wave == 0) {
    trace = a.trace + 8 * ((uint64_t)blockIdx.x + (uint64_t)gridDim.x * (blockIdx.y + (uint64_t)gridDim.y * blockIdx.z));
    uint32_t xcc, hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    trace_xcc = xcc;
    if (lane == 0) {
      trace[0] = wall_clock64();
      trace[1] = trace[2] = trace[3] = 0;
      trace[4] = xcc;
      trace[5] = hw;
      trace[6] = trace[7] = ~0ull;
    }
  }
  const int n_hg = a.n_kv_heads / HPT;  // head groups
  int row = blockIdx.x, hg = blockIdx.y, split = blockIdx.z, n_parts = a.n_splits;
  int flat_req = -1, flat_seq = 0;  // the row's page-table row and length, when found by the flat search
  // flat grid: lane i's request length and page-table row, requested first so that the
  // lane-only arithmetic below runs while they travel
  auto len_of = [&](int i) {
    const int64_t s64 = i < a.rows ? a.seq_lens[i] : 0;
    const int64_t cap = a.kv.table_cols < (1 << 20) ? a.kv.table_cols : (1 << 20);
    return (int)(s64 < 0 ? 0 : (s64 > cap ? cap : s64));
  };
  auto req_of = [&](int i) {
    const int64_t r64 = i < a.rows ? a.req_idx[i] : 0;
    return (int)(r64 < 0 ? 0 : (r64 >= a.kv.table_rows ? a.kv.table_rows - 1 : r64));
  };
  int len0 = 0, req0 = 0;
  if (a.balance_target > 0) {
#ifdef BSGL_CUT_KNOWN_LEN
    // timing experiment (tools/phase_cut.sh, CUTS=KNOWN_LEN=2048; only right for a batch of that one
    // length on page-table rows 0, 1, ...): what a piece list handed in from outside could save at
    // most -- no trip to memory for the batch's lengths before the first block's entries are asked for
    len0 = lane < a.rows ? BSGL_CUT_KNOWN_LEN : 0;
    req0 = lane;
#else
    len0 = len_of(lane);
    req0 = req_of(lane);
#endif
  }
  const int lq = lane & 15;   // q head (column of S^T / row of O) or d column
  const int lg = lane >> 4;   // lane group
  // ---- per-lane LDS read offsets -----------------------------------------------------
  // K rows (A operand): lane (t = lq, g = lg) reads chunk 4 s + g of token t
  const int k_rec = rec_of_token(lq);
  int k_off[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) k_off[s] = k_rec * REC + (((4 * s + lg) ^ xr_of_rec(k_rec)) << 4);
  // V blocks (B operand, transposed read): lane 4 q + p of group g addresses token 4 g + q,
  // 8-byte piece p of the 32 bytes of d-tile n
  const int v_rec = rec_of_token(4 * lg + ((lane >> 2) & 3));
  int v_off[8];
#pragma unroll
  for (int n = 0; n < 8; ++n)
    v_off[n] = v_rec * REC + HPT * 256 + (((2 * n + ((lane >> 1) & 1)) ^ xr_of_rec(v_rec)) << 4) + ((lane & 1) << 3);

  if (a.balance_target > 0) {
    // ---- flat grid: which (row, piece) is this? -------------------------------------
    // (head group fastest: the workgroups of a piece's kv heads run side by side and read the
    // same DRAM pages together; measured 1 % faster than all pieces of a head group first, and
    // equal to putting the two in one two-wave workgroup)
    // The head groups take the launch indices in DESCENDING order (group n_hg - 1 first).  Two
    // things were measured on every launch of the headline batch, each worth about 1.2 us of
    // streaming time: workgroups on odd launch indices (dealt to the odd XCDs) stream slower
    // than those on even ones, and so do the workgroups of head group 1 compared with head
    // group 0, whichever index they have.  In ascending order the two fall on the same
    // workgroups; in descending order they cancel (1.2 % of the step).  Placement only.
    // (the flat grid has at most rows * (target + 1) * n_hg <= 1024 * 65 * n_hg workgroups: small_div's range)
    const int piece = __builtin_amdgcn_readfirstlane(small_div((int)blockIdx.x, n_hg));
    hg = n_hg - 1 - ((int)blockIdx.x - piece * n_hg);
    // per-lane request data, 64 rows at a time; the first block was loaded above, once for both passes
    // Equal lengths (every lane of a batch of at most 64 rows holds the same length): the answer
    // in closed form.  The most pieces wanted per request never exceed n_splits, so the piece
    // length is ceil(len / target) rounded up to tiles, every request yields at most `target`
    // pieces, and the search below would stop at its first pass with exactly this result.
    const int len_first = __builtin_amdgcn_readfirstlane(len0);
    const bool equal_lens = a.rows <= 64 && a.balance_target <= a.n_splits && len_first < (1 << 22) &&
                            __all(lane >= a.rows || len0 == len_first);
    if (equal_lens) {
      int piece_len = small_div(len_first + a.balance_target - 1, a.balance_target);
      // (the finer cut is for launches of few pieces, which share a piece among several waves: the
      // one-wave form does not carry its code -- the headline launch was 0.4 % slower with it, for
      // no more than a handful of scalar instructions: this kernel sits at its scalar-register limit)
      if constexpr (W > 1) {
        if (a.balance_target_hi > a.balance_target)
          piece_len = max(small_div(len_first + a.balance_target_hi - 1, a.balance_target_hi), min(kFinePiece, piece_len));
      }
      if (len_first <= a.whole_len) piece_len = len_first;
      piece_len = max(kTile, (piece_len + kTile - 1) / kTile * kTile);
      const int parts = __builtin_amdgcn_readfirstlane(min(a.n_splits, max(1, small_div(len_first + piece_len - 1, piece_len))));
      row = __builtin_amdgcn_readfirstlane(small_div(piece, parts));
      if (row >= a.rows) return;  // past the batch's last piece
      split = piece - row * parts;
      n_parts = parts;
      flat_seq = len_first;
      flat_req = __builtin_amdgcn_readlane(req0, row);
    } else {
    int tot = 0, longest = 0;
    for (int r0 = 0; r0 < a.rows; r0 += 64) {
      const int s = r0 == 0 ? len0 : len_of(r0 + lane);
      tot += __builtin_amdgcn_readlane(wave_inclusive_scan<ScanAdd>(s), 63);
      longest = max(longest, __builtin_amdgcn_readlane(wave_inclusive_scan<ScanMax>(s), 63));
    }
    int want = a.rows * a.balance_target;
    int piece_len = max((tot + want - 1) / want, small_div(longest + a.n_splits - 1, a.n_splits));  // (lengths are capped at 2^20)
    if constexpr (W > 1) {
      if (a.balance_target_hi > a.balance_target) {
        want = a.rows * a.balance_target_hi;
        piece_len = max((tot + want - 1) / want, min(kFinePiece, piece_len));
      }
    }
    piece_len = max(kTile, (piece_len + kTile - 1) / kTile * kTile);
    // Rounding up per request can leave up to `rows` pieces more than wanted: a few CUs would
    // then hold one workgroup more than the others and finish a quarter later (ragged batches:
    // 18 % on the step).  The search below also counts the batch's pieces; while there are more
    // than `want`, the pieces are lengthened and the search repeated.
    for (int it = 0; it < 4; ++it) {
      int base = 0;
      row = -1;
      for (int r0 = 0; r0 < a.rows; r0 += 64) {
        const int s = r0 == 0 ? len0 : len_of(r0 + lane);
        const int parts = r0 + lane >= a.rows ? 0 : s <= a.whole_len ? 1 : min(a.n_splits, max(1, small_div(s + piece_len - 1, piece_len)));
        const int incl = wave_inclusive_scan<ScanAdd>(parts);
        const int blk_total = __builtin_amdgcn_readlane(incl, 63);
        if (row < 0 && piece < base + blk_total) {
          const int excl = base + incl - parts;
          const unsigned long long hit = __ballot(piece >= excl && piece < excl + parts);  // one lane: parts >= 1
          const int src = __ffsll((long long)hit) - 1;
          row = r0 + src;
          split = piece - __builtin_amdgcn_readlane(excl, src);
          n_parts = __builtin_amdgcn_readlane(parts, src);
          flat_seq = __builtin_amdgcn_readlane(s, src);
          flat_req = __builtin_amdgcn_readlane(r0 == 0 ? req0 : req_of(r0 + lane), src);
        }
        base += blk_total;
      }
      if (base <= want || it == 3) break;
      piece_len = (int)(((int64_t)piece_len * base + want - 1) / want);
      piece_len = (piece_len + kTile - 1) / kTile * kTile;
    }
    if (row < 0) return;  // past the batch's last piece
    }
  }
  const int head0 = hg * HPT;
  const int G = __builtin_amdgcn_readfirstlane(small_div(a.n_q_heads, a.n_kv_heads));
  // Folded heads (fold_group > 0): the caller's model has two kv heads of 64 values; their rows
  // stand side by side in a slab row, so the pair is streamed as ONE 128-wide kv head serving all
  // q heads.  q head h belongs to kv head h / fold_group: its query is zero outside that half of
  // the row (the scores are then exactly the 64-wide ones) and only that half of its output row
  // is kept.  Partial results stay 128 wide.
  const int fold = a.fold_group;

  // ---- this row's token range (multiples of the tile so splits start on tiles) ----
  int64_t req;
  int seq;
  if (a.balance_target > 0) {
    req = flat_req;
    seq = flat_seq;
  } else {
    req = a.req_idx[row];
    req = req < 0 ? 0 : (req >= a.kv.table_rows ? a.kv.table_rows - 1 : req);
    const int64_t seq64 = a.seq_lens[row];
    seq = (int)(seq64 < 0 ? 0 : (seq64 > a.kv.table_cols ? a.kv.table_cols : seq64));
  }
  // the row's pieces are equal among themselves (a piece past the end is empty)
  int chunk = a.balance_target > 0 ? small_div(seq + n_parts - 1, n_parts) : (seq + n_parts - 1) / n_parts;  // (flat grid: seq <= 2^20)
  chunk = (chunk + kTile - 1) / kTile * kTile;
  const int start = split * chunk;
  const int end = (split == n_parts - 1) ? seq : min(seq, start + chunk);
  const int32_t* table_row = a.kv.table + req * a.kv.table_row_stride;
  const uint32_t max_slot = (uint32_t)(a.kv.n_slots - 1);
  const uint32_t row_units = (uint32_t)(a.kv.slab_row_stride * 2 / 16);

  // ---- page-table entries of a block: requested (all loads together, clamped to the block so
  // that nothing branches on a lane's own index between them), later turned into row offsets
  // and written to LDS.  As a loop of load - write pairs the compiler waits for every load
  // before it issues the next: up to 8 memory round trips at the head of a workgroup.
  static_assert(kSlotBlock == 512, "entries per lane below");
  constexpr int NE = 8 / W;  // entries per lane of a full block
  const int tid = lane + 64 * wave;
  auto request_entries = [&](int blk, int n_blk, uint32_t (&ent)[NE]) {
#pragma unroll
    for (int j = 0; j < NE; ++j)
      if (64 * W * j < n_blk) ent[j] = (uint32_t)table_row[blk + min(tid + 64 * W * j, n_blk - 1)];
  };
  // The FIRST block's entries are requested here, together with the query fragments and the
  // new token's slot and rows below, so that everything this workgroup needs before its first
  // K/V request comes back in ONE memory round trip (requested inside the block loop they
  // left only after the new slot had arrived: a second trip, 0.7 us of a 29 us launch).
  uint32_t ent[NE];
  if (start < end) request_entries(start, min(kSlotBlock, end - start), ent);

  // ---- Q^T fragments: lane (q = lq, g = lg) holds Q[q][32 s + 8 g .. +8], zero for pad columns
  uint4 qf[HPT][4];
  {
    // One predicated load per fragment, address and predicate chosen by selects: written as
    // branches on `fold` the compiler put a wait for ALL pending loads in front of every
    // second fragment load (the branches' stores to the fragment registers met at a join),
    // i.e. two more memory round trips before the first K/V request.
    const char* qrow = (const char*)a.q + (int64_t)row * a.q_row_stride * 2;
    const int my_half = fold != 0 ? lq / fold : 0;  // folded heads: the half of the 128-wide row q head lq owns
#pragma unroll
    for (int hh = 0; hh < HPT; ++hh)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const int col = 32 * s + 8 * lg;  // first of this lane's 8 values of the row
        // folded heads: q head lq's 64 values stand in the half of the row that its kv head owns
        const int64_t plain = ((int64_t)(head0 + hh) * G + lq) * D + col;
        const int64_t folded = (int64_t)lq * 64 + (col & 63);
        const bool want_it = lq < G && (fold == 0 || (col >> 6) == my_half);
        const uint4* src = (const uint4*)(qrow + (fold == 0 ? plain : folded) * 2);
        qf[hh][s] = make_uint4(0, 0, 0, 0);
        if (want_it) qf[hh][s] = *src;
      }
  }
  const float scale_log2e = a.scale * kLog2e;

  // ---- per-lane DMA source: which piece of a token record this lane moves ----------
  const int piece = lane % (REC / 16);             // 16-byte piece inside the record
  const bool piece_is_v = piece >= HPT * 16;
  const int piece_head = (piece >> 4) % HPT;
  const int piece_chunk = piece & 15;              // chunk position in LDS
  const int lane_tok = TOK_PER_DMA == 2 ? (lane >> 5) : 0;  // which token of a DMA instruction
  // Source address of this lane's piece for slot 0 and record 0.  Slab rows are
  // 256-byte aligned (host check), so bits 7:4 of a piece address ARE its chunk
  // index and the record's swizzle is one XOR on the finished address; for two
  // tokens per instruction the odd token's record adds the lane-constant 4.
  const uint64_t piece_off = (uint64_t)(head0 + piece_head) * (D * 2) + (uint64_t)((piece_chunk ^ (lane_tok << 2)) << 4);
  const uint64_t lane_src = (uint64_t)(piece_is_v ? a.kv.v_slab : a.kv.k_slab) + piece_off;

  // ---- fused KV write: this row's new token ------------------------------------------
  constexpr uint32_t kNewTok = 0xffffffffu;        // row-offset value that marks "read the new rows"
  const bool fuse_kv = a.new_loc != nullptr;
  uint32_t new_slot = kNewTok;
  uint64_t lane_new_src = 0;
  if (fuse_kv) {
    const uint64_t new_row = piece_is_v ? (uint64_t)a.v_new + (uint64_t)row * a.v_new_stride * 2
                                        : (uint64_t)a.k_new + (uint64_t)row * a.k_new_stride * 2;
    lane_new_src = new_row + piece_off;
    // (the new rows are requested before the slot is, and the slot is looked at after both:
    // one round trip)
    const bool writer = split == 0 && wave == 0 && lane < REC / 16;
    const uint64_t o = (uint64_t)(head0 + piece_head) * (D * 2) + (uint64_t)(piece_chunk << 4);
    uint4 val = make_uint4(0, 0, 0, 0);
    if (writer) val = *(const uint4*)(new_row + o);
    const int32_t loc = a.new_loc[row];
    if (loc >= 0 && (uint32_t)loc <= max_slot) new_slot = (uint32_t)loc;
    if (writer && new_slot != kNewTok) {
      // copy this head group's new K and V rows into the slab (plain chunk order)
      char* slab = piece_is_v ? (char*)a.kv.v_slab : (char*)a.kv.k_slab;
      *(uint4*)(slab + (uint64_t)new_slot * row_units * 16 + o) = val;
    }
  }

  if (TRACE && trace != nullptr) trace_cyc0 = __builtin_readcyclecounter();
  if (trace != nullptr && lane == 0) {
    trace[1] = wall_clock64();
    trace[6] = (uint64_t)row | ((uint64_t)hg << 32);
    trace[7] = (uint64_t)split | ((uint64_t)n_parts << 32);
  }
  float m_run[HPT], l_run[HPT];
  f32x4 o_acc[HPT][8];
#pragma unroll
  for (int hh = 0; hh < HPT; ++hh) {
    m_run[hh] = kNegInf;
    l_run[hh] = 0.f;
#pragma unroll
    for (int n = 0; n < 8; ++n) o_acc[hh][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const bool direct = n_parts == 1 && a.out_dtype == a.dtype;
  const bool fused_merge = !direct && a.counters != nullptr;
  for (int blk = start; blk < end; blk += kSlotBlock) {
    const int n_blk = min(kSlotBlock, end - blk);
    const int n_tiles = (n_blk + kTile - 1) / kTile;
    // The block's page-table entries (the first block's were requested before the query
    // fragments): row offsets into LDS.
    if (W > 1) __syncthreads();  // every wave is done with the previous block's entries
    if (blk != start) request_entries(blk, n_blk, ent);
#pragma unroll
    for (int j = 0; j < NE; ++j) {
      if (tid + 64 * W * j < n_blk) {
        const uint32_t e = ent[j] > max_slot ? max_slot : ent[j];
        // row offset of the slot in 16-byte units (host checked that it stays below 2^32 - 1)
        s_slots[tid + 64 * W * j] = (int32_t)(e == new_slot ? kNewTok : e * row_units);
      }
    }
    // (same wave: the LDS writes above are ordered before the LDS reads below)
    // Retire every load the compiler knows about (q fragments, page-table entries)
    // HERE, with the builtin it models: inside the tile loop it must find nothing
    // of its own to wait for, or its vmcnt(0) would also drain the DMA prefetch.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) only
    if (W > 1) __syncthreads();  // the entries were written by all waves

    // (a wave's tiles: wave, wave + W, ...; its k-th tile goes through stage k % NSTAGE of its ring)
    auto issue_tile = [&](int tile) {
      const uint32_t stage = lds_ring + (uint32_t)((tile / W) % NSTAGE) * TILE_BYTES;
      // lane j holds the row offset of token j of the tile; tokens past the end
      // re-read the last valid one (finite data, weight 0)
      const uint32_t tok_off = (uint32_t)s_slots[min(tile * kTile + (lane & 15), n_blk - 1)];
#pragma unroll
      for (int i = 0; i < DMA_PER_TILE; ++i) {
        const int t0 = i * TOK_PER_DMA;  // first token of this instruction
        uint32_t off = __builtin_amdgcn_readlane(tok_off, t0);
        if constexpr (TOK_PER_DMA == 2) {
          const uint32_t off1 = __builtin_amdgcn_readlane(tok_off, t0 + 1);
          off = lane < 32 ? off : off1;
        }
        uint64_t src = off == kNewTok ? lane_new_src : ((uint64_t)off << 4) + lane_src;
        src ^= (uint64_t)(xr_of_rec(rec_of_token(t0)) << 4);
        dma_1k<NT>((const void*)src, __builtin_amdgcn_readfirstlane(stage + rec_of_token(t0) * REC));
      }
    };

#pragma unroll
    for (int t = 0; t < NSTAGE - 1; ++t)
      if (wave + t * W < n_tiles) issue_tile(wave + t * W);

    for (int tile = wave; tile < n_tiles; tile += W) {
      const int ahead = (n_tiles - 1 - tile) / W;  // this wave's tiles after this one (capped below)
      if (ahead >= NSTAGE - 1) {
        issue_tile(tile + (NSTAGE - 1) * W);
        dma_wait<(NSTAGE - 1) * DMA_PER_TILE>();
      } else if (NSTAGE > 3 && ahead == 2) {
        dma_wait<2 * DMA_PER_TILE>();
      } else if (NSTAGE > 2 && ahead == 1) {
        dma_wait<1 * DMA_PER_TILE>();
      } else {
        dma_wait<0>();
      }
      const char* st = smem + ((tile / W) % NSTAGE) * TILE_BYTES;
      const int n_valid = n_blk - tile * kTile;  // >= 1

#pragma unroll
      for (int hh = 0; hh < HPT; ++hh) {
        f32x4 sc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const uint4 kf = *(const uint4*)(st + hh * 256 + k_off[s]);
          sc = Mma<Tag>::qk(kf, qf[hh][s], sc);
        }
        // lane (q = lq, g = lg) now holds S^T[token 4 g + r][q], r = 0..3
        float sv[4];
        float tmax = kNegInf;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sv[r] = (4 * lg + r < n_valid) ? sc[r] * scale_log2e : kNegInf;
          tmax = fmaxf(tmax, sv[r]);
        }
        tmax = quad_group_max(tmax);
        if (__any(tmax > m_run[hh])) {
          const float m_new = fmaxf(m_run[hh], tmax);
          const float alpha = fast_exp2(m_run[hh] - m_new);  // m_new is finite: the tile has a valid token
          m_run[hh] = m_new;
          l_run[hh] *= alpha;
          // O rows are q = 4 g + r on lane group g: fetch that q's factor
          float ar[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) ar[r] = __shfl(alpha, 4 * lg + r);
#pragma unroll
          for (int n = 0; n < 8; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) o_acc[hh][n][r] *= ar[r];
        }
        float pr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          pr[r] = fast_exp2(sv[r] - m_run[hh]);
          l_run[hh] += pr[r];
        }
        // P as the A operand of the 16x16x16 product: lane (q, g) holds P[q][token 4 g + j]
        const PFrag<Tag> pf(pr);
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          const s16x4_t vf = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) s16x4_t*)(st + hh * 256 + v_off[n]));
#pragma unroll
          for (int k = 0; k < PFrag<Tag>::N; ++k) o_acc[hh][n] = Mma<Tag>::pv(pf.f[k], vf, o_acc[hh][n]);
        }
      }
    }
  }

  if constexpr (W > 1) {
    // ---- the workgroup's waves: states to wave 0 through LDS (each wave's own ring, idle now) --
    // state of a wave and kv head: 34 floats per lane, float k of lane l at [k][l]
    // (a wave's ring is its own and it has waited for its last tile)
    if (wave != 0) {
      float* stf = (float*)smem;
#pragma unroll
      for (int hh = 0; hh < HPT; ++hh) {
        stf[(hh * 34 + 0) * 64 + lane] = m_run[hh];
        stf[(hh * 34 + 1) * 64 + lane] = l_run[hh];
#pragma unroll
        for (int n = 0; n < 8; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) stf[(hh * 34 + 2 + 4 * n + r) * 64 + lane] = o_acc[hh][n][r];
      }
    }
    __syncthreads();
    if (wave != 0) return;
    for (int w = 1; w < W; ++w) {
      const float* stf = (const float*)(smem_all + w * RING);
#pragma unroll
      for (int hh = 0; hh < HPT; ++hh) {
        const float mw = stf[(hh * 34 + 0) * 64 + lane], lw = stf[(hh * 34 + 1) * 64 + lane];
        const float m_new = fmaxf(m_run[hh], mw);
        // (a wave that saw no token has m = -inf and nothing to add: factor 0, never exp2(nan))
        const float a0 = m_run[hh] == kNegInf ? 0.f : fast_exp2(m_run[hh] - m_new);
        const float aw = mw == kNegInf ? 0.f : fast_exp2(mw - m_new);
        m_run[hh] = m_new;
        l_run[hh] = l_run[hh] * a0 + lw * aw;
        float a0r[4], awr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          a0r[r] = __shfl(a0, 4 * lg + r);  // O rows are q = 4 g + r on lane group g
          awr[r] = __shfl(aw, 4 * lg + r);
        }
#pragma unroll
        for (int n = 0; n < 8; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            o_acc[hh][n][r] = o_acc[hh][n][r] * a0r[r] + stf[(hh * 34 + 2 + 4 * n + r) * 64 + lane] * awr[r];
      }
    }
  }
  if (trace != nullptr && lane == 0) {
    trace[2] = wall_clock64();
    // core-clock cycles of the streaming phase above the XCC id: cycles / wall time = the clock this XCD ran at
    if (TRACE) trace[4] = (uint64_t)(trace_xcc & 0xffu) | ((__builtin_readcyclecounter() - trace_cyc0) << 8);
  }
#ifdef BSGL_CUT_AFTER_STREAM
  // timing experiment (tools/phase_cut.sh; results are wrong): leave after the last tile.  The
  // never-true store keeps the streaming loop alive.
  if (o_acc[0][0][0] == 12345.f) a.ws_o[0] = l_run[0];
  return;
#endif
  // ---- finish: sum l over the lane groups, normalise, write ---------------------------
  const int64_t n_rh = (int64_t)a.rows * a.n_q_heads;
  constexpr int kOPad = D + 4;
  const auto ws_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ws_o + ((int64_t)split * n_rh + (int64_t)row * a.n_q_heads) * D), 0,
                                                         (int)(a.n_q_heads * D * 4), 0x00020000);
#pragma unroll
  for (int hh = 0; hh < HPT; ++hh) {
    float* o_lds = (float*)smem;  // the ring is idle now: [16 q][kOPad] floats
    const float l = quad_group_sum(l_run[hh]);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    const float lse = l > 0.f ? (m_run[hh] + __builtin_amdgcn_logf(l)) * kLn2 : kNegInf;
    float ir[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ir[r] = __shfl(inv, 4 * lg + r);
    const int hq0 = (head0 + hh) * G;
    if (direct) {
      if (lg == 0 && lq < G && a.lse != nullptr) a.lse[(int64_t)row * a.n_q_heads + hq0 + lq] = lse;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int q = 4 * lg + r;
        if (q < G) {
          if (fold == 0) {
            char* orow = (char*)a.out + ((int64_t)row * a.out_row_stride + (int64_t)(hq0 + q) * D) * 2;
#pragma unroll
            for (int n = 0; n < 8; ++n) *(uint16_t*)(orow + (16 * n + lq) * 2) = (uint16_t)E::pack2(o_acc[hh][n][r] * ir[r], 0.f);
          } else {
            char* orow = (char*)a.out + ((int64_t)row * a.out_row_stride + (int64_t)q * 64) * 2;
            const int half = q / fold;
#pragma unroll
            for (int n = 0; n < 8; ++n)
              if (n / 4 == half) *(uint16_t*)(orow + (16 * (n % 4) + lq) * 2) = (uint16_t)E::pack2(o_acc[hh][n][r] * ir[r], 0.f);
          }
        }
      }
      continue;
    }
    // partial state: rows through LDS so that every lane stores (or later reads) whole rows
#pragma unroll
    for (int n = 0; n < 8; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) o_lds[(4 * lg + r) * kOPad + 16 * n + lq] = o_acc[hh][n][r] * ir[r];
    if (lg == 0 && lq < G) {
      float* pl = a.ws_lse + (int64_t)split * n_rh + (int64_t)row * a.n_q_heads + hq0 + lq;
      if (fused_merge) __hip_atomic_store(pl, lse, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else *pl = lse;
    }
    // (folded heads: only the 64 values of the head's own kv head are kept, head h at floats
    // [64 h, 64 h + 64) of the request's block, so that two heads make one 128-wide row for the merge)
    for (int idx = lane; idx < G * (D / 4); idx += 64) {
      const int q = idx / (D / 4), c4 = idx % (D / 4);
      if (fold != 0 && c4 / (D / 8) != q / fold) continue;
      const f32x4 v = *(const f32x4*)(o_lds + q * kOPad + c4 * 4);
      // aux 16 = sc1: write-through, so the merging workgroup's sc1 loads see the bytes
      const int off = fold == 0 ? ((hq0 + q) * D + c4 * 4) * 4 : ((hq0 + q) * (D / 2) + (c4 % (D / 8)) * 4) * 4;
      if (fused_merge) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), ws_rsrc, off, 0, 16);
      else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, v), ws_rsrc, off, 0, 0);
    }
  }
  if (!fused_merge) {
    if (trace != nullptr && lane == 0) trace[3] = wall_clock64();
    return;
  }
  // ---- ticket: the last workgroup of this (row, head group) merges all splits ----------
#ifdef BSGL_CUT_AFTER_PARTIALS
  return;  // timing experiment (tools/phase_cut.sh; results are wrong): no ticket, no merge
#endif
#ifndef BSGL_CUT_NO_DRAIN  // (timing experiment, tools/phase_cut.sh: results may be wrong without the wait)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every partial byte of this wave has left
#endif
  uint32_t* counter = a.counters + (int64_t)row * n_hg + hg;
  uint32_t ticket = 0;
  if (lane == 0) ticket = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  ticket = __builtin_amdgcn_readfirstlane(ticket);
  if (ticket != (uint32_t)n_parts - 1) {
    if (trace != nullptr && lane == 0) trace[3] = wall_clock64();
    return;
  }

  // All partial bytes were stored sc1 and drained before their tickets; read them
  // back with sc1 loads only (aux 16), which bypass this CU's vector cache.
  // Lane p owns part p's weight (n_splits <= 64); loads of up to 6 heads x 8 parts
  // are issued together so the merge costs two memory round trips, not one per head.
  const int nh = HPT * G;
  const int64_t rh0 = (int64_t)row * a.n_q_heads + (int64_t)head0 * G;
  const auto lse_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ws_lse + rh0), 0, (int)(((int64_t)(n_parts - 1) * n_rh + nh) * 4), 0x00020000);
  const auto o_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(a.ws_o + rh0 * D), 0, (int)((((int64_t)(n_parts - 1) * n_rh + nh) * D) * 4), 0x00020000);
  const int part_lse_stride = (int)(n_rh * 4);
  const int part_o_stride = (int)(n_rh * D * 4);
  // A "row" of the merge is 128 floats of a part: one q head, or (folded heads) two q heads of 64
  // values, the lower 32 lanes on the first of them.  PAIR = heads per row.
  // Partial rows come in 16 bytes per lane: the lower half of the wave reads the even parts of a
  // round, the upper half the odd ones (a load instruction costs the wave some 80 cycles to issue
  // whatever its width, so the merge is paid in instructions), and the two halves' sums are
  // added at the end.
  // The weights: lane (head k of a load, part p) = P2 * k + p holds lse and weight of its (head,
  // part), P2 = 8 lanes per head for up to 8 parts (64 otherwise), so one
  // load instruction fetches 64 / P2 heads' values and the maxima and sums run inside P2 lanes.
  auto merge_rows = [&](auto pair_c, auto p2_c) {
    constexpr int PAIR = decltype(pair_c)::value;
    constexpr int P2 = decltype(p2_c)::value;
    constexpr int HPI = 64 / P2;  // heads per lse load
    constexpr int kMH = PAIR == 2 ? 7 : 6;  // rows per round (the GQA group of the 1.5B model, the 14 heads of the 0.5B model)
    constexpr int kMP = 8;                  // parts per round; kMH * kMP * 2 registers hold a round's partial rows
    const int n_mrows = nh / PAIR;
    const int hp = lane >> 5;               // parity of the parts this lane reads
    const int c4 = lane & 31;               // its four floats of a row: c4 * 4 ..
    const int sub = PAIR == 2 ? c4 >> 4 : 0;  // which head of the row they belong to
    for (int h0 = 0; h0 < n_mrows; h0 += kMH) {
      constexpr int NL = (kMH * PAIR + HPI - 1) / HPI;  // lse loads of a round
      float my_lse[NL], my_w[NL], mx[NL], wsum[NL];
      // (unconditional loads, clamped to a real head and part: they all leave together)
#pragma unroll
      for (int k = 0; k < NL; ++k) {
        const int h = min(h0 * PAIR + k * HPI + lane / P2, nh - 1);
        my_lse[k] = as_f32(__builtin_amdgcn_raw_buffer_load_b32(lse_rsrc, min(lane % P2, n_parts - 1) * part_lse_stride + h * 4, 0, 16));
      }
      f32x4 acc[kMH];
#pragma unroll
      for (int j = 0; j < kMH; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int p0 = 0; p0 < n_parts; p0 += kMP) {
        u32x4_t v[kMH][kMP / 2];
#pragma unroll
        for (int j = 0; j < kMH; ++j) {
          const int h = min(h0 + j, n_mrows - 1);
#pragma unroll
          for (int i = 0; i < kMP / 2; ++i) {
            const int p = min(p0 + 2 * i + hp, n_parts - 1);
            v[j][i] = __builtin_amdgcn_raw_buffer_load_b128(o_rsrc, p * part_o_stride + (h * D + c4 * 4) * 4, 0, 16);
          }
        }
        if (p0 == 0) {  // (first use of the lse loads: after the partial rows have been requested)
#pragma unroll
          for (int k = 0; k < NL; ++k) {
            my_lse[k] = lane % P2 < n_parts ? my_lse[k] : kNegInf;
            mx[k] = group_max<P2>(my_lse[k]);
            my_w[k] = (my_lse[k] == kNegInf) ? 0.f : fast_exp2((my_lse[k] - mx[k]) * kLog2e);
            wsum[k] = group_sum<P2>(my_w[k]);
          }
        }
#pragma unroll
        for (int j = 0; j < kMH; ++j) {
#pragma unroll
          for (int i = 0; i < kMP / 2; ++i) {
            const int p = min(p0 + 2 * i + hp, n_parts - 1);
            float w = __shfl(my_w[(j * PAIR) / HPI], ((j * PAIR) % HPI) * P2 + p);
            if (PAIR == 2) {
              const float w1 = __shfl(my_w[(j * PAIR + 1) / HPI], ((j * PAIR + 1) % HPI) * P2 + p);
              w = sub ? w1 : w;
            }
            w = p0 + 2 * i + hp < n_parts ? w : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(w, as_f32(v[j][i][e]), acc[j][e]);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < kMH; ++j) {
        if (h0 + j >= n_mrows) break;
        float r4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {  // even parts + odd parts
          float x = acc[j][e], y = acc[j][e];
          swap32(x, y);
          r4[e] = x + y;
        }
        float ws = __shfl(wsum[(j * PAIR) / HPI], ((j * PAIR) % HPI) * P2);
        float m = __shfl(mx[(j * PAIR) / HPI], ((j * PAIR) % HPI) * P2);
        if (PAIR == 2) {
          const float ws1 = __shfl(wsum[(j * PAIR + 1) / HPI], ((j * PAIR + 1) % HPI) * P2);
          const float m1 = __shfl(mx[(j * PAIR + 1) / HPI], ((j * PAIR + 1) % HPI) * P2);
          ws = sub ? ws1 : ws;
          m = sub ? m1 : m;
        }
        const float inv = ws > 0.f ? 1.f / ws : 0.f;
        const int64_t hq = (int64_t)head0 * G + (int64_t)(h0 + j) * PAIR + sub;
        // (PAIR == 2: head hq's 64 values are columns [64 hq, 64 hq + 64) of the output row as well)
        const int64_t col = PAIR == 2 ? hq * 64 + (c4 * 4) % 64 : hq * D + c4 * 4;
        if (hp == 0) {
          if (a.out_dtype == kF32) {
            *(float4*)((char*)a.out + ((int64_t)row * a.out_row_stride + col) * 4) =
                make_float4(r4[0] * inv, r4[1] * inv, r4[2] * inv, r4[3] * inv);
          } else {
            *(uint2*)((char*)a.out + ((int64_t)row * a.out_row_stride + col) * 2) =
                make_uint2(E::pack2(r4[0] * inv, r4[1] * inv), E::pack2(r4[2] * inv, r4[3] * inv));
          }
          if (a.lse != nullptr && (c4 & 15) == 0 && (PAIR == 2 || c4 == 0)) {
            a.lse[rh0 + (h0 + j) * PAIR + sub] = ws > 0.f ? m + __builtin_amdgcn_logf(ws) * kLn2 : kNegInf;
          }
        }
      }
    }
  };
  auto merge_p2 = [&](auto pair_c) {
    // (16 lanes per head for 9 - 16 parts was measured too: slower than one head per load)
    if (n_parts <= 8) merge_rows(pair_c, std::integral_constant<int, 8>{});
    else merge_rows(pair_c, std::integral_constant<int, 64>{});
  };
  if (fold != 0) merge_p2(std::integral_constant<int, 2>{});
  else merge_p2(std::integral_constant<int, 1>{});
  if (lane == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (trace != nullptr && lane == 0) trace[3] = wall_clock64() | (1ull << 63);  // top bit: this one merged
}

uint64_t* g_trace = nullptr;

template <typename Tag, int HPT, int NSTAGE>
hipError_t launch_inst(const DecodeArgs& a, hipStream_t stream) {
  dim3 grid(a.rows, a.n_kv_heads / HPT, a.n_splits);
  // flat list of length-balanced pieces: at most rows * target + rows of them
  if (a.balance_target > 0) grid = dim3((unsigned)((int64_t)a.rows * (a.balance_target + 1) * (a.n_kv_heads / HPT)), 1, 1);
  // (the per-workgroup timeline is a build of its own, kept for the default configuration only:
  // as a run-time flag its stamps cost the untraced launches 0.1 - 0.7 %)
  if (HPT == 1 && NSTAGE == 2 && a.trace != nullptr && knob(kKnobMfmaNT))
    hipLaunchKernelGGL((decode_mfma_kernel<Tag, 1, 2, true, 1, true>), grid, dim3(64), 0, stream, a);
  else if (knob(kKnobMfmaNT))
    hipLaunchKernelGGL((decode_mfma_kernel<Tag, HPT, NSTAGE, true>), grid, dim3(64), 0, stream, a);
  else
    hipLaunchKernelGGL((decode_mfma_kernel<Tag, HPT, NSTAGE, false>), grid, dim3(64), 0, stream, a);
  return hipGetLastError();
}

template <typename Tag>
hipError_t launch_tag(const DecodeArgs& a, hipStream_t stream) {
  // tuning knobs (defaults are the measured best)
  const int stages = knob(kKnobMfmaStages);
  const int hpt = (a.n_kv_heads % 2 == 0) ? knob(kKnobMfmaHPT) : 1;
  if (hpt == 2) {
    if (stages == 3) return launch_inst<Tag, 2, 3>(a, stream);
    return launch_inst<Tag, 2, 2>(a, stream);
  }
  if (stages == 4) return launch_inst<Tag, 1, 4>(a, stream);
  if (stages == 3) return launch_inst<Tag, 1, 3>(a, stream);
  // Waves per workgroup: one where the pieces alone come near a wave per SIMD (1024 on this chip),
  // else four waves share a piece.  (Two waves for 257 - 512 pieces were the rule before; four are
  // 2 - 8 % faster there: 24 requests x 2048 tokens 16.9 -> 15.6 us, 32 requests 17.9 -> 17.6.  Between
  // 512 and 1024 pieces two waves win at 1024 tokens and lose at 2048 and beyond.)
  const int64_t n_pieces = (int64_t)a.rows * a.n_kv_heads * (a.balance_target > 0 ? a.balance_target : a.n_splits);
  int waves = n_pieces > 512 ? 1 : 4;
  // (folded kv heads keep a wave busier per tile -- up to 16 q heads -- and gain from a second wave a
  // little longer: 80 - 96 requests of the 0.5B shapes, 1024 / 2048 / 4095 tokens: -8..-11 % / -2..-6 % / +-1 %)
  if (a.fold_group > 0 && n_pieces > 512 && n_pieces <= 768) waves = 2;
  if (a.balance_target > 0 && (a.hint_waves == 1 || a.hint_waves == 2 || a.hint_waves == 4)) waves = a.hint_waves;
  {
    const int forced = knob(kKnobMfmaWaves);  // tuning override
    if (forced == 1 || forced == 2 || forced == 4) waves = forced;
  }
  if (waves > 1 && knob(kKnobMfmaNT)) {
    dim3 grid(a.rows, a.n_kv_heads, a.n_splits);
    if (a.balance_target > 0) grid = dim3((unsigned)((int64_t)a.rows * (max(a.balance_target, a.balance_target_hi) + 1) * a.n_kv_heads), 1, 1);
    if (waves == 2)
      hipLaunchKernelGGL((decode_mfma_kernel<Tag, 1, 2, true, 2>), grid, dim3(128), 0, stream, a);
    else
      hipLaunchKernelGGL((decode_mfma_kernel<Tag, 1, 2, true, 4>), grid, dim3(256), 0, stream, a);
    return hipGetLastError();
  }
  return launch_inst<Tag, 1, 2>(a, stream);
}

}  // namespace

bool decode_mfma_supported(const DecodeArgs& a, const char** why) {
  if (a.dtype != kBF16 && a.dtype != kF16) {
    *why = "matrix-core decode needs bf16 or f16";
    return false;
  }
  if (a.head_dim != 128) {
    *why = "matrix-core decode needs head_dim 128";
    return false;
  }
  if (a.n_kv_heads <= 0 || a.n_q_heads % a.n_kv_heads != 0 || a.n_q_heads / a.n_kv_heads > 16) {
    *why = "matrix-core decode needs 1..16 query heads per kv head";
    return false;
  }
  const int64_t row_bytes = a.kv.slab_row_stride * 2;
  if (((uintptr_t)a.kv.k_slab | (uintptr_t)a.kv.v_slab) % 256 != 0 || row_bytes % 256 != 0) {
    *why = "matrix-core decode needs 256-byte aligned slab rows";
    return false;
  }
  if (a.kv.n_slots * (row_bytes / 16) >= 0xffffffffll) {
    *why = "slab larger than 64 GiB";
    return false;
  }
  return true;
}

void set_decode_trace(uint64_t* buf) { g_trace = buf; }

hipError_t launch_decode_mfma(const DecodeArgs& a_in, hipStream_t stream) {
  if (a_in.rows == 0) return hipSuccess;
  DecodeArgs a = a_in;
  a.trace = g_trace;
  hipError_t e = a.dtype == kBF16 ? launch_tag<bf16_tag>(a, stream) : launch_tag<f16_tag>(a, stream);
  if (e != hipSuccess || decode_direct_out(a) || a.counters != nullptr) return e;
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

}  // namespace bsgl
