// Matrix-core paged extend (prefill) attention for gfx950 on the 32x32x16 product shape
// (bf16 / f16, head_dim 128, at least two q heads per kv head; blocks of 64 positions, or of 16
// for batches of short extends: see the PW parameter of the kernel).
//
// Same contract, work list, K/V streaming and LDS image as paged_prefill_mfma.hip (reference:
// _prefill_attention, attn_backend.py:34-110); what differs is how a workgroup's 64 positions x
// 2 q heads are dealt to its four waves and which product shape they use.  A 16x16x32 product
// holds the SIMD's vector issue for 8 of its 16 cycles, a 32x32x16 product for 8 of its 32 at
// twice the arithmetic: with the softmax's exponentials, multiply-adds and conversions in the
// gaps, the smaller shape is bound by instruction issue, the larger one by the matrix pipe.
//
// A wave owns 32 consecutive positions of ONE q head (wave & 1: which half of the block,
// wave >> 1: which head).  Per step of 32 tokens:
//   S^T[token][position] = K . Q^T      8 products (k = 16 each over head_dim 128); a lane then
//                                       holds 16 of the 32 scores of ITS position (lane & 31),
//                                       the partner lane (lane ^ 32) the other 16
//   softmax in registers                row maximum = 15 maxima + one half-wave exchange
//   O[position][d] += P . V             8 products: the score accumulator, rounded to the dtype,
//                                       IS the A operand (position on the lane, tokens in the
//                                       registers -- the order of summation over tokens is free
//                                       as long as the V fragment uses the same one)
// The steps are software-pipelined inside the wave (the 16-row kernel's are not): iteration i turns
// scores i into probabilities while the matrix cores compute scores i + 1, then runs PV i.
#include "bsgl_device.h"
#include "bsgl_launch.h"
#include "bsgl_mfma.h"

#include <type_traits>

// Timing cuts of this translation unit (bit mask, tools/build_variant.sh sets it; RESULTS ARE WRONG
// with any of them, they exist to price a phase): 4 = no K/V request inside the step loop, 8 = no
// exponentials / multiply-adds / row sums / maxima, 16 = no LDS read of K / V fragments in the loop,
// 64 = half of a step's K/V requests, 256 = every
// request of the step loop re-reads the rows of the request's FIRST tokens (same instructions and LDS
// writes, but served by the CU's own cache: what the traffic between L2 and the CU costs), 512 = the
// whole K/V stream of the workgroup is run through TWICE (one prologue, one epilogue: the difference
// to the plain launch is what a pass over the stream costs without them), 1024 = no mask and no running-maximum
// check at the head of a step's body (after step 0).
#ifndef BSGL_CUT
#define BSGL_CUT 0
#endif
#ifndef BSGL_T2_MODE
#define BSGL_T2_MODE 1
#endif

namespace bsgl {

namespace {

using namespace mfma;

typedef __attribute__((ext_vector_type(16))) float f32x16;

template <typename Tag> struct Mma32;
template <> struct Mma32<bf16_tag> {
  __device__ static __forceinline__ f32x16 mm(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
};
template <> struct Mma32<f16_tag> {
  __device__ static __forceinline__ f32x16 mm(const uint4& a, const uint4& b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
};

// row of the 32x32 accumulator that register r of a lane in half `hi` holds
__device__ __forceinline__ constexpr int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// PW = positions among a wave's 32 columns.  32: the layout above (64 positions x 2 q heads per
// workgroup).  16: for batches of short extends a wave's columns are 2 q heads x 16 positions and
// the workgroup covers 16 positions x 8 q heads, so a kv head's K/V is streamed by ONE workgroup
// for up to 8 q heads instead of once per pair of heads.
// PROF: the build with cycle stamps (per-phase sums into a.prof).  A build of its own: where a
// stamped and an unstamped path join, the compiler waits for the stamp's value, and that wait
// also waits for every LDS read on its way -- in the unstamped path too.
// TEAMS: 1 = four waves, 64 positions x 2 q heads.  2 = eight waves, TWO such teams on consecutive
// 64-position halves of a 128-position block that share one K/V stream (one workgroup per CU, two
// waves per SIMD): half the K/V requests and half the bytes from L2 per product.
// one block slot of the launch (block_id: its index in the launch order); the kernel below is its only caller
template <typename Tag, int PW, bool PROF, int TEAMS = 1>
__device__ __forceinline__ void prefill_mfma32_item(const PrefillArgs& a, const unsigned block_id) {
  using E = Elem<Tag>;
  static_assert(TEAMS == 1 || (TEAMS == 2 && PW == 32), "teams are halves of a 128-position block");
  constexpr int D = 128;
  constexpr int REC = 512;  // K row | V row of one kv head
  constexpr int WAVES = 4 * TEAMS;
  constexpr int CH = 32 / PW;                    // q heads among a wave's columns
  constexpr int HPW = PW == 32 ? 2 : WAVES * CH;  // q heads per workgroup
  constexpr int ROWS = PW == 32 ? 64 * TEAMS : PW;  // positions per workgroup
  // SUB steps of 32 tokens form a group: the unit of the K/V requests, of the wait for them and of the
  // workgroup barrier.  Four waves: 1 (two workgroups per CU share the LDS).  Eight waves: 2 -- both
  // waves of a SIMD then sit behind the same barrier, so the time around a barrier is time the matrix
  // pipe idles, and the workgroup has the CU's LDS to itself for a ring twice as deep.
  // Eight waves, BSGL_T2_MODE (build switch, default 1): bit 0 = SUB 2, bit 1 = the second team runs
  // half a step behind the first (STAGGER): its wait / barrier / requests of step s + 1 sit between the
  // score half and the PV half of its step s, so that on every SIMD one wave is in its score half
  // (exponentials, K reads) while the other is in its PV half, instead of both in the same half.
  constexpr int SUB = (TEAMS == 2 && (BSGL_T2_MODE & 1)) ? 2 : 1;
  constexpr bool STAGGER = TEAMS == 2 && (BSGL_T2_MODE & 2) != 0 && SUB == 1;
  constexpr int NSTAGE = TEAMS == 2 ? 8 : 4;
  constexpr int TILE_BYTES = kTile * REC;
  constexpr int STEP = 2 * kTile;                  // tokens per step: two tiles
  constexpr int STEP_BYTES = 2 * TILE_BYTES;
  constexpr int DMA_PER_STEP = STEP_BYTES / 1024;  // 16 instructions, two tokens each
  constexpr int DPW = DMA_PER_STEP / WAVES;        // instructions per wave and step
  constexpr int RING = NSTAGE * STEP_BYTES;
  constexpr int kOPad = D + 4;
  constexpr int STAGE_OUT = 32 * kOPad * 4;        // output staging of one wave
  constexpr int BUF = RING > WAVES * STAGE_OUT ? RING : WAVES * STAGE_OUT;
  constexpr int SLOTS = kSlotBlock;
  constexpr int NTHR = 64 * WAVES;
  constexpr int EPT = SLOTS / NTHR;  // page-table entries of a block per thread: 2 (four waves) or 1 (eight)
  static_assert(SLOTS == EPT * NTHR && SLOTS / STEP == 16, "whole entries per thread and block; steps per block as shifted below");

  __shared__ __attribute__((aligned(1024))) char smem[BUF + 2 * SLOTS * 4];
  int32_t* s_slots = (int32_t*)(smem + BUF);
  const uint32_t lds_ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  const int tid = threadIdx.x;
  const uint64_t t_entry = PROF ? __builtin_readcyclecounter() : 0;
  const uint64_t rt_entry = PROF ? __builtin_amdgcn_s_memrealtime() : 0;  // 100 MHz ticks: the clock the chip holds
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lp = lane & 31;  // column of S^T, row of O
  const int pos = lp % PW;   // its position inside the wave's PW ...
  const int hsub = lp / PW;  // ... and its q head among the wave's CH
  const int hi = lane >> 5;
  const int G = a.n_q_heads / a.n_kv_heads;
  const int fold = a.fold_group;  // folded heads: see paged_prefill_mfma.hip

  // ---- which (request, block of 64 positions, kv head, pair of q heads) is this? ------------
  // (the launch order of paged_prefill_mfma.hip: index = 8 * slot + xcd)
  const int n_z = (G + HPW - 1) / HPW;
  const int per_group = a.grp_blocks * n_z;
  const unsigned slot = block_id >> 3;
  const int64_t g = (int64_t)(slot / (unsigned)per_group) * 8 + (block_id & 7);
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
  // (inconsistent metadata gives wrong numbers, not an access outside the tensors)
  const int64_t q_first = a.cu_ext[i_req];
  const int64_t ext_raw = max((int64_t)0, min(a.extended_lens[i_req], (int64_t)a.total_q_tokens - q_first));
  const int n_blocks = (int)((ext_raw + ROWS - 1) / ROWS);
  const int bg0 = (int)(pg - a.cu_blk[i_req]) * a.grp_blocks;
  const int nb = min(a.grp_blocks, n_blocks - bg0);
  const int bi = within / n_z;  // block of the group, counted from the request's end (last blocks first)
  if (bi >= nb) return;
  // The last pair of an odd GQA group holds one q head.  Its workgroups then take TWO blocks of
  // that head each (waves 2, 3 the block before the one of waves 0, 1; the stream is the longer
  // one's) instead of leaving half their waves without work, and every second one of them retires.
  const bool two_blocks = PW == 32 && n_rt == 1;
  if (two_blocks && (bi & 1)) return;
  const int q0_wg = (bg0 + nb - 1 - bi) * ROWS;
  const int rows_wg = (int)min((int64_t)ROWS, ext_raw - q0_wg);
  const int w4 = wave & 3, team = wave >> 2;       // (four waves: team 0)
  const int hw = PW == 32 ? w4 >> 1 : wave * CH;  // first of this wave's q heads among the workgroup's (or its block)
  const int my_bi = bi + (two_blocks ? hw : 0);
  const bool head_ok = two_blocks ? my_bi < nb : hw < n_rt;  // no such head / block: the wave only helps to load
  const int q0_blk = (bg0 + nb - 1 - (head_ok ? my_bi : bi)) * ROWS;
  const int rows_blk = (int)min((int64_t)ROWS, ext_raw - q0_blk);
  const int pos_off = PW == 32 ? 32 * (w4 & 1) + 64 * team : 0;
  const int q0 = q0_blk + pos_off;
  const int n_heads = PW == 32 ? (head_ok ? 1 : 0) : max(0, min(CH, n_rt - hw));  // q heads this wave really has
  const int n_rows = n_heads > 0 ? max(0, min(PW, rows_blk - pos_off)) : 0;       // positions it really has
  const bool col_ok = pos < n_rows && hsub < n_heads;
  const int64_t q_row0 = q_first + q0;
  int64_t req = a.req_idx[i_req];
  req = req < 0 ? 0 : (req >= a.kv.table_rows ? a.kv.table_rows - 1 : req);
  const int64_t seq64 = a.seq_lens[i_req];
  const int seq = (int)(seq64 < 0 ? 0 : (seq64 > a.kv.table_cols ? a.kv.table_cols : seq64));
  const int64_t pre64 = a.prefix_lens[i_req];
  const int prefix = (int)(pre64 < 0 ? 0 : (pre64 > seq ? seq : pre64));
  const int kv_len = min(seq, prefix + q0_wg + rows_wg);
  const int wave_len = n_rows > 0 ? min(kv_len, prefix + q0 + n_rows) : 0;
  const int my_last = min(prefix + q0 + min(pos, max(n_rows, 1) - 1), kv_len - 1);
  const int full_until = min(prefix + q0, kv_len - 1);
  const int32_t* table_row = a.kv.table + req * a.kv.table_row_stride;
  const uint32_t max_slot = (uint32_t)(a.kv.n_slots - 1);
  const uint32_t row_units = (uint32_t)(a.kv.slab_row_stride * 2 / 16);
  const int head0 = kvh * G + rt0 + (two_blocks ? 0 : hw);  // (folded: kvh = 0 and G counts the heads of both halves)
  const int head = head0 + hsub;                            // the q head of this lane's column

  // ---- Q^T fragments: lane (position lp, half hi) holds Q[q_row0 + lp][head][16 kb + 8 hi ..] ----
  uint4 qf[8];
  {
    const char* qrow = (const char*)a.q + (q_row0 + pos) * a.q_row_stride * 2;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      qf[kb] = make_uint4(0, 0, 0, 0);
      if (col_ok) {
        if (fold == 0) {
          qf[kb] = *(const uint4*)(qrow + ((int64_t)head * D + 16 * kb + 8 * hi) * 2);
        } else if ((16 * kb + 8 * hi) / 64 == head / fold) {
          qf[kb] = *(const uint4*)(qrow + ((int64_t)head * 64 + (16 * kb + 8 * hi) % 64) * 2);
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

  // K as the A operand: lane (token lp of the step, half hi) reads K[token][16 kb + 8 hi ..]
  const int k_rec = rec_of_token(lp & 15);
  int k_off[8];
#pragma unroll
  for (int kb = 0; kb < 8; ++kb)
    k_off[kb] = (lp >> 4) * TILE_BYTES + k_rec * REC + (((2 * kb + hi) ^ xr_of_rec(k_rec)) << 4);
  // V as the B operand through transposing reads: a 16-lane group fetches 4 tokens x 16 values
  // (lane i of it: token i >> 2, values 4 (i & 3) ..) and lane i receives value column i of the
  // four tokens.  The group's tokens: 8 jh + 4 hi + 0..3 of the tile, its columns
  // 32 db + 16 (group & 1) ..
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
  float m_run = kNegInf, neg_mc = 0.f;  // reference score of the lane's position, and -(that) * scale * log2(e)
  constexpr float kSlack = 6.f;
  const float slack_raw = kSlack / scale_log2e;
  float l_run = 0.f;  // sum of this lane's probabilities (its 16 tokens of every step)
  f32x16 o_acc[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[db][r] = 0.f;

  // The page-table entries of the sequence are staged SLOTS at a time into the two halves of
  // s_slots; block b + 1 is fetched while block b is being worked on, so the K/V stream never
  // drains between blocks.  Each thread fetches two entries of a block.
  const int n_steps = (kv_len + STEP - 1) / STEP;
  auto entry_ptr = [&](int e) { return table_row + max(min(e, kv_len - 1), 0); };  // entries past the end repeat the last one
  auto entry_off = [&](uint32_t s) { return (int32_t)((s > max_slot ? max_slot : s) * row_units); };
  const uint64_t t_stage = PROF ? __builtin_readcyclecounter() : 0;
  if (PROF) prof[6] = (t_stage - t_entry) & 0xffffffffu;
  {
    uint32_t e[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) e[k] = (uint32_t)*entry_ptr(tid + k * NTHR);
#pragma unroll
    for (int k = 0; k < EPT; ++k) s_slots[tid + k * NTHR] = entry_off(e[k]);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);  // retire every compiler-visible load before the step loop
  __syncthreads();
  // (the entries of block b + 1 come in by LDS-DMA, raw, and are turned into row offsets in place by
  // the threads whose wave brought them: no register ever holds data that is still on its way --
  // an asm load INTO a register counts as complete where its statement ends, and under register
  // pressure the compiler may copy that register before the data lands)
  const uint32_t lds_slots = lds_ring + BUF;

  // a step is loaded by all waves together: wave w issues instructions w*DPW .. w*DPW+DPW-1;
  // instruction i brings tokens 2i, 2i+1 of the step (tile i / 8)
  // (the row offsets of a step are read from s_slots one request early, so that the request
  // itself does not start with an LDS round trip)
  // a group's requests of this wave: request r (0 .. NREQ_ALL-1) belongs to step g * SUB + r / DPW of
  // group g and is instruction wave * DPW + r % DPW of that step
  constexpr int NREQ_ALL = SUB * DPW;
  static_assert(NREQ_ALL == 4 || NREQ_ALL == 2, "four or two requests per wave and group");
  uint32_t off[NREQ_ALL];
  auto read_offsets = [&](int group) {
#pragma unroll
    for (int r = 0; r < NREQ_ALL; ++r) {
      const int step = group * SUB + r / DPW;
      off[r] = (uint32_t)s_slots[((step * STEP) & (2 * SLOTS - 1)) + 2 * (wave * DPW + r % DPW) + lane_tok];
    }
  };
  // The four source addresses are made ahead of the wait and the barrier of the step that issues them
  // (make_src), and the requests then go out as ONE statement, M0 set per request and restored once
  // (four statements with their address arithmetic in between: 78 cycles per request, this form 54).
  constexpr int NREQ = (BSGL_CUT & 64) && NREQ_ALL == 4 ? NREQ_ALL / 2 : NREQ_ALL;
  uint64_t src4[NREQ_ALL];
  uint32_t off0[NREQ_ALL] = {};  // (BSGL_CUT & 256: the offsets of group 0)
  auto make_src = [&]() {
#pragma unroll
    for (int r = 0; r < NREQ_ALL; ++r) {
      const int i = wave * DPW + r % DPW;  // wave-uniform: instruction i brings tokens 2i, 2i+1 of its step (tile i / 8)
      const uint32_t xr = (uint32_t)xr_of_rec(rec_of_token(2 * (i & 7)));
      src4[r] = (((uint64_t)((BSGL_CUT & 256) ? off0[r] : off[r]) << 4) + lane_src) ^ (uint64_t)(xr << 4);
    }
  };
  auto issue_group = [&](int group) {
    uint32_t dst[NREQ_ALL];
#pragma unroll
    for (int r = 0; r < NREQ_ALL; ++r) {
      const int i = wave * DPW + r % DPW;
      const uint32_t stage = lds_ring + (uint32_t)((group * SUB + r / DPW) % NSTAGE) * STEP_BYTES;
      dst[r] = __builtin_amdgcn_readfirstlane(stage + (uint32_t)(i >> 3) * TILE_BYTES + (uint32_t)rec_of_token(2 * (i & 7)) * REC);
    }
    uint32_t keep;
    if constexpr (NREQ == 4) {
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
    } else {
      asm volatile(
          "s_mov_b32 %0, m0\n\t"
          "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
          "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
          "s_mov_b32 m0, %0"
          : "=&s"(keep)
          : "v"(src4[0]), "v"(src4[1]), "s"(dst[0]), "s"(dst[1])
          : "memory");
    }
  };

  if (PROF) prof[5] += (__builtin_readcyclecounter() - t_stage) << 32;
  const int n_groups = (n_steps + SUB - 1) / SUB;
  f32x16 sc_a, sc_b;
  float lmax_nx = kNegInf;
  for (int pass = 0; pass < ((BSGL_CUT & 512) ? 2 : 1); ++pass) {
  if (pass > 0) __syncthreads();
  if (n_groups > 0) {
    read_offsets(0);
    if constexpr ((BSGL_CUT & 256) != 0) {
#pragma unroll
      for (int r = 0; r < NREQ_ALL; ++r) off0[r] = off[r];
    }
    make_src();
    issue_group(0);
  }
  if (n_groups > 1) { read_offsets(1); make_src(); issue_group(1); }
  if (n_groups > 2) { read_offsets(2); make_src(); issue_group(2); }
  if (n_groups > 3) read_offsets(3);

  // sc_a, sc_b: raw scores of the step about to be consumed and of the one after it: the two trade
  // places from step to step (a copy would wait for the last score product)
  // lmax_nx: largest of the lane's 16 raw scores of the step that comes next, found in the shadow of that
  // step's predecessor's PV products (steps that need the mask look again, after masking)
  auto max16 = [&](const f32x16& v) {
    float m = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
    return fmaxf(m, fmaxf(fmaxf(fmaxf(v[8], v[9]), fmaxf(v[10], v[11])), fmaxf(fmaxf(v[12], v[13]), fmaxf(v[14], v[15]))));
  };
  auto load_k = [&](uint4 (&kf)[8], const char* st) {
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) kf[kb] = *(const uint4*)(st + k_off[kb]);
  };
  auto scores = [&](const uint4 (&kf)[8]) {
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) s = Mma32<Tag>::mm(kf[kb], qf[kb], s);
    return s;
  };
  {
    if (n_groups > 2) dma_wait<2 * NREQ>(); else if (n_groups > 1) dma_wait<NREQ>(); else dma_wait<0>();
    __syncthreads();
    if (0 < wave_len) {
      uint4 kf[8];
      load_k(kf, smem);
      sc_a = scores(kf);
      lmax_nx = max16(sc_a);
    }
  }

  {
    constexpr int blk = 0;
    // the head of a step: wait for this wave's share of the next group, the workgroup barrier, the
    // requests of the group three ahead
    auto top = [&](int step) {
      const uint64_t t0 = PROF ? __builtin_readcyclecounter() : 0;
      uint64_t t1 = t0, t2 = t0;
      // the entries of block b + 1 are asked for at the head of step 16 b + 2 (behind that group's K/V
      // requests), have arrived by the wait one group after the next and are turned into row offsets
      // there; their half of s_slots was last read for step 16 b - 1 and is first read for step
      // 16 b + 16, whose requests go out three groups ahead of it
      const int in_blk = step & (SLOTS / STEP - 1);
      const int group = step / SUB;                    // (SUB is 1 or 2)
      const bool head = step % SUB == 0;               // the group's first step: wait, barrier, requests
      const bool more_entries = ((step >> 4) + 1) * SLOTS < kv_len;  // wave-uniform
      const bool has_next = head && group + 1 < n_groups;           // wave-uniform
      const bool issue_ok = ((BSGL_CUT & 4) == 0) && has_next && group + 3 < n_groups;  // wave-uniform
      if (issue_ok) make_src();  // the addresses of this group's requests, ahead of the wait
      if (has_next) {
        // this wave's share of the next group has landed (one younger group may stay in flight, and
        // one group after the entries were asked for, those behind it) ...
        if (group + 2 >= n_groups) dma_wait<0>();
        else if (in_blk == 2 + SUB && more_entries) dma_wait<NREQ + EPT>();
        else dma_wait<NREQ>();
        t1 = PROF ? __builtin_readcyclecounter() : 0;
        if (in_blk == 2 + 2 * SUB && more_entries) {
          // (this wave's own pieces, landed by the wait above; the barrier below publishes them)
          int32_t* half = s_slots + (((step >> 4) + 1) & 1) * SLOTS;
          uint32_t e[EPT];
#pragma unroll
          for (int k = 0; k < EPT; ++k) e[k] = (uint32_t)half[tid + k * NTHR];
#pragma unroll
          for (int k = 0; k < EPT; ++k) half[tid + k * NTHR] = entry_off(e[k]);
        }
        // ... and so has everybody's; all waves are also done reading the group before this one
        __syncthreads();
        t2 = PROF ? __builtin_readcyclecounter() : 0;
        // the row offsets of the group after the one requested now are asked for first (any entries if
        // there is no such group: they are then never used) and travel while the requests go out
        read_offsets(group + 4);
        if (issue_ok) issue_group(group + 3);  // into the stages the group before this one used
        if (in_blk == 2 && more_entries) {
          const int e = ((step >> 4) + 1) * SLOTS + tid;
          const uint32_t half = lds_slots + (uint32_t)((((step >> 4) + 1) & 1) * SLOTS * 4) + (uint32_t)wave * 256u;
#pragma unroll
          for (int k = 0; k < EPT; ++k)
            dma_256((const void*)entry_ptr(e + k * NTHR), __builtin_amdgcn_readfirstlane(half + (uint32_t)(k * NTHR * 4)));
        }
      }
      const uint64_t t3 = PROF ? __builtin_readcyclecounter() : 0;
      if (PROF) { prof[0] += t1 - t0; prof[1] += t2 - t1; prof[2] += t3 - t2; prof[5] += 1; }
    };
    const bool late_team = STAGGER && team == 1;  // wave-uniform
    uint4 kf_cut[8] = {};
    if constexpr ((BSGL_CUT & 16) != 0) {
      load_k(kf_cut, smem);
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(kf_cut[i].x), "+v"(kf_cut[i].y), "+v"(kf_cut[i].z), "+v"(kf_cut[i].w));
    }
    auto do_step = [&](int step, auto stage_c, f32x16& sc, f32x16& sn) {
      constexpr int STAGE = decltype(stage_c)::value;
      constexpr int NEXT = (STAGE + 1) % NSTAGE;
      // (the late team has run this step's head already, between the halves of the step before)
      if (!late_team || step == 0) top(step);
      const uint64_t t3 = PROF ? __builtin_readcyclecounter() : 0;
      const char* st = smem + STAGE * STEP_BYTES;
      if (blk + step * STEP >= wave_len) {  // nothing this wave's rows may see (wave-uniform)
        if (late_team && step + 1 < n_steps) top(step + 1);
        return;
      }

      // The body runs at raised priority (four waves: the SIMD's other wave belongs to another workgroup and
      // is wherever it is): where both waves have an instruction ready, the one inside its products goes first
      // and the one at the head of its step (wait, barrier, requests) fills in.  + 0.6 to + 1.8 % by shape.
      if constexpr (TEAMS == 1) __builtin_amdgcn_s_setprio(2);
      // K fragments of step + 1 (whatever the stage holds if there is no such step: the scores
      // made of them are then never used); they travel while the maximum is looked at
      uint4 kf[8];
      if constexpr ((BSGL_CUT & 16) != 0) {  // timing cut: no LDS reads in the loop
#pragma unroll
        for (int i = 0; i < 8; ++i) kf[i] = kf_cut[i];
      } else {
        load_k(kf, smem + NEXT * STEP_BYTES);
      }

      const bool step_masked = blk + step * STEP + STEP - 1 > full_until;  // wave-uniform
      if (((BSGL_CUT & 1024) == 0) && __builtin_expect(step_masked, 0)) {  // (rare: the block's diagonal steps)
        int t0m = blk + step * STEP + 4 * hi;
        asm volatile("" : "+v"(t0m));
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = (t0m + acc_row(r, 0) <= my_last) ? sc[r] : kNegInf;
      }
      // lazy running maximum: see paged_prefill_mfma.hip
      const float lmax = step_masked ? max16(sc) : lmax_nx;
      if (((BSGL_CUT & 1024) == 0 || step == 0) && __builtin_expect(__any(lmax > m_run + slack_raw), 0)) {  // (rare after the first steps)
        float x = lmax, y = lmax;
        swap32(x, y);  // the partner lane holds the position's other 16 tokens
        const float m_new = fmaxf(m_run, fmaxf(x, y));
        // a position with nothing visible yet keeps m = -inf: its factor is 1, not exp2(nan)
        const float alpha = m_new == kNegInf ? 1.f : fast_exp2((m_run - m_new) * scale_log2e);
        m_run = m_new;
        neg_mc = -(m_new == kNegInf ? 0.f : m_new) * scale_log2e;  // masked scores: exp2(-inf) = 0
        l_run *= alpha;
        // O has its positions in the registers: register r of this lane belongs to position
        // acc_row(r, hi), whose factor lives on lane acc_row(r, hi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float ar = __shfl(alpha, acc_row(r, 0) + 4 * hi);
#pragma unroll
          for (int db = 0; db < 4; ++db) o_acc[db][r] *= ar;
        }
      }

      // ---- from here to the end of the step: one basic block, issued in the order written ----
      // (a scheduling barrier after every product: the compiler's own order puts the vector
      // work in clumps and waits for every LDS read right behind its issue).  A product holds
      // the vector issue for 8 of its 32 cycles; the gap takes one or two exponentials (8
      // cycles each) and a few 4-cycle instructions.
      auto v_frag = [&](int i) {  // V fragment of PV product i: tile i >> 2, value columns 32 (i & 3) ..
        if constexpr ((BSGL_CUT & 16) != 0) return kf_cut[i];
        const s16x4_t va = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(st + (i >> 2) * TILE_BYTES + v_off[0][i & 3]));
        const s16x4_t vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(st + (i >> 2) * TILE_BYTES + v_off[1][i & 3]));
        const uint2 va2 = __builtin_bit_cast(uint2, va), vb2 = __builtin_bit_cast(uint2, vb);
        return make_uint4(va2.x, va2.y, vb2.x, vb2.y);
      };
#pragma unroll
      for (int r = 0; r < 16; ++r) sn[r] = 0.f;
      float e[16], p[16];
      float mx0 = kNegInf, mx1 = kNegInf;
      uint32_t pw[8];  // probabilities, two to a word: words 4 kk .. 4 kk + 3 = the A operand of tile kk
      uint4 vf[4];  // V fragments travel three products ahead of their use
      e[0] = fmaf(sc[0], scale_log2e, neg_mc);
      e[1] = fmaf(sc[1], scale_log2e, neg_mc);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < 8; ++kb) {
        sn = Mma32<Tag>::mm(kf[kb], qf[kb], sn);
        if constexpr ((BSGL_CUT & 8) != 0) {  // timing cut: no exponentials, no multiply-adds
          p[2 * kb] = sc[2 * kb];
          p[2 * kb + 1] = sc[2 * kb + 1];
        } else {
          p[2 * kb] = fast_exp2(e[2 * kb]);
          p[2 * kb + 1] = fast_exp2(e[2 * kb + 1]);
          if (kb < 7) {
            e[2 * kb + 2] = fmaf(sc[2 * kb + 2], scale_log2e, neg_mc);
            e[2 * kb + 3] = fmaf(sc[2 * kb + 3], scale_log2e, neg_mc);
          }
        }
        if (kb >= 4) pw[kb - 4] = E::pack2(p[2 * (kb - 4)], p[2 * (kb - 4) + 1]);
        if (kb >= 5) vf[kb - 5] = v_frag(kb - 5);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (PROF) prof[4] += __builtin_readcyclecounter() - t3;  // the score half of the body
      if constexpr (STAGGER) {
        if (late_team && step + 1 < n_steps) top(step + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
      // k-values of a lane in PV product i: registers 8 kk .. 8 kk + 7 of the score accumulator,
      // i.e. tokens {0-3, 8-11} + 4 hi of tile kk = i >> 2
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int kk = i >> 2;
        o_acc[i & 3] = Mma32<Tag>::mm(make_uint4(pw[4 * kk], pw[4 * kk + 1], pw[4 * kk + 2], pw[4 * kk + 3]), vf[i % 4], o_acc[i & 3]);
        if (i + 3 < 8) vf[(i + 3) % 4] = v_frag(i + 3);
        if (i < 4) pw[4 + i] = E::pack2(p[8 + 2 * i], p[8 + 2 * i + 1]);
        // (as instructions of their own, here: the compiler otherwise pairs the sums into packed adds
        // and parks them at the head of the next step)
        if constexpr ((BSGL_CUT & 8) == 0) {
          asm volatile("v_add_f32 %0, %0, %1\n\tv_add_f32 %0, %0, %2" : "+v"(l_run) : "v"(p[2 * i]), "v"(p[2 * i + 1]));
          // (the next step's scores are complete by now: their largest, two maxima per gap)
          if (i == 2) { mx0 = fmaxf(sn[0], sn[1]); mx1 = fmaxf(sn[2], sn[3]); }
          if (i == 3) { mx0 = fmaxf(fmaxf(sn[4], sn[5]), mx0); mx1 = fmaxf(fmaxf(sn[6], sn[7]), mx1); }
          if (i == 4) { mx0 = fmaxf(fmaxf(sn[8], sn[9]), mx0); mx1 = fmaxf(fmaxf(sn[10], sn[11]), mx1); }
          if (i == 5) { mx0 = fmaxf(fmaxf(sn[12], sn[13]), mx0); mx1 = fmaxf(fmaxf(sn[14], sn[15]), mx1); }
          if (i == 6) lmax_nx = fmaxf(mx0, mx1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (TEAMS == 1) __builtin_amdgcn_s_setprio(0);
      if (PROF) prof[3] += __builtin_readcyclecounter() - t3;
    };
    static_assert(NSTAGE == 4 || NSTAGE == 8, "the step loop below is unrolled over the ring");
    for (int step = 0; step < n_steps; step += NSTAGE) {
      do_step(step, std::integral_constant<int, 0>{}, sc_a, sc_b);
      if (step + 1 < n_steps) do_step(step + 1, std::integral_constant<int, 1>{}, sc_b, sc_a);
      if (step + 2 < n_steps) do_step(step + 2, std::integral_constant<int, 2>{}, sc_a, sc_b);
      if (step + 3 < n_steps) do_step(step + 3, std::integral_constant<int, 3>{}, sc_b, sc_a);
      if constexpr (NSTAGE == 8) {
        if (step + 4 < n_steps) do_step(step + 4, std::integral_constant<int, 4>{}, sc_a, sc_b);
        if (step + 5 < n_steps) do_step(step + 5, std::integral_constant<int, 5>{}, sc_b, sc_a);
        if (step + 6 < n_steps) do_step(step + 6, std::integral_constant<int, 6>{}, sc_a, sc_b);
        if (step + 7 < n_steps) do_step(step + 7, std::integral_constant<int, 7>{}, sc_b, sc_a);
      }
    }
  }

  }  // (pass)
  const uint64_t t_loop_end = PROF ? __builtin_readcyclecounter() : 0;
  // ---- normalise and write: rows through LDS so that a lane stores 16 contiguous bytes ----
  __syncthreads();  // the ring is free: each wave stages its rows in its own region
  float* o_lds = (float*)(smem + wave * STAGE_OUT);
  {
    float x = l_run, y = l_run;
    swap32(x, y);
    const float l_pos = x + y;  // both halves of the position's tokens
    const float inv = l_pos > 0.f ? 1.f / l_pos : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float ir = __shfl(inv, acc_row(r, 0) + 4 * hi);
#pragma unroll
      for (int db = 0; db < 4; ++db) o_lds[(acc_row(r, 0) + 4 * hi) * kOPad + 32 * db + lp] = o_acc[db][r] * ir;
    }
    for (int idx = lane; idx < 32 * (D / 8); idx += 64) {
      const int r = idx / (D / 8), c8 = idx % (D / 8);  // r: column of S^T = (q head r / PW, position r % PW)
      if (r % PW >= n_rows || r / PW >= n_heads) continue;
      const int64_t orow = q_row0 + r % PW;
      const int oh = head0 + r / PW;
      const float* src = o_lds + r * kOPad + c8 * 8;
      const uint4 packed = make_uint4(E::pack2(src[0], src[1]), E::pack2(src[2], src[3]),
                                      E::pack2(src[4], src[5]), E::pack2(src[6], src[7]));
      if (fold == 0) {
        *(uint4*)((char*)a.out + (orow * a.out_row_stride + (int64_t)oh * D + c8 * 8) * 2) = packed;
      } else if ((c8 * 8) / 64 == oh / fold) {
        *(uint4*)((char*)a.out + (orow * a.out_row_stride + (int64_t)oh * 64 + (c8 * 8) % 64) * 2) = packed;
      }
    }
  }
  if (PROF && lane == 0) {
    prof[7] = (__builtin_readcyclecounter() - t_entry) & 0xffffffffu;
    prof[6] |= (uint64_t)(prof[7] - (t_loop_end - t_entry)) << 32;
    prof[7] |= ((__builtin_amdgcn_s_memrealtime() - rt_entry) & 0xffffffffu) << 32;
    uint64_t* dst = a.prof + 8 * ((uint64_t)block_id * WAVES + wave);
#pragma unroll
    for (int i = 0; i < 8; ++i) dst[i] = prof[i];
  }
}

template <typename Tag, int PW, bool PROF, int TEAMS = 1>
__global__ __launch_bounds__(256 * TEAMS) __attribute__((amdgpu_waves_per_eu(2)))
void prefill_mfma32_kernel(PrefillArgs a) {
  prefill_mfma32_item<Tag, PW, PROF, TEAMS>(a, blockIdx.x);
}

}  // namespace

bool prefill_mfma32_supported(const PrefillArgs& a) {
  return (a.blk_rows == 128 || a.blk_rows == 64 || a.blk_rows == 16) && a.n_kv_heads > 0 && a.n_q_heads / a.n_kv_heads >= 2;
}

// the grid of paged_prefill_mfma.hip's launch with 2 (64-position blocks) or 8 (16-position
// blocks) q heads per workgroup
hipError_t launch_prefill_mfma32(const PrefillArgs& a, hipStream_t stream) {
  const int group = a.n_q_heads / a.n_kv_heads;
  const int hpw = a.blk_rows >= 64 ? 2 : 8;
  const int n_z = (group + hpw - 1) / hpw;
  const int64_t local_groups = ((int64_t)a.max_groups * a.n_kv_heads + 7) / 8;
  const int64_t n_wg = 8 * local_groups * a.grp_blocks * n_z;
  if (n_wg > 0x7fffffffll) return hipErrorInvalidValue;
  dim3 grid((unsigned)n_wg, 1, 1);
  const unsigned pad = (unsigned)max(0, knob(kKnobPrefillLdsPad));  // tuning aid: extra LDS per workgroup, to lower the occupancy
  auto go = [&](auto kernel) { hipLaunchKernelGGL(kernel, grid, dim3(a.blk_rows == 128 ? 512 : 256), pad, stream, a); };
  const bool prof = a.prof != nullptr;
#ifndef BSGL_NO_TEAMS2
  if (a.blk_rows == 128) {
    if (a.dtype == kBF16) prof ? go(prefill_mfma32_kernel<bf16_tag, 32, true, 2>) : go(prefill_mfma32_kernel<bf16_tag, 32, false, 2>);
    else prof ? go(prefill_mfma32_kernel<f16_tag, 32, true, 2>) : go(prefill_mfma32_kernel<f16_tag, 32, false, 2>);
  } else
#endif
  if (a.blk_rows == 64) {
    if (a.dtype == kBF16) prof ? go(prefill_mfma32_kernel<bf16_tag, 32, true>) : go(prefill_mfma32_kernel<bf16_tag, 32, false>);
    else prof ? go(prefill_mfma32_kernel<f16_tag, 32, true>) : go(prefill_mfma32_kernel<f16_tag, 32, false>);
  } else {
    if (a.dtype == kBF16) prof ? go(prefill_mfma32_kernel<bf16_tag, 16, true>) : go(prefill_mfma32_kernel<bf16_tag, 16, false>);
    else prof ? go(prefill_mfma32_kernel<f16_tag, 16, true>) : go(prefill_mfma32_kernel<f16_tag, 16, false>);
  }
  return hipGetLastError();
}

}  // namespace bsgl
