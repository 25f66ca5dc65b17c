// Host-side launch interface between the C-ABI layer (capi.hip) and the kernel
// translation units.  Plain structs of device pointers and sizes; every
// launcher returns hipSuccess or the failing HIP error and never synchronises.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace bsgl {

enum DType : int32_t { kBF16 = 0, kF16 = 1, kF32 = 2 };

// Tuning knobs.  Each is read from the environment variable of its name ONCE per process (the
// first time any knob is asked for; thread safe) and kept as an int; bsgl_debug_set_knob replaces a
// value later.  No entry point calls getenv on its per-call path.
enum Knob : int {
  kKnobNoFold,          // BSGL_NO_FOLD: never stream two kv heads of 64 values as one of 128
  kKnobNoFusedKV,       // BSGL_NO_FUSED_KV: the decode step's KV write is always its own launch
  kKnobNoFusedMerge,    // BSGL_NO_FUSED_MERGE: split partials are merged by the merge kernel
  kKnobDecodeBalance,   // BSGL_DECODE_BALANCE: length-balanced pieces (default 1)
  kKnobDecodeTarget,    // BSGL_DECODE_TARGET: pieces wanted per request (0 = heuristic)
  kKnobFoldPieces,      // BSGL_FOLD_PIECES: most pieces per request with folded heads (default 8)
  kKnobDecodeTargetWgs, // BSGL_DECODE_TARGET_WGS: workgroups a decode launch aims for (default 1024)
  kKnobDecodeMinChunk,  // BSGL_DECODE_MIN_CHUNK: shortest split of the longest possible request (default 128)
  kKnobPrefillBlk,      // BSGL_PREFILL_BLK: force 16- or 64-position extend blocks
  kKnobPrefillGroup,    // BSGL_PREFILL_GROUP: force the blocks per group of the extend launch order
  kKnobMfmaNT,          // BSGL_MFMA_NT: non-temporal K/V requests in the decode kernel (default 1)
  kKnobMfmaStages,      // BSGL_MFMA_STAGES: ring depth of the decode kernel (default 2)
  kKnobMfmaHPT,         // BSGL_MFMA_HPT: kv heads per decode workgroup (default 1)
  kKnobMfmaWaves,       // BSGL_MFMA_WAVES: force the waves that share a decode piece (0 = heuristic)
  kKnobPrefillRT,       // BSGL_PREFILL_RT: q heads per workgroup of the 16x16x32 extend kernel (0 = heuristic)
  kKnobPrefill16,       // BSGL_PREFILL_16: keep the 16x16x32 extend kernel where the 32x32x16 form applies
  kKnobPrefillLdsPad,   // BSGL_PREFILL_LDS_PAD: extra LDS bytes per extend workgroup (lowers occupancy)
  kKnobP2PWaitMs,       // BSGL_P2P_WAIT_MS: how long a one-shot all-reduce waits for a peer (default 10000)
  kKnobDecodeWhole,     // BSGL_DECODE_WHOLE: requests up to this many tokens stay in one piece where several waves share a piece (default 256)
  kKnobCount
};
int knob(Knob k);

inline int dtype_bytes(int32_t dt) { return dt == kF32 ? 4 : 2; }

struct PagedKV {
  const void* k_slab;
  const void* v_slab;
  int64_t slab_row_stride;  // elements between consecutive slots
  int64_t n_slots;          // rows in a slab (slot ids are clamped to n_slots-1)
  const int32_t* table;     // req_to_token
  int64_t table_row_stride;
  int64_t table_rows;
  int64_t table_cols;
};

// One attention "query row": decode has one per request, prefill one per
// extend token (req / len arrays are then the expanded per-token ones).
struct DecodeArgs {
  void* out;
  int64_t out_row_stride;
  float* lse;  // optional [rows, n_q_heads], natural log
  const void* q;
  int64_t q_row_stride;
  PagedKV kv;
  const int64_t* req_idx;   // [rows]
  const int64_t* seq_lens;  // [rows]
  int32_t rows;
  int32_t n_q_heads;
  int32_t n_kv_heads;
  int32_t head_dim;
  float scale;
  int32_t dtype;
  int32_t out_dtype;  // dtype, or kF32 (then results always pass through the merge kernel)
  int32_t n_splits;
  float* ws_o;    // [n_splits, rows * n_q_heads, head_dim]   (unless direct)
  float* ws_lse;  // [n_splits, rows * n_q_heads]
  uint32_t* counters;  // [rows, head groups] tickets of the in-kernel merge; zero between launches; may be NULL
  // > 0: splits are cut by LENGTH, not by count: every row is cut into pieces of one
  // batch-wide size chosen so that the batch yields about rows * balance_target pieces
  // (and no row more than n_splits); needs the in-kernel merge.  0: n_splits equal
  // pieces per row.
  int32_t balance_target;
  // > balance_target: small batches whose requests turn out long may be cut finer, into up to this
  // many pieces per request, as long as a piece keeps at least 256 tokens (0: no such allowance)
  int32_t balance_target_hi;
  // requests of up to this many tokens are not cut at all (0: every request is cut like the others)
  int32_t whole_len;
  // the caller's word that no request is longer than this (0: unknown); shapes the launch only
  int32_t len_hint;
  // waves that share a piece, chosen by the host layer from len_hint (0: by the piece count)
  int32_t hint_waves;
  // matrix-core kernel, set by the host layer together with n_kv_heads = 1 and head_dim = 128 when
  // a model with two kv heads of 64 values is streamed as one 128-wide kv head: q heads per REAL kv
  // head (0 = not folded)
  int32_t fold_group;
  // fused KV write of the new token of each row (decode only); new_loc == NULL disables it
  const void* k_new;
  const void* v_new;
  int64_t k_new_stride;
  int64_t v_new_stride;
  const int32_t* new_loc;
  // tuning aid, normally NULL (looked at by the one-wave-per-piece kernel in its default
  // configuration only): 8 x uint64 per workgroup (in launch order): start, after
  // the prologue, after the streaming loop, end (100 MHz ticks), XCC id, HW id, row, piece
  uint64_t* trace;
};

struct MergeArgs {
  void* out;
  int64_t out_row_stride;
  float* lse_out;  // optional
  const float* parts_o;
  const float* parts_lse;
  int32_t n_parts;
  int32_t rows;
  int32_t n_heads;
  int32_t head_dim;
  int32_t dtype;
};

struct ScatterArgs {
  void* k_slab;
  void* v_slab;
  int64_t slab_row_stride;
  int64_t n_slots;
  const int32_t* loc;
  const void* k_new;
  const void* v_new;
  int64_t k_row_stride;
  int64_t v_row_stride;
  int32_t n_tok;
  int32_t row_bytes;  // n_kv_heads * head_dim * sizeof(elem)
  int32_t elem_bytes;
};

struct ExpandArgs {
  const int64_t* req_pool_indices;
  const int64_t* seq_lens;
  const int64_t* prefix_lens;
  const int64_t* extended_lens;
  int32_t batch;
  int32_t total_q_tokens;
  int64_t* cu_ext;   // [batch + 1] running sum of extended_lens
  int64_t* cu_blk;   // [batch + 1] running sum of block GROUPS: ceil(ceil(extended_lens / blk_rows) / grp_blocks); may be NULL
  int32_t blk_rows;  // 16 or 64 (0 counts as 16)
  int32_t grp_blocks;  // blocks per group (0 counts as 1)
  int64_t* row_req;  // [total_q_tokens]; may be NULL (then only the running sums are made)
  int64_t* row_len;  // [total_q_tokens]
};

// extend attention over blocks of 16 query positions (matrix-core kernel)
struct PrefillArgs {
  void* out;
  int64_t out_row_stride;
  const void* q;
  int64_t q_row_stride;
  PagedKV kv;
  const int64_t* req_idx;
  const int64_t* seq_lens;
  const int64_t* prefix_lens;
  const int64_t* extended_lens;
  const int64_t* cu_ext;
  const int64_t* cu_blk;  // running sum of block groups per request (ExpandArgs::cu_blk)
  int32_t batch;
  int32_t total_q_tokens;
  int32_t blk_rows;    // query positions per workgroup: 16 or 64
  int32_t grp_blocks;  // consecutive blocks of a request that form one group (share an XCD)
  int32_t max_groups;  // bound on cu_blk[batch]
  uint64_t* prof;      // tuning aid, normally NULL: 8 x uint64 of phase cycles per wave
  int32_t fold_group;  // two kv heads of 64 values streamed as one of 128 (see DecodeArgs::fold_group); 0 = no
  int32_t n_q_heads;
  int32_t n_kv_heads;
  int32_t head_dim;
  float scale;
  int32_t dtype;
};

inline bool decode_direct_out(const DecodeArgs& a) { return a.n_splits == 1 && a.out_dtype == a.dtype; }

// returns false (and leaves *why) when no VALU kernel instance covers the shape
bool decode_valu_supported(const DecodeArgs& a, const char** why);
hipError_t launch_decode_valu(const DecodeArgs& a, hipStream_t stream);

bool decode_mfma_supported(const DecodeArgs& a, const char** why);
hipError_t launch_decode_mfma(const DecodeArgs& a, hipStream_t stream);
// tuning aid: device buffer that the matrix-core decode kernel writes per-workgroup
// timestamps into (NULL switches it off); process-wide, not thread safe
void set_decode_trace(uint64_t* buf);

struct NormArgs {
  void* out;
  void* residual_out;    // optional: x + residual
  const void* x;
  const void* residual;  // optional
  const void* weight;
  int64_t x_row_stride;
  int64_t out_row_stride;
  int32_t rows;
  int32_t hidden;
  float eps;
  int32_t dtype;
};

struct RopeArgs {
  void* q_out;
  int64_t q_out_row_stride;
  void* k_out;  // optional rotated k rows
  void* v_out;  // optional v rows
  int64_t kv_out_row_stride;
  const void* qkv;  // [n_tok, (n_q_heads + 2 n_kv_heads) * head_dim]
  int64_t qkv_row_stride;
  const int64_t* positions;
  const float* inv_freq;  // [head_dim / 2]
  void* k_slab;
  void* v_slab;
  int64_t slab_row_stride;
  int64_t n_slots;
  const int32_t* out_cache_loc;  // optional: slots to write rotated k and v into
  int32_t n_tok;
  int32_t n_q_heads;
  int32_t n_kv_heads;
  int32_t head_dim;
  int32_t dtype;
};

hipError_t launch_rmsnorm(const NormArgs& a, hipStream_t stream);
int rmsnorm_max_hidden(int dtype);
hipError_t launch_rope_kv(const RopeArgs& a, hipStream_t stream);

hipError_t launch_merge(const MergeArgs& a, hipStream_t stream);
hipError_t launch_kv_scatter(const ScatterArgs& a, hipStream_t stream);
hipError_t launch_prefill_expand(const ExpandArgs& a, hipStream_t stream);

bool prefill_mfma_supported(const PrefillArgs& a, const char** why);
hipError_t launch_prefill_mfma(const PrefillArgs& a, hipStream_t stream);
// the 32x32x16 form of the same kernel (64-position blocks, two or more q heads per kv head);
// launch_prefill_mfma picks it where it applies
bool prefill_mfma32_supported(const PrefillArgs& a);
hipError_t launch_prefill_mfma32(const PrefillArgs& a, hipStream_t stream);

}  // namespace bsgl
