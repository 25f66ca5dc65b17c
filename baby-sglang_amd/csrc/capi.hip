// extern "C" surface of libbsgl_attn.so (declared in include/bsgl_attn.h):
// argument validation, kernel selection, workspace carving, error reporting.
// No entry point allocates, synchronises or reads device memory.
#include "../../include/bsgl_attn.h"

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "bsgl_launch.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int hip_fail(const char* what, hipError_t e) {
  return fail(BSGL_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

}  // namespace

namespace bsgl {
// error reporting for the other translation units of the C surface (comm.hip, p2p.hip)
int set_error(int code, const char* msg) { return fail(code, "%s", msg); }

namespace {
struct KnobDef {
  const char* name;
  int dflt;
};
// (same order as enum Knob)
constexpr KnobDef kKnobDefs[kKnobCount] = {
    {"BSGL_NO_FOLD", 0},          {"BSGL_NO_FUSED_KV", 0},         {"BSGL_NO_FUSED_MERGE", 0},
    {"BSGL_DECODE_BALANCE", 1},   {"BSGL_DECODE_TARGET", 0},       {"BSGL_FOLD_PIECES", 8},
    {"BSGL_DECODE_TARGET_WGS", 1024}, {"BSGL_DECODE_MIN_CHUNK", 128}, {"BSGL_PREFILL_BLK", 0},
    {"BSGL_PREFILL_GROUP", 0},    {"BSGL_MFMA_NT", 1},             {"BSGL_MFMA_STAGES", 2},
    {"BSGL_MFMA_HPT", 1},         {"BSGL_MFMA_WAVES", 0},          {"BSGL_PREFILL_RT", 0},
    {"BSGL_PREFILL_16", 0},       {"BSGL_PREFILL_LDS_PAD", 0},     {"BSGL_P2P_WAIT_MS", 10000},
    {"BSGL_DECODE_WHOLE", 256},
};
std::atomic<int> g_knobs[kKnobCount];
std::once_flag g_knobs_once;
void load_knobs() {
  for (int i = 0; i < kKnobCount; ++i) {
    const char* s = getenv(kKnobDefs[i].name);
    g_knobs[i].store((s == nullptr || *s == '\0') ? kKnobDefs[i].dflt : atoi(s), std::memory_order_relaxed);
  }
}
}  // namespace

int knob(Knob k) {
  std::call_once(g_knobs_once, load_knobs);
  return g_knobs[k].load(std::memory_order_relaxed);
}
}  // namespace bsgl

namespace {

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int check_dtype(int32_t dtype) {
  if (dtype != BSGL_DTYPE_BF16 && dtype != BSGL_DTYPE_F16 && dtype != BSGL_DTYPE_F32)
    return fail(BSGL_ERR_UNSUPPORTED, "dtype %d is not bf16/f16/f32 (fp8 KV slabs are not supported)", dtype);
  return BSGL_OK;
}


// shared validation of the paged-KV description and the q/out rows
int check_paged(const void* out, int64_t out_row_stride, const void* q, int64_t q_row_stride,
                const bsgl::PagedKV& kv, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim,
                int32_t dtype) {
  if (int rc = check_dtype(dtype)) return rc;
  if (!out || !q || !kv.k_slab || !kv.v_slab || !kv.table)
    return fail(BSGL_ERR_BAD_ARG, "null device pointer (out/q/k_slab/v_slab/req_to_token)");
  if (n_q_heads <= 0 || n_kv_heads <= 0 || head_dim <= 0)
    return fail(BSGL_ERR_BAD_ARG, "head counts and head_dim must be positive");
  if (n_q_heads % n_kv_heads != 0)
    return fail(BSGL_ERR_BAD_ARG, "n_q_heads %d is not a multiple of n_kv_heads %d", n_q_heads, n_kv_heads);
  if (kv.n_slots <= 0 || kv.table_rows <= 0 || kv.table_cols <= 0)
    return fail(BSGL_ERR_BAD_ARG, "slab / page-table extents must be positive");
  const int eb = bsgl::dtype_bytes(dtype);
  if (!aligned16(out) || !aligned16(q) || !aligned16(kv.k_slab) || !aligned16(kv.v_slab))
    return fail(BSGL_ERR_BAD_ARG, "out, q and KV slabs must be 16-byte aligned");
  if ((out_row_stride * eb) % 16 || (q_row_stride * eb) % 16 || (kv.slab_row_stride * eb) % 16)
    return fail(BSGL_ERR_BAD_ARG, "row strides must be multiples of 16 bytes");
  if (out_row_stride < (int64_t)n_q_heads * head_dim || q_row_stride < (int64_t)n_q_heads * head_dim)
    return fail(BSGL_ERR_BAD_ARG, "q/out row stride shorter than n_q_heads * head_dim");
  if (kv.slab_row_stride < (int64_t)n_kv_heads * head_dim)
    return fail(BSGL_ERR_BAD_ARG, "slab row stride shorter than n_kv_heads * head_dim");
  if (kv.table_row_stride < kv.table_cols)
    return fail(BSGL_ERR_BAD_ARG, "page-table row stride shorter than its column count");
  return BSGL_OK;
}

// fixed-size ticket area at the head of every decode workspace, so that its place
// does not move when one workspace serves calls of different batch sizes
constexpr size_t kCounterBytes = 65536;

int run_decode(bsgl::DecodeArgs& a, int32_t variant, uint32_t* counters, hipStream_t stream) {
  const char* why = "";
  bool use_mfma = false;
  const bool may_fuse_kv = (variant & BSGL_DECODE_FUSE_KV_WRITE) != 0;
  const bool uniform_splits = (variant & BSGL_DECODE_UNIFORM_SPLITS) != 0;
  variant &= 0xff;
  // Two kv heads of 64 values (Qwen2.5-0.5B): their rows are neighbours in a slab row and in the
  // new-token rows, so the matrix-core kernel can stream the pair as ONE 128-wide kv head for all
  // q heads (at most 16), with the in-kernel merge (its partial rows are 128 wide).
  bsgl::DecodeArgs folded = a;
  bool can_fold = false;
  if (a.head_dim == 64 && a.n_kv_heads == 2 && a.n_q_heads % 2 == 0 && a.n_q_heads <= 16 &&
      (counters != nullptr || bsgl::decode_direct_out(a)) && bsgl::knob(bsgl::kKnobNoFold) == 0) {
    folded.fold_group = a.n_q_heads / 2;
    folded.n_kv_heads = 1;
    folded.head_dim = 128;
    can_fold = bsgl::decode_mfma_supported(folded, &why);
  }
  if (variant == BSGL_DECODE_MFMA) {
    if (can_fold) a = folded;
    else if (!bsgl::decode_mfma_supported(a, &why)) return fail(BSGL_ERR_UNSUPPORTED, "MFMA decode: %s", why);
    use_mfma = true;
  } else if (variant == BSGL_DECODE_AUTO) {
    // (at every batch size: 1 - 24 requests of 1 - 4095 tokens take 5.9 - 17.4 us per layer call folded,
    // 13.5 - 31 us with the generic kernel and its separate KV-write and merge launches)
    if (can_fold) a = folded;
    use_mfma = a.fold_group > 0 || bsgl::decode_mfma_supported(a, &why);
  } else if (variant != BSGL_DECODE_VALU) {
    return fail(BSGL_ERR_BAD_ARG, "unknown decode variant %d", variant);
  }
  if (!use_mfma && !bsgl::decode_valu_supported(a, &why)) return fail(BSGL_ERR_UNSUPPORTED, "decode: %s", why);
  // KV write of the new tokens: inside the matrix-core kernel when the caller
  // allows it and the rows qualify, otherwise as its own launch first
  if (a.new_loc != nullptr) {
    const int eb = bsgl::dtype_bytes(a.dtype);
    const bool rows_ok = (((uintptr_t)a.k_new | (uintptr_t)a.v_new) % 256 == 0) &&
                         (a.k_new_stride * eb) % 256 == 0 && (a.v_new_stride * eb) % 256 == 0;
    if (!(use_mfma && may_fuse_kv && rows_ok && bsgl::knob(bsgl::kKnobNoFusedKV) == 0)) {
      bsgl::ScatterArgs sc;
      sc.k_slab = const_cast<void*>(a.kv.k_slab);
      sc.v_slab = const_cast<void*>(a.kv.v_slab);
      sc.slab_row_stride = a.kv.slab_row_stride;
      sc.n_slots = a.kv.n_slots;
      sc.loc = a.new_loc;
      sc.k_new = a.k_new;
      sc.v_new = a.v_new;
      sc.k_row_stride = a.k_new_stride;
      sc.v_row_stride = a.v_new_stride;
      sc.n_tok = a.rows;
      sc.row_bytes = a.n_kv_heads * a.head_dim * eb;
      sc.elem_bytes = eb;
      const hipError_t es = bsgl::launch_kv_scatter(sc, stream);
      if (es != hipSuccess) return hip_fail("kv_scatter launch", es);
      a.new_loc = nullptr;
    }
  }
  // one ticket per (row, kv head): fall back to the merge kernel when they do not fit
  const bool tickets_fit = (size_t)a.rows * (size_t)a.n_kv_heads * 4 <= kCounterBytes &&
                           (size_t)a.n_splits * a.rows * a.n_q_heads * a.head_dim * 4 < (1ull << 31);
  a.counters = (use_mfma && tickets_fit && (a.fold_group > 0 || bsgl::knob(bsgl::kKnobNoFusedMerge) == 0)) ? counters : nullptr;
  if (a.fold_group > 0 && a.counters == nullptr && !bsgl::decode_direct_out(a))
    return fail(BSGL_ERR_UNSUPPORTED, "folded kv heads need the in-kernel merge (batch too large for the ticket area)");
  // length-balanced pieces: needs the in-kernel merge (a row's piece count varies) and a
  // batch small enough for every workgroup to scan its seq_lens
  a.balance_target = 0;
  a.balance_target_hi = 0;
  a.whole_len = 0;
  if (a.counters != nullptr && a.n_splits > 1 && a.rows <= 1024 && a.kv.table_cols <= (1 << 20) && !uniform_splits && bsgl::knob(bsgl::kKnobDecodeBalance) != 0)
    a.balance_target = (a.n_splits + 1) / 2;
  if (a.balance_target > 0) {
    // Small batches get many splits, but pieces much shorter than 256 tokens cost more in the merge
    // than they gain (a piece can be shared by up to four waves instead): no more pieces than
    // 512-token stretches of a full page-table row -- 256-token stretches for one or two requests,
    // which leave most of the chip idle either way.  (Measured in a 4096-column table: 8 pieces
    // against 16 is 7 - 15 % faster for 4 - 32 requests of 2048 tokens; for one request 16 pieces are
    // 4 % faster at 2048 tokens and 14 % at 4095.)
    const int64_t stretch = a.rows <= 2 ? 256 : 512;
    const int64_t cap = a.kv.table_cols / stretch > 4 ? a.kv.table_cols / stretch : 4;
    if (a.balance_target > cap) a.balance_target = (int32_t)cap;
    // A few more requests (up to 16 (request, kv head) units): whether 8 or 16 pieces are right depends
    // on their lengths (eight requests of 2048 tokens: 8 pieces 12.1 us, 16 pieces 12.9; of 4095 tokens:
    // 15.4 against 14.8; three requests of 4095 tokens: 14.4 against 12.6), which only the kernel sees: it
    // may cut finer as long as a piece keeps 256 tokens.  (12 and 16 requests of two kv heads: 5 - 10 %
    // slower that way at 3000 - 4095 tokens.)
    if (a.rows > 2 && (int64_t)a.rows * a.n_kv_heads <= 16) {
      const int64_t fine = a.kv.table_cols / 256;
      const int64_t hi = fine < (a.n_splits + 1) / 2 ? fine : (a.n_splits + 1) / 2;
      if (hi > a.balance_target) a.balance_target_hi = (int32_t)hi;
    }
    // (where a piece is shared by four waves: 256 pieces or fewer, as launch_decode_mfma decides)
    {
      // Short requests are not cut at all: one piece has no merge to wait for.  Measured (µs per
      // layer call, 8 pieces against 1): 64 tokens 8.1 - 8.9 against 5.7 - 6.0 at any batch up to 32
      // (64 requests: 8.8 against 7.1); 128 tokens 8.1 - 10.3 against 6.5 - 6.7 (64 requests: 11.5 against
      // 10.1); 256 tokens 9.2 - 11.0 against 8.0 - 8.3 where four waves share a piece (512 pieces or
      // fewer, as launch_decode_mfma decides), but 11.9 against 16.1 beyond.
      const int whole = bsgl::knob(bsgl::kKnobDecodeWhole);
      a.whole_len = (int64_t)a.rows * a.n_kv_heads * a.balance_target <= 512 ? whole : whole / 2;
    }
    // A caller that knows its longest request (bsgl_decode_call::max_seq_len) and has only short ones:
    // the rules above are made for requests as long as the page table is wide.  Up to 256 tokens every
    // request is one piece, so the launch has rows x kv heads pieces, shared by four waves each up to 256
    // of them and by two up to 1024 (64 requests of 128 tokens: 10.1 -> 6.9 us per layer call, 64 x 256:
    // 11.8 -> 8.8, 128 x 256: 13.4 -> 10.8, 256 x 256: 19.3 -> 16.6; two kv heads of 64 values folded into one:
    // 128 x 256: 17.8 -> 10.4; at 512 tokens little is left to gain).
    a.hint_waves = 0;
    if (a.len_hint > 0 && a.len_hint <= 256) {
      const int64_t units = (int64_t)a.rows * a.n_kv_heads;
      a.whole_len = 256;
      a.hint_waves = units <= 256 ? 4 : units <= 1024 ? 2 : 1;
      // (one piece per request: the flat grid need not hold room for more)
      a.balance_target = 1;
      a.balance_target_hi = 0;
    }
    const int t = bsgl::knob(bsgl::kKnobDecodeTarget);  // tuning override: pieces wanted per request
    if (t > 0 && t <= a.n_splits) {
      a.balance_target = t;
      a.balance_target_hi = 0;
    }
  }
  // (folded heads: the workgroup that merges a request reads the partial rows of ALL its q heads,
  // 14 for Qwen2.5-0.5B: more than 8 pieces per request cost more in that merge than they gain)
  // -- and with half as many kv heads to cut, up to n_splits pieces are wanted below that
  if (a.fold_group > 0 && a.balance_target > 0 && a.hint_waves == 0) {
    const int cap = bsgl::knob(bsgl::kKnobFoldPieces);
    a.balance_target = a.n_splits < cap ? a.n_splits : cap;
    a.balance_target_hi = 0;
  }
  const hipError_t e = use_mfma ? bsgl::launch_decode_mfma(a, stream) : bsgl::launch_decode_valu(a, stream);
  if (e != hipSuccess) return hip_fail("decode launch", e);
  return BSGL_OK;
}

}  // namespace

extern "C" {

int bsgl_abi_version(void) { return 2; }

void bsgl_debug_decode_trace(void* device_buf) { bsgl::set_decode_trace((uint64_t*)device_buf); }

static uint64_t* g_prefill_prof = nullptr;
void bsgl_debug_prefill_profile(void* device_buf) { g_prefill_prof = (uint64_t*)device_buf; }

const char* bsgl_last_error(void) { return g_err; }

int bsgl_debug_set_knob(const char* name, int32_t value, int32_t* previous) {
  if (name == nullptr) return fail(BSGL_ERR_BAD_ARG, "null knob name");
  for (int i = 0; i < bsgl::kKnobCount; ++i) {
    if (strcmp(name, bsgl::kKnobDefs[i].name) == 0) {
      const int old = bsgl::knob((bsgl::Knob)i);  // (loads the environment's values first)
      if (previous != nullptr) *previous = old;
      bsgl::g_knobs[i].store(value, std::memory_order_relaxed);
      return BSGL_OK;
    }
  }
  return fail(BSGL_ERR_BAD_ARG, "unknown knob %s", name);
}

int bsgl_kv_scatter(void* k_slab, void* v_slab, int64_t slab_row_stride, int64_t n_slots,
                    const int32_t* loc, const void* k_new, const void* v_new,
                    int64_t k_row_stride, int64_t v_row_stride, int32_t n_tok,
                    int32_t n_kv_heads, int32_t head_dim, int32_t dtype, void* stream) {
  if (int rc = check_dtype(dtype)) return rc;
  if (n_tok < 0) return fail(BSGL_ERR_BAD_ARG, "n_tok < 0");
  if (n_tok == 0) return BSGL_OK;
  if (!k_slab || !v_slab || !loc || !k_new || !v_new)
    return fail(BSGL_ERR_BAD_ARG, "null device pointer (k_slab/v_slab/loc/k_new/v_new)");
  if (n_kv_heads <= 0 || head_dim <= 0 || n_slots <= 0)
    return fail(BSGL_ERR_BAD_ARG, "n_kv_heads, head_dim and n_slots must be positive");
  const int eb = bsgl::dtype_bytes(dtype);
  const int64_t row_bytes = (int64_t)n_kv_heads * head_dim * eb;
  if (row_bytes % 16) return fail(BSGL_ERR_UNSUPPORTED, "KV row of %lld bytes is not a multiple of 16", (long long)row_bytes);
  if (!aligned16(k_slab) || !aligned16(v_slab) || !aligned16(k_new) || !aligned16(v_new))
    return fail(BSGL_ERR_BAD_ARG, "slabs and new K/V rows must be 16-byte aligned");
  if ((slab_row_stride * eb) % 16 || (k_row_stride * eb) % 16 || (v_row_stride * eb) % 16)
    return fail(BSGL_ERR_BAD_ARG, "row strides must be multiples of 16 bytes");
  if (slab_row_stride * eb < row_bytes || k_row_stride * eb < row_bytes || v_row_stride * eb < row_bytes)
    return fail(BSGL_ERR_BAD_ARG, "a row stride is shorter than the KV row");
  bsgl::ScatterArgs a;
  a.k_slab = k_slab;
  a.v_slab = v_slab;
  a.slab_row_stride = slab_row_stride;
  a.n_slots = n_slots;
  a.loc = loc;
  a.k_new = k_new;
  a.v_new = v_new;
  a.k_row_stride = k_row_stride;
  a.v_row_stride = v_row_stride;
  a.n_tok = n_tok;
  a.row_bytes = (int32_t)row_bytes;
  a.elem_bytes = eb;
  const hipError_t e = bsgl::launch_kv_scatter(a, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail("kv_scatter launch", e);
  return BSGL_OK;
}

// pieces per request that fill the machine when every request has the same length;
// depth_factor 2 leaves room for requests of twice the mean length
static int32_t decode_splits(int32_t batch, int32_t n_kv_heads, int32_t max_context_len, int depth_factor) {
  if (batch <= 0) return 1;
  if (n_kv_heads < 1) n_kv_heads = 1;
  // enough workgroups to put several on each of the 256 CUs, but no split
  // shorter than min_chunk tokens of the longest possible request
  const int target_wgs = bsgl::knob(bsgl::kKnobDecodeTargetWgs);
  const int min_chunk = bsgl::knob(bsgl::kKnobDecodeMinChunk);
  // one workgroup per (request, kv head, split)
  const int64_t per_split = (int64_t)batch * n_kv_heads;
  int splits = (int)((target_wgs + per_split - 1) / per_split);
  // twice that as grid depth: with length-balanced pieces a request of twice the
  // batch's mean length is cut into twice the mean number of pieces
  if (splits > 1) splits *= depth_factor;
  const int by_len = max_context_len > 0 ? (max_context_len + min_chunk - 1) / min_chunk : 1;
  if (splits > by_len) splits = by_len;
  if (splits > 64) splits = 64;
  if (splits < 1) splits = 1;
  return splits;
}

int32_t bsgl_decode_num_splits(int32_t batch, int32_t n_kv_heads, int32_t max_context_len) {
  return decode_splits(batch, n_kv_heads, max_context_len, 2);
}

size_t bsgl_decode_counter_bytes(void) { return kCounterBytes; }

// partial output rows in the decode workspace: head_dim floats, but never fewer than 128 (two kv
// heads of 64 values may run folded into one 128-wide head)
static size_t partial_row_width(int32_t head_dim) { return head_dim < 128 ? 128 : (size_t)head_dim; }

size_t bsgl_decode_workspace_bytes(int32_t batch, int32_t n_q_heads, int32_t head_dim, int32_t n_splits,
                                   int32_t out_is_f32) {
  if (batch <= 0 || n_splits < 1 || (n_splits == 1 && !out_is_f32)) return 0;
  const size_t rh = (size_t)batch * (size_t)n_q_heads;
  return kCounterBytes + align_up((size_t)n_splits * rh * (size_t)partial_row_width(head_dim) * 4, 256) +
         align_up((size_t)n_splits * rh * 4, 256);
}

// Scratch of bsgl_paged_decode_static: the decode workspace (tickets, partial rows, their
// log-sum-exps) as an array of the code object instead of caller memory.  Every GPU the library
// runs on holds its own copy, at its own address, until the process ends.
constexpr size_t kStaticScratchBytes = BSGL_STATIC_SCRATCH_BYTES;
static __device__ __align__(256) __attribute__((used)) unsigned char scratch[kStaticScratchBytes];

namespace {

// The array's name is a host-side stand-in, not a device address: the address is asked for once per
// device (with that device current) and only that pointer is ever handed to a kernel or a memset.
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void* g_scratch_ptr[kMaxDevices] = {};

// (`capturing`: the caller's stream is being captured.  The address is then only handed out if an
// earlier, eager call on this device has looked it up: the first look-up loads the code object's
// symbol table, which is no operation for a stream capture to run into.)
int scratch_of_current_device(void** out, bool capturing) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return hip_fail("hipGetDevice", e);
  if (dev < 0 || dev >= kMaxDevices) return fail(BSGL_ERR_UNSUPPORTED, "device ordinal %d outside [0, %d)", dev, kMaxDevices);
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (g_scratch_ptr[dev] == nullptr) {
    if (capturing)
      return fail(BSGL_ERR_UNSUPPORTED, "bsgl_paged_decode_static: first call on device %d while its stream is being captured; "
                  "make one eager call on the device first (the warm-up run a capture needs anyway)", dev);
    void* p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return hip_fail("hipGetSymbolAddress(scratch)", e);
    if (p == nullptr) return fail(BSGL_ERR_HIP, "hipGetSymbolAddress(scratch) gave a null address");
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return BSGL_OK;
}

int decode_entry(void* out, int64_t out_row_stride, float* lse, const void* q, int64_t q_row_stride,
                 const void* k_new, const void* v_new, int64_t k_new_row_stride, int64_t v_new_row_stride,
                 const int32_t* out_cache_loc,
                 void* k_slab, void* v_slab, int64_t slab_row_stride, int64_t n_slots,
                 const int32_t* req_to_token, int64_t table_row_stride, int64_t table_rows,
                 int64_t table_cols, const int64_t* req_pool_indices, const int64_t* seq_lens,
                 int32_t batch, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim, float scale,
                 int32_t dtype, int32_t out_dtype, int32_t n_splits, int32_t variant,
                 void* workspace, size_t workspace_bytes, bool static_scratch, int32_t max_seq_len, void* stream) {
  if (batch < 0) return fail(BSGL_ERR_BAD_ARG, "batch < 0");
  if (out_dtype != dtype && out_dtype != BSGL_DTYPE_F32)
    return fail(BSGL_ERR_BAD_ARG, "out_dtype must equal dtype or be f32");
  if (batch == 0) return BSGL_OK;
  bsgl::DecodeArgs a{};
  a.kv = {k_slab, v_slab, slab_row_stride, n_slots, req_to_token, table_row_stride, table_rows, table_cols};
  if (int rc = check_paged(out, out_row_stride, q, q_row_stride, a.kv, n_q_heads, n_kv_heads, head_dim, dtype)) return rc;
  if (!req_pool_indices || !seq_lens) return fail(BSGL_ERR_BAD_ARG, "null req_pool_indices / seq_lens");
  const int n_new = (k_new != nullptr) + (v_new != nullptr) + (out_cache_loc != nullptr);
  if (n_new != 0 && n_new != 3) return fail(BSGL_ERR_BAD_ARG, "k_new, v_new and out_cache_loc must be given together");
  if (n_new == 3) {
    const int64_t eb = bsgl::dtype_bytes(dtype);
    const int64_t row_bytes = (int64_t)n_kv_heads * head_dim * eb;
    if (!aligned16(k_new) || !aligned16(v_new) || (k_new_row_stride * eb) % 16 || (v_new_row_stride * eb) % 16)
      return fail(BSGL_ERR_BAD_ARG, "new K/V rows must be 16-byte aligned with 16-byte-multiple strides");
    if (k_new_row_stride * eb < row_bytes || v_new_row_stride * eb < row_bytes)
      return fail(BSGL_ERR_BAD_ARG, "new K/V row stride shorter than the KV row");
    a.k_new = k_new;
    a.v_new = v_new;
    a.k_new_stride = k_new_row_stride;
    a.v_new_stride = v_new_row_stride;
    a.new_loc = out_cache_loc;
  }
  if (n_splits < 1 || n_splits > 64) return fail(BSGL_ERR_BAD_ARG, "n_splits %d outside [1, 64]", n_splits);
  const int f32_out = out_dtype != dtype;
  if (f32_out && (((uintptr_t)out & 15u) || (out_row_stride * 4) % 16))
    return fail(BSGL_ERR_BAD_ARG, "f32 out must be 16-byte aligned with a 16-byte-multiple row stride");
  const size_t need = bsgl_decode_workspace_bytes(batch, n_q_heads, head_dim, n_splits, f32_out);
  if (static_scratch && need > 0) {
    if (need > kStaticScratchBytes)
      return fail(BSGL_ERR_WORKSPACE, "bsgl_paged_decode_static: batch %d x %d q heads x %d splits needs %zu bytes of scratch, "
                  "the library's array holds %zu; use bsgl_paged_decode with a workspace, or fewer splits",
                  batch, n_q_heads, n_splits, need, kStaticScratchBytes);
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    const hipError_t ec = hipStreamIsCapturing((hipStream_t)stream, &capturing);
    if (ec != hipSuccess) return hip_fail("hipStreamIsCapturing", ec);
    if (int rc = scratch_of_current_device(&workspace, capturing != hipStreamCaptureStatusNone)) return rc;
    workspace_bytes = kStaticScratchBytes;
    // The array keeps the last call's contents, and the tickets of this call's rows must start at
    // zero.  They do after every launch that ran to its end (the workgroup that merges a row stores
    // 0 to its ticket) and in a freshly loaded array; an eager call clears them nevertheless, so that
    // nothing an aborted launch left behind is counted.  While the stream is being CAPTURED the
    // clearing is left out: as a memset node it was found not to be ordered in front of the kernel
    // node when the graph is replayed (ROCm 7.0 / 7.2: rows merged from stale partial results), and a
    // replayed graph starts from the zeros its own last replay, or the eager warm-up call, left.
#ifdef BSGL_STATIC_MEMSET_IN_CAPTURE  // (probe build: the clearing as a memset node of the captured graph)
    {
#else
    if (capturing == hipStreamCaptureStatusNone) {
#endif
      const size_t ticket_bytes = align_up((size_t)batch * (size_t)n_kv_heads * 4, 256);
      const hipError_t em = hipMemsetAsync(workspace, 0, ticket_bytes < kCounterBytes ? ticket_bytes : kCounterBytes, (hipStream_t)stream);
      if (em != hipSuccess) return hip_fail("hipMemsetAsync(scratch tickets)", em);
    }
  }
  if (need > 0 && (workspace == nullptr || workspace_bytes < need || !aligned16(workspace)))
    return fail(BSGL_ERR_WORKSPACE, "decode with %d splits needs a 16-byte aligned workspace of %zu bytes (got %zu)", n_splits, need, workspace ? workspace_bytes : (size_t)0);
  a.out = out;
  a.out_row_stride = out_row_stride;
  a.lse = lse;
  a.q = q;
  a.q_row_stride = q_row_stride;
  a.req_idx = req_pool_indices;
  a.seq_lens = seq_lens;
  a.rows = batch;
  a.n_q_heads = n_q_heads;
  a.n_kv_heads = n_kv_heads;
  a.head_dim = head_dim;
  a.scale = scale;
  a.dtype = dtype;
  a.out_dtype = out_dtype;
  a.n_splits = n_splits;
  a.len_hint = max_seq_len > 0 ? max_seq_len : 0;
  const size_t rh = (size_t)batch * (size_t)n_q_heads;
  const size_t cb = kCounterBytes;
  a.ws_o = need ? (float*)((char*)workspace + cb) : nullptr;
  a.ws_lse = need ? (float*)((char*)workspace + cb + align_up((size_t)n_splits * rh * partial_row_width(head_dim) * 4, 256)) : nullptr;
  return run_decode(a, variant, need ? (uint32_t*)workspace : nullptr, (hipStream_t)stream);
}

}  // namespace

int bsgl_paged_decode(void* out, int64_t out_row_stride, float* lse, const void* q, int64_t q_row_stride,
                      const void* k_new, const void* v_new, int64_t k_new_row_stride, int64_t v_new_row_stride,
                      const int32_t* out_cache_loc,
                      void* k_slab, void* v_slab, int64_t slab_row_stride, int64_t n_slots,
                      const int32_t* req_to_token, int64_t table_row_stride, int64_t table_rows,
                      int64_t table_cols, const int64_t* req_pool_indices, const int64_t* seq_lens,
                      int32_t batch, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim, float scale,
                      int32_t dtype, int32_t out_dtype, int32_t n_splits, int32_t variant,
                      void* workspace, size_t workspace_bytes, void* stream) {
  return decode_entry(out, out_row_stride, lse, q, q_row_stride, k_new, v_new, k_new_row_stride, v_new_row_stride,
                      out_cache_loc, k_slab, v_slab, slab_row_stride, n_slots, req_to_token, table_row_stride,
                      table_rows, table_cols, req_pool_indices, seq_lens, batch, n_q_heads, n_kv_heads, head_dim,
                      scale, dtype, out_dtype, n_splits, variant, workspace, workspace_bytes, false, 0, stream);
}

int bsgl_paged_decode_call(const bsgl_decode_call* c) {
  if (c == nullptr) return fail(BSGL_ERR_BAD_ARG, "null call descriptor");
  return decode_entry(c->out, c->out_row_stride, c->lse, c->q, c->q_row_stride, c->k_new, c->v_new, c->k_new_row_stride,
                      c->v_new_row_stride, c->out_cache_loc, c->k_slab, c->v_slab, c->slab_row_stride, c->n_slots,
                      c->req_to_token, c->table_row_stride, c->table_rows, c->table_cols, c->req_pool_indices, c->seq_lens,
                      c->batch, c->n_q_heads, c->n_kv_heads, c->head_dim, c->scale, c->dtype, c->out_dtype, c->n_splits,
                      c->variant, c->workspace, c->workspace_bytes, false, c->max_seq_len, c->stream);
}

int bsgl_paged_decode_static(void* out, int64_t out_row_stride, float* lse, const void* q, int64_t q_row_stride,
                             const void* k_new, const void* v_new, int64_t k_new_row_stride, int64_t v_new_row_stride,
                             const int32_t* out_cache_loc,
                             void* k_slab, void* v_slab, int64_t slab_row_stride, int64_t n_slots,
                             const int32_t* req_to_token, int64_t table_row_stride, int64_t table_rows,
                             int64_t table_cols, const int64_t* req_pool_indices, const int64_t* seq_lens,
                             int32_t batch, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim, float scale,
                             int32_t dtype, int32_t out_dtype, int32_t n_splits, int32_t variant, void* stream) {
  return decode_entry(out, out_row_stride, lse, q, q_row_stride, k_new, v_new, k_new_row_stride, v_new_row_stride,
                      out_cache_loc, k_slab, v_slab, slab_row_stride, n_slots, req_to_token, table_row_stride,
                      table_rows, table_cols, req_pool_indices, seq_lens, batch, n_q_heads, n_kv_heads, head_dim,
                      scale, dtype, out_dtype, n_splits, variant, nullptr, 0, true, 0, stream);
}

static int32_t prefill_row_splits(int32_t total_q_tokens, int32_t max_context_len) {
  return decode_splits(total_q_tokens, 1, max_context_len, 1);  // equal pieces per row (generic kernel)
}


size_t bsgl_prefill_workspace_bytes(int32_t batch, int32_t total_q_tokens, int32_t n_q_heads, int32_t head_dim) {
  if (batch <= 0 || total_q_tokens <= 0) return 0;
  // cu_ext[batch + 1], row_req[T], row_len[T], then split partials sized for the
  // worst case over context lengths (64 splits only when T is tiny)
  size_t bytes = 2 * align_up((size_t)(batch + 1) * 8, 256) + 2 * align_up((size_t)total_q_tokens * 8, 256);
  const int32_t splits = prefill_row_splits(total_q_tokens, 1 << 30);
  bytes += bsgl_decode_workspace_bytes(total_q_tokens, n_q_heads, head_dim, splits, 0);
  return bytes;
}

int bsgl_paged_prefill(void* out, int64_t out_row_stride, const void* q, int64_t q_row_stride,
                       const void* k_slab, const void* v_slab, int64_t slab_row_stride, int64_t n_slots,
                       const int32_t* req_to_token, int64_t table_row_stride, int64_t table_rows,
                       int64_t table_cols, const int64_t* req_pool_indices, const int64_t* seq_lens,
                       const int64_t* prefix_lens, const int64_t* extended_lens, int32_t batch,
                       int32_t total_q_tokens, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim,
                       float scale, int32_t dtype, int32_t variant, void* workspace,
                       size_t workspace_bytes, void* stream) {
  if (batch < 0 || total_q_tokens < 0) return fail(BSGL_ERR_BAD_ARG, "batch / total_q_tokens < 0");
  if (batch == 0 || total_q_tokens == 0) return BSGL_OK;
  bsgl::DecodeArgs a{};
  a.kv = {k_slab, v_slab, slab_row_stride, n_slots, req_to_token, table_row_stride, table_rows, table_cols};
  if (int rc = check_paged(out, out_row_stride, q, q_row_stride, a.kv, n_q_heads, n_kv_heads, head_dim, dtype)) return rc;
  if (!req_pool_indices || !seq_lens || !prefix_lens || !extended_lens)
    return fail(BSGL_ERR_BAD_ARG, "null req_pool_indices / seq_lens / prefix_lens / extended_lens");
  const size_t need = bsgl_prefill_workspace_bytes(batch, total_q_tokens, n_q_heads, head_dim);
  if (workspace == nullptr || workspace_bytes < need || !aligned16(workspace))
    return fail(BSGL_ERR_WORKSPACE, "prefill needs a 16-byte aligned workspace of %zu bytes (got %zu)", need, workspace ? workspace_bytes : (size_t)0);

  char* ws = (char*)workspace;
  bsgl::ExpandArgs ex{};
  ex.req_pool_indices = req_pool_indices;
  ex.seq_lens = seq_lens;
  ex.prefix_lens = prefix_lens;
  ex.extended_lens = extended_lens;
  ex.batch = batch;
  ex.total_q_tokens = total_q_tokens;
  ex.cu_ext = (int64_t*)ws;
  ws += align_up((size_t)(batch + 1) * 8, 256);
  ex.cu_blk = (int64_t*)ws;
  ws += align_up((size_t)(batch + 1) * 8, 256);
  ex.row_req = (int64_t*)ws;
  ws += align_up((size_t)total_q_tokens * 8, 256);
  ex.row_len = (int64_t*)ws;
  ws += align_up((size_t)total_q_tokens * 8, 256);

  // matrix-core kernel over blocks of 16 query positions when the shape allows it
  bsgl::PrefillArgs pa{};
  pa.out = out;
  pa.out_row_stride = out_row_stride;
  pa.q = q;
  pa.q_row_stride = q_row_stride;
  pa.kv = a.kv;
  pa.req_idx = req_pool_indices;
  pa.seq_lens = seq_lens;
  pa.prefix_lens = prefix_lens;
  pa.extended_lens = extended_lens;
  pa.cu_ext = ex.cu_ext;
  pa.cu_blk = ex.cu_blk;
  pa.batch = batch;
  pa.total_q_tokens = total_q_tokens;
  // 64 query positions per workgroup unless the extends are short on average.  With several q
  // heads per kv head a 16-position workgroup serves 8 of them from one K/V stream (measured
  // against 64 positions x 2 heads: faster up to 32 new tokens per request, equal at 48, slower
  // from 64); with one q head per kv head it is the single-wave kernel, worth it below 24.
  const int64_t short_below = n_kv_heads > 0 && n_q_heads / n_kv_heads >= 2 ? 44 : 24;
  pa.blk_rows = (int64_t)total_q_tokens >= short_below * batch ? 64 : 16;
  {
    const int forced = bsgl::knob(bsgl::kKnobPrefillBlk);  // tuning override
    if (forced == 16 || forced == 64) pa.blk_rows = forced;
    // (128: two four-wave teams share a K/V stream; only where the 32x32x16 kernel applies)
    if (forced == 128 && n_kv_heads > 0 && n_q_heads / n_kv_heads >= 2) pa.blk_rows = 128;
  }
  ex.blk_rows = pa.blk_rows;
  // Groups of up to 8 consecutive blocks of a request share an XCD (and with it an L2).  The grid
  // holds a workgroup for every block slot of every group, and the slots a request does not fill
  // retire at once -- but dispatching them is not free (batches of one-block requests in groups
  // of 8 spent 4/5 of the launch on it), so the group is no larger than the mean request needs.
  {
    const int64_t per = (int64_t)pa.blk_rows * (batch > 0 ? batch : 1);
    const int64_t mean_blocks = ((int64_t)total_q_tokens + per - 1) / per;
    int grp = 1;
    while (grp < 8 && grp < mean_blocks) grp *= 2;
    const int forced = bsgl::knob(bsgl::kKnobPrefillGroup);  // tuning override
    pa.grp_blocks = forced > 0 ? forced : grp;
  }
  ex.grp_blocks = pa.grp_blocks;
  // bound on the number of (request, block group) entries: blocks <= total / rows + batch, and
  // sum ceil(blocks_i / g) <= blocks / g + batch (for g = 1 the groups are the blocks)
  {
    const int64_t blocks = ((int64_t)total_q_tokens + pa.blk_rows - 1) / pa.blk_rows + batch;
    pa.max_groups = (int32_t)(pa.grp_blocks == 1 ? blocks : blocks / pa.grp_blocks + batch + 1);
  }
  pa.prof = g_prefill_prof;
  pa.n_q_heads = n_q_heads;
  pa.n_kv_heads = n_kv_heads;
  pa.head_dim = head_dim;
  pa.scale = scale;
  pa.dtype = dtype;
  const char* why = "";
  const int32_t kind = variant & 0xff;
  if (kind != BSGL_DECODE_AUTO && kind != BSGL_DECODE_VALU && kind != BSGL_DECODE_MFMA)
    return fail(BSGL_ERR_BAD_ARG, "unknown variant %d", variant);
  // two kv heads of 64 values (Qwen2.5-0.5B): streamed as one 128-wide kv head (see run_decode)
  if (head_dim == 64 && n_kv_heads == 2 && n_q_heads % 2 == 0 && n_q_heads <= 16 && bsgl::knob(bsgl::kKnobNoFold) == 0) {
    bsgl::PrefillArgs folded = pa;
    folded.fold_group = n_q_heads / 2;
    folded.n_kv_heads = 1;
    folded.head_dim = 128;
    if (bsgl::prefill_mfma_supported(folded, &why)) pa = folded;
  }
  const bool mfma_ok = bsgl::prefill_mfma_supported(pa, &why);
  if (kind == BSGL_DECODE_MFMA && !mfma_ok) return fail(BSGL_ERR_UNSUPPORTED, "matrix-core extend: %s", why);
  if (kind != BSGL_DECODE_VALU && mfma_ok) {
    ex.row_req = nullptr;  // only the running sums are needed
    hipError_t e = hipSuccess;
    if ((variant & BSGL_PREFILL_LIST_READY) == 0) {  // (else: an earlier call for this batch left them in the workspace)
      e = bsgl::launch_prefill_expand(ex, (hipStream_t)stream);
      if (e != hipSuccess) return hip_fail("prefill running-sum launch", e);
    }
    e = bsgl::launch_prefill_mfma(pa, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail("prefill launch", e);
    return BSGL_OK;
  }
  hipError_t e = bsgl::launch_prefill_expand(ex, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail("prefill expand launch", e);

  // every q row is an independent attention row over a length-limited page-table row
  const int32_t splits = prefill_row_splits(total_q_tokens, (int32_t)(table_cols > (1 << 30) ? (1 << 30) : table_cols));
  a.out = out;
  a.out_row_stride = out_row_stride;
  a.lse = nullptr;
  a.q = q;
  a.q_row_stride = q_row_stride;
  a.req_idx = ex.row_req;
  a.seq_lens = ex.row_len;
  a.rows = total_q_tokens;
  a.n_q_heads = n_q_heads;
  a.n_kv_heads = n_kv_heads;
  a.head_dim = head_dim;
  a.scale = scale;
  a.dtype = dtype;
  a.out_dtype = dtype;
  a.n_splits = splits;
  const size_t rh = (size_t)total_q_tokens * (size_t)n_q_heads;
  const size_t cb = kCounterBytes;
  a.ws_o = (float*)(ws + cb);
  a.ws_lse = splits > 1 ? (float*)(ws + cb + align_up((size_t)splits * rh * (size_t)head_dim * 4, 256)) : nullptr;
  return run_decode(a, BSGL_DECODE_VALU, nullptr, (hipStream_t)stream);
}

int bsgl_merge_attn_states(void* out, int64_t out_row_stride, float* lse_out, const float* parts_o,
                           const float* parts_lse, int32_t n_parts, int32_t rows, int32_t n_heads,
                           int32_t head_dim, int32_t dtype, void* stream) {
  if (int rc = check_dtype(dtype)) return rc;
  if (rows < 0 || n_parts < 1 || n_parts > 64 || n_heads <= 0)
    return fail(BSGL_ERR_BAD_ARG, "rows < 0, n_parts outside [1, 64] or n_heads <= 0");
  if (rows == 0) return BSGL_OK;
  if (head_dim != 64 && head_dim != 128) return fail(BSGL_ERR_UNSUPPORTED, "head_dim must be 64 or 128");
  if (!out || !parts_o || !parts_lse) return fail(BSGL_ERR_BAD_ARG, "null device pointer (out/parts_o/parts_lse)");
  const int eb = bsgl::dtype_bytes(dtype);
  if (!aligned16(out) || !aligned16(parts_o) || (out_row_stride * eb) % 16)
    return fail(BSGL_ERR_BAD_ARG, "out / parts_o must be 16-byte aligned with 16-byte-multiple row stride");
  if (out_row_stride < (int64_t)n_heads * head_dim) return fail(BSGL_ERR_BAD_ARG, "out row stride shorter than n_heads * head_dim");
  bsgl::MergeArgs m;
  m.out = out;
  m.out_row_stride = out_row_stride;
  m.lse_out = lse_out;
  m.parts_o = parts_o;
  m.parts_lse = parts_lse;
  m.n_parts = n_parts;
  m.rows = rows;
  m.n_heads = n_heads;
  m.head_dim = head_dim;
  m.dtype = dtype;
  const hipError_t e = bsgl::launch_merge(m, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail("merge launch", e);
  return BSGL_OK;
}

int bsgl_rmsnorm(void* out, int64_t out_row_stride, void* residual_out, const void* x, int64_t x_row_stride,
                 const void* residual, const void* weight, int32_t rows, int32_t hidden, float eps,
                 int32_t dtype, void* stream) {
  if (int rc = check_dtype(dtype)) return rc;
  if (rows < 0 || hidden <= 0) return fail(BSGL_ERR_BAD_ARG, "rows < 0 or hidden <= 0");
  if (rows == 0) return BSGL_OK;
  if (!out || !x || !weight) return fail(BSGL_ERR_BAD_ARG, "null device pointer (out/x/weight)");
  if (residual_out != nullptr && residual == nullptr)
    return fail(BSGL_ERR_BAD_ARG, "residual_out without residual");
  const int eb = bsgl::dtype_bytes(dtype);
  if ((hidden * eb) % 16) return fail(BSGL_ERR_UNSUPPORTED, "hidden size %d is not a multiple of 16 bytes", hidden);
  if (hidden > bsgl::rmsnorm_max_hidden(dtype))
    return fail(BSGL_ERR_UNSUPPORTED, "hidden size %d above %d", hidden, bsgl::rmsnorm_max_hidden(dtype));
  if (!aligned16(out) || !aligned16(x) || !aligned16(weight) || !aligned16(residual) || !aligned16(residual_out))
    return fail(BSGL_ERR_BAD_ARG, "out, x, weight and residuals must be 16-byte aligned");
  if ((x_row_stride * eb) % 16 || (out_row_stride * eb) % 16 || x_row_stride < hidden || out_row_stride < hidden)
    return fail(BSGL_ERR_BAD_ARG, "row strides must be multiples of 16 bytes and at least hidden");
  bsgl::NormArgs a{};
  a.out = out;
  a.residual_out = residual_out;
  a.x = x;
  a.residual = residual;
  a.weight = weight;
  a.x_row_stride = x_row_stride;
  a.out_row_stride = out_row_stride;
  a.rows = rows;
  a.hidden = hidden;
  a.eps = eps;
  a.dtype = dtype;
  const hipError_t e = bsgl::launch_rmsnorm(a, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail("rmsnorm launch", e);
  return BSGL_OK;
}

int bsgl_rope_kv_write(void* q_out, int64_t q_out_row_stride, void* k_out, void* v_out, int64_t kv_out_row_stride,
                       const void* qkv, int64_t qkv_row_stride, const int64_t* positions, const float* inv_freq,
                       void* k_slab, void* v_slab, int64_t slab_row_stride, int64_t n_slots,
                       const int32_t* out_cache_loc, int32_t n_tok, int32_t n_q_heads, int32_t n_kv_heads,
                       int32_t head_dim, int32_t dtype, void* stream) {
  if (int rc = check_dtype(dtype)) return rc;
  if (n_tok < 0) return fail(BSGL_ERR_BAD_ARG, "n_tok < 0");
  if (n_tok == 0) return BSGL_OK;
  if (!q_out || !qkv || !positions || !inv_freq) return fail(BSGL_ERR_BAD_ARG, "null device pointer (q_out/qkv/positions/inv_freq)");
  if (n_q_heads <= 0 || n_kv_heads <= 0 || head_dim <= 0) return fail(BSGL_ERR_BAD_ARG, "head counts and head_dim must be positive");
  const int eb = bsgl::dtype_bytes(dtype);
  if (head_dim > 256 || (head_dim / 2 * eb) % 16 || head_dim % 2)
    return fail(BSGL_ERR_UNSUPPORTED, "head_dim %d: half a head row must be a multiple of 16 bytes, head_dim <= 256", head_dim);
  if ((k_out == nullptr) != (v_out == nullptr)) return fail(BSGL_ERR_BAD_ARG, "k_out and v_out go together");
  if (out_cache_loc != nullptr && (!k_slab || !v_slab || n_slots <= 0))
    return fail(BSGL_ERR_BAD_ARG, "out_cache_loc given without KV slabs");
  if (out_cache_loc == nullptr && k_out == nullptr)
    return fail(BSGL_ERR_BAD_ARG, "neither a cache write (out_cache_loc) nor k_out / v_out requested");
  const int64_t q_cols = (int64_t)n_q_heads * head_dim, kv_cols = (int64_t)n_kv_heads * head_dim;
  if (!aligned16(q_out) || !aligned16(qkv) || !aligned16(k_out) || !aligned16(v_out) || !aligned16(k_slab) || !aligned16(v_slab))
    return fail(BSGL_ERR_BAD_ARG, "q_out, qkv, k_out, v_out and slabs must be 16-byte aligned");
  if ((q_out_row_stride * eb) % 16 || (qkv_row_stride * eb) % 16 || (kv_out_row_stride * eb) % 16 || (slab_row_stride * eb) % 16)
    return fail(BSGL_ERR_BAD_ARG, "row strides must be multiples of 16 bytes");
  if (q_out_row_stride < q_cols || qkv_row_stride < q_cols + 2 * kv_cols || (k_out && kv_out_row_stride < kv_cols) ||
      (out_cache_loc && slab_row_stride < kv_cols))
    return fail(BSGL_ERR_BAD_ARG, "a row stride is shorter than its row");
  bsgl::RopeArgs a{};
  a.q_out = q_out;
  a.q_out_row_stride = q_out_row_stride;
  a.k_out = k_out;
  a.v_out = v_out;
  a.kv_out_row_stride = kv_out_row_stride;
  a.qkv = qkv;
  a.qkv_row_stride = qkv_row_stride;
  a.positions = positions;
  a.inv_freq = inv_freq;
  a.k_slab = k_slab;
  a.v_slab = v_slab;
  a.slab_row_stride = slab_row_stride;
  a.n_slots = n_slots;
  a.out_cache_loc = out_cache_loc;
  a.n_tok = n_tok;
  a.n_q_heads = n_q_heads;
  a.n_kv_heads = n_kv_heads;
  a.head_dim = head_dim;
  a.dtype = dtype;
  const hipError_t e = bsgl::launch_rope_kv(a, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail("rope launch", e);
  return BSGL_OK;
}

}  // extern "C"
