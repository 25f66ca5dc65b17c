/*
 * bsgl_attn.h -- C ABI of libbsgl_attn.so, the MI355X (gfx950) paged-attention
 * library that stands behind baby-sglang's attention backend.
 *
 * Conventions
 *   - Every pointer marked "device" is a HIP device address owned by the caller
 *     for the duration of the call; the library keeps no pointer after return.
 *   - Every launch goes onto `stream` (a hipStream_t passed as void*; NULL is the
 *     default stream).  No entry point synchronises the device, reads device
 *     memory on the host, or allocates: callable from any host thread, with
 *     several streams in flight, and under HIP graph capture.
 *   - Strides are in ELEMENTS of the tensor's dtype (rows of q/k/v/out/slabs)
 *     or of int32 (page table).  Inner dimensions are dense: a row is
 *     n_heads * head_dim consecutive elements.
 *   - Return value: BSGL_OK or a negative BSGL_ERR_* code; bsgl_last_error()
 *     then returns a thread-local, NUL-terminated description.  Nothing is thrown
 *     across the ABI.
 *   - Page-table and slot indices are clamped inside the kernels to the table
 *     and slab extents given here, so inconsistent metadata gives wrong
 *     numbers, never a GPU memory fault.
 *
 * Reference interfaces these entry points stand behind are cited per function
 * as file:line of the baby-sglang sources.
 */
#ifndef BSGL_ATTN_H
#define BSGL_ATTN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSGL_OK 0
#define BSGL_ERR_BAD_ARG (-1)     /* NULL pointer, negative size, misaligned row */
#define BSGL_ERR_UNSUPPORTED (-2) /* dtype / head_dim / group size without a kernel */
#define BSGL_ERR_HIP (-3)         /* a HIP runtime call failed */
#define BSGL_ERR_WORKSPACE (-4)   /* workspace missing or too small */
#define BSGL_ERR_COMM (-5)        /* RCCL missing or an RCCL call failed */

/* element types of q / k / v / out / KV slabs */
#define BSGL_DTYPE_BF16 0
#define BSGL_DTYPE_F16 1
#define BSGL_DTYPE_F32 2

/* decode kernel selection (low byte of bsgl_paged_decode `variant`) */
#define BSGL_DECODE_AUTO 0
#define BSGL_DECODE_VALU 1 /* generic: any dtype, head_dim 64/128, group <= 8 */
#define BSGL_DECODE_MFMA 2 /* bf16/f16, head_dim 128: QK^T and PV on matrix cores */
/* flag or-ed into `variant`: the caller asserts that every slot in out_cache_loc
 * is referenced by no page-table row other than its own request's (what the
 * reference scheduler produces: scheduler.py:459-490 hands out fresh slots).
 * The KV write may then be fused into the decode kernel instead of being a
 * launch of its own; without the flag the write is always a separate launch. */
#define BSGL_DECODE_FUSE_KV_WRITE 0x100
/* OR into `variant` of bsgl_paged_decode to cut every request into exactly
 * n_splits equal pieces.  By default (matrix-core kernel) n_splits is only the
 * MOST pieces a request may get: pieces have one batch-wide length, about
 * sum(seq_lens) / (batch * ceil(n_splits / 2)) tokens, so that ragged batches
 * load every workgroup alike.  Results agree to rounding either way. */
#define BSGL_DECODE_UNIFORM_SPLITS 0x200
/* OR into `variant` of bsgl_paged_prefill when `workspace` still holds the work list (the running sums
 * of extended_lens and of the requests' block groups) that an earlier call of bsgl_paged_prefill wrote
 * there for the SAME batch -- same batch, total_q_tokens, contents of extended_lens, head counts, head_dim,
 * dtype and tuning knobs -- and nothing has written the workspace since: the launch that makes the list is
 * then left out.  That is every layer after the first of one model forward (the reference calls its
 * backend once per layer with one batch descriptor, qwen2.py:292-315): 4.7 us less per layer call.
 * Ignored where the generic path runs (it rebuilds its row list every time). */
#define BSGL_PREFILL_LIST_READY 0x400

/* ABI version of this header (bumped on any signature change). */
int bsgl_abi_version(void);

/* Tuning aid with no counterpart in the reference: when device_buf is not NULL,
 * every later matrix-core decode launch of this process writes 8 x uint64 per
 * workgroup into it (launch order): start, end of prologue, end of streaming, end
 * (100 MHz ticks; top bit of the last = this workgroup merged), XCC id, hardware
 * id, row, piece | pieces << 32.  The buffer must hold the largest grid. */
void bsgl_debug_decode_trace(void* device_buf);

/* Tuning aid with no counterpart in the reference: when device_buf is not NULL, every
 * later matrix-core extend launch writes 6 x uint64 per wave (workgroup launch index x
 * waves + wave): shader cycles spent waiting for its own DMA, at the barrier, issuing
 * DMA, in the score / softmax half, in the PV half, and its step count. */
void bsgl_debug_prefill_profile(void* device_buf);

/* Tuning aid with no counterpart in the reference: the library's tuning knobs are read from the
 * environment variables of their names (BSGL_MFMA_WAVES, BSGL_NO_FUSED_MERGE, ...) once per
 * process; this call replaces one knob's value afterwards and hands back the value it had
 * (previous may be NULL).  BSGL_ERR_BAD_ARG for a name the library does not know.  Meant for
 * tests and tuning scripts: a change while other threads are inside the library is seen by
 * their calls at an unspecified moment. */
int bsgl_debug_set_knob(const char* name, int32_t value, int32_t* previous);

/* Thread-local message of the last failing call on this thread ("" if none). */
const char* bsgl_last_error(void);

/*
 * KV scatter: k_slab[loc[t]] = k_new[t], v_slab[loc[t]] = v_new[t], t < n_tok.
 * Replaces MHATokenToKVPool.set_kv_buffer (memory_pool.py:152-171; called from
 * attn_backend.py:20-23).  k_new / v_new rows may be strided views (v is a
 * column slice of the merged qkv projection, qwen2.py:288); slab rows are
 * n_kv_heads * head_dim elements, slab_row_stride apart.  A loc outside
 * [0, n_slots) is skipped.  Rows must be 16-byte aligned.
 */
int bsgl_kv_scatter(void* k_slab /*device*/, void* v_slab /*device*/,
                    int64_t slab_row_stride, int64_t n_slots,
                    const int32_t* loc /*device, [n_tok]*/,
                    const void* k_new /*device*/, const void* v_new /*device*/,
                    int64_t k_row_stride, int64_t v_row_stride,
                    int32_t n_tok, int32_t n_kv_heads, int32_t head_dim,
                    int32_t dtype, void* stream);

/*
 * Number of sequence splits bsgl_paged_decode will use for this problem, and
 * the workspace it needs.  Both are pure host arithmetic.
 */
int32_t bsgl_decode_num_splits(int32_t batch, int32_t n_kv_heads, int32_t max_context_len);
size_t bsgl_decode_workspace_bytes(int32_t batch, int32_t n_q_heads, int32_t head_dim,
                                   int32_t n_splits, int32_t out_is_f32);
size_t bsgl_decode_counter_bytes(void);

/*
 * Paged decode attention, one query row per request.  Replaces
 * SimpleAttentionBackend._decode_attention (attn_backend.py:112-149):
 *   idx  = req_to_token[req_pool_indices[i], 0:seq_lens[i]]
 *   out[i, h, :] = softmax(scale * q[i, h, :] . K[idx, h / group, :]^T) . V[idx, h / group, :]
 * with group = n_q_heads / n_kv_heads (repeat_interleave head mapping,
 * attn_backend.py:132-135) and no mask.  seq_lens[i] counts the token written
 * this step; a request with seq_lens[i] <= 0 gets a zero row.
 *
 * out is [batch, n_q_heads * head_dim] in `out_dtype`, which is `dtype` or
 * BSGL_DTYPE_F32 (unrounded partial results for a later merge).  lse (optional, may be NULL)
 * receives the natural-log-sum-exp of the scaled scores, [batch, n_q_heads]
 * fp32, which lets partial results over disjoint token sets be merged with
 * bsgl_merge_attn_states.  n_splits comes from bsgl_decode_num_splits (or any
 * value in [1, 64]); workspace holds bsgl_decode_workspace_bytes bytes and may
 * be NULL when that is 0 (one split, out_dtype == dtype).
 *
 * k_new / v_new / out_cache_loc (all three or none, NULL = none): the new
 * token's rows, [batch, n_kv_heads * head_dim] with the given row strides, to be
 * stored at slab slot out_cache_loc[i] BEFORE attention reads the cache -- the
 * set_kv_buffer call the reference makes first (attn_backend.py:20-23).
 *
 * Workspace contract: its first bsgl_decode_counter_bytes() bytes (a constant)
 * hold the tickets of the in-kernel split merge.  They must be zero when a
 * workspace is first used; every call leaves them zero.  Two calls that may run
 * concurrently (different streams) need different workspaces.
 */
int bsgl_paged_decode(void* out /*device*/, int64_t out_row_stride,
                      float* lse /*device or NULL*/,
                      const void* q /*device*/, int64_t q_row_stride,
                      const void* k_new /*device or NULL*/, const void* v_new /*device or NULL*/,
                      int64_t k_new_row_stride, int64_t v_new_row_stride,
                      const int32_t* out_cache_loc /*device or NULL, [batch]*/,
                      void* k_slab /*device*/, void* v_slab /*device*/,
                      int64_t slab_row_stride, int64_t n_slots,
                      const int32_t* req_to_token /*device*/, int64_t table_row_stride,
                      int64_t table_rows, int64_t table_cols,
                      const int64_t* req_pool_indices /*device, [batch]*/,
                      const int64_t* seq_lens /*device, [batch]*/,
                      int32_t batch, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim,
                      float scale, int32_t dtype, int32_t out_dtype, int32_t n_splits,
                      int32_t variant,
                      void* workspace /*device*/, size_t workspace_bytes, void* stream);

/*
 * bsgl_paged_decode with its arguments in a struct (same meaning, same checks, same launch).
 * Stands behind the same reference call (attn_backend.py:112-149).  For host languages in which
 * every argument of a foreign call is converted one by one (Python's ctypes): a caller that
 * decodes layer after layer keeps one bsgl_decode_call per (layer, batch shape), fills the fields
 * that never change once (slabs, page table, head counts, splits, workspace) and per call only
 * the few that do (q, out, the new rows, the batch's index arrays, the stream).
 */
typedef struct bsgl_decode_call {
  void* out;
  int64_t out_row_stride;
  float* lse;
  const void* q;
  int64_t q_row_stride;
  const void* k_new;
  const void* v_new;
  int64_t k_new_row_stride;
  int64_t v_new_row_stride;
  const int32_t* out_cache_loc;
  void* k_slab;
  void* v_slab;
  int64_t slab_row_stride;
  int64_t n_slots;
  const int32_t* req_to_token;
  int64_t table_row_stride;
  int64_t table_rows;
  int64_t table_cols;
  const int64_t* req_pool_indices;
  const int64_t* seq_lens;
  int32_t batch;
  int32_t n_q_heads;
  int32_t n_kv_heads;
  int32_t head_dim;
  float scale;
  int32_t dtype;
  int32_t out_dtype;
  int32_t n_splits;
  int32_t variant;
  int32_t max_seq_len; /* 0 = unknown; else no request of this call is longer.  A HINT for the shape of
                          the launch only (a host sees the page table's width, not the lengths): results
                          never depend on it, a wrong value costs time */
  void* workspace;
  size_t workspace_bytes;
  void* stream;
} bsgl_decode_call;
int bsgl_paged_decode_call(const bsgl_decode_call* call);

/*
 * bsgl_paged_decode without a caller workspace and without any allocation.  No counterpart in
 * the reference: it is for callers that cannot allocate device memory -- code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc -- and that have no workspace
 * tensor to hand in.  Same arguments, kernels and results (bit for bit) as bsgl_paged_decode;
 * the tickets, partial rows and log-sum-exps of the split merge live in an array inside the
 * library's code object instead: BSGL_STATIC_SCRATCH_BYTES per GPU, held by every GPU the library
 * runs on until the process ends.  The call makes no hipMalloc / hipFree and calls no allocating
 * routine; it enqueues one hipMemsetAsync (this call's tickets) and the decode launch on `stream`.
 * While `stream` is being captured into a HIP graph the memset is left out -- as a graph node it
 * is not ordered in front of the launch when the graph is replayed (seen on ROCm 7.0 / 7.2) -- and
 * the tickets are the zeros that every finished launch leaves behind (the merging workgroup
 * resets its row's ticket); make one call outside the capture first, as a warm-up.
 * The first call on a device looks the array's address up (hipGetSymbolAddress, once per device,
 * with that device current -- the caller must have the stream's device current, as for every
 * entry point).
 * Limits: bsgl_decode_workspace_bytes(batch, n_q_heads, head_dim, n_splits, out_dtype != dtype)
 * must not exceed BSGL_STATIC_SCRATCH_BYTES, else BSGL_ERR_WORKSPACE (nothing is launched).  All
 * calls on one GPU share that GPU's array: calls that may overlap on one GPU (two streams, two
 * threads) race on it and must use bsgl_paged_decode with workspaces of their own.
 */
#define BSGL_STATIC_SCRATCH_BYTES ((size_t)16 << 20)
int bsgl_paged_decode_static(void* out /*device*/, int64_t out_row_stride,
                             float* lse /*device or NULL*/,
                             const void* q /*device*/, int64_t q_row_stride,
                             const void* k_new /*device or NULL*/, const void* v_new /*device or NULL*/,
                             int64_t k_new_row_stride, int64_t v_new_row_stride,
                             const int32_t* out_cache_loc /*device or NULL, [batch]*/,
                             void* k_slab /*device*/, void* v_slab /*device*/,
                             int64_t slab_row_stride, int64_t n_slots,
                             const int32_t* req_to_token /*device*/, int64_t table_row_stride,
                             int64_t table_rows, int64_t table_cols,
                             const int64_t* req_pool_indices /*device, [batch]*/,
                             const int64_t* seq_lens /*device, [batch]*/,
                             int32_t batch, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim,
                             float scale, int32_t dtype, int32_t out_dtype, int32_t n_splits,
                             int32_t variant, void* stream);

/*
 * Workspace for bsgl_paged_prefill (host arithmetic).
 */
size_t bsgl_prefill_workspace_bytes(int32_t batch, int32_t total_q_tokens, int32_t n_q_heads,
                                    int32_t head_dim);

/*
 * Paged prefill / extend attention.  Replaces
 * SimpleAttentionBackend._prefill_attention (attn_backend.py:34-110): request i
 * owns q rows [off_i, off_i + extended_lens[i]) with off_i the running sum of
 * extended_lens over earlier requests; row r of request i attends to page-table
 * positions 0 .. prefix_lens[i] + r (mask: col > prefix + row is -inf,
 * attn_backend.py:74-92) of the full sequence 0 .. seq_lens[i]-1, K/V read
 * through the page table.  Requests with extended_lens[i] == 0 own no rows.
 * total_q_tokens is the number of rows of q and out.
 */
int bsgl_paged_prefill(void* out /*device*/, int64_t out_row_stride,
                       const void* q /*device*/, int64_t q_row_stride,
                       const void* k_slab /*device*/, const void* v_slab /*device*/,
                       int64_t slab_row_stride, int64_t n_slots,
                       const int32_t* req_to_token /*device*/, int64_t table_row_stride,
                       int64_t table_rows, int64_t table_cols,
                       const int64_t* req_pool_indices /*device, [batch]*/,
                       const int64_t* seq_lens /*device, [batch]*/,
                       const int64_t* prefix_lens /*device, [batch]*/,
                       const int64_t* extended_lens /*device, [batch]*/,
                       int32_t batch, int32_t total_q_tokens,
                       int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim,
                       float scale, int32_t dtype, int32_t variant,
                       void* workspace /*device*/, size_t workspace_bytes, void* stream);

/*
 * Merge n_parts attention states computed over disjoint token sets:
 *   w_p = exp(lse_p - max_p lse_p);  out = sum_p w_p o_p / sum_p w_p
 * parts_o is [n_parts, rows, n_heads * head_dim] fp32 (already normalised),
 * parts_lse is [n_parts, rows, n_heads] fp32 (natural log; -inf marks an empty
 * part).  out is [rows, n_heads * head_dim] in `dtype`; lse_out (optional) gets
 * the merged lse.  Used for sequence-sliced tensor parallelism, where each rank
 * of a kv-head group holds a slice of every request's tokens; the reference
 * has no counterpart (tp_runtime.py:10-51 only wraps the collectives).
 */
int bsgl_merge_attn_states(void* out /*device*/, int64_t out_row_stride,
                           float* lse_out /*device or NULL*/,
                           const float* parts_o /*device*/, const float* parts_lse /*device*/,
                           int32_t n_parts, int32_t rows, int32_t n_heads, int32_t head_dim,
                           int32_t dtype, void* stream);

/*
 * The steps either side of the attention call (the reference's next callers).
 *
 * RMSNorm with optional residual add.  Replaces RMSNorm.forward (qwen2.py:125-141):
 *   h = x + residual (rounded to dtype; h = x when residual is NULL); residual_out = h
 *   out = weight * dtype(float(h) * rsqrt(mean(float(h)^2) + eps))
 * x / out rows may be strided views (strides in elements); residual, residual_out and
 * weight are dense.  residual_out may alias residual, out may alias x.
 */
int bsgl_rmsnorm(void* out /*device*/, int64_t out_row_stride, void* residual_out /*device or NULL*/,
                 const void* x /*device*/, int64_t x_row_stride, const void* residual /*device or NULL*/,
                 const void* weight /*device, [hidden]*/, int32_t rows, int32_t hidden, float eps,
                 int32_t dtype, void* stream);

/*
 * Rotary embedding of q and k, fused with the QKV split and the KV-cache write.
 * Replaces RotaryEmbedding.forward (qwen2.py:246-268: half-split rotation, cos / sin of
 * float(position) * inv_freq cast to dtype before the multiply), the split of the
 * projection row into q | k | v (qwen2.py:285-290) and MHATokenToKVPool.set_kv_buffer
 * (memory_pool.py:152-171).  qkv is [n_tok, (n_q_heads + 2 n_kv_heads) * head_dim];
 * rotated q goes to q_out [n_tok, n_q_heads * head_dim]; rotated k and v go to
 * k_slab / v_slab rows out_cache_loc[t] (slots outside [0, n_slots) are skipped) when
 * out_cache_loc is not NULL, and to k_out / v_out [n_tok, n_kv_heads * head_dim] when
 * those are not NULL (at least one destination is required).
 */
int bsgl_rope_kv_write(void* q_out /*device*/, int64_t q_out_row_stride,
                       void* k_out /*device or NULL*/, void* v_out /*device or NULL*/, int64_t kv_out_row_stride,
                       const void* qkv /*device*/, int64_t qkv_row_stride,
                       const int64_t* positions /*device, [n_tok]*/, const float* inv_freq /*device, [head_dim/2]*/,
                       void* k_slab /*device or NULL*/, void* v_slab /*device or NULL*/,
                       int64_t slab_row_stride, int64_t n_slots,
                       const int32_t* out_cache_loc /*device, [n_tok], or NULL*/,
                       int32_t n_tok, int32_t n_q_heads, int32_t n_kv_heads, int32_t head_dim,
                       int32_t dtype, void* stream);

/*
 * Tensor-parallel collectives on RCCL (xGMI), one communicator per process (one process per
 * GPU).  They stand behind the reference's torch.distributed wrappers: bsgl_comm_init behind
 * init_tp (tp_runtime.py:10-24; the unique id made by rank 0 with bsgl_comm_unique_id reaches
 * the other ranks by the caller's own means, e.g. the launcher's store), bsgl_all_reduce_sum
 * behind tp_all_reduce_sum (tp_runtime.py:27-33, in place), bsgl_all_gather behind
 * tp_all_gather (tp_runtime.py:36-44: rank r's `count_per_rank` elements land at
 * out + r * count_per_rank, i.e. concatenation along dim 0).  The communicator is made on the
 * CURRENT HIP device; collectives are enqueued on `stream` and return at once (no host wait:
 * they may be captured into a HIP graph behind the attention launches).  RCCL is loaded on the
 * first bsgl_comm_* call only (librccl.so.1); BSGL_ERR_COMM when it is missing or a call fails.
 * bsgl_comm_init blocks until every rank of the group has called it.
 */
#define BSGL_COMM_ID_BYTES 128
int bsgl_comm_unique_id(void* id_out /*host, BSGL_COMM_ID_BYTES*/);
int bsgl_comm_init(int32_t rank, int32_t world, const void* unique_id /*host, BSGL_COMM_ID_BYTES*/);
int bsgl_comm_world(void); /* ranks of the live communicator, 0 if none */
int bsgl_all_reduce_sum(void* buf /*device, count elements, in place*/, int64_t count, int32_t dtype, void* stream);
int bsgl_all_gather(void* out /*device, world * count_per_rank*/, const void* in /*device*/,
                    int64_t count_per_rank, int32_t dtype, void* stream);
int bsgl_comm_destroy(void);

/*
 * One-shot all-reduce for the small messages of the tensor-parallel boundary, without a
 * collective library -- an alternative to bsgl_all_reduce_sum behind tp_all_reduce_sum
 * (tp_runtime.py:27-33: the [batch, hidden] sum after a row-parallel projection, 192 KiB at
 * 64 x 1536 bf16, far below the size at which a ring pays).  Every rank owns a window of device
 * memory that its peers map; a call is ONE launch per rank: the rank writes its addend into its
 * window, raises a flag, waits for the peers' flags and reads all addends straight from the peers'
 * windows (every byte crosses a link once); all ranks add in rank order in fp32, so all get the
 * same bits.  Up to BSGL_P2P_MAX_WORLD ranks, messages up to BSGL_P2P_MAX_BYTES, a multiple of 16
 * bytes at a 16-byte boundary (BSGL_ERR_UNSUPPORTED otherwise: use bsgl_all_reduce_sum).
 *
 *   bsgl_p2p_create   on the CURRENT device: allocates the window (the one allocation; set-up,
 *                     may wait for the device) and returns a communicator; `handle_out` (host,
 *                     BSGL_P2P_HANDLE_BYTES, or NULL) receives the window's IPC handle for peers in
 *                     other processes, `window_out` (or NULL) its address for peers in this process.
 *   bsgl_p2p_connect  the peers' windows in rank order, EITHER as `handles` (world x
 *                     BSGL_P2P_HANDLE_BYTES, host; the own entry is ignored) OR as `windows` (world
 *                     pointers valid on this device); the other argument NULL.  How handles travel
 *                     between the ranks is the caller's business (the launcher's store, as for
 *                     bsgl_comm_init).
 *   bsgl_p2p_all_reduce_sum  in place on `buf`, enqueued on `stream`, no host wait, no allocation;
 *                     the call number lives in the window, so a captured launch can be replayed.
 *                     All calls of a communicator on ONE stream, every rank the same sequence of
 *                     (count, dtype).  A rank whose peer does not arrive within 10 s (knob BSGL_P2P_WAIT_MS) gives up:
 *                     `buf` is then undefined, the communicator's status becomes 1 and every later
 *                     call returns at once without touching `buf`.
 *   bsgl_p2p_status   copies the status (0 = good) and the number of completed calls to the host:
 *                     WAITS for the device; for set-up checks and tests, not for the serving loop.
 *   bsgl_p2p_status_async  enqueues the copy of {status, completed calls} into `host_words` (two
 *                     int32 of PINNED host memory) on `stream` and returns: the caller records an
 *                     event behind it and looks at the words once the event has passed -- how a serving
 *                     loop notices, a few steps late but without ever waiting, that a peer did not
 *                     arrive (from then on every sum is skipped: keep going and the model runs on
 *                     unsummed activations).  Not to be called while the stream is being captured.
 *                     Between GPUs the protocol rests on fine-grained memory, system-scope
 *                     release / acquire and cache-bypassing accesses as the ISA defines them; it has
 *                     only ever run with all ranks on ONE GPU (no link in between): cross-device
 *                     operation is unmeasured.
 *   bsgl_p2p_destroy  unmaps the peers' windows and frees the own one (after the peers are done).
 */
#define BSGL_P2P_HANDLE_BYTES 64
#define BSGL_P2P_MAX_WORLD 8
#define BSGL_P2P_MAX_BYTES (1 << 20)
typedef struct bsgl_p2p_comm bsgl_p2p_comm;
int bsgl_p2p_create(int32_t rank, int32_t world, bsgl_p2p_comm** comm_out,
                    void* handle_out /*host, BSGL_P2P_HANDLE_BYTES, or NULL*/, void** window_out /*or NULL*/);
int bsgl_p2p_connect(bsgl_p2p_comm* comm, const void* handles /*host, world x BSGL_P2P_HANDLE_BYTES, or NULL*/,
                     void* const* windows /*world device pointers, or NULL*/);
int bsgl_p2p_all_reduce_sum(bsgl_p2p_comm* comm, void* buf /*device, count elements, in place*/, int64_t count,
                            int32_t dtype, void* stream);
int bsgl_p2p_status(bsgl_p2p_comm* comm, int32_t* status_out /*or NULL*/, int64_t* calls_out /*or NULL*/);
int bsgl_p2p_status_async(bsgl_p2p_comm* comm, int32_t* host_words /*pinned host, 2 x int32*/, void* stream);
int bsgl_p2p_destroy(bsgl_p2p_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* BSGL_ATTN_H */
