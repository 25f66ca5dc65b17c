// KV scatter and prefill row expansion for gfx950.
//
// kv_scatter_kernel: k_slab[loc[t]] = k_new[t]; v_slab[loc[t]] = v_new[t]
// (reference: memory_pool.py:152-171).  Pure byte movement, 16 B per lane, so one
// instance serves every dtype; source rows may be strided views.
//
// prefill_expand: turns the per-request (prefix_lens, extended_lens) of an
// extend batch into one (page-table row, visible length) pair per q row, so
// that row r of request i sees positions 0 .. prefix_lens[i] + r -- the
// causal-with-prefix mask of attn_backend.py:74-92 expressed as a length.
#include "bsgl_device.h"
#include "bsgl_launch.h"

namespace bsgl {

namespace {

__global__ __launch_bounds__(256) void kv_scatter_kernel(ScatterArgs a) {
  const int pieces = a.row_bytes >> 4;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)a.n_tok * pieces;
  if (gid >= total) return;
  const int t = (int)(gid / pieces);
  const int p = (int)(gid % pieces);
  const int64_t slot = a.loc[t];
  if (slot < 0 || slot >= a.n_slots) return;
  const int64_t dst = slot * a.slab_row_stride * a.elem_bytes + (int64_t)p * 16;
  const uint4 kv = *(const uint4*)((const char*)a.k_new + (int64_t)t * a.k_row_stride * a.elem_bytes + (int64_t)p * 16);
  const uint4 vv = *(const uint4*)((const char*)a.v_new + (int64_t)t * a.v_row_stride * a.elem_bytes + (int64_t)p * 16);
  *(uint4*)((char*)a.k_slab + dst) = kv;
  *(uint4*)((char*)a.v_slab + dst) = vv;
}

// exclusive running sums of extended_lens and of the block groups of each request,
// ceil(ceil(extended_lens / blk_rows) / grp_blocks)  (negative lengths count as 0, lengths above
// the number of q rows as that number)
__global__ void prefill_cu_kernel(ExpandArgs a) {
  __shared__ int64_t s_ext[256], s_blk[256];
  const int tid = threadIdx.x;
  const int per = (a.batch + 255) / 256;
  const int lo = min(tid * per, a.batch), hi = min(lo + per, a.batch);
  // (lengths are bounded by the number of q rows, an int: 32-bit unsigned divisions, a fraction of the
  // instructions of 64-bit ones)
  const uint32_t br = a.blk_rows > 0 ? (uint32_t)a.blk_rows : 16u;
  const uint32_t gb = a.grp_blocks > 0 ? (uint32_t)a.grp_blocks : 1u;
  auto groups_of = [&](int64_t e) { return (int64_t)((((uint32_t)e + br - 1) / br + gb - 1) / gb); };
  int64_t sum = 0, blocks = 0;
  for (int i = lo; i < hi; ++i) {
    const int64_t e = min(max((int64_t)0, a.extended_lens[i]), (int64_t)a.total_q_tokens);  // (bounded: the sums cannot overflow)
    sum += e;
    blocks += groups_of(e);
  }
  // exclusive scan over the 256 threads' partial sums: inside a wave by shuffles, across the four waves
  // through LDS (one thread walking all 256 entries took 4 of this launch's 8 microseconds)
  const int lane = tid & 63, wave = tid >> 6;
  int64_t inc_e = sum, inc_b = blocks;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int64_t oe = __shfl_up(inc_e, d), ob = __shfl_up(inc_b, d);
    if (lane >= d) { inc_e += oe; inc_b += ob; }
  }
  if (lane == 63) { s_ext[wave] = inc_e; s_blk[wave] = inc_b; }
  __syncthreads();
  int64_t run = inc_e - sum, brun = inc_b - blocks;
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    if (w < wave) { run += s_ext[w]; brun += s_blk[w]; }
  }
  if (tid == 255) {
    a.cu_ext[a.batch] = run + sum;
    if (a.cu_blk != nullptr) a.cu_blk[a.batch] = brun + blocks;
  }
  for (int i = lo; i < hi; ++i) {
    const int64_t e = min(max((int64_t)0, a.extended_lens[i]), (int64_t)a.total_q_tokens);  // (bounded: the sums cannot overflow)
    a.cu_ext[i] = run;
    if (a.cu_blk != nullptr) a.cu_blk[i] = brun;
    run += e;
    brun += groups_of(e);
  }
}

__global__ __launch_bounds__(256) void prefill_rows_kernel(ExpandArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= a.total_q_tokens) return;
  // last request i with cu_ext[i] <= t
  int lo = 0, hi = a.batch;  // invariant: cu_ext[lo] <= t (cu_ext[0] == 0), answer in [lo, hi)
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (a.cu_ext[mid] <= t) lo = mid; else hi = mid;
  }
  // requests with extended_lens == 0 share a start with their successor; the
  // search above lands on the LAST request starting at or before t, which is
  // the one that owns the row.
  int64_t len = 0, req = 0;
  if (a.batch > 0 && t < a.cu_ext[a.batch]) {
    const int64_t r = t - a.cu_ext[lo];
    req = a.req_pool_indices[lo];
    len = a.prefix_lens[lo] + r + 1;
    const int64_t full = a.seq_lens[lo];
    len = len > full ? full : len;
    len = len < 0 ? 0 : len;
  }
  a.row_req[t] = req;
  a.row_len[t] = len;
}

}  // namespace

hipError_t launch_kv_scatter(const ScatterArgs& a, hipStream_t stream) {
  if (a.n_tok == 0) return hipSuccess;
  const int64_t total = (int64_t)a.n_tok * (a.row_bytes >> 4);
  hipLaunchKernelGGL(kv_scatter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_prefill_expand(const ExpandArgs& a, hipStream_t stream) {
  hipLaunchKernelGGL(prefill_cu_kernel, dim3(1), dim3(256), 0, stream, a);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || a.total_q_tokens == 0 || a.row_req == nullptr) return e;
  hipLaunchKernelGGL(prefill_rows_kernel, dim3((a.total_q_tokens + 255) / 256), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace bsgl
