// One-shot all-reduce for the tensor-parallel boundary's small messages (the [batch, hidden] sum
// behind a row-parallel projection: 192 KiB at 64 x 1536 bf16, tp_runtime.py:27-33), without a
// collective library: every rank owns a WINDOW of device memory that its peers map (IPC handles
// between processes, plain pointers inside one process), writes its addend there, raises a flag,
// waits for the peers' flags and reads all addends straight out of the peers' windows.  One
// launch per call, every byte crosses a link once, and every rank adds the addends in rank order
// in fp32, so all ranks get the same bits.
//
// Window layout:   [0, 4096)            64 flags, 64 bytes apart: flag g = the number of the last
//                                       call whose chunk g of this rank's addend is in place
//                  [4096, 4100)         number of calls completed by this rank
//                  [4160, 4164)         workgroups of the running call that have finished
//                  [4224, 4228)         status: 0, or kTimedOut once a wait for a peer gave up
//                  [8192, 8192 + 2 MiB) two slots of kMaxBytes; call n uses slot n & 1
//
// Protocol of call n on every rank (n is read from the window, so a captured launch replays right):
// workgroup g copies chunk g of the addend into its own slot n & 1 with system-scope write-through
// stores, waits for them to leave, and one lane stores flag g = n (system scope).  One lane per
// peer polls that peer's flag g until it reads >= n; behind a workgroup barrier the chunk of every
// rank is loaded with system-scope loads (they bypass the caches of the reading GPU), summed in
// rank order and written to the caller's buffer.  The last workgroup to finish stores n as the
// number of completed calls.  A slot is written again two calls later: by then this rank has
// seen a flag of every peer for call n + 1, and a peer raises those only after its call n has
// ended (calls of one rank are ordered by its stream), so nobody still reads the old bytes.
// Every wait is bounded (knob BSGL_P2P_WAIT_MS, 10 s by default): a peer that never arrives makes
// the call give up, leave garbage in `buf` and set the window's status word, and every later call
// returns at once.
//
// The calls of one communicator must all be made on ONE stream (or otherwise ordered), by every
// rank in the same order with the same count and dtype.
#include "../../include/bsgl_attn.h"
#include "bsgl_device.h"
#include "bsgl_launch.h"

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

namespace bsgl {
int set_error(int code, const char* msg);
}

struct bsgl_p2p_comm {
  int rank = -1, world = 0, device = -1;
  bool connected = false;
  void* window = nullptr;                    // this rank's window (hipExtMallocWithFlags)
  void* peers[BSGL_P2P_MAX_WORLD] = {};      // every rank's window as this process sees it (own one included)
  bool opened[BSGL_P2P_MAX_WORLD] = {};      // mapped by hipIpcOpenMemHandle: closed on destroy
};

namespace {

using namespace bsgl;

constexpr int kMaxWorld = BSGL_P2P_MAX_WORLD;
constexpr size_t kMaxBytes = BSGL_P2P_MAX_BYTES;
constexpr int kFlagStride = 64;              // bytes between two flags
constexpr int kMaxGroups = 64;               // workgroups (= chunks = flags) per call at most
constexpr size_t kDoneOffset = 4096, kArrivedOffset = 4160, kStatusOffset = 4224, kSlotOffset = 8192;
constexpr size_t kWindowBytes = kSlotOffset + 2 * kMaxBytes;
constexpr int kThreads = 256;
constexpr uint64_t kTicksPerMs = 100000;     // the wall clock the kernel reads runs at 100 MHz
constexpr uint32_t kTimedOut = 1;
constexpr int kSys = 17;                     // buffer aux bits sc0 | sc1: system scope, write-through / cache bypass

static_assert(BSGL_P2P_HANDLE_BYTES == sizeof(hipIpcMemHandle_t), "IPC handle size");
static_assert(kMaxGroups * kFlagStride <= (int)kDoneOffset, "flags fit below the counters");

struct Windows {
  uint8_t* w[kMaxWorld];
};

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

__device__ __forceinline__ uint32_t* word(uint8_t* window, size_t offset) { return (uint32_t*)(window + offset); }

template <typename Tag>
__global__ __launch_bounds__(kThreads) void p2p_all_reduce_kernel(Windows win, int rank, int world, uint8_t* buf, int n_vec,
                                                                 uint64_t wait_ticks) {
  using E = Elem<Tag>;
  uint8_t* mine = win.w[rank];
  const int g = blockIdx.x, n_groups = gridDim.x, tid = threadIdx.x;
  // (nobody writes the call counter while a call runs: every workgroup reads the same number)
  const uint32_t done = __hip_atomic_load(word(mine, kDoneOffset), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t status = __hip_atomic_load(word(mine, kStatusOffset), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t call = done + 1;
  const int per = (n_vec + n_groups - 1) / n_groups;
  const int v0 = g * per, v1 = min(n_vec, v0 + per);
  const size_t slot = kSlotOffset + (size_t)(call & 1) * kMaxBytes;
  __shared__ int s_failed;
  if (tid == 0) s_failed = 0;
  if (status == 0) {
    // -- this rank's chunk into its own window, write-through at system scope
    const auto own = __builtin_amdgcn_make_buffer_rsrc((void*)(mine + slot), 0, (int)kMaxBytes, 0x00020000);
    const uint4* in = (const uint4*)buf;
    for (int v = v0 + tid; v < v1; v += kThreads)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, in[v]), own, v * 16, 0, kSys);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every byte of this wave has left
    __syncthreads();
    if (tid == 0) {
      // system-scope release in front of the flag: the stores above are write-through already, the
      // fence gives the same guarantee through the memory model instead of by hand (its own wait is
      // written out: the compiler may drop the one it implies when it sees the counter at zero)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(word(mine, (size_t)g * kFlagStride), call, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // -- one lane per peer waits for that peer's chunk
    if (tid < world && tid != rank) {
      const uint32_t* flag = word(win.w[tid], (size_t)g * kFlagStride);
      const uint64_t t0 = wall_clock64();
      while ((int32_t)(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - call) < 0) {
        if (wall_clock64() - t0 > wait_ticks) {
          s_failed = 1;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    if (tid < 64) {
      // one relaxed poll per peer, then ONE system-scope acquire by the polling wave, waited for in
      // front of the barrier that lets the other waves load (the loads below also bypass the caches)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (s_failed == 0) {
      // -- all ranks' chunks, summed in rank order
      for (int v = v0 + tid; v < v1; v += kThreads) {
        float acc[E::VEC];
#pragma unroll
        for (int i = 0; i < E::VEC; ++i) acc[i] = 0.f;
        u32x4_t part[kMaxWorld];
#pragma unroll
        for (int r = 0; r < kMaxWorld; ++r) {
          if (r < world) {
            const auto src = __builtin_amdgcn_make_buffer_rsrc((void*)(win.w[r] + slot), 0, (int)kMaxBytes, 0x00020000);
            part[r] = __builtin_amdgcn_raw_buffer_load_b128(src, v * 16, 0, kSys);
          }
        }
#pragma unroll
        for (int r = 0; r < kMaxWorld; ++r) {
          if (r < world) {
            float f[E::VEC];
            E::unpack(__builtin_bit_cast(uint4, part[r]), f);
#pragma unroll
            for (int i = 0; i < E::VEC; ++i) acc[i] += f[i];
          }
        }
        ((uint4*)buf)[v] = E::pack(acc);
      }
    } else if (tid == 0) {
      __hip_atomic_store(word(mine, kStatusOffset), kTimedOut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // -- the last workgroup to get here closes the call
  __syncthreads();
  if (tid == 0) {
    const uint32_t arrived = __hip_atomic_fetch_add(word(mine, kArrivedOffset), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived == (uint32_t)n_groups - 1) {
      __hip_atomic_store(word(mine, kArrivedOffset), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(word(mine, kDoneOffset), call, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int fail(int code, const char* fmt, const char* a = "") {
  char msg[300];
  snprintf(msg, sizeof(msg), fmt, a);
  return bsgl::set_error(code, msg);
}

int hip_fail(const char* what, hipError_t e) {
  char msg[300];
  snprintf(msg, sizeof(msg), "%s: %s", what, hipGetErrorString(e));
  return bsgl::set_error(BSGL_ERR_HIP, msg);
}

}  // namespace

extern "C" {

int bsgl_p2p_create(int32_t rank, int32_t world, bsgl_p2p_comm** comm_out, void* handle_out, void** window_out) {
  if (comm_out == nullptr) return fail(BSGL_ERR_BAD_ARG, "null communicator pointer%s");
  *comm_out = nullptr;
  if (world < 1 || world > kMaxWorld || rank < 0 || rank >= world)
    return fail(BSGL_ERR_BAD_ARG, "rank outside [0, world) or world beyond BSGL_P2P_MAX_WORLD%s");
  int device = -1;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return hip_fail("hipGetDevice", e);
  void* window = nullptr;
  // fine-grained: stores of one GPU become visible to loads of another while kernels run
  e = hipExtMallocWithFlags(&window, kWindowBytes, hipDeviceMallocFinegrained);
  if (e != hipSuccess) return hip_fail("hipExtMallocWithFlags(window)", e);
  e = hipMemset(window, 0, kSlotOffset);  // flags, counters, status (set-up: this call may wait for the device)
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    (void)hipFree(window);
    return hip_fail("hipMemset(window)", e);
  }
  if (handle_out != nullptr) {
    hipIpcMemHandle_t h;
    e = hipIpcGetMemHandle(&h, window);
    if (e != hipSuccess) {
      (void)hipFree(window);
      return hip_fail("hipIpcGetMemHandle", e);
    }
    memcpy(handle_out, &h, sizeof(h));
  }
  bsgl_p2p_comm* c = new bsgl_p2p_comm();
  c->rank = rank;
  c->world = world;
  c->device = device;
  c->window = window;
  c->peers[rank] = window;
  if (window_out != nullptr) *window_out = window;
  *comm_out = c;
  return BSGL_OK;
}

int bsgl_p2p_connect(bsgl_p2p_comm* c, const void* handles, void* const* windows) {
  if (c == nullptr) return fail(BSGL_ERR_BAD_ARG, "null communicator%s");
  if (c->connected) return fail(BSGL_ERR_BAD_ARG, "communicator already connected%s");
  if ((handles == nullptr) == (windows == nullptr))
    return fail(BSGL_ERR_BAD_ARG, "give the peers' windows either as IPC handles or as pointers%s");
  int device = -1;
  hipError_t e = hipGetDevice(&device);
  if (e != hipSuccess) return hip_fail("hipGetDevice", e);
  if (device != c->device) return fail(BSGL_ERR_BAD_ARG, "connect on the device the communicator was created on%s");
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank) continue;
    if (windows != nullptr) {
      if (windows[r] == nullptr) return fail(BSGL_ERR_BAD_ARG, "null peer window%s");
      c->peers[r] = windows[r];
      continue;
    }
    hipIpcMemHandle_t h;
    memcpy(&h, (const uint8_t*)handles + (size_t)r * sizeof(h), sizeof(h));
    void* p = nullptr;
    e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return hip_fail("hipIpcOpenMemHandle", e);
    c->peers[r] = p;
    c->opened[r] = true;
  }
  c->connected = true;
  return BSGL_OK;
}

int bsgl_p2p_all_reduce_sum(bsgl_p2p_comm* c, void* buf, int64_t count, int32_t dtype, void* stream) {
  if (c == nullptr || !c->connected) return fail(BSGL_ERR_BAD_ARG, "no connected communicator (bsgl_p2p_create, bsgl_p2p_connect first)%s");
  if (count < 0) return fail(BSGL_ERR_BAD_ARG, "count < 0%s");
  if (count == 0 || c->world == 1) return BSGL_OK;
  if (buf == nullptr) return fail(BSGL_ERR_BAD_ARG, "null device pointer (buf)%s");
  const int64_t esz = dtype == BSGL_DTYPE_F32 ? 4 : 2;
  if (dtype != BSGL_DTYPE_F32 && dtype != BSGL_DTYPE_BF16 && dtype != BSGL_DTYPE_F16)
    return fail(BSGL_ERR_UNSUPPORTED, "p2p all-reduce: dtype is not bf16/f16/f32%s");
  const int64_t bytes = count * esz;
  if (bytes > (int64_t)kMaxBytes) return fail(BSGL_ERR_UNSUPPORTED, "p2p all-reduce: message beyond BSGL_P2P_MAX_BYTES%s");
  if (bytes % 16 != 0 || ((uintptr_t)buf & 15) != 0)
    return fail(BSGL_ERR_UNSUPPORTED, "p2p all-reduce: the message must be a multiple of 16 bytes at a 16-byte boundary%s");
  const int n_vec = (int)(bytes / 16);
  // four 16-byte pieces per thread before another workgroup (another flag round trip) pays
  int groups = (n_vec + 4 * kThreads - 1) / (4 * kThreads);
  groups = groups < 1 ? 1 : (groups > kMaxGroups ? kMaxGroups : groups);
  Windows win;
  for (int r = 0; r < kMaxWorld; ++r) win.w[r] = (uint8_t*)c->peers[r < c->world ? r : c->rank];
  const dim3 grid(groups), block(kThreads);
  hipStream_t s = (hipStream_t)stream;
  const int wait_ms = bsgl::knob(bsgl::kKnobP2PWaitMs);
  const uint64_t ticks = (uint64_t)(wait_ms < 1 ? 1 : wait_ms) * kTicksPerMs;
  switch (dtype) {
    case BSGL_DTYPE_BF16: p2p_all_reduce_kernel<bf16_tag><<<grid, block, 0, s>>>(win, c->rank, c->world, (uint8_t*)buf, n_vec, ticks); break;
    case BSGL_DTYPE_F16: p2p_all_reduce_kernel<f16_tag><<<grid, block, 0, s>>>(win, c->rank, c->world, (uint8_t*)buf, n_vec, ticks); break;
    default: p2p_all_reduce_kernel<f32_tag><<<grid, block, 0, s>>>(win, c->rank, c->world, (uint8_t*)buf, n_vec, ticks); break;
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail("p2p all-reduce launch", e);
  return BSGL_OK;
}

int bsgl_p2p_status(bsgl_p2p_comm* c, int32_t* status_out, int64_t* calls_out) {
  if (c == nullptr) return fail(BSGL_ERR_BAD_ARG, "null communicator%s");
  uint32_t words[(kStatusOffset - kDoneOffset) / 4 + 1];
  const hipError_t e = hipMemcpy(words, (const uint8_t*)c->window + kDoneOffset, sizeof(words), hipMemcpyDeviceToHost);
  if (e != hipSuccess) return hip_fail("hipMemcpy(window status)", e);
  if (calls_out != nullptr) *calls_out = words[0];
  if (status_out != nullptr) *status_out = (int32_t)words[(kStatusOffset - kDoneOffset) / 4];
  return BSGL_OK;
}

int bsgl_p2p_status_async(bsgl_p2p_comm* c, int32_t* host_words, void* stream) {
  if (c == nullptr || host_words == nullptr) return fail(BSGL_ERR_BAD_ARG, "null communicator or host buffer%s");
  // status word and call counter: two copies of 4 bytes behind the calls already enqueued on the stream
  hipError_t e = hipMemcpyAsync(host_words, (const uint8_t*)c->window + kStatusOffset, 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess)
    e = hipMemcpyAsync(host_words + 1, (const uint8_t*)c->window + kDoneOffset, 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e != hipSuccess) return hip_fail("hipMemcpyAsync(window status)", e);
  return BSGL_OK;
}

int bsgl_p2p_destroy(bsgl_p2p_comm* c) {
  if (c == nullptr) return BSGL_OK;
  hipError_t first = hipSuccess;
  for (int r = 0; r < c->world; ++r) {
    if (!c->opened[r]) continue;
    const hipError_t e = hipIpcCloseMemHandle(c->peers[r]);
    if (e != hipSuccess && first == hipSuccess) first = e;
  }
  if (c->window != nullptr) {
    const hipError_t e = hipFree(c->window);
    if (e != hipSuccess && first == hipSuccess) first = e;
  }
  delete c;
  if (first != hipSuccess) return hip_fail("bsgl_p2p_destroy", first);
  return BSGL_OK;
}

}  // extern "C"
