// Collectives of the tensor-parallel boundary on RCCL, behind the C ABI (include/bsgl_attn.h):
// one communicator per process (one process per GPU), every collective enqueued on the
// caller's HIP stream -- nothing here synchronises, so the calls can be captured into a HIP
// graph together with the attention launches.  They stand behind the reference's
// torch.distributed wrappers (tp_runtime.py:10-51: init_tp, tp_all_reduce_sum, tp_all_gather).
//
// RCCL is bound at run time (dlopen of librccl.so.1 on the first bsgl_comm_* call): the
// attention entry points of this library neither load nor need it, and a process that already
// holds RCCL (PyTorch ships one under the same soname) shares that copy.
#include "../../include/bsgl_attn.h"

#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <mutex>

namespace bsgl {
int set_error(int code, const char* msg);
}

namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclAllGather) all_gather = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
};

Rccl g_rccl;
std::mutex g_mu;
ncclComm_t g_comm = nullptr;
int g_world = 0, g_rank = -1;

static_assert(BSGL_COMM_ID_BYTES == sizeof(ncclUniqueId), "unique id size");

int fail(int code, const char* fmt, const char* a = "", const char* b = "") {
  char buf[400];
  snprintf(buf, sizeof(buf), fmt, a, b);
  return bsgl::set_error(code, buf);
}

// (called with g_mu held)
int bind_rccl() {
  if (g_rccl.handle != nullptr) return BSGL_OK;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h != nullptr) break;
  }
  if (h == nullptr) return fail(BSGL_ERR_COMM, "RCCL not found: %s", dlerror());
  Rccl r;
  r.handle = h;
  r.get_unique_id = (decltype(r.get_unique_id))dlsym(h, "ncclGetUniqueId");
  r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(h, "ncclCommInitRank");
  r.comm_destroy = (decltype(r.comm_destroy))dlsym(h, "ncclCommDestroy");
  r.all_reduce = (decltype(r.all_reduce))dlsym(h, "ncclAllReduce");
  r.all_gather = (decltype(r.all_gather))dlsym(h, "ncclAllGather");
  r.error_string = (decltype(r.error_string))dlsym(h, "ncclGetErrorString");
  if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_reduce || !r.all_gather || !r.error_string)
    return fail(BSGL_ERR_COMM, "RCCL library lacks a collective entry point%s", "");
  g_rccl = r;
  return BSGL_OK;
}

int rccl_fail(const char* what, ncclResult_t e) { return fail(BSGL_ERR_COMM, "%s: %s", what, g_rccl.error_string(e)); }

bool to_nccl_type(int32_t dtype, ncclDataType_t* t) {
  switch (dtype) {
    case BSGL_DTYPE_BF16: *t = ncclBfloat16; return true;
    case BSGL_DTYPE_F16: *t = ncclFloat16; return true;
    case BSGL_DTYPE_F32: *t = ncclFloat32; return true;
    default: return false;
  }
}

}  // namespace

extern "C" {

int bsgl_comm_unique_id(void* id_out) {
  if (id_out == nullptr) return fail(BSGL_ERR_BAD_ARG, "null unique-id buffer%s", "");
  std::lock_guard<std::mutex> lock(g_mu);
  if (int rc = bind_rccl()) return rc;
  ncclUniqueId id;
  const ncclResult_t e = g_rccl.get_unique_id(&id);
  if (e != ncclSuccess) return rccl_fail("ncclGetUniqueId", e);
  memcpy(id_out, &id, sizeof(id));
  return BSGL_OK;
}

int bsgl_comm_init(int32_t rank, int32_t world, const void* unique_id) {
  if (world < 1 || rank < 0 || rank >= world) return fail(BSGL_ERR_BAD_ARG, "rank outside [0, world)%s", "");
  if (unique_id == nullptr) return fail(BSGL_ERR_BAD_ARG, "null unique id%s", "");
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_comm != nullptr) return fail(BSGL_ERR_BAD_ARG, "communicator already initialised (bsgl_comm_destroy first)%s", "");
  if (int rc = bind_rccl()) return rc;
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclComm_t comm = nullptr;
  const ncclResult_t e = g_rccl.comm_init_rank(&comm, world, id, rank);  // on the current HIP device
  if (e != ncclSuccess) return rccl_fail("ncclCommInitRank", e);
  g_comm = comm;
  g_world = world;
  g_rank = rank;
  return BSGL_OK;
}

int bsgl_comm_world(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  return g_comm != nullptr ? g_world : 0;
}

int bsgl_all_reduce_sum(void* buf, int64_t count, int32_t dtype, void* stream) {
  ncclDataType_t t;
  if (!to_nccl_type(dtype, &t)) return fail(BSGL_ERR_UNSUPPORTED, "all-reduce: dtype is not bf16/f16/f32%s", "");
  if (count < 0) return fail(BSGL_ERR_BAD_ARG, "count < 0%s", "");
  if (count == 0) return BSGL_OK;
  if (buf == nullptr) return fail(BSGL_ERR_BAD_ARG, "null device pointer (buf)%s", "");
  ncclComm_t comm;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    comm = g_comm;
  }
  if (comm == nullptr) return fail(BSGL_ERR_BAD_ARG, "no communicator (bsgl_comm_init first)%s", "");
  const ncclResult_t e = g_rccl.all_reduce(buf, buf, (size_t)count, t, ncclSum, comm, (hipStream_t)stream);
  if (e != ncclSuccess) return rccl_fail("ncclAllReduce", e);
  return BSGL_OK;
}

int bsgl_all_gather(void* out, const void* in, int64_t count_per_rank, int32_t dtype, void* stream) {
  ncclDataType_t t;
  if (!to_nccl_type(dtype, &t)) return fail(BSGL_ERR_UNSUPPORTED, "all-gather: dtype is not bf16/f16/f32%s", "");
  if (count_per_rank < 0) return fail(BSGL_ERR_BAD_ARG, "count_per_rank < 0%s", "");
  if (count_per_rank == 0) return BSGL_OK;
  if (out == nullptr || in == nullptr) return fail(BSGL_ERR_BAD_ARG, "null device pointer (out/in)%s", "");
  ncclComm_t comm;
  {
    std::lock_guard<std::mutex> lock(g_mu);
    comm = g_comm;
  }
  if (comm == nullptr) return fail(BSGL_ERR_BAD_ARG, "no communicator (bsgl_comm_init first)%s", "");
  const ncclResult_t e = g_rccl.all_gather(in, out, (size_t)count_per_rank, t, comm, (hipStream_t)stream);
  if (e != ncclSuccess) return rccl_fail("ncclAllGather", e);
  return BSGL_OK;
}

int bsgl_comm_destroy(void) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_comm == nullptr) return BSGL_OK;
  const ncclResult_t e = g_rccl.comm_destroy(g_comm);
  g_comm = nullptr;
  g_world = 0;
  g_rank = -1;
  if (e != ncclSuccess) return rccl_fail("ncclCommDestroy", e);
  return BSGL_OK;
}

}  // extern "C"
