// Is a hipMemsetAsync node, captured in front of a kernel node on the same stream, ordered in front of
// it when the graph is replayed -- on the address of a `static __device__` array of the code object
// (what bsgl_paged_decode_static clears) and on hipMalloc memory?
//
// The kernel reads word 0 of the area into out[i], then stores 1 there.  With the memset ordered in
// front of every kernel (and behind the kernel of the replay before), every out[i] is 0; a 1 means the
// kernel ran before its memset, or the memset before the previous kernel's store.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/memset_capture_probe.hip -o tools/probes/memset_capture_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static __device__ __align__(256) unsigned char area[1 << 20];

__global__ void probe(unsigned* words, unsigned* out, int i, int spin) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    out[i] = __hip_atomic_load(words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // (some work between the read and the store, as a decode launch has)
    unsigned acc = 0;
    for (int k = 0; k < spin; ++k) acc += __hip_atomic_load(words + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(words, 1u + (acc & 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// torch_like: capture in global mode, instantiate with hipGraphInstantiateFlagAutoFreeOnLaunch and launch on
// another stream than the one captured (what torch.cuda.graph + CUDAGraph.replay do)
static int run(const char* what, void* ptr, size_t bytes, int steps_per_graph, int replays, bool torch_like) {
  hipStream_t s, launch_stream;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&launch_stream, hipStreamNonBlocking));
  unsigned* out;
  const int n = steps_per_graph * replays;
  CHECK(hipMalloc(&out, n * sizeof(unsigned)));
  CHECK(hipMemset(out, 0xff, n * sizeof(unsigned)));
  CHECK(hipMemset(ptr, 0, bytes));
  CHECK(hipDeviceSynchronize());
  std::vector<hipGraphExec_t> execs;
  for (int r = 0; r < replays; ++r) {
    hipGraph_t g;
    CHECK(hipStreamBeginCapture(s, torch_like ? hipStreamCaptureModeGlobal : hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < steps_per_graph; ++k) {
      CHECK(hipMemsetAsync(ptr, 0, bytes, s));
      hipLaunchKernelGGL(probe, dim3(64), dim3(64), 0, s, (unsigned*)ptr, out, r * steps_per_graph + k, 2000);
    }
    CHECK(hipStreamEndCapture(s, &g));
    hipGraphExec_t ex;
    if (torch_like) CHECK(hipGraphInstantiateWithFlags(&ex, g, hipGraphInstantiateFlagAutoFreeOnLaunch));
    else CHECK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    execs.push_back(ex);
    CHECK(hipGraphDestroy(g));
  }
  for (int r = 0; r < replays; ++r) CHECK(hipGraphLaunch(execs[r], torch_like ? launch_stream : s));
  CHECK(hipStreamSynchronize(torch_like ? launch_stream : s));
  std::vector<unsigned> h(n);
  CHECK(hipMemcpy(h.data(), out, n * sizeof(unsigned), hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < n; ++i) bad += h[i] != 0;
  printf("%-44s %7zu bytes, %d memset+kernel pairs per graph x %d graphs: %d of %d kernels saw a word that was not cleared\n",
         what, bytes, steps_per_graph, replays, bad, n);
  for (auto ex : execs) CHECK(hipGraphExecDestroy(ex));
  CHECK(hipFree(out));
  CHECK(hipStreamDestroy(s));
  CHECK(hipStreamDestroy(launch_stream));
  return bad;
}

int main() {
  void* sym = nullptr;
  CHECK(hipGetSymbolAddress(&sym, HIP_SYMBOL(area)));
  void* heap = nullptr;
  CHECK(hipMalloc(&heap, 1 << 20));
  int bad = 0;
  for (size_t bytes : {256ul, 1024ul, 65536ul}) {
    bad += run("static __device__ array (symbol address)", sym, bytes, 28, 20, false);
    bad += run("hipMalloc memory", heap, bytes, 28, 20, false);
    bad += run("symbol address, captured / launched as torch does", sym, bytes, 1, 40, true);
    bad += run("hipMalloc, captured / launched as torch does", heap, bytes, 1, 40, true);
  }
  printf(bad ? "UNORDERED memset nodes seen\n" : "every memset node was ordered in front of its kernel\n");
  return 0;
}
