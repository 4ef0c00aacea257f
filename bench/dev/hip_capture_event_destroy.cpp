// Stand-alone experiment: does destroying an event that a capturing stream recorded, BEFORE the capture ends, break
// hipStreamEndCapture?  (CUDA allows cudaEventDestroy at any time; torch.cuda.Event objects created inside a captured
// region and dropped before the region ends do exactly this.)
//
//   hipcc --offload-arch=gfx950 -O2 bench/dev/hip_capture_event_destroy.cpp -o bench/micro/bin/hip_capture_event_destroy
//   ./hip_capture_event_destroy keep      # control: events destroyed after hipStreamEndCapture
//   ./hip_capture_event_destroy destroy   # events destroyed while the capture is still open
//
// The captured graph is a fork / join of two streams around two empty kernels; nothing runs on the GPU during the capture
// and the instantiated graph is launched once.  Prints one line per stage, so the last line before a crash names the call.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#define CHECK(expr)                                                                    \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      std::printf("%s -> %s\n", #expr, hipGetErrorString(e_));                         \
      return 1;                                                                        \
    }                                                                                  \
  } while (0)

__global__ void nothing(int* p) {
  if (p != nullptr && threadIdx.x == 0 && blockIdx.x == 0) *p += 1;
}

int main(int argc, char** argv) {
  const bool early = argc > 1 && std::strcmp(argv[1], "destroy") == 0;
  int* word = nullptr;
  CHECK(hipMalloc(&word, sizeof(int)));
  CHECK(hipMemset(word, 0, sizeof(int)));
  hipStream_t s0, s1;
  CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t fork, join;
  CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  CHECK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  CHECK(hipStreamBeginCapture(s0, hipStreamCaptureModeGlobal));
  nothing<<<1, 64, 0, s0>>>(word);
  CHECK(hipEventRecord(fork, s0));
  CHECK(hipStreamWaitEvent(s1, fork, 0));
  nothing<<<1, 64, 0, s1>>>(nullptr);
  CHECK(hipEventRecord(join, s1));
  CHECK(hipStreamWaitEvent(s0, join, 0));
  nothing<<<1, 64, 0, s0>>>(word);
  std::printf("captured fork/join; events destroyed %s the capture ends\n", early ? "BEFORE" : "after");
  std::fflush(stdout);
  if (early) {
    CHECK(hipEventDestroy(fork));
    CHECK(hipEventDestroy(join));
  }
  hipGraph_t graph = nullptr;
  CHECK(hipStreamEndCapture(s0, &graph));
  std::printf("hipStreamEndCapture returned\n");
  std::fflush(stdout);
  if (!early) {
    CHECK(hipEventDestroy(fork));
    CHECK(hipEventDestroy(join));
  }
  hipGraphExec_t exec = nullptr;
  CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  std::printf("hipGraphInstantiate returned\n");
  std::fflush(stdout);
  CHECK(hipGraphLaunch(exec, s0));
  CHECK(hipStreamSynchronize(s0));
  int host = -1;
  CHECK(hipMemcpy(&host, word, sizeof(int), hipMemcpyDeviceToHost));
  std::printf("graph launched: counter %d (expected 2)\n", host);
  CHECK(hipGraphExecDestroy(exec));
  CHECK(hipGraphDestroy(graph));
  return host == 2 ? 0 : 1;
}
