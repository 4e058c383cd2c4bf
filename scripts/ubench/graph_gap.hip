// How much of a chain of small dependent kernels is launch gap, and does a hipGraph shorten it?  N dependent launches of a kernel that
// spins ~W us, (a) stream launches back to back, (b) the same chain captured into a graph and launched once.  Reports the GPU time from
// the first kernel's start to the last one's end (HIP events) and the host time of the enqueue.
// Build: hipcc -O2 --offload-arch=gfx950 scripts/ubench/graph_gap.hip -o /tmp/graph_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void spin_kernel(unsigned long long* p, int ticks) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < static_cast<unsigned long long>(ticks)) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1;
}
int main() {
  unsigned long long* d;
  CK(hipMalloc(&d, 64));
  CK(hipMemset(d, 0, 64));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int us : {0, 5, 10}) {
    for (int blocks : {1, 64}) {
      const int N = 30, ticks = us * 100;  // wall_clock64: 100 MHz
      auto chain = [&]() { for (int i = 0; i < N; i++) hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(256), 0, s, d, ticks); };
      double best_stream = 1e9, best_graph = 1e9, host_stream = 1e9, host_graph = 1e9;
      for (int rep = 0; rep < 20; rep++) {
        CK(hipStreamSynchronize(s));
        auto h0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, s));
        chain();
        CK(hipEventRecord(e1, s));
        auto h1 = std::chrono::steady_clock::now();
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best_stream = std::min(best_stream, 1e3 * ms);
        host_stream = std::min(host_stream, std::chrono::duration<double, std::micro>(h1 - h0).count());
      }
      hipGraph_t g;
      hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      chain();
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      for (int rep = 0; rep < 20; rep++) {
        CK(hipStreamSynchronize(s));
        auto h0 = std::chrono::steady_clock::now();
        CK(hipEventRecord(e0, s));
        CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        auto h1 = std::chrono::steady_clock::now();
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best_graph = std::min(best_graph, 1e3 * ms);
        host_graph = std::min(host_graph, std::chrono::duration<double, std::micro>(h1 - h0).count());
      }
      printf("%2d launches of a %2d us kernel, %2d workgroups: stream %.1f us on the GPU (%.2f per launch beyond the kernel), host %.1f us | graph %.1f us (%.2f), host %.1f us\n", N, us, blocks, best_stream,
             best_stream / N - us, host_stream, best_graph, best_graph / N - us, host_graph);
      CK(hipGraphExecDestroy(ge));
      CK(hipGraphDestroy(g));
    }
  }
  return 0;
}
