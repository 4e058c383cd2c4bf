// Dependent-load latency on MI355X: every lane chases its own random cycle through a table of 16-byte records.
// Usage: chase <table MB> <waves per SIMD (0 = one wave on the whole chip)>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
__global__ void chase(const uint4* __restrict__ tab, unsigned n, int hops, unsigned long long* out, unsigned* sink) {
  unsigned idx = (blockIdx.x * blockDim.x + threadIdx.x) * 977u % n;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int h = 0; h < hops; h++) idx = tab[idx].x;
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (idx == 0xffffffffu) *sink = idx;
}
int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? atoi(argv[1]) : 32;
  const int wps = argc > 2 ? atoi(argv[2]) : 0;
  const unsigned n = mb * (1u << 20) / 16;
  std::vector<unsigned> perm(n);
  std::iota(perm.begin(), perm.end(), 0u);
  std::mt19937 rng(1);
  std::shuffle(perm.begin(), perm.end(), rng);
  std::vector<uint4> h(n);
  for (unsigned i = 0; i < n; i++) h[perm[i]] = make_uint4(perm[(i + 1) % n], 0, 0, 0);
  uint4* d; unsigned long long* out; unsigned* sink;
  hipMalloc(&d, n * sizeof(uint4)); hipMalloc(&out, 65536 * 8); hipMalloc(&sink, 4);
  hipMemcpy(d, h.data(), n * sizeof(uint4), hipMemcpyHostToDevice);
  const int blocks = wps == 0 ? 1 : 256 * 4 * wps;
  const int hops = 200;
  for (int rep = 0; rep < 3; rep++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(chase, dim3(blocks), dim3(64), 0, 0, d, n, hops, out, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> ho(blocks);
    hipMemcpy(ho.data(), out, blocks * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : ho) avg += v; avg /= blocks;
    printf("table %zu MB, %d waves (%d/SIMD): kernel %.1f us, %.0f ticks/hop (s_memtime), %.3f us/hop wall\n", mb, blocks, wps, ms * 1e3, avg / hops, ms * 1e3 / hops);
  }
  return 0;
}
