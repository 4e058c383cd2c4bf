// How fast can a wave fetch 64 random 128-byte leaves (8 x 16 B each)?  MI355X, one wave per workgroup like nn_search_kernel.
//   A: every lane reads its own leaf with 8 dwordx4 loads (what kd_scan_leaf does)
//   B: octets: in round r the 8 lanes of an octet read the 8 points of the leaf owned by lane r of the octet (coalesced 128 B)
//   C: like B, transposed through LDS so that the owner ends up with its 8 points
//   D: one random 16-byte record per lane (a pair record / a box corner)
// Usage: leafload <table MB> <waves per SIMD> <dependent 0|1>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
template <int MODE, int DEP>
__global__ __launch_bounds__(64) void k(const uint4* __restrict__ tab, unsigned nleaf, int rounds, unsigned* sink) {
  __shared__ uint4 sh[8][64];
  const int lane = threadIdx.x;
  unsigned s = (blockIdx.x * 64 + lane) * 2654435761u + 12345u;
  unsigned acc = 0;
  unsigned leaf = s % nleaf;
  for (int it = 0; it < rounds; it++) {
    uint4 p[8];
    if (MODE == 0) {
#pragma unroll
      for (int j = 0; j < 8; j++) p[j] = tab[leaf * 8u + j];
    } else if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const unsigned ol = __shfl(leaf, (lane & ~7) + r);
        p[r] = tab[ol * 8u + (lane & 7)];
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int r = 0; r < 8; r++) {
        const unsigned ol = __shfl(leaf, (lane & ~7) + r);
        sh[lane & 7][(lane & ~7) + r] = tab[ol * 8u + (lane & 7)];
      }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int j = 0; j < 8; j++) p[j] = sh[j][lane];
      __builtin_amdgcn_wave_barrier();
    } else {
      p[0] = tab[leaf * 8u + (s >> 29)];
#pragma unroll
      for (int j = 1; j < 8; j++) p[j] = p[0];
    }
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) m += p[j].x ^ p[j].w;
    acc += m;
    s = s * 1664525u + 1013904223u;
    leaf = (DEP ? (s ^ (m & 1u)) : s) % nleaf;
  }
  if (acc == 0x12345u) *sink = acc;
}
template <int MODE, int DEP>
static void run(const char* name, const uint4* d, unsigned nleaf, int blocks, unsigned* sink) {
  const int rounds = 64;
  float best = 1e9f;
  for (int rep = 0; rep < 4; rep++) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, DEP>), dim3(blocks), dim3(64), 0, 0, d, nleaf, rounds, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  const double fetches = double(blocks) * 64 * rounds;
  printf("  %-28s dep=%d: %8.1f us  -> %7.1f G leaf-or-record fetches/s\n", name, DEP, best * 1e3, fetches / (best * 1e-3) * 1e-9);
}
int main(int argc, char** argv) {
  const size_t mb = argc > 1 ? atoi(argv[1]) : 16;
  const int wps = argc > 2 ? atoi(argv[2]) : 8;
  const unsigned nleaf = mb * (1u << 20) / 128;
  std::vector<uint4> h(size_t(nleaf) * 8);
  std::mt19937 rng(1);
  for (auto& v : h) v = make_uint4(rng(), rng(), rng(), rng());
  uint4* d; unsigned* sink;
  hipMalloc(&d, h.size() * sizeof(uint4)); hipMalloc(&sink, 4);
  hipMemcpy(d, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice);
  const int blocks = 256 * 4 * wps;
  printf("table %zu MB (%u leaves), %d waves (%d/SIMD)\n", mb, nleaf, blocks, wps);
  run<0, 0>("A own leaf, 8 loads", d, nleaf, blocks, sink);
  run<0, 1>("A own leaf, 8 loads", d, nleaf, blocks, sink);
  run<1, 0>("B octet-cooperative", d, nleaf, blocks, sink);
  run<1, 1>("B octet-cooperative", d, nleaf, blocks, sink);
  run<2, 0>("C octet-coop + LDS transpose", d, nleaf, blocks, sink);
  run<2, 1>("C octet-coop + LDS transpose", d, nleaf, blocks, sink);
  run<3, 0>("D one 16-B record", d, nleaf, blocks, sink);
  run<3, 1>("D one 16-B record", d, nleaf, blocks, sink);
  return 0;
}
