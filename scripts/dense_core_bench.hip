// Micro-benchmark of the CORE of a dense (tile x candidate-block) nearest-neighbour search on gfx950, to put a hardware number
// under scripts/sim_dense.py's candidate counts:  per tile of 64 queries and per block of 32 candidates (one kd group, contiguous)
//   - stage the block: one float4 per candidate, centred on the tile, |c|^2, split into f16 hi / lo (an fp32 coordinate of a few metres
//     needs both halves for 1e-4 relative on d^2),
//   - ONE v_mfma_f32_32x32x16_f16 per half-tile of 32 queries: K slots = {xh*qh, xh*ql, xl*qh, xl*ql} x 3 axes + |c|^2 hi, lo (14 of 16),
//     accumulator preloaded with |q|^2: D[candidate][query] ~ |c - q|^2,
//   - extraction: each lane holds 16 candidates of ONE query: min3 chain, then (best, block of the best, second-best block minimum),
//   - after the last block: nothing (the exact scan of the winning block and the certificate are not part of this loop).
// Candidates are random points around the tile (the memory system sees one 512-byte run per block, as a kd group would be).
//   hipcc -O3 --offload-arch=gfx950 scripts/dense_core_bench.hip -o /tmp/dense_core && /tmp/dense_core [tiles] [blocks per tile]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e = (x);                                                           \
    if (e != hipSuccess) {                                                        \
      std::printf("%s -> %s\n", #x, hipGetErrorString(e));                        \
      std::exit(1);                                                               \
    }                                                                             \
  } while (0)

__device__ __forceinline__ void split(float v, _Float16& hi, _Float16& lo) {
  hi = static_cast<_Float16>(v);
  lo = static_cast<_Float16>(v - static_cast<float>(hi));
}

__global__ __launch_bounds__(64) void dense_core(const float4* __restrict__ queries, const float4* __restrict__ cands, int blocks, float* __restrict__ out_best, int* __restrict__ out_block) {
  const int lane = threadIdx.x, tile = blockIdx.x;
  const int col = lane & 31, hi_half = lane >> 5;
  // tile centre: the mean of the 64 queries (wave reduction) — coordinates relative to it stay within a few metres
  const float4 qm = queries[tile * 64 + lane];
  float cx = qm.x, cy = qm.y, cz = qm.z;
  for (int o = 32; o > 0; o >>= 1) cx += __shfl_xor(cx, o), cy += __shfl_xor(cy, o), cz += __shfl_xor(cz, o);
  cx *= 1.f / 64.f, cy *= 1.f / 64.f, cz *= 1.f / 64.f;
  // B operands of the two half-tiles (query `col` of half h): lanes < 32 hold K slots 0..7, lanes >= 32 slots 8..15
  half8 bq[2];
  float q2[2];
  for (int h = 0; h < 2; h++) {
    const float4 q = queries[tile * 64 + 32 * h + col];
    const float x = q.x - cx, y = q.y - cy, z = q.z - cz;
    q2[h] = x * x + y * y + z * z + 1e-5f;  // bias: the result stays positive under the rounding of the expansion
    _Float16 xh, xl, yh, yl, zh, zl;
    split(-2.f * x, xh, xl), split(-2.f * y, yh, yl), split(-2.f * z, zh, zl);
    const _Float16 one = static_cast<_Float16>(1.f), zero = static_cast<_Float16>(0.f);
    bq[h] = hi_half ? half8{zh, zl, zh, zl, one, one, zero, zero} : half8{xh, xl, xh, xl, yh, yl, yh, yl};
  }
  float best[2] = {INFINITY, INFINITY}, second[2] = {INFINITY, INFINITY};
  int best_block[2] = {-1, -1};
  const float4* cb = cands + static_cast<size_t>(tile) * blocks * 32;
  float4 c = cb[col];
  for (int b = 0; b < blocks; b++) {
    const float4 cn = b + 1 < blocks ? cb[(b + 1) * 32 + col] : c;  // the next block's load in flight behind this block's arithmetic
    const float x = c.x - cx, y = c.y - cy, z = c.z - cz;
    const float n2 = x * x + y * y + z * z;
    _Float16 xh, xl, yh, yl, zh, zl, nh, nl;
    split(x, xh, xl), split(y, yh, yl), split(z, zh, zl), split(n2, nh, nl);
    const _Float16 zero = static_cast<_Float16>(0.f);
    const half8 a = hi_half ? half8{zh, zh, zl, zl, nh, nl, zero, zero} : half8{xh, xh, xl, xl, yh, yh, yl, yl};
#pragma unroll
    for (int h = 0; h < 2; h++) {
      float16v acc;
#pragma unroll
      for (int v = 0; v < 16; v++) acc[v] = q2[h];
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bq[h], acc, 0, 0, 0);
      // 16 candidates of query `col` (rows (v & 3) + 8 (v >> 2) + 4 hi_half): their minimum, three at a time
      float m = __builtin_fminf(__builtin_fminf(acc[0], acc[1]), acc[2]);
#pragma unroll
      for (int v = 3; v + 1 < 16; v += 2) m = __builtin_fminf(__builtin_fminf(m, acc[v]), acc[v + 1]);
      m = __builtin_fminf(m, acc[15]);
      m = __builtin_fminf(m, __shfl_xor(m, 32));  // the other half-wave holds the other 16 candidates of the same query
      const bool better = m < best[h];
      second[h] = better ? best[h] : __builtin_fminf(second[h], m);
      best_block[h] = better ? b : best_block[h];
      best[h] = better ? m : best[h];
    }
    c = cn;
  }
  if (hi_half == 0) {
    for (int h = 0; h < 2; h++) {
      out_best[tile * 64 + 32 * h + col] = best[h] + 0.f * second[h];
      out_block[tile * 64 + 32 * h + col] = best_block[h];
    }
  }
}

int main(int argc, char** argv) {
  const int tiles = argc > 1 ? std::atoi(argv[1]) : 15625, blocks = argc > 2 ? std::atoi(argv[2]) : 45;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> u(-1.f, 1.f), big(-50.f, 50.f);
  std::vector<float4> q(static_cast<size_t>(tiles) * 64), c(static_cast<size_t>(tiles) * blocks * 32);
  for (int t = 0; t < tiles; t++) {
    const float ox = big(rng), oy = big(rng), oz = 0.1f * big(rng);
    for (int i = 0; i < 64; i++) q[static_cast<size_t>(t) * 64 + i] = make_float4(ox + 0.7f * u(rng), oy + 0.7f * u(rng), oz + 0.1f * u(rng), 0.f);
    for (int i = 0; i < blocks * 32; i++) c[(static_cast<size_t>(t) * blocks) * 32 + i] = make_float4(ox + 2.f * u(rng), oy + 2.f * u(rng), oz + 0.3f * u(rng), 0.f);
  }
  float4 *dq, *dc;
  float* db;
  int* dk;
  CHECK(hipMalloc(&dq, q.size() * sizeof(float4)));
  CHECK(hipMalloc(&dc, c.size() * sizeof(float4)));
  CHECK(hipMalloc(&db, q.size() * sizeof(float)));
  CHECK(hipMalloc(&dk, q.size() * sizeof(int)));
  CHECK(hipMemcpy(dq, q.data(), q.size() * sizeof(float4), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dc, c.data(), c.size() * sizeof(float4), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 3; w++) hipLaunchKernelGGL(dense_core, dim3(tiles), dim3(64), 0, 0, dq, dc, blocks, db, dk);
  CHECK(hipEventRecord(e0));
  const int reps = 20;
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL(dense_core, dim3(tiles), dim3(64), 0, 0, dq, dc, blocks, db, dk);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  // exactness of the approximation on a sample: the block of the minimum and its value against brute force in double
  std::vector<float> hb(q.size());
  std::vector<int> hk(q.size());
  CHECK(hipMemcpy(hb.data(), db, hb.size() * sizeof(float), hipMemcpyDeviceToHost));
  CHECK(hipMemcpy(hk.data(), dk, hk.size() * sizeof(int), hipMemcpyDeviceToHost));
  int wrong_block = 0, checked = 0;
  double worst_rel = 0.0;
  for (int t = 0; t < tiles; t += std::max(1, tiles / 64)) {
    for (int i = 0; i < 64; i++) {
      const float4 p = q[static_cast<size_t>(t) * 64 + i];
      double best = 1e300;
      int bb = -1;
      for (int k = 0; k < blocks * 32; k++) {
        const float4 m = c[(static_cast<size_t>(t) * blocks) * 32 + k];
        const double d = (double(m.x) - p.x) * (double(m.x) - p.x) + (double(m.y) - p.y) * (double(m.y) - p.y) + (double(m.z) - p.z) * (double(m.z) - p.z);
        if (d < best) best = d, bb = k / 32;
      }
      checked++;
      wrong_block += bb != hk[static_cast<size_t>(t) * 64 + i];
      worst_rel = std::max(worst_rel, std::abs(double(hb[static_cast<size_t>(t) * 64 + i]) - 1e-5 - best) / best);
    }
  }
  const double us = 1e3 * ms / reps;
  std::printf("{\"tiles\": %d, \"blocks_per_tile\": %d, \"candidates_per_tile\": %d, \"kernel_us\": %.1f, \"ns_per_block_of_32_candidates_x_64_queries\": %.2f, \"pair_tests_per_s\": %.3e, \"checked\": %d, \"minimum_in_another_block\": %d, \"worst_rel_err_of_d2\": %.2e}\n",
              tiles, blocks, blocks * 32, us, 1e3 * us / (double(tiles) * blocks), double(tiles) * blocks * 32 * 64 / (us * 1e-6), checked, wrong_block, worst_rel);
  return 0;
}
