// Exact nearest neighbour over a uniform CELL GRID of the target (the second search structure of a kd-tree index; gfx950 device code).
//
// Why a second structure.  The kd walk (kd_search.hpp) is a chain of ~65 dependent loop trips per wave executed at the pace of its slowest
// lane.  A flat grid has a chain of depth TWO — cell headers, then candidate points, all addresses computable from
// the query alone — so the loads of a query are independent and a wave's lanes differ only in how many candidates their cells hold.
// What it cannot do is adapt to the density (a kd leaf always holds 8 points, a cell of a wall holds six times what a cell of the ground
// holds) or to queries far from any surface.  It is therefore used where it wins: cold passes whose queries lie NEAR the target (every
// cold pass but the first of a registration), with the kd walk kept for the first pass, for the walkers of warm passes and for k-NN.
//
// Layout.  Cell (cx, cy, cz) of edge h, x fastest: c = (cz * ny + cy) * nx + cx, cx = floor((x - ox) / h) clamped to [1, nx - 2]
// (cells 0 and n - 1 of every axis are empty padding, so a block of cells around any clamped cell stays inside the arrays).
// `start[c]` = first position of cell c in `pts`, the target points sorted by cell (stable: kd order inside a cell), w = kd position.
// The cells (cx - r .. cx + r) of one (cy, cz) row are ONE contiguous run of `pts`: [start[row + cx - r], start[row + cx + r + 1]).
//
// Search = scan the (2r + 1)^3 block of cells around the query's cell; every target point outside the block is farther than
// rho = the distance from the query to the nearest face of the block that has cells beyond it.  If the nearest point found is closer than
// rho it is THE nearest neighbour — the same canonical one the kd walk returns (minimal (kd_dist2, kd position)) — and min(third-nearest
// distance, rho^2) is the exclusion bound of the warm pass's certificate.  Ring 1 (27 cells = 9 runs) settles the queries near a
// surface; the others are collected and finished by a second kernel in full waves with the ring their first result calls for.
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "kd_search.hpp"

namespace sga {

struct GridView {
  const float4* __restrict__ pts;      // cell order, w = kd position bits
  const uint32_t* __restrict__ start;  // nx * ny * nz + 1 entries
  float ox, oy, oz;                    // low corner of cell (0, 0, 0)
  float h, inv_h;
  float eps;                           // absolute slack of the geometric bounds (rounding of the cell arithmetic)
  int nx, ny, nz;
};

inline GridView make_grid_view(const sga_index* idx) {
  GridView g;
  g.pts = idx->grid_pts.p;
  g.start = idx->grid_start.p;
  g.ox = idx->grid_org[0], g.oy = idx->grid_org[1], g.oz = idx->grid_org[2];
  g.h = idx->grid_h;
  g.inv_h = 1.0f / idx->grid_h;
  g.eps = idx->grid_eps;
  g.nx = idx->grid_dim[0], g.ny = idx->grid_dim[1], g.nz = idx->grid_dim[2];
  return g;
}

// THE cell of a coordinate: the build and the search evaluate exactly this expression (fp32), which is monotone in x.
__device__ __forceinline__ int grid_cell(float x, float o, float inv_h, int n) {
  const int c = static_cast<int>(floorf((x - o) * inv_h));  // the conversion saturates; NaN -> 0
  return min(max(c, 1), n - 2);
}

// The nearest two candidates seen as 64-bit keys (distance bits << 32 | kd position: distances are >= 0, so the order of the keys is the
// lexicographic order of (distance, position) — exact, nothing to resolve afterwards) and the distance of the third.
struct GridTop3 {
  unsigned long long k1, k2;
  float d3;
};
constexpr unsigned long long kGridNoKey = (0x7f800000ull << 32) | 0x7fffffffull;  // (+inf, no position)
__device__ __forceinline__ GridTop3 grid_top3() { return {kGridNoKey, kGridNoKey, INFINITY}; }
__device__ __forceinline__ float grid_key_dist(unsigned long long k) { return __uint_as_float(static_cast<uint32_t>(k >> 32)); }
__device__ __forceinline__ int grid_key_pos(unsigned long long k) { return static_cast<int>(static_cast<uint32_t>(k)); }
__device__ __forceinline__ void grid_offer(GridTop3& s, float d, uint32_t pos) {
  const unsigned long long key = (static_cast<unsigned long long>(__float_as_uint(d)) << 32) | pos;
  const bool lt1 = key < s.k1, lt2 = key < s.k2;
  s.d3 = __builtin_amdgcn_fmed3f(grid_key_dist(s.k2), s.d3, d);  // third smallest of {d1 <= d2 <= d3, d}
  s.k2 = lt1 ? s.k1 : (lt2 ? key : s.k2);
  s.k1 = lt1 ? key : s.k1;
}

// Four candidates of a run, loaded together (four independent 16-byte gathers in flight per lane); slots at or beyond `e` re-read
// position 0 (always there, one line for all such lanes) and are offered as (+inf, no position), which changes nothing: no branches.
struct GridBatch {
  float4 c[4];
};
__device__ __forceinline__ GridBatch grid_load4(const GridView& g, uint32_t j, uint32_t e) {
  GridBatch b;
#pragma unroll
  for (int u = 0; u < 4; u++) b.c[u] = g.pts[j + u < e ? j + u : 0u];
  return b;
}
__device__ __forceinline__ void grid_offer4(const GridBatch& b, uint32_t j, uint32_t e, float qx, float qy, float qz, GridTop3& t) {
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const bool v = j + u < e;
    grid_offer(t, v ? kd_dist2(b.c[u].x, b.c[u].y, b.c[u].z, qx, qy, qz) : INFINITY, v ? __float_as_uint(b.c[u].w) : 0x7fffffffu);
  }
}
// One run of candidates [s, e) per lane; the trip count is the wave's longest run.
__device__ __forceinline__ void grid_scan_run(const GridView& g, uint32_t s, uint32_t e, float qx, float qy, float qz, GridTop3& t) {
  uint32_t j = s;
  while (__ballot(j < e) != 0ull) {
    const GridBatch b = grid_load4(g, j, e);
    grid_offer4(b, j, e, qx, qy, qz, t);
    j += 4u;
  }
}

// Squared distance beyond which the block [c - r, c + r]^3 (clamped to the grid) says nothing: the distance from the query to the nearest
// face that still has non-padding cells behind it, minus the rounding slack.  INFINITY when the block covers the whole grid.
__device__ __forceinline__ float grid_axis_rho(float q, float o, float h, int c, int r, int n) {
  float rho = INFINITY;
  const int a = c - r, b = c + r;
  if (a > 1) rho = fminf(rho, q - fmaf(static_cast<float>(a), h, o));          // cells below a exist: low face at o + a h
  if (b < n - 2) rho = fminf(rho, fmaf(static_cast<float>(b + 1), h, o) - q);  // cells above b exist: high face at o + (b + 1) h
  return rho;
}
__device__ __forceinline__ float grid_rho2(const GridView& g, float qx, float qy, float qz, int cx, int cy, int cz, int r) {
  const float rho = fminf(fminf(grid_axis_rho(qx, g.ox, g.h, cx, r, g.nx), grid_axis_rho(qy, g.oy, g.h, cy, r, g.ny)), grid_axis_rho(qz, g.oz, g.h, cz, r, g.nz));
  if (!(rho < 3.0e38f)) return INFINITY;
  const float safe = (rho - g.eps) * 0.99999f;
  return safe > 0.f ? safe * safe : 0.f;
}

__device__ __forceinline__ float grid_rex_from_r2(float r2) { return sqrtf(r2) * 0.9999995f; }  // as linearize.hip: sqrt of the bound, rounded down

// nn / nn2 / rex of a query whose block has been scanned.  Returns false if the block does not settle it.
__device__ __forceinline__ bool grid_settle(const GridTop3& t, float rho2, float bound2, int& nn, int& nn2, float& rex) {
  const float d1 = grid_key_dist(t.k1);
  const bool hit = d1 < bound2;
  const bool settled = d1 < rho2 || rho2 >= bound2;  // the nearest point lies inside the certified ball, or the ball covers the whole reach
  nn = hit ? grid_key_pos(t.k1) : -1;
  nn2 = (hit && t.k2 != kGridNoKey) ? grid_key_pos(t.k2) : -1;
  rex = grid_rex_from_r2(fminf(hit ? t.d3 : d1, rho2));
  return settled;
}

// Ring 1 for the query of one lane (callable by any subset of a wave's lanes): 9 independent header loads, then the candidates of the
// 9 runs, four at a time, the first four of the NEXT run already in flight while a run's candidates are compared.  Returns whether the
// 27 cells settle the query (then nn / nn2 / rex are its exact neighbour, runner-up and exclusion radius); otherwise `seen` = the distance
// of the nearest point the ring saw (+inf: none).  A chain of two dependent loads where the seeded kd walk has a dozen: this is what the
// walkers of warm passes use (linearize.hip: certify_linearize_kernel).
__device__ __forceinline__ bool grid_ring1_lane(const GridView& g, float qx, float qy, float qz, float bound2, int& nn, int& nn2, float& rex, float& seen) {
  const int cx = grid_cell(qx, g.ox, g.inv_h, g.nx), cy = grid_cell(qy, g.oy, g.inv_h, g.ny), cz = grid_cell(qz, g.oz, g.inv_h, g.nz);
  uint32_t s[9], e[9];
#pragma unroll
  for (int r = 0; r < 9; r++) {
    // the query's own row first, then its neighbours in y, then the rows above and below
    constexpr int order[9] = {4, 3, 5, 1, 7, 0, 2, 6, 8};
    const int k = order[r];
    const int row = ((cz + k / 3 - 1) * g.ny + (cy + k % 3 - 1)) * g.nx + cx;
    s[r] = g.start[row - 1];
    e[r] = g.start[row + 2];
  }
  GridTop3 t = grid_top3();
  GridBatch nxt = grid_load4(g, s[0], e[0]);
#pragma unroll
  for (int r = 0; r < 9; r++) {
    const GridBatch cur = nxt;
    if (r + 1 < 9) nxt = grid_load4(g, s[r + 1], e[r + 1]);
    grid_offer4(cur, s[r], e[r], qx, qy, qz, t);
    grid_scan_run(g, s[r] + 4u, e[r], qx, qy, qz, t);  // the rest of a run longer than four (wave-uniform loop; skipped when no lane has one)
  }
  seen = t.k1 != kGridNoKey ? sqrtf(grid_key_dist(t.k1)) : INFINITY;
  return grid_settle(t, grid_rho2(g, qx, qy, qz, cx, cy, cz, 1), bound2, nn, nn2, rex);
}

}  // namespace sga
