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

// nn / nn2 / rex of a query whose block has been scanned.  Returns false if the block does not settle it.  by_face (optional): the
// exclusion radius ends at the block's FACE, not at a third target point — a radius a tree walk could widen.
__device__ __forceinline__ bool grid_settle(const GridTop3& t, float rho2, float bound2, int& nn, int& nn2, float& rex, bool* by_face = nullptr) {
  const float d1 = grid_key_dist(t.k1);
  const bool hit = d1 < bound2;
  const bool settled = d1 < rho2 || rho2 >= bound2;  // the nearest point lies inside the certified ball, or the ball covers the whole reach
  nn = hit ? grid_key_pos(t.k1) : -1;
  nn2 = (hit && t.k2 != kGridNoKey) ? grid_key_pos(t.k2) : -1;
  rex = grid_rex_from_r2(fminf(hit ? t.d3 : d1, rho2));
  if (by_face != nullptr) *by_face = rho2 < (hit ? t.d3 : d1);
  return settled;
}

// Ring 1 for the query of one lane (callable by any subset of a wave's lanes): 9 independent header loads, then the candidates of the
// 9 runs, four at a time, the first four of the NEXT run already in flight while a run's candidates are compared.  Returns whether the
// 27 cells settle the query (then nn / nn2 / rex are its exact neighbour, runner-up and exclusion radius); otherwise `seen` = the distance
// of the nearest point the ring saw (+inf: none).  A chain of two dependent loads where the seeded kd walk has a dozen: this is what the
// walkers of warm passes use (linearize.hip: certify_linearize_kernel).
__device__ __forceinline__ bool grid_ring1_lane(const GridView& g, float qx, float qy, float qz, float bound2, int& nn, int& nn2, float& rex, float& seen, bool* by_face = nullptr) {
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
  return grid_settle(t, grid_rho2(g, qx, qy, qz, cx, cy, cz, 1), bound2, nn, nn2, rex, by_face);
}

__device__ __forceinline__ unsigned long long grid_wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) {
    const uint32_t lo = __shfl_xor(static_cast<uint32_t>(v), sft, 64), hi = __shfl_xor(static_cast<uint32_t>(v >> 32), sft, 64);
    const unsigned long long o = (static_cast<unsigned long long>(hi) << 32) | lo;
    v = o < v ? o : v;
  }
  return v;
}
__device__ __forceinline__ float grid_wave_min_f32(float v) {
#pragma unroll
  for (int sft = 32; sft >= 1; sft >>= 1) v = fminf(v, __shfl_xor(v, sft, 64));
  return v;
}

// ring whose block certifies a ball of radius `need` around the query: (r + the distance to the nearest face of its own cell) * h >= need
__device__ __forceinline__ int grid_ring_for(const GridView& g, float need, float qx, float qy, float qz, int cx, int cy, int cz) {
  const float fx = qx - fmaf(static_cast<float>(cx), g.h, g.ox), fy = qy - fmaf(static_cast<float>(cy), g.h, g.oy), fz = qz - fmaf(static_cast<float>(cz), g.h, g.oz);
  const float inside = fminf(fminf(fminf(fx, g.h - fx), fminf(fy, g.h - fy)), fminf(fz, g.h - fz));  // negative for a query outside the grid
  const float rr = ceilf((need - inside + 2.f * g.eps) * g.inv_h * 1.0001f);
  return static_cast<int>(fminf(fmaxf(rr, 2.f), 1.0e6f));
}

// One query per WAVE (all 64 lanes pass the same query): the ring that settles it for certain — as far as the nearest point seen so far
// (`seen`, +inf: none), or as the whole reach — with its (2r + 1)^2 rows spread over the lanes: every lane looks up and scans its own rows,
// the lanes' three nearest are merged with wave-wide minima.  Hundreds of mostly empty rows cost a handful of independent loads per lane
// instead of a serial sweep by one lane.  A block that does not settle the query (a face closer than the arithmetic slack allows) grows and
// is scanned again; the block that covers the grid settles everything.  Returns the sum of the rings scanned (lane 0; statistics).
__device__ __forceinline__ unsigned grid_settle_wave(const GridView& g, int lane, float ux, float uy, float uz, float useen, float bound2, int& nn, int& nn2, float& rex) {
  const float reach = sqrtf(bound2) * 1.00001f;
  const int rcap = max(max(g.nx, g.ny), g.nz);  // a block never needs to reach beyond the grid
  const int ucx = grid_cell(ux, g.ox, g.inv_h, g.nx), ucy = grid_cell(uy, g.oy, g.inv_h, g.ny), ucz = grid_cell(uz, g.oz, g.inv_h, g.nz);
  int ur = min(grid_ring_for(g, fminf(useen * 1.00001f, reach), ux, uy, uz, ucx, ucy, ucz), rcap);
  unsigned ring_sum = 0;
  for (;;) {
    ring_sum += lane == 0 ? static_cast<unsigned>(ur) : 0u;
    const int W = 2 * ur + 1, rows = W * W;
    const float inv_w = 1.0f / static_cast<float>(W);
    const int xlo = max(ucx - ur, 0), xhi = min(ucx + ur, g.nx - 1);
    GridTop3 t = grid_top3();
    for (int base = 0; base < rows; base += 64) {
      const int k = base + lane;
      int a = static_cast<int>((static_cast<float>(k) + 0.5f) * inv_w);  // k / W for k < 2^20
      a -= a * W > k ? 1 : 0;
      a += (a + 1) * W <= k ? 1 : 0;
      const int zz = ucz + a - ur, yy = ucy + (k - a * W) - ur;
      uint32_t s = 0u, e = 0u;
      if (k < rows && zz >= 0 && zz < g.nz && yy >= 0 && yy < g.ny) {
        const int row = (zz * g.ny + yy) * g.nx;
        s = g.start[row + xlo];
        e = g.start[row + xhi + 1];
      }
      grid_scan_run(g, s, e, ux, uy, uz, t);
    }
    // the three nearest over the lanes' rows (every point lies in exactly one row: no key occurs twice)
    const unsigned long long g1 = grid_wave_min_u64(t.k1);
    const unsigned long long c2 = t.k1 == g1 ? t.k2 : t.k1;
    const unsigned long long g2 = grid_wave_min_u64(c2);
    const float third = t.k1 == g1 ? (t.k2 == g2 ? t.d3 : grid_key_dist(t.k2)) : (t.k1 == g2 ? grid_key_dist(t.k2) : grid_key_dist(t.k1));
    GridTop3 m;
    m.k1 = g1;
    m.k2 = g2;
    m.d3 = grid_wave_min_f32(third);
    if (grid_settle(m, grid_rho2(g, ux, uy, uz, ucx, ucy, ucz, ur), bound2, nn, nn2, rex)) return ring_sum;  // wave-uniform
    const float d1 = sqrtf(grid_key_dist(g1));
    ur = min(max(ur + 1, d1 < 3.0e38f ? grid_ring_for(g, fminf(d1 * 1.00001f, reach), ux, uy, uz, ucx, ucy, ucz) : ur + 1), rcap + 1);
  }
}

// Ring 1 for ONE query by a GROUP of G lanes (G a power of two, 2 .. 64; the groups of a wave work on different queries, every lane of a
// group passes the same query).  The candidates of the 9 runs are numbered through (prefix sums of the run lengths) and dealt to the
// lanes round-robin, four per lane and trip in flight: 300 candidates of a dense wall cost a group of 64 two trips where one lane needs 75,
// and the latency of a walker — all that matters when a pass has a handful of them — drops from tens of dependent loads to about four.
// The lanes' three nearest are merged with minima over the group (shuffles on the 64-bit keys; every point lies in exactly one run, so no
// key occurs twice).  All lanes of the wave must call this together (`has` = this lane's group has a query); the result is uniform
// within a group.
__device__ __forceinline__ bool grid_ring1_group(const GridView& g, int G, int gl, bool has, float qx, float qy, float qz, float bound2, int& nn, int& nn2, float& rex, float& seen, bool* by_face = nullptr) {
  const int cx = grid_cell(qx, g.ox, g.inv_h, g.nx), cy = grid_cell(qy, g.oy, g.inv_h, g.ny), cz = grid_cell(qz, g.oz, g.inv_h, g.nz);
  uint32_t off[9], pre[10];  // off[r] = start of run r - candidates before it; pre[r] = candidates before run r
  pre[0] = 0u;
#pragma unroll
  for (int r = 0; r < 9; r++) {
    const int row = ((cz + r / 3 - 1) * g.ny + (cy + r % 3 - 1)) * g.nx + cx;
    const uint32_t s = has ? g.start[row - 1] : 0u, e = has ? g.start[row + 2] : 0u;  // one address per group: a broadcast load
    off[r] = s - pre[r];
    pre[r + 1] = pre[r] + (e - s);
  }
  const uint32_t total = pre[9];
  GridTop3 t = grid_top3();
  for (uint32_t c0 = static_cast<uint32_t>(gl); __ballot(c0 < total) != 0ull; c0 += 4u * static_cast<uint32_t>(G)) {
    float4 cand[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t c = c0 + static_cast<uint32_t>(u * G);
      ok[u] = c < total;
      uint32_t o = off[0];
#pragma unroll
      for (int r = 1; r < 9; r++) o = c >= pre[r] ? off[r] : o;
      cand[u] = g.pts[ok[u] ? c + o : 0u];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) grid_offer(t, ok[u] ? kd_dist2(cand[u].x, cand[u].y, cand[u].z, qx, qy, qz) : INFINITY, ok[u] ? __float_as_uint(cand[u].w) : 0x7fffffffu);
  }
  // the three nearest over the group's lanes
  auto min_u64 = [G](unsigned long long v) {
    for (int sft = G >> 1; sft >= 1; sft >>= 1) {
      const uint32_t lo = __shfl_xor(static_cast<uint32_t>(v), sft, 64), hi = __shfl_xor(static_cast<uint32_t>(v >> 32), sft, 64);
      const unsigned long long o = (static_cast<unsigned long long>(hi) << 32) | lo;
      v = o < v ? o : v;
    }
    return v;
  };
  const unsigned long long g1 = min_u64(t.k1);
  const unsigned long long c2 = t.k1 == g1 ? t.k2 : t.k1;  // (kGridNoKey == kGridNoKey for lanes that saw nothing while the group saw nothing: then g2 = no key as well)
  const unsigned long long g2 = min_u64(c2);
  float third = t.k1 == g1 ? (t.k2 == g2 ? t.d3 : grid_key_dist(t.k2)) : (t.k1 == g2 ? grid_key_dist(t.k2) : grid_key_dist(t.k1));
  for (int sft = G >> 1; sft >= 1; sft >>= 1) third = fminf(third, __shfl_xor(third, sft, 64));
  GridTop3 m;
  m.k1 = g1, m.k2 = g2, m.d3 = third;
  seen = g1 != kGridNoKey ? sqrtf(grid_key_dist(g1)) : INFINITY;
  return grid_settle(m, grid_rho2(g, qx, qy, qz, cx, cy, cz, 1), bound2, nn, nn2, rex, by_face);
}

}  // namespace sga
