// Exact nearest-neighbour search over the cell-sorted uniform grid (replaces the kd-tree descent of
// ann/kdtree.hpp:193-233, reference tree /root/reference) and the GaussianVoxelMap probe
// (ann/incremental_voxelmap.hpp:99-119).  Device code, gfx950.
//
// Grid layout: linear cell id = (z*ny + y)*nx + x, target points sorted by cell id, cell_start[ncells+1].
// The cells of one (y,z) row are contiguous, so a query's 3x3x3 neighbourhood is 9 contiguous runs of points.
//
// Exactness: after scanning the cube of cells [c-r, c+r]^3 around the query's cell c, every target point closer than
// r*h has been seen.  The search stops when best <= (r*h)^2 (exact hit), when (r*h)^2 >= max_sq (anything farther is
// rejected by DistanceRejector anyway, rejector.hpp:24-26) or when the cube covers the whole grid.
// Ties: strict '<' in scan order (the reference's tie rule is traversal-order dependent, knn_result.hpp:81-83).
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace sga {

struct GridView {
  const float4* __restrict__ pts;
  const uint32_t* __restrict__ cell_start;
  float ox, oy, oz, inv_cell, cell;
  int nx, ny, nz;
  int debug;  // SGA_DEBUG_MODE bit mask, timing experiments only (results are wrong when set)
  unsigned long long* stats;  // optional device counters (SGA_DEBUG_STATS=1), null otherwise
};

inline GridView make_grid_view(const sga_index* idx) {
  GridView g;
  g.pts = idx->pts.p;
  g.cell_start = idx->cell_start.p;
  g.ox = idx->grid.origin[0];
  g.oy = idx->grid.origin[1];
  g.oz = idx->grid.origin[2];
  g.inv_cell = idx->grid.inv_cell;
  g.cell = idx->grid.cell;
  g.nx = idx->grid.dims[0];
  g.ny = idx->grid.dims[1];
  g.nz = idx->grid.dims[2];
  const char* dbg = getenv("SGA_DEBUG_MODE");
  g.debug = dbg ? atoi(dbg) : 0;
  g.stats = nullptr;
  if (getenv("SGA_DEBUG_STATS")) {
    static unsigned long long* d_stats = nullptr;
    if (!d_stats) {
      (void)hipMalloc(reinterpret_cast<void**>(&d_stats), 16 * sizeof(unsigned long long));
      (void)hipMemset(d_stats, 0, 16 * sizeof(unsigned long long));
    }
    g.stats = d_stats;
  }
  return g;
}

__device__ __forceinline__ int cell_coord(float q, float o, float inv) { return static_cast<int>(floorf((q - o) * inv)); }

struct NNBest {
  float d2;
  int idx;  // position in the cell-sorted target, -1 = none
  float x, y, z;
};

// Scan the run [s, e) of the cell-sorted target.  The loads of one batch are independent of the running minimum, so U of them
// are issued back to back and waited for once (memory-level parallelism: the walk is latency-, not bandwidth-bound).  The
// tail batch re-reads the last point; a duplicate can never beat itself under the strict '<'.
template <int U = 8>
__device__ __forceinline__ void scan_run(const float4* __restrict__ pts, uint32_t s, uint32_t e, float qx, float qy, float qz, NNBest& best) {
  for (uint32_t j = s; j < e; j += U) {
    float4 p[U];
#pragma unroll
    for (int k = 0; k < U; k++) p[k] = pts[min(j + k, e - 1)];
#pragma unroll
    for (int k = 0; k < U; k++) {
      const float dx = p[k].x - qx, dy = p[k].y - qy, dz = p[k].z - qz;
      const float d2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
      if (d2 < best.d2) {
        best.d2 = d2;
        best.idx = static_cast<int>(min(j + k, e - 1));
        best.x = p[k].x;
        best.y = p[k].y;
        best.z = p[k].z;
      }
    }
  }
}

// max_sq: squared search radius (INFINITY = unbounded).  Returns the exact nearest neighbour among points with d2 <= max_sq
// (and possibly a farther one, which the caller rejects).
__device__ __forceinline__ NNBest grid_nearest(const GridView& g, float qx, float qy, float qz, float max_sq) {
  NNBest best;
  best.d2 = INFINITY;
  best.idx = -1;
  best.x = best.y = best.z = 0.f;
  const int cx = cell_coord(qx, g.ox, g.inv_cell), cy = cell_coord(qy, g.oy, g.inv_cell), cz = cell_coord(qz, g.oz, g.inv_cell);
  // rings needed to cover the whole grid from this cell
  int r_all = max(max(max(cx, g.nx - 1 - cx), max(cy, g.ny - 1 - cy)), max(cz, g.nz - 1 - cz));
  r_all = max(r_all, 1);
  const float h = g.cell * 0.9999f;  // guard band for the float cell assignment
  // own row first (x-1 .. x+1): the nearest neighbour is almost always here, which lets the row test below prune most of ring 1
  if (cy >= 0 && cy < g.ny && cz >= 0 && cz < g.nz) {
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
    const uint32_t row = (static_cast<uint32_t>(cz) * g.ny + cy) * g.nx;
    if (x0 <= x1) scan_run(g.pts, g.cell_start[row + x0], g.cell_start[row + x1 + 1], qx, qy, qz, best);
  }
  for (int r = 1;; ++r) {
    const int zlo = max(cz - r, 0), zhi = min(cz + r, g.nz - 1);
    const int ylo = max(cy - r, 0), yhi = min(cy + r, g.ny - 1);
    for (int z = zlo; z <= zhi; ++z) {
      const int adz = abs(z - cz);
      // distance from the query to the slab of cells at this z (0 inside the own slab)
      const float ddz = (z == cz) ? 0.f : (z > cz ? (g.oz + z * g.cell) - qz : qz - (g.oz + (z + 1) * g.cell));
      for (int y = ylo; y <= yhi; ++y) {
        const int ady = abs(y - cy);
        const float ddy = (y == cy) ? 0.f : (y > cy ? (g.oy + y * g.cell) - qy : qy - (g.oy + (y + 1) * g.cell));
        const float row_d2 = fmaxf(ddy, 0.f) * fmaxf(ddy, 0.f) + fmaxf(ddz, 0.f) * fmaxf(ddz, 0.f);
        if (row_d2 * 0.9999f >= best.d2 || row_d2 * 0.9999f > max_sq) continue;
        const uint32_t row = (static_cast<uint32_t>(z) * g.ny + y) * g.nx;
        if (r == 1 && ady == 0 && adz == 0) continue;  // own row: done above
        if (max(ady, adz) == r) {
          const int x0 = max(cx - r, 0), x1 = min(cx + r, g.nx - 1);
          if (x0 <= x1) scan_run(g.pts, g.cell_start[row + x0], g.cell_start[row + x1 + 1], qx, qy, qz, best);
        } else {
          // interior row of the shell: only the two end cells are new
          const int xa = cx - r, xb = cx + r;
          if (xa >= 0 && xa < g.nx) scan_run(g.pts, g.cell_start[row + xa], g.cell_start[row + xa + 1], qx, qy, qz, best);
          if (xb >= 0 && xb < g.nx) scan_run(g.pts, g.cell_start[row + xb], g.cell_start[row + xb + 1], qx, qy, qz, best);
        }
      }
    }
    const float reach = r * h;
    const float reach2 = reach * reach;
    if (best.d2 <= reach2 || reach2 >= max_sq || r >= r_all) break;
  }
  return best;
}

// ---- wave-cooperative exact nearest neighbour ------------------------------------------------------------------------------------
// Phase A: every lane scans the 3x3x3 cells around its own query (own row first, rows pruned by distance).  That settles
// every query whose neighbour is closer than one cell — the bulk of a registration workload.
// Phase B: queries that are still open (sparse regions, outliers that will end up rejected) are finished ONE AT A TIME BY THE
// WHOLE WAVE: the leader's query is broadcast, the rows of the cube of radius R cells that have not been scanned yet are dealt
// round-robin to the 64 lanes, and a wave64 shuffle arg-min merges the lanes' candidates.  R comes straight from the best
// distance known so far (or doubles while nothing has been found), so one or two passes finish a query instead of the lanes
// of a wave idling behind one lane that walks up to (2r+1)^2 rows on its own.
// Must be called by all 64 lanes of a wave (lanes without a query pass active = false).
__device__ __forceinline__ float wave_min_f32(float v) {
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
  for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off));
  return v;
}

// Phase B (see above): finish every query of the wave that ring 1 left open.  best / cx,cy,cz / r_all / done describe the
// calling lane's query after ring 1.  Must be called by all 64 lanes.
__device__ __forceinline__ NNBest grid_finish_wave(const GridView& g, float qx, float qy, float qz, float max_sq, NNBest best, int cx, int cy, int cz, int r_all, bool done, int lane, int r_scanned = 1) {
  const float h = g.cell * 0.9999f;
  if (g.debug & 1) return best;
  unsigned long long todo = __ballot(!done);
  if (g.stats && lane == 0) atomicAdd(&g.stats[1], static_cast<unsigned long long>(__popcll(todo)));
  while (todo) {
    const int leader = __ffsll(static_cast<long long>(todo)) - 1;
    todo &= todo - 1;
    const float lqx = __shfl(qx, leader), lqy = __shfl(qy, leader), lqz = __shfl(qz, leader);
    const int lcx = __shfl(cx, leader), lcy = __shfl(cy, leader), lcz = __shfl(cz, leader), l_all = __shfl(r_all, leader);
    float cur = __shfl(best.d2, leader);  // best squared distance known for the leader's query (wave-uniform)
    NNBest loc;
    if (lane == leader) {
      loc = best;
    } else {
      loc.d2 = cur;
      loc.idx = -1;
      loc.x = loc.y = loc.z = 0.f;
    }
    int R_prev = __shfl(r_scanned, leader);  // rings the leader has already covered
    for (;;) {
      const float lim2 = fminf(cur, max_sq);
      int R = (lim2 < 3.0e38f) ? static_cast<int>(ceilf(sqrtf(lim2) / h)) : 2 * R_prev;
      R = min(max(R, R_prev + 1), l_all);
      const int side = 2 * R + 1, ntasks = side * side;
      if (g.stats && lane == 0) {
        atomicAdd(&g.stats[4], 1ull);
        atomicAdd(&g.stats[5], static_cast<unsigned long long>(ntasks));
      }
      for (int t = lane; t < ntasks; t += 64) {
        const int dz = t / side - R, dy = t % side - R;
        const int y = lcy + dy, z = lcz + dz;
        if (y < 0 || y >= g.ny || z < 0 || z >= g.nz) continue;
        const float ddz = (dz == 0) ? 0.f : (dz > 0 ? (g.oz + z * g.cell) - lqz : lqz - (g.oz + (z + 1) * g.cell));
        const float ddy = (dy == 0) ? 0.f : (dy > 0 ? (g.oy + y * g.cell) - lqy : lqy - (g.oy + (y + 1) * g.cell));
        const float row_d2 = fmaxf(ddy, 0.f) * fmaxf(ddy, 0.f) + fmaxf(ddz, 0.f) * fmaxf(ddz, 0.f);
        if (row_d2 * 0.9999f >= loc.d2 || row_d2 * 0.9999f > max_sq) continue;
        const uint32_t row = (static_cast<uint32_t>(z) * g.ny + y) * g.nx;
        if (max(abs(dy), abs(dz)) > R_prev) {
          const int x0 = max(lcx - R, 0), x1 = min(lcx + R, g.nx - 1);
          if (x0 <= x1) scan_run(g.pts, g.cell_start[row + x0], g.cell_start[row + x1 + 1], lqx, lqy, lqz, loc);
        } else {
          const int a0 = max(lcx - R, 0), a1 = min(lcx - R_prev - 1, g.nx - 1);
          if (a0 <= a1) scan_run(g.pts, g.cell_start[row + a0], g.cell_start[row + a1 + 1], lqx, lqy, lqz, loc);
          const int b0 = max(lcx + R_prev + 1, 0), b1 = min(lcx + R, g.nx - 1);
          if (b0 <= b1) scan_run(g.pts, g.cell_start[row + b0], g.cell_start[row + b1 + 1], lqx, lqy, lqz, loc);
        }
      }
      // wave arg-min: smallest distance, ties to the smallest target position (independent of the lane assignment)
      const float mn = wave_min_f32(loc.d2);
      const int cand = (loc.d2 == mn && loc.idx >= 0) ? loc.idx : 0x7fffffff;
      const int mi = wave_min_i32(cand);
      if (mi != 0x7fffffff) {
        const unsigned long long owners = __ballot(cand == mi);
        const int owner = __ffsll(static_cast<long long>(owners)) - 1;
        const float wx = __shfl(loc.x, owner), wy = __shfl(loc.y, owner), wz = __shfl(loc.z, owner);
        loc.d2 = mn;
        loc.idx = mi;
        loc.x = wx;
        loc.y = wy;
        loc.z = wz;
      }
      cur = mn;
      R_prev = R;
      const float reach = R * h;
      const float reach2 = reach * reach;
      if (cur <= reach2 || reach2 >= max_sq || R >= l_all) break;
    }
    if (lane == leader) best = loc;
  }
  return best;
}

__device__ __forceinline__ NNBest grid_nearest_wave(const GridView& g, float qx, float qy, float qz, float max_sq, bool active, int lane) {
  NNBest best;
  best.d2 = INFINITY;
  best.idx = -1;
  best.x = best.y = best.z = 0.f;
  const float h = g.cell * 0.9999f;
  int cx = 0, cy = 0, cz = 0, r_all = 1;
  bool done = true;
  if (active) {
    cx = cell_coord(qx, g.ox, g.inv_cell);
    cy = cell_coord(qy, g.oy, g.inv_cell);
    cz = cell_coord(qz, g.oz, g.inv_cell);
    r_all = max(max(max(max(cx, g.nx - 1 - cx), max(cy, g.ny - 1 - cy)), max(cz, g.nz - 1 - cz)), 1);
    if (cy >= 0 && cy < g.ny && cz >= 0 && cz < g.nz) {
      const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
      const uint32_t row = (static_cast<uint32_t>(cz) * g.ny + cy) * g.nx;
      if (x0 <= x1) scan_run(g.pts, g.cell_start[row + x0], g.cell_start[row + x1 + 1], qx, qy, qz, best);
    }
    const int zlo = max(cz - 1, 0), zhi = min(cz + 1, g.nz - 1);
    const int ylo = max(cy - 1, 0), yhi = min(cy + 1, g.ny - 1);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, g.nx - 1);
    for (int z = zlo; z <= zhi; ++z) {
      const float ddz = (z == cz) ? 0.f : (z > cz ? (g.oz + z * g.cell) - qz : qz - (g.oz + (z + 1) * g.cell));
      for (int y = ylo; y <= yhi; ++y) {
        if (y == cy && z == cz) continue;
        const float ddy = (y == cy) ? 0.f : (y > cy ? (g.oy + y * g.cell) - qy : qy - (g.oy + (y + 1) * g.cell));
        const float row_d2 = fmaxf(ddy, 0.f) * fmaxf(ddy, 0.f) + fmaxf(ddz, 0.f) * fmaxf(ddz, 0.f);
        if (row_d2 * 0.9999f >= best.d2 || row_d2 * 0.9999f > max_sq) continue;
        const uint32_t row = (static_cast<uint32_t>(z) * g.ny + y) * g.nx;
        if (x0 <= x1) scan_run(g.pts, g.cell_start[row + x0], g.cell_start[row + x1 + 1], qx, qy, qz, best);
      }
    }
    const float reach2 = h * h;
    done = best.d2 <= reach2 || reach2 >= max_sq || 1 >= r_all;
  }
  return grid_finish_wave(g, qx, qy, qz, max_sq, best, cx, cy, cz, r_all, done, lane);
}

// k nearest neighbours of one query per lane, k-best kept sorted in LDS laid out [k][block] (conflict-free: lane-contiguous).
// Semantics of KnnResult<-1>::push (ann/knn_result.hpp:80-100): ascending distances, a candidate that ties the current
// worst never displaces it.  sd/si must be initialised to +inf / -1.  Returns the k-th best squared distance.
template <int BLOCK>
__device__ __forceinline__ float grid_knn_lds(const GridView& g, float qx, float qy, float qz, int k, float max_sq, float* __restrict__ sd, int* __restrict__ si, int lane) {
  const int cx = cell_coord(qx, g.ox, g.inv_cell), cy = cell_coord(qy, g.oy, g.inv_cell), cz = cell_coord(qz, g.oz, g.inv_cell);
  int r_all = max(max(max(cx, g.nx - 1 - cx), max(cy, g.ny - 1 - cy)), max(cz, g.nz - 1 - cz));
  r_all = max(r_all, 0);
  const float h = g.cell * 0.9999f;
  float worst = INFINITY;  // k-th best so far
  auto push_run = [&](uint32_t s, uint32_t e) {
    for (uint32_t j = s; j < e; ++j) {
      const float4 p = g.pts[j];
      const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
      const float d2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
      if (d2 >= worst) continue;
      int loc = k - 1;
      for (; loc > 0 && d2 < sd[(loc - 1) * BLOCK + lane]; loc--) {
        sd[loc * BLOCK + lane] = sd[(loc - 1) * BLOCK + lane];
        si[loc * BLOCK + lane] = si[(loc - 1) * BLOCK + lane];
      }
      sd[loc * BLOCK + lane] = d2;
      si[loc * BLOCK + lane] = static_cast<int>(j);
      worst = sd[(k - 1) * BLOCK + lane];
    }
  };
  for (int r = 0;; ++r) {
    const int zlo = max(cz - r, 0), zhi = min(cz + r, g.nz - 1);
    const int ylo = max(cy - r, 0), yhi = min(cy + r, g.ny - 1);
    for (int z = zlo; z <= zhi; ++z) {
      const int adz = abs(z - cz);
      const float ddz = (z == cz) ? 0.f : (z > cz ? (g.oz + z * g.cell) - qz : qz - (g.oz + (z + 1) * g.cell));
      for (int y = ylo; y <= yhi; ++y) {
        const int ady = abs(y - cy);
        const float ddy = (y == cy) ? 0.f : (y > cy ? (g.oy + y * g.cell) - qy : qy - (g.oy + (y + 1) * g.cell));
        const float row_d2 = fmaxf(ddy, 0.f) * fmaxf(ddy, 0.f) + fmaxf(ddz, 0.f) * fmaxf(ddz, 0.f);
        if (row_d2 * 0.9999f >= worst || row_d2 * 0.9999f > max_sq) continue;
        const uint32_t row = (static_cast<uint32_t>(z) * g.ny + y) * g.nx;
        if (max(ady, adz) == r) {
          const int x0 = max(cx - r, 0), x1 = min(cx + r, g.nx - 1);
          if (x0 <= x1) push_run(g.cell_start[row + x0], g.cell_start[row + x1 + 1]);
        } else {
          const int xa = cx - r, xb = cx + r;
          if (xa >= 0 && xa < g.nx) push_run(g.cell_start[row + xa], g.cell_start[row + xa + 1]);
          if (xb >= 0 && xb < g.nx) push_run(g.cell_start[row + xb], g.cell_start[row + xb + 1]);
        }
      }
    }
    const float reach = r * h;
    const float reach2 = reach * reach;
    if (worst <= reach2 || reach2 >= max_sq || r >= r_all) break;
  }
  return worst;
}

// ---- Gaussian voxel map: open-addressing hash on the packed voxel coordinate -------------------------------------------------
__host__ __device__ __forceinline__ unsigned long long voxel_key(int x, int y, int z) {
  return (static_cast<unsigned long long>(static_cast<uint32_t>(x + (1 << 20)) & 0x1fffffu)) | (static_cast<unsigned long long>(static_cast<uint32_t>(y + (1 << 20)) & 0x1fffffu) << 21) |
         (static_cast<unsigned long long>(static_cast<uint32_t>(z + (1 << 20)) & 0x1fffffu) << 42);
}
__host__ __device__ __forceinline__ uint32_t voxel_hash(unsigned long long k) {
  k ^= k >> 33;
  k *= 0xff51afd7ed558ccdull;
  k ^= k >> 33;
  k *= 0xc4ceb9fe1a85ec53ull;
  k ^= k >> 33;
  return static_cast<uint32_t>(k);
}
#define SGA_HASH_EMPTY 0xffffffffffffffffull

struct VoxelView {
  const unsigned long long* __restrict__ hkeys;
  const uint32_t* __restrict__ hvals;
  uint32_t hmask;
  double inv_leaf;
};

// util/fast_floor.hpp:12-15 on doubles (the reference floors pt * inv_leaf_size in double)
__device__ __forceinline__ int fast_floor_d(double x) {
  const int n = static_cast<int>(x);
  return n - (x < static_cast<double>(n));
}

// returns voxel id or -1
__device__ __forceinline__ int voxel_lookup(const VoxelView& v, float qx, float qy, float qz) {
  const int cx = fast_floor_d(static_cast<double>(qx) * v.inv_leaf);
  const int cy = fast_floor_d(static_cast<double>(qy) * v.inv_leaf);
  const int cz = fast_floor_d(static_cast<double>(qz) * v.inv_leaf);
  if (abs(cx) >= (1 << 20) || abs(cy) >= (1 << 20) || abs(cz) >= (1 << 20)) return -1;
  const unsigned long long key = voxel_key(cx, cy, cz);
  uint32_t slot = voxel_hash(key) & v.hmask;
  for (uint32_t probe = 0; probe <= v.hmask; ++probe) {
    const unsigned long long k = v.hkeys[slot];
    if (k == key) return static_cast<int>(v.hvals[slot]);
    if (k == SGA_HASH_EMPTY) return -1;
    slot = (slot + 1) & v.hmask;
  }
  return -1;
}

}  // namespace sga
