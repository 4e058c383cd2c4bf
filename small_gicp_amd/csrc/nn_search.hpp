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
  return g;
}

__device__ __forceinline__ int cell_coord(float q, float o, float inv) { return static_cast<int>(floorf((q - o) * inv)); }

struct NNBest {
  float d2;
  int idx;  // position in the cell-sorted target, -1 = none
  float x, y, z;
};

__device__ __forceinline__ void scan_run(const float4* __restrict__ pts, uint32_t s, uint32_t e, float qx, float qy, float qz, NNBest& best) {
  for (uint32_t j = s; j < e; ++j) {
    const float4 p = pts[j];
    const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
    const float d2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
    if (d2 < best.d2) {
      best.d2 = d2;
      best.idx = static_cast<int>(j);
      best.x = p.x;
      best.y = p.y;
      best.z = p.z;
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
