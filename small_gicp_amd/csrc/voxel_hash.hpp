// GaussianVoxelMap probe (ann/incremental_voxelmap.hpp:99-119, reference tree /root/reference): open-addressing hash on the
// packed voxel coordinate.  Device code, gfx950.  (The point-cloud target is searched through kd_search.hpp.)
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace sga {

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
  double org[3];  // origin of the device frame the queries and the stored means live in (common.hpp): voxel coordinates are the CALLER's
  int offsets;    // 1 (the query's own voxel), 7 or 27 (incremental_voxelmap.hpp:157-186)
};

// util/fast_floor.hpp:12-15 on doubles (the reference floors pt * inv_leaf_size in double)
__device__ __forceinline__ int fast_floor_d(double x) {
  const int n = static_cast<int>(x);
  return n - (x < static_cast<double>(n));
}

// returns voxel id or -1
__device__ __forceinline__ int voxel_lookup(const VoxelView& v, float qx, float qy, float qz) {
  const int cx = fast_floor_d((static_cast<double>(qx) + v.org[0]) * v.inv_leaf);
  const int cy = fast_floor_d((static_cast<double>(qy) + v.org[1]) * v.inv_leaf);
  const int cz = fast_floor_d((static_cast<double>(qz) + v.org[2]) * v.inv_leaf);
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

// Gaussian maps searched over 7 / 27 voxels (incremental_voxelmap.hpp:99-119: every voxel at the search offsets offers its mean,
// gaussian_voxelmap.hpp:83-86; KnnResult<1>::push keeps the FIRST of equal distances; offsets in the reference's order: centre, +x +y +z
// -x -y -z (7), or centre and then the 3 x 3 x 3 cube in i, j, k order (27).  Why the centre comes FIRST for 27 as well (ADVICE r5 read the
// reference's case 27 as the plain cube): incremental_voxelmap.hpp:176-183 APPENDS the 27 cube offsets to whatever list the map held
// (emplace_back without a clear), and every list the map can hold starts with the centre (the default {0}, :163-164, or the 7 pattern,
// :166-174) — so after set_search_offsets(27) the search visits the centre, then the cube; the cube's own centre entry is a second visit
// of the same voxel and, with KnnResult<1>::push keeping the first of equal distances, cannot change the result: it is skipped here).
// Returns the voxel id or -1.  Distances between the fp32 records the device holds, evaluated in Real.
template <typename Real>
__device__ __forceinline__ int voxel_nearest(const VoxelView& v, const float4* __restrict__ means, Real qx, Real qy, Real qz) {
  const int cx = fast_floor_d((static_cast<double>(qx) + v.org[0]) * v.inv_leaf);
  const int cy = fast_floor_d((static_cast<double>(qy) + v.org[1]) * v.inv_leaf);
  const int cz = fast_floor_d((static_cast<double>(qz) + v.org[2]) * v.inv_leaf);
  Real best = static_cast<Real>(INFINITY);
  int j = -1;
  auto offer = [&](int ox, int oy, int oz) {
    const int x = cx + ox, y = cy + oy, z = cz + oz;
    if (abs(x) >= (1 << 20) || abs(y) >= (1 << 20) || abs(z) >= (1 << 20)) return;
    const unsigned long long key = voxel_key(x, y, z);
    uint32_t slot = voxel_hash(key) & v.hmask;
    int vox = -1;
    for (uint32_t probe = 0; probe <= v.hmask; ++probe) {
      const unsigned long long k = v.hkeys[slot];
      if (k == key) {
        vox = static_cast<int>(v.hvals[slot]);
        break;
      }
      if (k == SGA_HASH_EMPTY) break;
      slot = (slot + 1) & v.hmask;
    }
    if (vox < 0) return;
    const float4 m = means[vox];
    const Real dx = static_cast<Real>(m.x) - qx, dy = static_cast<Real>(m.y) - qy, dz = static_cast<Real>(m.z) - qz;
    const Real d2 = dx * dx + dy * dy + dz * dz;
    if (d2 >= best) return;
    best = d2;
    j = vox;
  };
  offer(0, 0, 0);
  if (v.offsets == 7) {
    offer(1, 0, 0);
    offer(0, 1, 0);
    offer(0, 0, 1);
    offer(-1, 0, 0);
    offer(0, -1, 0);
    offer(0, 0, -1);
  } else if (v.offsets == 27) {
    for (int a = -1; a <= 1; a++)
      for (int b = -1; b <= 1; b++)
        for (int c = -1; c <= 1; c++)
          if (a || b || c) offer(a, b, c);
  }
  return j;
}

// Flat maps: nearest stored point over the search-offset pattern (incremental_voxelmap.hpp:99-119 + flat_container.hpp:84-93 with
// KnnResult<1>::push, which keeps the FIRST of equal distances).  Offsets in the reference's order: centre, then +x +y +z -x -y -z
// (7) or the 3x3x3 cube in i, j, k order (27; its second visit of the centre cannot change the result and is skipped).
struct FlatView {
  const unsigned long long* __restrict__ hkeys;
  const uint32_t* __restrict__ hvals;
  uint32_t hmask;
  double inv_leaf;
  const uint32_t* __restrict__ vnum;  // points per voxel
  int offsets;                        // 1, 7, 27
  double org[3];                      // origin of the device frame (see VoxelView)
};

__device__ __forceinline__ int flat_voxel_at(const FlatView& v, int cx, int cy, int cz) {
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

// returns the slot (voxel * kFlatCap + i) of the nearest stored point or -1; t = that point
template <typename Real>
__device__ __forceinline__ int flat_nearest(const FlatView& v, const float4* __restrict__ pts, Real qx, Real qy, Real qz, float4& t) {
  const int cx = fast_floor_d((static_cast<double>(qx) + v.org[0]) * v.inv_leaf);
  const int cy = fast_floor_d((static_cast<double>(qy) + v.org[1]) * v.inv_leaf);
  const int cz = fast_floor_d((static_cast<double>(qz) + v.org[2]) * v.inv_leaf);
  Real best = static_cast<Real>(INFINITY);
  int j = -1;
  auto scan = [&](int ox, int oy, int oz) {
    const int vox = flat_voxel_at(v, cx + ox, cy + oy, cz + oz);
    if (vox < 0) return;
    const uint32_t n = v.vnum[vox];
    for (uint32_t i = 0; i < n; i++) {
      const float4 p = pts[static_cast<size_t>(vox) * kFlatCap + i];
      const Real dx = static_cast<Real>(p.x) - qx, dy = static_cast<Real>(p.y) - qy, dz = static_cast<Real>(p.z) - qz;
      const Real d2 = dx * dx + dy * dy + dz * dz;
      if (d2 >= best) continue;
      best = d2;
      j = vox * kFlatCap + static_cast<int>(i);
      t = p;
    }
  };
  scan(0, 0, 0);
  if (v.offsets == 7) {
    scan(1, 0, 0);
    scan(0, 1, 0);
    scan(0, 0, 1);
    scan(-1, 0, 0);
    scan(0, -1, 0);
    scan(0, 0, -1);
  } else if (v.offsets == 27) {
    for (int a = -1; a <= 1; a++)
      for (int b = -1; b <= 1; b++)
        for (int c = -1; c <= 1; c++)
          if (a || b || c) scan(a, b, c);
  }
  return j;
}

}  // namespace sga
