// Preprocessing on the GPU so that a whole align() is device resident (registration_helper.cpp:22-34 preprocess_points):
//   sga_voxelgrid_sampling            <- util/downsampling.hpp:23-78   (serial voxel-grid mean; ascending packed-key order)
//   sga_estimate_normals_covariances  <- util/normal_estimation.hpp:13-92 (kNN incl. self -> mean/cov (1/n) -> eigvecs -> n, C)
// The covariance / eigen step runs in fp64 like the reference (sum p p^T - mean sum p^T cancels catastrophically in fp32);
// results are stored fp32.  The 3x3 eigen-solver restates Eigen 3.4.0 SelfAdjointEigenSolver<Matrix3d>::computeDirect.
#include "common.hpp"

#include <memory>
#include <rocprim/rocprim.hpp>

#include "kd_search.hpp"
#include "knn_wave.hpp"
#include "notes.hpp"
#include "sort_util.hpp"

namespace sga {

int ensure_temp(sga_context* ctx, size_t bytes);

// ---- voxel grid ----------------------------------------------------------------------------------------------------------------
// util/downsampling.hpp:23-78: key = x | y << 21 | z << 42 of the voxel coordinates + 2^20, points sorted by key, one centroid per run,
// output in ascending key order.  Round 6 (the chain of a C5 scan was 240 us of wall time for 120 us of kernels):
//   * SHORT KEYS.  An uploaded cloud knows the bounding box of its records (sga_cloud::has_box), so the host knows the range of voxel
//     coordinates per axis and the kernel packs (c - c_min) into just as many bits as the range needs, x lowest, z highest: the same
//     ORDER as the reference's 63-bit key (same axes, same significance), in 27 bits for a KITTI scan at 0.25 m — a 32-bit radix sort over
//     28 bits instead of a 64-bit one over 64.  Clouds without a box (made on the device) keep the 63-bit key.
//   * ONE KERNEL from sorted keys to run starts (ds_segments_kernel: head flags, a single-pass scan with decoupled look-back, the
//     compacted starts, the number of runs) instead of heads + a three-launch library scan + starts.
//   * NO copy commands, no stream synchronisation: the number of runs reaches the host as a note (notes.hpp) while the centroid kernel
//     — launched for the largest possible number of runs, it reads the true one on the device — is already running.
__device__ __forceinline__ int fast_floor_dd(double x) {
  const int n = static_cast<int>(x);
  return n - (x < static_cast<double>(n));
}

struct VoxelKeyLayout {
  int cmin[3];   // smallest voxel coordinate (+ 2^20) the keys can hold, per axis
  int bits[3];   // bits per axis; 21 / 21 / 21 with cmin = 0 is the reference's key
  int total;     // bits[0] + bits[1] + bits[2]; the key of a dropped point is 1 << total: it sorts behind every voxel
};

// scratch words of a voxel-grid call (per context, kept between calls): [0] arrival counter of ds_segments_kernel (0 between launches),
// [1] number of runs, [2] number of points with a valid key
template <typename Key>
__global__ void downsample_keys_kernel(const float4* __restrict__ pts, uint32_t n, double inv_leaf, double ox, double oy, double oz, VoxelKeyLayout L, Key* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t* __restrict__ scratch) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) scratch[2] = n;  // lowered by the first dropped point ds_segments_kernel meets
  if (i >= n) return;
  const float4 p = pts[i];
  // downsampling.hpp:36-49: coord = fast_floor(p * inv_leaf) + 2^20, valid iff 0 <= coord <= 2^21-1.  (ox, oy, oz): origin of the cloud's
  // device frame (common.hpp) — the voxel a point falls into is a property of its position in the CALLER's frame
  const int cx = fast_floor_dd((static_cast<double>(p.x) + ox) * inv_leaf) + (1 << 20);
  const int cy = fast_floor_dd((static_cast<double>(p.y) + oy) * inv_leaf) + (1 << 20);
  const int cz = fast_floor_dd((static_cast<double>(p.z) + oz) * inv_leaf) + (1 << 20);
  const int mask = (1 << 21) - 1;
  bool bad = cx < 0 || cy < 0 || cz < 0 || cx > mask || cy > mask || cz > mask;
  // a non-finite coordinate has no voxel: the reference's fast_floor turns it into INT_MIN on x86 (cvttsd2si), i.e. an invalid coordinate
  // that is dropped (downsampling.hpp:41-46); the conversion on the device would give 0 for a NaN — a voxel next to the origin
  bad = bad || !(fabsf(p.x) <= 3.4028234e38f) || !(fabsf(p.y) <= 3.4028234e38f) || !(fabsf(p.z) <= 3.4028234e38f);
  const unsigned ux = static_cast<unsigned>(cx - L.cmin[0]), uy = static_cast<unsigned>(cy - L.cmin[1]), uz = static_cast<unsigned>(cz - L.cmin[2]);
  bad = bad || (ux >> L.bits[0]) != 0u || (uy >> L.bits[1]) != 0u || (uz >> L.bits[2]) != 0u;  // (outside the cloud's box: only a non-finite coordinate gets here)
  const Key key = static_cast<Key>(ux) | (static_cast<Key>(uy) << L.bits[0]) | (static_cast<Key>(uz) << (L.bits[0] + L.bits[1]));
  keys[i] = bad ? (static_cast<Key>(1) << L.total) : key;
  vals[i] = i;
}

// Sorted keys -> the start of every run (seg_start[r], r in ascending key order), the number of runs (scratch[1]) and of valid points
// (scratch[2]); the number of runs also goes to the host as a note.  Single pass: a workgroup takes the next tile of 2048 keys (arrival
// order = tile order, so a workgroup only ever waits for workgroups that are already running), counts its run heads, publishes the count
// and looks back over its predecessors' published counts / prefixes 64 at a time (decoupled look-back).  status[]: one word per tile,
// {epoch : 30, state : 2, value : 32}; words of earlier launches carry an older epoch and read as "nothing yet", so nothing is cleared.
constexpr int kSegThreads = 256, kSegItems = 8, kSegTile = kSegThreads * kSegItems;
template <typename Key>
__global__ __launch_bounds__(kSegThreads) void ds_segments_kernel(const Key* __restrict__ keys, uint32_t n, Key bad, unsigned long long* __restrict__ status, unsigned epoch, uint32_t* __restrict__ scratch, uint32_t* __restrict__ seg_start,
                                                                   unsigned long long* __restrict__ note_slot, unsigned long long seq) {
  __shared__ uint32_t sh_tile, sh_wave[kSegThreads / 64], sh_excl;
  if (threadIdx.x == 0) sh_tile = atomicAdd(&scratch[0], 1u);
  __syncthreads();
  const uint32_t tile = sh_tile, num_tiles = gridDim.x;
  const uint32_t first = tile * kSegTile + threadIdx.x * kSegItems;
  Key k[kSegItems + 1];
  k[0] = (first > 0 && first - 1 < n) ? keys[first - 1] : bad;
#pragma unroll
  for (int j = 0; j < kSegItems; j++) k[j + 1] = first + j < n ? keys[first + j] : bad;
  unsigned heads = 0;
#pragma unroll
  for (int j = 0; j < kSegItems; j++) {
    const uint32_t i = first + j;
    if (i < n && k[j + 1] != bad && (i == 0 || k[j] != k[j + 1])) heads |= 1u << j;
    if (i < n && k[j + 1] == bad && (i == 0 || k[j] != bad)) scratch[2] = i;  // the first dropped point: one writer
  }
  const uint32_t mine = __popc(heads);
  // workgroup scan of the head counts
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t incl = mine;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  if (lane == 63) sh_wave[wave] = incl;
  __syncthreads();
  uint32_t wave_base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < kSegThreads / 64; w++) {
    if (w < wave) wave_base += sh_wave[w];
    total += sh_wave[w];
  }
  const unsigned long long tag = static_cast<unsigned long long>(epoch) << 34;
  constexpr unsigned long long kAgg = 1ull << 32, kPrefix = 2ull << 32;
  if (wave == 0) {
    uint32_t excl = 0;
    if (tile > 0) {
      if (lane == 0) __hip_atomic_store(&status[tile], tag | kAgg | total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      int look = static_cast<int>(tile) - 1;
      for (;;) {
        const int j = look - lane;
        unsigned long long w;
        do {
          w = j >= 0 ? __hip_atomic_load(&status[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) : (tag | kPrefix);
        } while (__ballot((w >> 34) != (tag >> 34)) != 0ull);  // every predecessor of this window has taken its tile: it publishes without waiting for anybody
        const unsigned long long prefixes = __ballot((w & kPrefix) != 0ull);
        const int stop = prefixes != 0ull ? __ffsll(static_cast<long long>(prefixes)) - 1 : 63;  // nearest predecessor holding an inclusive prefix
        uint32_t v = lane <= stop ? static_cast<uint32_t>(w) : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        excl += v;
        if (prefixes != 0ull) break;
        look -= 64;
      }
    }
    if (lane == 0) {
      __hip_atomic_store(&status[tile], tag | kPrefix | (excl + total), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      sh_excl = excl;
    }
  }
  __syncthreads();
  uint32_t r = sh_excl + wave_base + incl - mine;  // rank of this thread's first head
#pragma unroll
  for (int j = 0; j < kSegItems; j++)
    if (heads & (1u << j)) seg_start[r++] = first + j;
  if (tile == num_tiles - 1 && threadIdx.x == 0) {  // the last tile to be TAKEN: every arrival has happened, its prefix is the number of runs
    const uint32_t nseg = sh_excl + total;
    scratch[1] = nseg;
    __hip_atomic_store(&scratch[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    note_slot[1] = nseg;
    note_publish(note_slot, seq);
  }
}

// Eight lanes per voxel: a LiDAR scan has a few voxels with hundreds of points next to the sensor, and one lane walking such a
// segment alone was the tail of the whole kernel.  Lane g sums the points g, g+8, ... of the segment in fp64, then the eight partial
// sums are added in a fixed order (bit-reproducible; the grouping differs from a serial sum by rounding of the last bit at most).
// Launched for `capacity` voxels; the number there are (scratch[1]) and the end of the last run (scratch[2]) are read here.
__global__ void ds_mean_kernel(const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ scratch, const uint32_t* __restrict__ order, const float4* __restrict__ pts, float4* __restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t v = t >> 3, g = t & 7u;
  const uint32_t nseg = scratch[1];
  if ((blockIdx.x * blockDim.x) >> 3 >= nseg) return;  // workgroup-uniform
  const bool valid = v < nseg;
  double sx = 0, sy = 0, sz = 0;
  uint32_t cnt = 0;
  if (valid) {
    const uint32_t s = seg_start[v], e = v + 1 < nseg ? seg_start[v + 1] : scratch[2];
    uint32_t i = s + g;
    for (; i + 24 < e; i += 32) {  // four points of this lane in flight
      const float4 p0 = pts[order[i]], p1 = pts[order[i + 8]], p2 = pts[order[i + 16]], p3 = pts[order[i + 24]];
      sx += p0.x, sy += p0.y, sz += p0.z;
      sx += p1.x, sy += p1.y, sz += p1.z;
      sx += p2.x, sy += p2.y, sz += p2.z;
      sx += p3.x, sy += p3.y, sz += p3.z;
      cnt += 4;
    }
    for (; i < e; i += 8) {
      const float4 p = pts[order[i]];
      sx += p.x;
      sy += p.y;
      sz += p.z;
      cnt++;
    }
  }
  for (int off = 1; off < 8; off <<= 1) {  // lanes of one voxel are adjacent: butterfly inside the group of 8
    sx += __shfl_xor(sx, off);
    sy += __shfl_xor(sy, off);
    sz += __shfl_xor(sz, off);
    cnt += __shfl_xor(cnt, off);
  }
  if (valid && g == 0) {
    const double inv = 1.0 / cnt;
    out[v] = make_float4(static_cast<float>(sx * inv), static_cast<float>(sy * inv), static_cast<float>(sz * inv), __uint_as_float(v));
  }
}

// ---- 3x3 symmetric eigen-decomposition (Eigen 3.4.0 computeDirect, restated), fp64 -------------------------------------------------
struct Eig3 {
  double val[3];
  double vec[3][3];  // vec[c] = eigenvector c (ascending eigenvalues)
};

__device__ __forceinline__ void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

// m is symmetric, full 3x3 row-major; finds a unit vector in the kernel of m and a representative column
__device__ __forceinline__ void extract_kernel(const double m[3][3], double* res, double* representative) {
  int i0 = 0;
  double best = fabs(m[0][0]);
  if (fabs(m[1][1]) > best) {
    best = fabs(m[1][1]);
    i0 = 1;
  }
  if (fabs(m[2][2]) > best) i0 = 2;
  const int i1 = (i0 + 1) % 3, i2 = (i0 + 2) % 3;
  double col1[3], col2[3];
  for (int r = 0; r < 3; r++) {
    representative[r] = m[r][i0];
    col1[r] = m[r][i1];
    col2[r] = m[r][i2];
  }
  double c0[3], c1[3];
  cross3(representative, col1, c0);
  cross3(representative, col2, c1);
  const double n0 = c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2];
  const double n1 = c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2];
  if (n0 > n1) {
    const double s = 1.0 / sqrt(n0);
    for (int r = 0; r < 3; r++) res[r] = c0[r] * s;
  } else {
    const double s = 1.0 / sqrt(n1);
    for (int r = 0; r < 3; r++) res[r] = c1[r] * s;
  }
}

__device__ __forceinline__ Eig3 eigen_sym3(const double a[3][3] /* lower triangle is read */) {
  Eig3 out;
  double m[3][3] = {{a[0][0], a[1][0], a[2][0]}, {a[1][0], a[1][1], a[2][1]}, {a[2][0], a[2][1], a[2][2]}};
  const double shift = (m[0][0] + m[1][1] + m[2][2]) / 3.0;
  for (int i = 0; i < 3; i++) m[i][i] -= shift;
  double scale = 0.0;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j <= i; j++) scale = fmax(scale, fabs(m[i][j]));
  if (scale > 0.0) {
    const double inv = 1.0 / scale;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) m[i][j] *= inv;
  }
  // roots of the characteristic polynomial (trigonometric closed form)
  {
    const double s_inv3 = 1.0 / 3.0, s_sqrt3 = sqrt(3.0);
    const double c0 = m[0][0] * m[1][1] * m[2][2] + 2.0 * m[1][0] * m[2][0] * m[2][1] - m[0][0] * m[2][1] * m[2][1] - m[1][1] * m[2][0] * m[2][0] - m[2][2] * m[1][0] * m[1][0];
    const double c1 = m[0][0] * m[1][1] - m[1][0] * m[1][0] + m[0][0] * m[2][2] - m[2][0] * m[2][0] + m[1][1] * m[2][2] - m[2][1] * m[2][1];
    const double c2 = m[0][0] + m[1][1] + m[2][2];
    const double c2_over_3 = c2 * s_inv3;
    double a_over_3 = (c2 * c2_over_3 - c1) * s_inv3;
    a_over_3 = fmax(a_over_3, 0.0);
    const double half_b = 0.5 * (c0 + c2_over_3 * (2.0 * c2_over_3 * c2_over_3 - c1));
    double q = a_over_3 * a_over_3 * a_over_3 - half_b * half_b;
    q = fmax(q, 0.0);
    const double rho = sqrt(a_over_3);
    const double theta = atan2(sqrt(q), half_b) * s_inv3;
    const double ct = cos(theta), st = sin(theta);
    out.val[0] = c2_over_3 - rho * (ct + s_sqrt3 * st);
    out.val[1] = c2_over_3 - rho * (ct - s_sqrt3 * st);
    out.val[2] = c2_over_3 + 2.0 * rho * ct;
  }
  const double eps = 2.220446049250313e-16;
  if ((out.val[2] - out.val[0]) <= eps) {
    for (int c = 0; c < 3; c++)
      for (int r = 0; r < 3; r++) out.vec[c][r] = (r == c) ? 1.0 : 0.0;
  } else {
    double d0 = out.val[2] - out.val[1];
    const double d1 = out.val[1] - out.val[0];
    int k = 0, l = 2;
    if (d0 > d1) {
      k = 2;
      l = 0;
      d0 = d1;
    }
    {
      double t[3][3];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[i][j] = m[i][j] - (i == j ? out.val[k] : 0.0);
      extract_kernel(t, out.vec[k], out.vec[l]);
    }
    if (d0 <= 2.0 * eps * d1) {
      const double dp = out.vec[k][0] * out.vec[l][0] + out.vec[k][1] * out.vec[l][1] + out.vec[k][2] * out.vec[l][2];
      for (int r = 0; r < 3; r++) out.vec[l][r] -= dp * out.vec[l][r];
      const double nn = 1.0 / sqrt(out.vec[l][0] * out.vec[l][0] + out.vec[l][1] * out.vec[l][1] + out.vec[l][2] * out.vec[l][2]);
      for (int r = 0; r < 3; r++) out.vec[l][r] *= nn;
    } else {
      double t[3][3], dummy[3];
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[i][j] = m[i][j] - (i == j ? out.val[l] : 0.0);
      extract_kernel(t, out.vec[l], dummy);
    }
    cross3(out.vec[2], out.vec[0], out.vec[1]);
    const double nn = 1.0 / sqrt(out.vec[1][0] * out.vec[1][0] + out.vec[1][1] * out.vec[1][1] + out.vec[1][2] * out.vec[1][2]);
    for (int r = 0; r < 3; r++) out.vec[1][r] *= nn;
  }
  for (int i = 0; i < 3; i++) out.val[i] = out.val[i] * scale + shift;
  return out;
}

// ---- normals / covariances -----------------------------------------------------------------------------------------------------
constexpr int kFeatBlock = 64;
constexpr int kFeatWindow = 128;

// mean / covariance of the neighbourhood -> normal, regularised covariance (normal_estimation.hpp:65-92, :13-63), written to the index's
// kd-ordered arrays and, through the original index in p.w, to the caller's cloud.  id_at(j): kd position of the j-th neighbour, < 0 = no more.
template <int KS = 0, class IdAt>  // KS > 0: exactly KS neighbours, the loop unrolled (id_at may then index a register array)
__device__ __forceinline__ void features_from_neighbours(const KdView& g, size_t i, const float4 p, IdAt id_at, int k, int flags, float4* __restrict__ idx_nrm, Cov8* __restrict__ idx_cov, float4* __restrict__ cloud_nrm,
                                                         Cov8* __restrict__ cloud_cov, double ox, double oy, double oz) {
  int found = 0;
  double sp[3] = {0, 0, 0}, sc[6] = {0, 0, 0, 0, 0, 0};
  auto add = [&](int id) {
    const float4 q = g.pts[id];
    const double x = q.x, y = q.y, z = q.z;
    sp[0] += x;
    sp[1] += y;
    sp[2] += z;
    sc[0] += x * x;
    sc[1] += x * y;
    sc[2] += x * z;
    sc[3] += y * y;
    sc[4] += y * z;
    sc[5] += z * z;
    found++;
  };
  if constexpr (KS > 0) {
#pragma unroll
    for (int j = 0; j < KS; j++) {
      const int id = id_at(j);
      if (id >= 0) add(id);  // (the list is filled from the front: the entries behind the last neighbour are -1)
    }
  } else {
    for (int j = 0; j < k; j++) {
      const int id = id_at(j);
      if (id < 0) break;
      add(id);
    }
  }
  const uint32_t orig = __float_as_uint(p.w);
  float4 nrm = make_float4(0.f, 0.f, 0.f, 0.f);
  Cov8 cov;
  cov.xx = cov.yy = cov.zz = 1.f;  // normal_estimation.hpp:33-37: identity when < 5 neighbours
  cov.xy = cov.xz = cov.yz = cov.pad0 = cov.pad1 = 0.f;
  if (found >= 5) {
    const double invn = 1.0 / found;
    const double mx = sp[0] * invn, my = sp[1] * invn, mz = sp[2] * invn;
    // cov = (sum_cross - mean * sum_points^T) / n   (normal_estimation.hpp:85-86); only the lower triangle is consumed
    double a[3][3];
    a[0][0] = (sc[0] - mx * sp[0]) * invn;
    a[1][0] = (sc[1] - my * sp[0]) * invn;
    a[2][0] = (sc[2] - mz * sp[0]) * invn;
    a[1][1] = (sc[3] - my * sp[1]) * invn;
    a[2][1] = (sc[4] - mz * sp[1]) * invn;
    a[2][2] = (sc[5] - mz * sp[2]) * invn;
    a[0][1] = a[1][0];
    a[0][2] = a[2][0];
    a[1][2] = a[2][1];
    const Eig3 eg = eigen_sym3(a);
    {
      double v0[3] = {eg.vec[0][0], eg.vec[0][1], eg.vec[0][2]};
      const double nn = 1.0 / sqrt(v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2]);
      for (int r = 0; r < 3; r++) v0[r] *= nn;
      // the normal looks towards the origin of the CALLER's frame (the sensor, normal_estimation.hpp:20-25): (ox, oy, oz) = the device frame's origin
      const double dp = (static_cast<double>(p.x) + ox) * v0[0] + (static_cast<double>(p.y) + oy) * v0[1] + (static_cast<double>(p.z) + oz) * v0[2];
      const double sgn = dp > 0 ? -1.0 : 1.0;  // normal_estimation.hpp:20-25
      nrm = make_float4(static_cast<float>(sgn * v0[0]), static_cast<float>(sgn * v0[1]), static_cast<float>(sgn * v0[2]), 0.f);
    }
    {
      const double w[3] = {1e-3, 1.0, 1.0};  // normal_estimation.hpp:42-45: V diag(1e-3,1,1) V^T
      double c[6] = {0, 0, 0, 0, 0, 0};
      for (int e = 0; e < 3; e++) {
        const double* v = eg.vec[e];
        c[0] += w[e] * v[0] * v[0];
        c[1] += w[e] * v[0] * v[1];
        c[2] += w[e] * v[0] * v[2];
        c[3] += w[e] * v[1] * v[1];
        c[4] += w[e] * v[1] * v[2];
        c[5] += w[e] * v[2] * v[2];
      }
      cov.xx = static_cast<float>(c[0]);
      cov.xy = static_cast<float>(c[1]);
      cov.xz = static_cast<float>(c[2]);
      cov.yy = static_cast<float>(c[3]);
      cov.yz = static_cast<float>(c[4]);
      cov.zz = static_cast<float>(c[5]);
    }
  }
  if (flags & 1) {
    if (idx_nrm) idx_nrm[i] = nrm;
    cloud_nrm[orig] = nrm;
  }
  if (flags & 2) {
    if (idx_cov) idx_cov[i] = cov;
    cloud_cov[orig] = cov;
  }
}

// One lane per point of the kd-ordered index; neighbours come from the same tree.  Results are written both to the index's
// kd-ordered attribute arrays and, through the original index kept in pts.w, to the caller's cloud.
#ifndef SGA_FEAT_WAVES
#define SGA_FEAT_WAVES 4
#endif
template <int K>  // K > 0: k = K neighbours in registers (kd_knn_own_points); K = 0: any k, list in LDS (kd_knn)
__global__ __launch_bounds__(kFeatBlock) __attribute__((amdgpu_waves_per_eu(SGA_FEAT_WAVES, SGA_FEAT_WAVES))) void local_features_kernel(
  const KdView g, size_t n, int k, int flags, float4* __restrict__ idx_nrm, Cov8* __restrict__ idx_cov, float4* __restrict__ cloud_nrm, Cov8* __restrict__ cloud_cov, double ox, double oy, double oz) {
  extern __shared__ float sh[];
  __shared__ float4 window[kFeatWindow];  // the kd positions around the wave's own, scanned before the walk
  // K > 0: the list lives in registers from the search to the sums; the dynamic LDS is the traversal stack alone (6 KB per wave instead
  // of 16: the k-best arrays were the kernel's occupancy limit)
  const int kpad = K > 0 ? 0 : (k + 3) & ~3;  // kd_knn sweeps the list four slots at a time
  float* sd = sh;
  int* si = reinterpret_cast<int*>(sh + static_cast<size_t>(kpad) * kFeatBlock);
  uint32_t* stack = reinterpret_cast<uint32_t*>(sh + 2 * static_cast<size_t>(kpad) * kFeatBlock);
  const int lane = threadIdx.x;
  const size_t i = blockIdx.x * static_cast<size_t>(kFeatBlock) + lane;
  for (int j = 0; j < kpad; j++) {
    sd[j * kFeatBlock + lane] = INFINITY;
    si[j * kFeatBlock + lane] = -1;
  }
  // candidates scanned before the walk: the wave's own 64 positions and 32 on either side (kFeatWindow = 128), fetched with two coalesced loads
  const uint32_t base = blockIdx.x * kFeatBlock;
  const uint32_t pre_first = base > (kFeatWindow - kFeatBlock) / 2 ? base - (kFeatWindow - kFeatBlock) / 2 : 0u;
  const uint32_t pre_end = static_cast<uint32_t>(min(static_cast<size_t>(pre_first) + kFeatWindow, n));
  for (uint32_t w = lane; w < pre_end - pre_first; w += kFeatBlock) window[w] = g.pts[pre_first + w];
  __syncthreads();
  if (i >= n) return;
  const float4 p = g.pts[i];
  if constexpr (K > 0) {
    KnnRegs<K> L;
    kd_knn_own_points<K, kFeatBlock>(g, p.x, p.y, p.z, L, window, pre_first, pre_end, min(base, pre_end), min(base + kFeatBlock, pre_end), stack, lane);
    features_from_neighbours<K>(g, i, p, [&](int j) { return L.id[j]; }, K, flags, idx_nrm, idx_cov, cloud_nrm, cloud_cov, ox, oy, oz);
  } else {
    kd_knn<kFeatBlock>(g, p.x, p.y, p.z, k, INFINITY, sd, si, stack, lane, false, pre_first, pre_end, window, base, base + kFeatBlock);  // unsorted: the sums below do not depend on the order
    features_from_neighbours(g, i, p, [&](int j) { return si[j * kFeatBlock + lane]; }, k, flags, idx_nrm, idx_cov, cloud_nrm, cloud_cov, ox, oy, oz);
  }
}

// ---- small clouds: one wave per query (knn_wave.hpp), then one lane per point for the eigen-decompositions ------------------------
__global__ __launch_bounds__(64) void knn_wave_kernel(const KdView g, uint32_t n, int k, int* __restrict__ nbr /* n x k kd positions, nearest first, -1 = none */) {
  __shared__ uint32_t stack[2 * kKdMaxDepth + 2];
  const int lane = threadIdx.x;
  const uint32_t i = blockIdx.x;
  float bd;
  int bid;
  knn_wave_query(g, i, k, lane, stack, bd, bid);
  if (lane < k) nbr[static_cast<size_t>(i) * k + lane] = bd < INFINITY ? bid : -1;
}

__global__ __launch_bounds__(64) void features_from_list_kernel(const KdView g, size_t n, int k, const int* __restrict__ nbr, int flags, float4* __restrict__ idx_nrm, Cov8* __restrict__ idx_cov, float4* __restrict__ cloud_nrm,
                                                                Cov8* __restrict__ cloud_cov, double ox, double oy, double oz) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const int* __restrict__ mine = nbr + i * k;
  features_from_neighbours(g, i, g.pts[i], [&](int j) { return mine[j]; }, k, flags, idx_nrm, idx_cov, cloud_nrm, cloud_cov, ox, oy, oz);
}

__global__ void refresh_attributes_kernel(const float4* __restrict__ idx_pts, size_t n, const float4* __restrict__ cloud_nrm, const Cov8* __restrict__ cloud_cov, float4* __restrict__ idx_nrm, Cov8* __restrict__ idx_cov) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const uint32_t orig = __float_as_uint(idx_pts[i].w);
  if (cloud_nrm) idx_nrm[i] = cloud_nrm[orig];
  if (cloud_cov) idx_cov[i] = cloud_cov[orig];
}

}  // namespace sga

using namespace sga;

// clouds of at most this many points estimate their normals / covariances with one wave per query (knn_wave.hpp); SGA_KNN_WAVE_MAX, sga_set_knn_wave_max
// (81 920: the measured crossover with the one-query-per-lane kernel, whose time is flat ~200 us while it under-fills the chip — 33k points 194
// against 98 us, 60k 182 against 138, 80k 230 against 196, 100k 215 against 240 us for k = 20)
static long long g_knn_wave_max = getenv("SGA_KNN_WAVE_MAX") ? atoll(getenv("SGA_KNN_WAVE_MAX")) : 81920;

extern "C" {

void sga_set_knn_wave_max(long long max_points) { g_knn_wave_max = max_points; }

}  // extern "C"

namespace {
// bits needed for the values 0 .. range - 1
int bits_for(long long range) {
  int b = 0;
  while ((1ll << b) < range) b++;
  return b;
}

template <typename Key>
int voxelgrid_run(sga_context* ctx, const sga_cloud* in, double leaf, const VoxelKeyLayout& L, sga_cloud* res) {
  const size_t n = in->n;
  const uint32_t n32 = static_cast<uint32_t>(n);
  DevBuf<Key> keys, keys_sorted;
  DevBuf<uint32_t> vals, order, seg_start;
  SGA_TRY(keys.alloc(n));
  SGA_TRY(keys_sorted.alloc(n));
  SGA_TRY(vals.alloc(n));
  SGA_TRY(order.alloc(n));
  SGA_TRY(seg_start.alloc(n + 1));
  const uint32_t tiles = (n32 + kSegTile - 1) / kSegTile;
  if (ctx->vg_status.n < tiles || ctx->vg_scratch.n < 4 || ctx->vg_epoch >= (1u << 30) - 1u) {  // grow-only; a fresh array reads "nothing yet" for every epoch > 0
    if (ctx->vg_status.n < tiles) SGA_TRY(ctx->vg_status.alloc(std::max<size_t>(2 * tiles, 1024)));
    SGA_HIP(hipMemsetAsync(ctx->vg_status.p, 0, ctx->vg_status.n * sizeof(unsigned long long), ctx->stream));
    if (ctx->vg_scratch.n < 4) {
      SGA_TRY(ctx->vg_scratch.alloc(4));
      SGA_HIP(hipMemsetAsync(ctx->vg_scratch.p, 0, 4 * sizeof(uint32_t), ctx->stream));
    }
    ctx->vg_epoch = 0;
  }
  const unsigned epoch = ++ctx->vg_epoch;
  const dim3 grid((n + 255) / 256), block(256);
  hipLaunchKernelGGL((downsample_keys_kernel<Key>), grid, block, 0, ctx->stream, in->pts.p, n32, 1.0 / leaf, in->origin[0], in->origin[1], in->origin[2], L, keys.p, vals.p, ctx->vg_scratch.p);
  const unsigned end_bit = static_cast<unsigned>(std::min<int>(L.total + 1, 8 * static_cast<int>(sizeof(Key))));
  // (sort_util.hpp: a merge sort for a scan's 115k keys — rocPRIM's onesweep radix sort was measured slower there: 82 us of passes + 33 us
  // of state fills against 54 us — and the radix sort from 262144 keys on.  The short keys halve the bytes either of them moves.)
  SGA_TRY(sort_pairs(ctx, keys.p, keys_sorted.p, vals.p, order.p, n, 0, end_bit));
  unsigned long long* slot = nullptr;
  const unsigned long long seq = note_begin(ctx, &slot);
  hipLaunchKernelGGL((ds_segments_kernel<Key>), dim3(tiles), dim3(kSegThreads), 0, ctx->stream, keys_sorted.p, n32, static_cast<Key>(1) << L.total, ctx->vg_status.p, epoch, ctx->vg_scratch.p, seg_start.p, slot, seq);
  SGA_HIP(hipGetLastError());
  // Small clouds (a LiDAR scan): the centroid kernel is launched for n voxels before the host knows how many there are, so the device
  // never waits for the host; the output then keeps room for n points.  Large clouds wait for the count and allocate what they need.
  constexpr size_t kSpeculativeMax = 262144;
  unsigned long long payload[kNoteWords - 1];
  uint32_t nseg = 0;
  if (n <= kSpeculativeMax) {
    SGA_TRY(res->pts.alloc(n));
    hipLaunchKernelGGL(ds_mean_kernel, dim3((n * 8 + 255) / 256), dim3(256), 0, ctx->stream, seg_start.p, ctx->vg_scratch.p, order.p, in->pts.p, res->pts.p);
    SGA_HIP(hipGetLastError());
    SGA_TRY(note_wait(ctx, seq, payload));
    nseg = static_cast<uint32_t>(payload[0]);
  } else {
    SGA_TRY(note_wait(ctx, seq, payload));
    nseg = static_cast<uint32_t>(payload[0]);
    if (nseg > 0) {
      SGA_TRY(res->pts.alloc(nseg));
      hipLaunchKernelGGL(ds_mean_kernel, dim3((static_cast<size_t>(nseg) * 8 + 255) / 256), dim3(256), 0, ctx->stream, seg_start.p, ctx->vg_scratch.p, order.p, in->pts.p, res->pts.p);
      SGA_HIP(hipGetLastError());
    }
  }
  res->n = nseg;
  return SGA_OK;
}
}  // namespace

extern "C" {

int sga_voxelgrid_sampling(sga_context* ctx, const sga_cloud* in, double leaf, sga_cloud** out) {
  if (!ctx || !in || !out) return fail(SGA_ERR_INVALID, "null argument");
  if (!(leaf > 0)) return fail(SGA_ERR_INVALID, "leaf size must be positive");
  if (in->device != ctx->device) return fail(SGA_ERR_INVALID, "cloud lives on another device");
  *out = nullptr;
  SGA_ENTER(ctx);
  const size_t n = in->n;
  std::unique_ptr<sga_cloud> res(new sga_cloud);
  res->device = ctx->device;
  for (int k = 0; k < 3; k++) res->origin[k] = in->origin[k];  // the centroids stay in the input's device frame
  SGA_TRY(wait_ready(ctx, in->ready));
  if (n == 0) {  // downsampling.hpp:24-26
    *out = res.release();
    return SGA_OK;
  }
  // the reference's key layout, or — when the box of the records is known — as many bits per axis as the cloud's voxel range needs
  VoxelKeyLayout L{{0, 0, 0}, {21, 21, 21}, 63};
  if (in->has_box) {
    long long lo[3], hi[3];
    bool ok = true;
    for (int k = 0; k < 3; k++) {
      const double a = std::floor((static_cast<double>(in->box_lo[k]) + in->origin[k]) / leaf), b = std::floor((static_cast<double>(in->box_hi[k]) + in->origin[k]) / leaf);
      ok = ok && std::isfinite(a) && std::isfinite(b) && std::fabs(a) < 1e15 && std::fabs(b) < 1e15;
      // one voxel of slack on either side: p * (1 / leaf) in the kernel and p / leaf here may round to different sides of an integer
      const long long top = (1 << 21) - 1;
      lo[k] = std::min(std::max<long long>(ok ? static_cast<long long>(a) - 1 + (1 << 20) : 0, 0), top);
      hi[k] = std::max(std::min<long long>(ok ? static_cast<long long>(b) + 1 + (1 << 20) : top, top), lo[k]);  // (a box outside the grid: its points are dropped by the range test)
    }
    if (ok) {
      L.total = 0;
      for (int k = 0; k < 3; k++) {
        L.cmin[k] = static_cast<int>(lo[k]);
        L.bits[k] = std::max(1, bits_for(hi[k] - lo[k] + 1));
        L.total += L.bits[k];
      }
    }
  }
  SGA_TRY(L.total <= 31 ? voxelgrid_run<uint32_t>(ctx, in, leaf, L, res.get()) : voxelgrid_run<unsigned long long>(ctx, in, leaf, L, res.get()));
  if (!ctx->stream_ordered) SGA_HIP(hipStreamSynchronize(ctx->stream));
  SGA_TRY(mark_ready(ctx, res->ready));
  *out = res.release();
  return SGA_OK;
}

int sga_index_refresh_attributes(sga_context* ctx, sga_index* index, const sga_cloud* cloud) {
  if (!ctx || !index || !cloud) return fail(SGA_ERR_INVALID, "null argument");
  if (index->kind != SGA_INDEX_KDTREE) return fail(SGA_ERR_INVALID, "not a kd-tree index");
  if (index->n != cloud->n) return fail(SGA_ERR_INVALID, "index was built over a cloud of %zu points, got %zu", index->n, cloud->n);
  SGA_ENTER(ctx);
  SGA_TRY(wait_ready(ctx, index->ready));  // produced on another context in stream-ordered mode (common.hpp: Ready)
  SGA_TRY(wait_ready(ctx, cloud->ready));
  const size_t n = index->n;
  if (cloud->has_normals && index->nrm.n < n) SGA_TRY(index->nrm.alloc(n));
  if (cloud->has_covs && index->cov.n < n) SGA_TRY(index->cov.alloc(n));
  if (n > 0 && (cloud->has_normals || cloud->has_covs)) {
    hipLaunchKernelGGL(refresh_attributes_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, index->kd_pts.p, n, cloud->has_normals ? cloud->nrm.p : nullptr, cloud->has_covs ? cloud->cov.p : nullptr, index->nrm.p, index->cov.p);
    SGA_HIP(hipGetLastError());
    SGA_HIP(hipStreamSynchronize(ctx->stream));
  }
  index->has_normals = cloud->has_normals;
  index->has_covs = cloud->has_covs;
  return SGA_OK;
}

int sga_estimate_normals_covariances(sga_context* ctx, sga_cloud* cloud, const sga_index* index_in, int k, int flags) {
  if (!ctx || !cloud) return fail(SGA_ERR_INVALID, "null argument");
  // LDS per workgroup: the k-best list (kpad * 8 bytes per lane, kpad = k rounded up to 4) + the traversal stack (kKdMaxDepth words per
  // lane) + the static candidate window (kFeatWindow float4) must fit the 64 KB a workgroup may allocate
  constexpr int kMaxK = ((64 * 1024 - kFeatWindow * 16) / 64 - kKdMaxDepth * 4) / 8 / 4 * 4;  // 112
  if (k < 1 || k > kMaxK) return fail(SGA_ERR_INVALID, "num_neighbors must be in [1,%d] (k-best list + traversal stack + candidate window must fit 64 KB of LDS per workgroup)", kMaxK);
  if ((flags & 3) == 0) return SGA_OK;
  if (cloud->device != ctx->device) return fail(SGA_ERR_INVALID, "cloud lives on another device");
  SGA_ENTER(ctx);
  const size_t n = cloud->n;
  sga_index* index = const_cast<sga_index*>(index_in);
  sga_index* temp = nullptr;
  if (!index) {
    SGA_TRY(sga_index_build_kdtree(ctx, cloud, &temp));
    index = temp;
  } else {
    if (index->kind != SGA_INDEX_KDTREE) return fail(SGA_ERR_INVALID, "a kd-tree index is required");
    if (index->n != n) return fail(SGA_ERR_INVALID, "index was built over a cloud of %zu points, got %zu", index->n, n);
    for (int a = 0; a < 3; a++)
      if (index->origin[a] != cloud->origin[a]) return fail(SGA_ERR_INVALID, "the index was not built over this cloud (their device frames differ)");
    SGA_TRY(wait_ready(ctx, index->ready));  // built on another context that returned before its kernels had run
  }
  SGA_TRY(wait_ready(ctx, cloud->ready));
  int rc = SGA_OK;
  if ((flags & 1) && cloud->nrm.n < n) rc = cloud->nrm.alloc(n);
  if (rc == SGA_OK && (flags & 2) && cloud->cov.n < n) rc = cloud->cov.alloc(n);
  if (rc == SGA_OK && n > 0) {
    const size_t shmem = (static_cast<size_t>((k == 20 || k == 10) ? 0 : ((k + 3) & ~3)) * 8 + kKdMaxDepth * 4) * kFeatBlock;  // (k = 10 / 20: the list lives in registers)
    if ((flags & 1) && !temp && index->nrm.n < n) rc = index->nrm.alloc(n);
    if (rc == SGA_OK && (flags & 2) && !temp && index->cov.n < n) rc = index->cov.alloc(n);
    KdView kv = make_kd_view(index);
    const dim3 fgrid((n + kFeatBlock - 1) / kFeatBlock), fblock(kFeatBlock);
    float4* inrm = temp ? nullptr : index->nrm.p;
    Cov8* icov = temp ? nullptr : index->cov.p;
    // clouds that do not fill the chip (a LiDAR scan after the voxel grid): one wave per query (knn_wave.hpp), then one lane per point
    if (n <= static_cast<size_t>(g_knn_wave_max) && k <= 64) {
      DevBuf<int> nbr;
      rc = nbr.alloc(n * static_cast<size_t>(k));
      if (rc == SGA_OK) {
        hipLaunchKernelGGL(knn_wave_kernel, dim3(static_cast<unsigned>(n)), dim3(64), 0, ctx->stream, kv, static_cast<uint32_t>(n), k, nbr.p);
        hipLaunchKernelGGL(features_from_list_kernel, fgrid, fblock, 0, ctx->stream, kv, n, k, nbr.p, flags, inrm, icov, cloud->nrm.p, cloud->cov.p, cloud->origin[0], cloud->origin[1], cloud->origin[2]);
      }
    } else if (k == 20)
      hipLaunchKernelGGL((local_features_kernel<20>), fgrid, fblock, shmem, ctx->stream, kv, n, k, flags, inrm, icov, cloud->nrm.p, cloud->cov.p, cloud->origin[0], cloud->origin[1], cloud->origin[2]);
    else if (k == 10)
      hipLaunchKernelGGL((local_features_kernel<10>), fgrid, fblock, shmem, ctx->stream, kv, n, k, flags, inrm, icov, cloud->nrm.p, cloud->cov.p, cloud->origin[0], cloud->origin[1], cloud->origin[2]);
    else
      hipLaunchKernelGGL((local_features_kernel<0>), fgrid, fblock, shmem, ctx->stream, kv, n, k, flags, inrm, icov, cloud->nrm.p, cloud->cov.p, cloud->origin[0], cloud->origin[1], cloud->origin[2]);
    hipError_t e = rc == SGA_OK ? hipGetLastError() : hipSuccess;
    if (rc == SGA_OK && e == hipSuccess && !ctx->stream_ordered) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) rc = fail(SGA_ERR_HIP, "local_features_kernel: %s", hipGetErrorString(e));
  }
  if (rc == SGA_OK) {
    if (flags & 1) cloud->has_normals = true;
    if (flags & 2) cloud->has_covs = true;
  }
  if (rc == SGA_OK && !temp) {  // the kernel wrote the index's kd-ordered copies as well
    if (flags & 1) index->has_normals = true;
    if (flags & 2) index->has_covs = true;
    rc = mark_ready(ctx, index->ready);
  }
  if (rc == SGA_OK) rc = mark_ready(ctx, cloud->ready);
  if (temp) sga_index_destroy(temp);
  return rc;
}

}  // extern "C"
