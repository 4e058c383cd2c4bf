// Preprocessing on the GPU so that a whole align() is device resident (registration_helper.cpp:22-34 preprocess_points):
//   sga_voxelgrid_sampling            <- util/downsampling.hpp:23-78   (serial voxel-grid mean; ascending packed-key order)
//   sga_estimate_normals_covariances  <- util/normal_estimation.hpp:13-92 (kNN incl. self -> mean/cov (1/n) -> eigvecs -> n, C)
// The covariance / eigen step runs in fp64 like the reference (sum p p^T - mean sum p^T cancels catastrophically in fp32);
// results are stored fp32.  The 3x3 eigen-solver restates Eigen 3.4.0 SelfAdjointEigenSolver<Matrix3d>::computeDirect.
#include "common.hpp"

#include <memory>
#include <rocprim/rocprim.hpp>

#include "kd_search.hpp"

namespace sga {

int ensure_temp(sga_context* ctx, size_t bytes);

// ---- voxel grid ----------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int fast_floor_dd(double x) {
  const int n = static_cast<int>(x);
  return n - (x < static_cast<double>(n));
}

// (ox, oy, oz): the origin of the cloud's device frame (common.hpp) — the voxel a point falls into is a property of its position in the
// CALLER's frame, so the origin is added back in double before the floor
__global__ void downsample_keys_kernel(const float4* __restrict__ pts, size_t n, double inv_leaf, double ox, double oy, double oz, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  // downsampling.hpp:36-49: coord = fast_floor(p * inv_leaf) + 2^20, valid iff 0 <= coord <= 2^21-1, key = x | y<<21 | z<<42
  const int cx = fast_floor_dd((static_cast<double>(p.x) + ox) * inv_leaf) + (1 << 20);
  const int cy = fast_floor_dd((static_cast<double>(p.y) + oy) * inv_leaf) + (1 << 20);
  const int cz = fast_floor_dd((static_cast<double>(p.z) + oz) * inv_leaf) + (1 << 20);
  const int mask = (1 << 21) - 1;
  const bool bad = cx < 0 || cy < 0 || cz < 0 || cx > mask || cy > mask || cz > mask;
  keys[i] = bad ? ~0ull : (static_cast<unsigned long long>(cx) | (static_cast<unsigned long long>(cy) << 21) | (static_cast<unsigned long long>(cz) << 42));
  vals[i] = static_cast<uint32_t>(i);
}

__global__ void ds_heads_kernel(const unsigned long long* __restrict__ keys, size_t n, uint32_t* __restrict__ flags) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  flags[i] = (k != ~0ull && (i == 0 || keys[i - 1] != k)) ? 1u : 0u;
}

__global__ void ds_starts_kernel(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ seg_id, size_t n, uint32_t* __restrict__ seg_start) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  if (flags[i]) seg_start[seg_id[i]] = static_cast<uint32_t>(i);
}

// Eight lanes per voxel: a LiDAR scan has a few voxels with hundreds of points next to the sensor, and one lane walking such a
// segment alone was the tail of the whole kernel.  Lane g sums the points g, g+8, ... of the segment in fp64, then the eight partial
// sums are added in a fixed order (bit-reproducible; the grouping differs from a serial sum by rounding of the last bit at most).
__global__ void ds_mean_kernel(const uint32_t* __restrict__ seg_start, uint32_t nseg, const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ order, size_t n, const float4* __restrict__ pts, float4* __restrict__ out) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t v = t >> 3, g = t & 7u;
  const bool valid = v < nseg;
  double sx = 0, sy = 0, sz = 0;
  uint32_t cnt = 0;
  if (valid) {
    const uint32_t s = seg_start[v];
    const unsigned long long key = keys[s];
    for (size_t i = s + g; i < n && keys[i] == key; i += 8) {
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

// One lane per point of the kd-ordered index; neighbours come from the same tree.  Results are written both to the index's
// kd-ordered attribute arrays and, through the original index kept in pts.w, to the caller's cloud.
template <int K>  // K > 0: k = K neighbours in registers (kd_knn_own_points); K = 0: any k, list in LDS (kd_knn)
__global__ __launch_bounds__(kFeatBlock) void local_features_kernel(
  const KdView g, size_t n, int k, int flags, float4* __restrict__ idx_nrm, Cov8* __restrict__ idx_cov, float4* __restrict__ cloud_nrm, Cov8* __restrict__ cloud_cov, double ox, double oy, double oz) {
  extern __shared__ float sh[];
  __shared__ float4 window[kFeatWindow];  // the kd positions around the wave's own, scanned before the walk
  const int kpad = (k + 3) & ~3;  // kd_knn sweeps the list four slots at a time
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
#pragma unroll
    for (int j = 0; j < K; j++) {  // hand the list to the common code below
      sd[j * kFeatBlock + lane] = L.d[j];
      si[j * kFeatBlock + lane] = L.id[j];
    }
  } else {
    kd_knn<kFeatBlock>(g, p.x, p.y, p.z, k, INFINITY, sd, si, stack, lane, false, pre_first, pre_end, window, base, base + kFeatBlock);  // unsorted: the sums below do not depend on the order
  }
  int found = 0;
  double sp[3] = {0, 0, 0}, sc[6] = {0, 0, 0, 0, 0, 0};
  for (int j = 0; j < k; j++) {
    const int id = si[j * kFeatBlock + lane];
    if (id < 0) break;
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

__global__ void refresh_attributes_kernel(const float4* __restrict__ idx_pts, size_t n, const float4* __restrict__ cloud_nrm, const Cov8* __restrict__ cloud_cov, float4* __restrict__ idx_nrm, Cov8* __restrict__ idx_cov) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const uint32_t orig = __float_as_uint(idx_pts[i].w);
  if (cloud_nrm) idx_nrm[i] = cloud_nrm[orig];
  if (cloud_cov) idx_cov[i] = cloud_cov[orig];
}

}  // namespace sga

using namespace sga;

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
  DevBuf<unsigned long long> keys, keys_sorted;
  DevBuf<uint32_t> vals, order, flags, seg_id, seg_start;
  SGA_TRY(keys.alloc(n));
  SGA_TRY(keys_sorted.alloc(n));
  SGA_TRY(vals.alloc(n));
  SGA_TRY(order.alloc(n));
  SGA_TRY(flags.alloc(n));
  SGA_TRY(seg_id.alloc(n));
  const dim3 grid((n + 255) / 256), block(256);
  hipLaunchKernelGGL(downsample_keys_kernel, grid, block, 0, ctx->stream, in->pts.p, n, 1.0 / leaf, in->origin[0], in->origin[1], in->origin[2], keys.p, vals.p);
  size_t tb = 0;
  SGA_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys.p, keys_sorted.p, vals.p, order.p, n, 0, 64, ctx->stream));
  SGA_TRY(ensure_temp(ctx, tb));
  SGA_HIP(rocprim::radix_sort_pairs(ctx->d_temp.p, tb, keys.p, keys_sorted.p, vals.p, order.p, n, 0, 64, ctx->stream));
  hipLaunchKernelGGL(ds_heads_kernel, grid, block, 0, ctx->stream, keys_sorted.p, n, flags.p);
  size_t tb2 = 0;
  SGA_HIP(rocprim::exclusive_scan(nullptr, tb2, flags.p, seg_id.p, 0u, n, rocprim::plus<uint32_t>(), ctx->stream));
  SGA_TRY(ensure_temp(ctx, tb2));
  SGA_HIP(rocprim::exclusive_scan(ctx->d_temp.p, tb2, flags.p, seg_id.p, 0u, n, rocprim::plus<uint32_t>(), ctx->stream));
  uint32_t last_flag = 0, last_seg = 0;
  SGA_HIP(hipMemcpyAsync(&last_flag, flags.p + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
  SGA_HIP(hipMemcpyAsync(&last_seg, seg_id.p + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
  SGA_HIP(hipStreamSynchronize(ctx->stream));
  const uint32_t nseg = last_seg + last_flag;
  res->n = nseg;
  if (nseg > 0) {
    SGA_TRY(seg_start.alloc(nseg));
    SGA_TRY(res->pts.alloc(nseg));
    hipLaunchKernelGGL(ds_starts_kernel, grid, block, 0, ctx->stream, flags.p, seg_id.p, n, seg_start.p);
    hipLaunchKernelGGL(ds_mean_kernel, dim3((static_cast<size_t>(nseg) * 8 + 255) / 256), dim3(256), 0, ctx->stream, seg_start.p, nseg, keys_sorted.p, order.p, n, in->pts.p, res->pts.p);
    SGA_HIP(hipGetLastError());
    SGA_HIP(hipStreamSynchronize(ctx->stream));
  }
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
    const size_t shmem = (static_cast<size_t>((k + 3) & ~3) * 8 + kKdMaxDepth * 4) * kFeatBlock;
    if ((flags & 1) && !temp && index->nrm.n < n) rc = index->nrm.alloc(n);
    if (rc == SGA_OK && (flags & 2) && !temp && index->cov.n < n) rc = index->cov.alloc(n);
    KdView kv = make_kd_view(index);
    const dim3 fgrid((n + kFeatBlock - 1) / kFeatBlock), fblock(kFeatBlock);
    float4* inrm = temp ? nullptr : index->nrm.p;
    Cov8* icov = temp ? nullptr : index->cov.p;
    // the reference's two neighbourhood sizes (registration_helper.cpp:60-61 k = 10, the benchmarks' k = 20) keep the list in registers
    if (k == 20)
      hipLaunchKernelGGL((local_features_kernel<20>), fgrid, fblock, shmem, ctx->stream, kv, n, k, flags, inrm, icov, cloud->nrm.p, cloud->cov.p, cloud->origin[0], cloud->origin[1], cloud->origin[2]);
    else if (k == 10)
      hipLaunchKernelGGL((local_features_kernel<10>), fgrid, fblock, shmem, ctx->stream, kv, n, k, flags, inrm, icov, cloud->nrm.p, cloud->cov.p, cloud->origin[0], cloud->origin[1], cloud->origin[2]);
    else
      hipLaunchKernelGGL((local_features_kernel<0>), fgrid, fblock, shmem, ctx->stream, kv, n, k, flags, inrm, icov, cloud->nrm.p, cloud->cov.p, cloud->origin[0], cloud->origin[1], cloud->origin[2]);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && !ctx->stream_ordered) e = hipStreamSynchronize(ctx->stream);
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
