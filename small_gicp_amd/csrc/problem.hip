// sga_problem: the (target index, source cloud) pairing that Registration<>::align creates per call
// (registration/registration.hpp:41 `std::vector<PointFactor> factors(size(source))`), device resident.
// The source is copied once in a spatially coherent order (sorted by the target-grid cell of init_T * p, Morton order
// over cells) so that the 64 lanes of a wave query neighbouring cells and share cache lines of the target.
#include "common.hpp"

#include <cmath>
#include <memory>
#include <rocprim/rocprim.hpp>
#include "device_math.hpp"
#include "kd_search.hpp"
#include "sort_util.hpp"
#include "voxel_hash.hpp"

namespace sga {

int ensure_temp(sga_context* ctx, size_t bytes);
size_t problem_partials_doubles(size_t n);
int problem_ensure_maha(sga_context* ctx, sga_problem* pb);
int cloud_bbox(sga_context* ctx, const float4* pts, size_t n, float lo[3], float hi[3]);

__device__ __forceinline__ unsigned long long spread3(unsigned long long v) {
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x1f00000000ffffull;
  v = (v | (v << 16)) & 0x1f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}

__global__ void source_keys_kernel(const float4* __restrict__ pts, size_t n, Rigid<float> T, float ox, float oy, float oz, float inv, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  float qx, qy, qz;
  transform_point<float>(T, p.x, p.y, p.z, qx, qy, qz);
  // cell coordinates relative to the target grid, biased so that sources sticking out of the grid stay ordered
  const long long bias = 1 << 20;
  long long cx = static_cast<long long>(floorf((qx - ox) * inv)) + bias, cy = static_cast<long long>(floorf((qy - oy) * inv)) + bias, cz = static_cast<long long>(floorf((qz - oz) * inv)) + bias;
  cx = min(max(cx, 0ll), (1ll << 21) - 1);
  cy = min(max(cy, 0ll), (1ll << 21) - 1);
  cz = min(max(cz, 0ll), (1ll << 21) - 1);
  keys[i] = spread3(cx) | (spread3(cy) << 1) | (spread3(cz) << 2);
  vals[i] = static_cast<uint32_t>(i);
}

// Sort key for a kd-tree target: the leaf the transformed point descends into (high bits) refined by the Morton code of its
// position inside the target's bounding box (low bits).  The 64 lanes of a wave then start in the same or neighbouring leaves
// and walk nearly the same nodes in the same order: less divergence, better cache reuse than plain Morton order.
// (Measured and dropped: grouping the source by search work — leaves scanned by a probe search — on top of this; the lanes of a
// wave then finish together, but their loads scatter and the kernel got 11 % slower.)
__global__ void source_kd_keys_kernel(const float4* __restrict__ pts, size_t n, Rigid<float> T, KdView kd, float ox, float oy, float oz, float inv, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  float qx, qy, qz;
  transform_point<float>(T, p.x, p.y, p.z, qx, qy, qz);
  uint32_t node = 1;
  for (int d = 0; d < kd.depth; d++) {
    const float2 nd = kd.nodes[node];
    const int axis = __float_as_int(nd.y);
    const float qa = axis == 0 ? qx : (axis == 1 ? qy : qz);
    node = 2 * node + (qa - nd.x < 0.f ? 0u : 1u);
  }
  const unsigned long long leaf = node - (1u << kd.depth);
  long long cx = static_cast<long long>(floorf((qx - ox) * inv)), cy = static_cast<long long>(floorf((qy - oy) * inv)), cz = static_cast<long long>(floorf((qz - oz) * inv));
  cx = min(max(cx, 0ll), 1023ll);
  cy = min(max(cy, 0ll), 1023ll);
  cz = min(max(cz, 0ll), 1023ll);
  const unsigned long long fine = (spread3(cx) | (spread3(cy) << 1) | (spread3(cz) << 2)) & 0x3fffffffull;  // 30 bits
  keys[i] = (leaf << 30) | fine;
  vals[i] = static_cast<uint32_t>(i);
}

__global__ void gather_source_kernel(const uint32_t* __restrict__ order, size_t n, const float4* __restrict__ pts, const Cov8* __restrict__ cov, float4* __restrict__ opts, Cov8* __restrict__ ocov) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = order[i];
  opts[i] = pts[s];
  if (cov) ocov[i] = cov[s];
}

// Factor state back in the caller's source order with original target indices.
template <typename Real>
__global__ void export_factors_kernel(const float4* __restrict__ src_pts, const int* __restrict__ corr, const Real* __restrict__ maha, size_t n, const float4* __restrict__ tgt_pts, int flat, long long* __restrict__ out_idx, float* __restrict__ out_m) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const uint32_t orig = __float_as_uint(src_pts[i].w);
  const int j = corr[i];
  if (out_idx) {
    long long t = j < 0 ? -1ll : static_cast<long long>(__float_as_uint(tgt_pts[j].w));
    if (flat && t >= 0) t = ((t / kFlatCap) << 32) | (t % kFlatCap);  // incremental_voxelmap.hpp:151: (voxel_id << 32) | point_id
    out_idx[orig] = t;
  }
  if (out_m) {
    for (int k = 0; k < 6; k++) out_m[6 * static_cast<size_t>(orig) + k] = (j >= 0 && maha) ? static_cast<float>(maha[6 * i + k]) : 0.f;
  }
}

// ---- standalone kNN (traits::knn_search).  One lane per query, k-best in LDS [k][64], traversal stack in LDS [level][64]. ----
constexpr int kKnnBlock = 64;
constexpr int kKnnMaxK = 116;  // (k * 8 + 96) * 64 bytes of LDS per workgroup must stay below 64 KB

// queries64 / out_d2_64 (optional): the squared distances of the neighbours found are re-evaluated in double against the double
// query — the coordinates stored on the device are fp32, the reference returns double distances (ann/kdtree.hpp:193-233)
__global__ __launch_bounds__(kKnnBlock) void knn_kernel(const KdView t, const float* __restrict__ queries, size_t m, int k, float max_sq, long long* __restrict__ out_idx, float* __restrict__ out_d2, const double* __restrict__ queries64, double* __restrict__ out_d2_64) {
  extern __shared__ float sh[];  // k*64 distances, k*64 indices, kKdMaxDepth*64 stack words
  const int kpad = (k + 3) & ~3;  // kd_knn sweeps the list four slots at a time
  float* sd = sh;
  int* si = reinterpret_cast<int*>(sh + static_cast<size_t>(kpad) * kKnnBlock);
  uint32_t* stack = reinterpret_cast<uint32_t*>(sh + 2 * static_cast<size_t>(kpad) * kKnnBlock);
  const int lane = threadIdx.x;
  const size_t qi = blockIdx.x * static_cast<size_t>(kKnnBlock) + lane;
  for (int j = 0; j < kpad; j++) {
    sd[j * kKnnBlock + lane] = INFINITY;
    si[j * kKnnBlock + lane] = -1;
  }
  if (qi >= m) return;
  const float qx = queries[3 * qi], qy = queries[3 * qi + 1], qz = queries[3 * qi + 2];
  const float bound2 = max_sq < 3.0e38f ? max_sq * 1.0000002f : INFINITY;
  kd_knn<kKnnBlock>(t, qx, qy, qz, k, bound2, sd, si, stack, lane);
  for (int j = 0; j < k; j++) {
    const float d2 = sd[j * kKnnBlock + lane];
    const int id = si[j * kKnnBlock + lane];
    const bool ok = id >= 0 && !(d2 > max_sq);
    const float4 c = ok ? t.pts[id] : make_float4(0.f, 0.f, 0.f, 0.f);
    out_idx[qi * k + j] = ok ? static_cast<long long>(__float_as_uint(c.w)) : -1ll;
    out_d2[qi * k + j] = ok ? d2 : INFINITY;
    if (out_d2_64 != nullptr) {
      const double dx = static_cast<double>(c.x) - queries64[3 * qi], dy = static_cast<double>(c.y) - queries64[3 * qi + 1], dz = static_cast<double>(c.z) - queries64[3 * qi + 2];
      out_d2_64[qi * k + j] = ok ? dx * dx + dy * dy + dz * dz : INFINITY;
    }
  }
}

// kNN over a voxel map (ann/incremental_voxelmap.hpp:127-149): the voxels at the search offsets around the query's voxel, in the
// reference's order (1: the voxel itself; 7: centre, +x, +y, +z, -x, -y, -z; 27: centre, then the 3 x 3 x 3 cube in i, j, k order), every stored
// point of each (a Gaussian voxel: its mean, gaussian_voxelmap.hpp:84-86; a flat container: its points in insertion order,
// flat_container.hpp:98-107) pushed into a k-best list with KnnResult::push (knn_result.hpp:80-100: sorted ascending, a candidate as far
// as the current worst is dropped, equal distances keep their push order).  Index = the reference's global index
// (voxel_id << 32) | point_id (incremental_voxelmap.hpp:152).  One thread per query; the list lives in the output arrays.  Distances
// in double from the fp32 data the map holds.  Not on the registration path (that searches inside the factor kernel, k = 1).
template <bool FLAT>
__global__ void voxel_knn_kernel(
  const FlatView v, const float4* __restrict__ pts, const float* __restrict__ queries, size_t m, int k, float max_sq, long long* __restrict__ out_idx, float* __restrict__ out_d2) {
  const size_t qi = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (qi >= m) return;
  const float qx = queries[3 * qi], qy = queries[3 * qi + 1], qz = queries[3 * qi + 2];
  long long* idx = out_idx + qi * static_cast<size_t>(k);
  float* d2 = out_d2 + qi * static_cast<size_t>(k);
  for (int j = 0; j < k; j++) {
    idx[j] = -1;
    d2[j] = INFINITY;
  }
  const int cx = fast_floor_d((static_cast<double>(qx) + v.org[0]) * v.inv_leaf);
  const int cy = fast_floor_d((static_cast<double>(qy) + v.org[1]) * v.inv_leaf);
  const int cz = fast_floor_d((static_cast<double>(qz) + v.org[2]) * v.inv_leaf);
  int found = 0;
  auto push = [&](long long index, float d) {
    if (d > max_sq || d >= d2[k - 1]) return;
    int loc = min(found, k - 1);
    for (; loc > 0 && d < d2[loc - 1]; loc--) {
      idx[loc] = idx[loc - 1];
      d2[loc] = d2[loc - 1];
    }
    idx[loc] = index;
    d2[loc] = d;
    found = min(found + 1, k);
  };
  auto scan = [&](int ox, int oy, int oz) {
    const int vox = flat_voxel_at(v, cx + ox, cy + oy, cz + oz);
    if (vox < 0) return;
    const uint32_t n = FLAT ? v.vnum[vox] : 1u;
    for (uint32_t i = 0; i < n; i++) {
      const float4 p = FLAT ? pts[static_cast<size_t>(vox) * kFlatCap + i] : pts[vox];
      const double dx = static_cast<double>(p.x) - qx, dy = static_cast<double>(p.y) - qy, dz = static_cast<double>(p.z) - qz;
      push((static_cast<long long>(vox) << 32) | static_cast<long long>(i), static_cast<float>(dx * dx + dy * dy + dz * dz));
    }
  };
  scan(0, 0, 0);
  if (v.offsets == 27) {
    // the reference's set_search_offsets(27) APPENDS the cube to the default list {(0, 0, 0)} (incremental_voxelmap.hpp:176-184:
    // emplace_back without a clear): 28 offsets, the query's own voxel twice — so its points appear twice in a k > 1 result.
    // Reproduced as it is: parity is with what the reference returns.
    for (int a = -1; a <= 1; a++)
      for (int b = -1; b <= 1; b++)
        for (int c = -1; c <= 1; c++) scan(a, b, c);
  } else {
    if (v.offsets == 7) {
      scan(1, 0, 0);
      scan(0, 1, 0);
      scan(0, 0, 1);
      scan(-1, 0, 0);
      scan(0, -1, 0);
      scan(0, 0, -1);
    }
  }
}

}  // namespace sga

using namespace sga;

extern "C" {

void sga_factor_params_default(sga_factor_params* p) {
  if (!p) return;
  p->factor_kind = SGA_GICP;
  p->robust_kind = SGA_ROBUST_NONE;
  p->robust_c = 1.0;
  p->max_dist_sq = 1.0;
  p->math_mode = SGA_MATH_FP32;
}

// the per-point factor state of a problem over n source points: correspondences and certificates start as "none"
__global__ void problem_state_init_kernel(int* __restrict__ corr, int* __restrict__ hint, int* __restrict__ hint2, uint32_t* __restrict__ walked, size_t n) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i < n) corr[i] = hint[i] = hint2[i] = -1;
  if (i < n / 64 + 1) walked[i] = 0u;
}

static int problem_alloc_state(sga_context* ctx, sga_problem* pb, size_t n, bool has_covs, bool own_arrays = true) {
  SGA_TRY(pb->partials.alloc(problem_partials_doubles(n)));
  SGA_TRY(pb->walked.alloc(n / 64 + 1));
  if (n > 0 && own_arrays) {
    SGA_TRY(pb->pts.alloc(n));
    if (has_covs) SGA_TRY(pb->cov.alloc(n));
  }
  if (n > 0) {
    SGA_TRY(pb->corr.alloc(n));
    SGA_TRY(pb->hint.alloc(n));
    SGA_TRY(pb->hint2.alloc(n));
    SGA_TRY(pb->rex.alloc(n));
    SGA_TRY(pb->maha.alloc(n * 6));  // only ever read where corr >= 0, i.e. after a pass has written it: no fill
  }
  // correspondences and certificates start as "none": one launch (four fills cost four launches, which is what a 15k-point scan pays for)
  hipLaunchKernelGGL(problem_state_init_kernel, dim3((n + 255) / 256 + 1), dim3(256), 0, ctx->stream, pb->corr.p, pb->hint.p, pb->hint2.p, pb->walked.p, n);
  SGA_HIP(hipGetLastError());
  pb->state_fresh = true;
  return SGA_OK;
}

// The source given by ITS OWN kd-tree index (a scan that has just been indexed for its covariances and as the next target — the
// odometry loop, odometry_benchmark_small_gicp_omp.cpp:22-38): the index's kd order is spatially coherent, so the problem takes the
// kd-ordered points and covariances as they are — no sort keys, no sort, no gather, no bounding-box pass (the index knows its box).
int sga_problem_create_from_index(sga_context* ctx, const sga_index* target, const sga_index* source, const double init_T[16], sga_problem** out) {
  if (!ctx || !target || !source || !out) return fail(SGA_ERR_INVALID, "null argument");
  if (target->device != ctx->device || source->device != ctx->device) return fail(SGA_ERR_INVALID, "target/source live on another device");
  if (source->kind != SGA_INDEX_KDTREE) return fail(SGA_ERR_INVALID, "the source index must be a kd-tree");
  (void)init_T;  // the kd order does not depend on the initial guess
  *out = nullptr;
  SGA_ENTER(ctx);
  SGA_TRY(wait_ready(ctx, target->ready));  // inputs produced on another context in stream-ordered mode (common.hpp: Ready)
  SGA_TRY(wait_ready(ctx, source->ready));
  std::unique_ptr<sga_problem> pb(new sga_problem);
  pb->device = ctx->device;
  pb->target = target;
  pb->n = source->n;
  pb->has_normals = source->has_normals;
  pb->has_covs = source->has_covs;
  const size_t n = source->n;
  SGA_TRY(problem_alloc_state(ctx, pb.get(), n, source->has_covs, /*own_arrays=*/false));
  if (n > 0) {
    for (int k = 0; k < 3; k++) pb->src_origin[k] = source->origin[k];
    pb->pts_view = source->kd_pts.p;  // borrowed: the source index outlives the problem
    pb->cov_view = source->has_covs ? source->cov.p : nullptr;
    for (int k = 0; k < 3; k++) {
      pb->bbox_lo[k] = source->bbox_lo[k];
      pb->bbox_hi[k] = source->bbox_hi[k];
    }
  }
  *out = pb.release();
  return SGA_OK;
}

int sga_problem_create(sga_context* ctx, const sga_index* target, const sga_cloud* source, const double init_T[16], sga_problem** out) {
  if (!ctx || !target || !source || !out) return fail(SGA_ERR_INVALID, "null argument");
  if (target->device != ctx->device || source->device != ctx->device) return fail(SGA_ERR_INVALID, "target/source live on another device");
  *out = nullptr;
  SGA_ENTER(ctx);
  SGA_TRY(wait_ready(ctx, target->ready));  // inputs produced on another context in stream-ordered mode (common.hpp: Ready)
  SGA_TRY(wait_ready(ctx, source->ready));
  static const double I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  double T[16];
  pose_to_device(init_T ? init_T : I16, source->origin, target->origin, T);  // the sort keys are computed from device-frame records (common.hpp)
  std::unique_ptr<sga_problem> pb(new sga_problem);
  pb->device = ctx->device;
  pb->target = target;
  pb->n = source->n;
  for (int k = 0; k < 3; k++) pb->src_origin[k] = source->origin[k];
  pb->has_normals = source->has_normals;
  pb->has_covs = source->has_covs;
  const size_t n = source->n;
  SGA_TRY(problem_alloc_state(ctx, pb.get(), n, source->has_covs));
  if (n > 0) {
    DevBuf<unsigned long long> keys, keys_sorted;
    DevBuf<uint32_t> vals, order;
    SGA_TRY(keys.alloc(n));
    SGA_TRY(keys_sorted.alloc(n));
    SGA_TRY(vals.alloc(n));
    SGA_TRY(order.alloc(n));
    float ox = 0, oy = 0, oz = 0, inv = 1.f;
    if (target->kind == SGA_INDEX_KDTREE) {
      ox = target->bbox_lo[0];
      oy = target->bbox_lo[1];
      oz = target->bbox_lo[2];
      const float ext = fmaxf(fmaxf(target->bbox_hi[0] - ox, target->bbox_hi[1] - oy), fmaxf(target->bbox_hi[2] - oz, 1e-6f));
      inv = 512.f / ext;  // 10-bit Morton cells over the target's extent refine the kd-leaf key
    } else {
      inv = static_cast<float>(4.0 / target->leaf);  // quarter-voxel cells: neighbouring lanes probe the same voxel
    }
    if (target->kind == SGA_INDEX_KDTREE && target->n > 0) {
      KdView kv = make_kd_view(target);
      hipLaunchKernelGGL(source_kd_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, source->pts.p, n, rigid_from_colmajor<float>(T), kv, ox, oy, oz, inv, keys.p, vals.p);
    } else {
      hipLaunchKernelGGL(source_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, source->pts.p, n, rigid_from_colmajor<float>(T), ox, oy, oz, inv, keys.p, vals.p);
    }
    SGA_HIP(hipGetLastError());
    SGA_TRY(sort_pairs(ctx, keys.p, keys_sorted.p, vals.p, order.p, n, 0, 63));
    hipLaunchKernelGGL(gather_source_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, order.p, n, source->pts.p, source->cov.p, pb->pts.p, pb->cov.p);
    SGA_HIP(hipGetLastError());
    SGA_TRY(cloud_bbox(ctx, source->pts.p, n, pb->bbox_lo, pb->bbox_hi));  // synchronises the stream
    for (int k = 0; k < 3; k++)  // the box bounds the motion between two poses (warm passes): a non-finite point would make that bound meaningless
      if (!std::isfinite(pb->bbox_lo[k]) || !std::isfinite(pb->bbox_hi[k])) return fail(SGA_ERR_INVALID, "source cloud contains non-finite coordinates");
  }
  *out = pb.release();
  return SGA_OK;
}

int sga_problem_destroy(sga_problem* problem) {
  if (problem) {
    (void)hipSetDevice(problem->device);
    delete problem;
  }
  return SGA_OK;
}

int sga_problem_get_pass_stats(sga_context* ctx, const sga_problem* pb, uint64_t* cold_passes, uint64_t* warm_passes, uint64_t* walked_points) {
  if (!ctx || !pb) return fail(SGA_ERR_INVALID, "null argument");
  if (cold_passes) *cold_passes = pb->cold_passes;
  if (warm_passes) *warm_passes = pb->warm_passes;
  if (walked_points) {
    std::vector<uint32_t> w(pb->walked.n);
    SGA_ENTER(ctx);
    SGA_HIP(hipMemcpyAsync(w.data(), pb->walked.p, w.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    SGA_HIP(hipStreamSynchronize(ctx->stream));
    uint64_t sum = 0;
    for (uint32_t v : w) sum += v;
    *walked_points = sum;
  }
  return SGA_OK;
}

int sga_problem_get_grid_stats(const sga_problem* pb, uint64_t out[6]) {
  if (!pb || !out) return fail(SGA_ERR_INVALID, "null argument");
  out[0] = pb->grid_passes;
  out[1] = pb->grid_open_total;
  out[2] = pb->grid_ring_total;
  out[3] = pb->target ? static_cast<uint64_t>(static_cast<double>(pb->target->grid_h) * 1e6) : 0;
  out[4] = out[5] = 0;  // (round 4 / 5 diagnostics of experiments that left the product kernels in round 6)
  return SGA_OK;
}

int sga_problem_get_factors(sga_context* ctx, const sga_problem* pb, int64_t* target_index, float* mahalanobis6) {
  if (!ctx || !pb) return fail(SGA_ERR_INVALID, "null argument");
  const size_t n = pb->n;
  if (n == 0) return SGA_OK;
  SGA_ENTER(ctx);
  DevBuf<long long> d_idx;
  DevBuf<float> d_m;
  if (target_index) SGA_TRY(d_idx.alloc(n));
  if (mahalanobis6) SGA_TRY(d_m.alloc(n * 6));
  if (mahalanobis6) SGA_TRY(problem_ensure_maha(ctx, const_cast<sga_problem*>(pb)));  // written on demand (linearize.hip)
  const float4* tpts = pb->target->kind != SGA_INDEX_KDTREE ? pb->target->pts.p : pb->target->kd_pts.p;
  const int is_flat = pb->target->kind == SGA_INDEX_FLATMAP ? 1 : 0;
  const bool has_maha = pb->lin_factor == SGA_GICP && pb->maha_valid;  // only GICP has a mahalanobis matrix (gicp_factor.hpp:57-60); the cache is never pre-filled: zeros otherwise
  if (has_maha && pb->last_math == SGA_MATH_FP64 && pb->maha64.p != nullptr)  // the last linearize cached its mahalanobis in fp64
    hipLaunchKernelGGL((export_factors_kernel<double>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, pb->src_pts(), pb->corr.p, pb->maha64.p, n, tpts, is_flat, d_idx.p, d_m.p);
  else
    hipLaunchKernelGGL((export_factors_kernel<float>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, pb->src_pts(), pb->corr.p, has_maha ? pb->maha.p : static_cast<const float*>(nullptr), n, tpts, is_flat, d_idx.p, d_m.p);
  SGA_HIP(hipGetLastError());
  if (target_index) SGA_HIP(hipMemcpyAsync(target_index, d_idx.p, n * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
  if (mahalanobis6) SGA_HIP(hipMemcpyAsync(mahalanobis6, d_m.p, n * 6 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  SGA_HIP(hipStreamSynchronize(ctx->stream));
  return SGA_OK;
}

static int index_knn_impl(sga_context* ctx, const sga_index* index, const float* queries, const double* queries64, size_t m, int k, double max_sq_dist, int64_t* idx, float* sq_dist, double* sq_dist64) {
  if (!ctx || !index || (m > 0 && ((!queries && !queries64) || !idx || (!sq_dist && !sq_dist64)))) return fail(SGA_ERR_INVALID, "null argument");
  if (k < 1 || k > 128) return fail(SGA_ERR_INVALID, "k must be in [1,128]");
  if (m == 0) return SGA_OK;
  SGA_ENTER(ctx);
  SGA_TRY(wait_ready(ctx, index->ready));
  std::vector<float> qf;
  std::vector<double> qd;
  const bool framed = !origin_is_zero(index->origin);
  if (!queries || framed) {  // double queries: the search itself runs on their fp32 roundings — in the index's device frame (common.hpp)
    qf.resize(m * 3);
    if (queries64 && framed) qd.resize(m * 3);
    for (size_t i = 0; i < m * 3; i++) {
      const double v = (queries64 ? queries64[i] : static_cast<double>(queries[i])) - index->origin[i % 3];
      qf[i] = static_cast<float>(v);
      if (!qd.empty()) qd[i] = v;
    }
    queries = qf.data();
    if (!qd.empty()) queries64 = qd.data();
  }
  DevBuf<float> d_q, d_d;
  DevBuf<double> d_q64, d_d64;
  DevBuf<long long> d_i;
  SGA_TRY(d_q.alloc(m * 3));
  SGA_TRY(d_d.alloc(m * k));
  SGA_TRY(d_i.alloc(m * k));
  SGA_HIP(hipMemcpyAsync(d_q.p, queries, m * 3 * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  const bool want64 = sq_dist64 != nullptr && queries64 != nullptr && index->kind == SGA_INDEX_KDTREE && index->n > 0;
  if (want64) {
    SGA_TRY(d_q64.alloc(m * 3));
    SGA_TRY(d_d64.alloc(m * k));
    SGA_HIP(hipMemcpyAsync(d_q64.p, queries64, m * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  }
  const float max_sq = max_sq_dist < 0 ? INFINITY : static_cast<float>(max_sq_dist);
  if (index->kind == SGA_INDEX_VOXELMAP || index->kind == SGA_INDEX_FLATMAP) {
    const FlatView v{index->hkeys.p, index->hvals.p, index->hmask, 1.0 / index->leaf, index->vcounts.p, index->search_offsets, {index->origin[0], index->origin[1], index->origin[2]}};
    if (index->n == 0 || index->hkeys.p == nullptr) {  // an empty map: nothing found
      SGA_HIP(hipMemsetAsync(d_i.p, 0xff, m * k * sizeof(long long), ctx->stream));
      std::vector<float> inf(m * k, INFINITY);
      SGA_HIP(hipMemcpyAsync(d_d.p, inf.data(), m * k * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
      SGA_HIP(hipStreamSynchronize(ctx->stream));
    } else if (index->kind == SGA_INDEX_FLATMAP)
      hipLaunchKernelGGL(voxel_knn_kernel<true>, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, v, index->pts.p, d_q.p, m, k, max_sq, d_i.p, d_d.p);
    else
      hipLaunchKernelGGL(voxel_knn_kernel<false>, dim3((m + 255) / 256), dim3(256), 0, ctx->stream, v, index->pts.p, d_q.p, m, k, max_sq, d_i.p, d_d.p);
  } else if (index->n == 0) {
    SGA_HIP(hipMemsetAsync(d_i.p, 0xff, m * k * sizeof(long long), ctx->stream));
    std::vector<float> inf(m * k, INFINITY);
    SGA_HIP(hipMemcpyAsync(d_d.p, inf.data(), m * k * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    SGA_HIP(hipStreamSynchronize(ctx->stream));
  } else {
    if (k > kKnnMaxK) return fail(SGA_ERR_INVALID, "k must be <= %d for a kd-tree (LDS per workgroup)", kKnnMaxK);
    const size_t shmem = (static_cast<size_t>((k + 3) & ~3) * 8 + kKdMaxDepth * 4) * kKnnBlock;
    KdView kv = make_kd_view(index);
    hipLaunchKernelGGL(knn_kernel, dim3((m + kKnnBlock - 1) / kKnnBlock), dim3(kKnnBlock), shmem, ctx->stream, kv, d_q.p, m, k, max_sq, d_i.p, d_d.p, want64 ? d_q64.p : nullptr, want64 ? d_d64.p : nullptr);
  }
  SGA_HIP(hipGetLastError());
  SGA_HIP(hipMemcpyAsync(idx, d_i.p, m * k * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
  std::vector<float> tmp;
  float* dst = sq_dist;
  if (!dst) {
    tmp.resize(m * k);
    dst = tmp.data();
  }
  SGA_HIP(hipMemcpyAsync(dst, d_d.p, m * k * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (want64) SGA_HIP(hipMemcpyAsync(sq_dist64, d_d64.p, m * k * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  SGA_HIP(hipStreamSynchronize(ctx->stream));
  if (sq_dist64 && !want64)
    for (size_t i = 0; i < m * k; i++) sq_dist64[i] = dst[i];
  return SGA_OK;
}

int sga_index_knn(sga_context* ctx, const sga_index* index, const float* queries, size_t m, int k, double max_sq_dist, int64_t* idx, float* sq_dist) {
  return index_knn_impl(ctx, index, queries, nullptr, m, k, max_sq_dist, idx, sq_dist, nullptr);
}

int sga_index_knn_f64(sga_context* ctx, const sga_index* index, const double* queries, size_t m, int k, double max_sq_dist, int64_t* idx, double* sq_dist) {
  return index_knn_impl(ctx, index, nullptr, queries, m, k, max_sq_dist, idx, nullptr, sq_dist);
}

}  // extern "C"
