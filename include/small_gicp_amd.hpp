// small_gicp_amd.hpp — header-only C++17 host layer over the C-ABI (small_gicp_amd.h) that mirrors the reference's C++ surface
// for the registration hot path, name for name and default for default (reference tree /root/reference, koide3/small_gicp v1.0.1):
//
//   Registration<PointFactor, Reduction, GeneralFactor, CorrespondenceRejector, Optimizer>::align   registration/registration.hpp:17-54
//   ICPFactor / PointToPlaneICPFactor / GICPFactor / RobustFactor<Huber|Cauchy, F>                   factors/*.hpp
//   DistanceRejector / NullRejector, TerminationCriteria, NullFactor / RestrictDoFFactor             registration/rejector.hpp, termination_criteria.hpp, factors/general_factor.hpp
//   LevenbergMarquardtOptimizer / GaussNewtonOptimizer (fields)                                      registration/optimizer.hpp:24-149
//   RegistrationResult, RegistrationSetting, preprocess_points, create_gaussian_voxelmap, align x3    registration_result.hpp, registration_helper.hpp:37-90
//   PointCloud, KdTree, GaussianVoxelMap, voxelgrid_sampling, estimate_{normals,covariances,...}      points/point_cloud.hpp, ann/*.hpp, util/*.hpp
//
// The one new name is the reduction policy `ParallelReductionHIP` (the slot SerialReduction / ParallelReductionOMP / ...TBB fill
// in the reference): with it the whole align() runs on the GPU and only the 6x6 solve stays on the host.
// Eigen is not required (it is absent from this image); fixed-size values are plain column-major double arrays, and thin
// adaptors are enabled when <Eigen/Core> is on the include path.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <type_traits>
#include <string>
#include <utility>
#include <vector>

#include "small_gicp_amd.h"

#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#include <Eigen/Geometry>
#define SMALL_GICP_AMD_HAS_EIGEN 1
#endif

namespace small_gicp_amd {

inline void check(int rc, const char* what) {
  if (rc != SGA_OK) throw std::runtime_error(std::string(what) + ": " + sga_last_error());
}

/// 4x4 rigid transform, column-major like Eigen::Isometry3d::matrix().data().
struct Isometry3d {
  std::array<double, 16> m{{1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}};
  static Isometry3d Identity() { return Isometry3d(); }
  double& operator()(int r, int c) { return m[4 * c + r]; }
  double operator()(int r, int c) const { return m[4 * c + r]; }
  const double* data() const { return m.data(); }
  Isometry3d operator*(const Isometry3d& o) const {
    Isometry3d r;
    for (int c = 0; c < 4; c++)
      for (int i = 0; i < 4; i++) {
        double s = 0;
        for (int k = 0; k < 4; k++) s += (*this)(i, k) * o(k, c);
        r(i, c) = s;
      }
    return r;
  }
  Isometry3d inverse() const {
    Isometry3d r;
    for (int i = 0; i < 3; i++) {
      for (int j = 0; j < 3; j++) r(i, j) = (*this)(j, i);
      r(i, 3) = 0;
    }
    for (int i = 0; i < 3; i++)
      for (int k = 0; k < 3; k++) r(i, 3) -= (*this)(k, i) * (*this)(k, 3);
    return r;
  }
#ifdef SMALL_GICP_AMD_HAS_EIGEN
  Isometry3d(const Eigen::Isometry3d& T) { std::memcpy(m.data(), T.matrix().data(), sizeof(double) * 16); }
  operator Eigen::Isometry3d() const {
    Eigen::Isometry3d T;
    std::memcpy(T.matrix().data(), m.data(), sizeof(double) * 16);
    return T;
  }
  Isometry3d() = default;
#endif
};

/// One context (GPU + stream) per device, shared by the objects of this header.
inline sga_context* default_context(int device = 0) {
  static std::array<sga_context*, 16> ctxs{};
  if (device < 0 || device >= 16) throw std::runtime_error("device out of range");
  if (!ctxs[device]) check(sga_context_create(device, &ctxs[device]), "sga_context_create");
  return ctxs[device];
}

/// points/point_cloud.hpp:15-71 — device-resident; host copies on demand (points(i) etc. read a cached download).
struct PointCloud {
  using Ptr = std::shared_ptr<PointCloud>;
  using ConstPtr = std::shared_ptr<const PointCloud>;

  PointCloud() { check(sga_cloud_create_f32(ctx, nullptr, nullptr, nullptr, 0, &h), "sga_cloud_create_f32"); }
  template <typename T, size_t D>
  explicit PointCloud(const std::vector<std::array<T, D>>& pts, int device = 0) : ctx(default_context(device)) {
    static_assert(D == 3 || D == 4, "points must be 3- or 4-vectors");
    // double points may lie kilometres from the origin (points/point_cloud.hpp:69-71): the cloud's origin is subtracted in double before the
    // cast (small_gicp_amd.h: device frames); float points are what they are and go through the library's own recentring
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, origin[3] = {0, 0, 0};
    for (size_t i = 0; i < pts.size(); i++)
      for (int k = 0; k < 3; k++) {
        const double v = static_cast<double>(pts[i][k]);
        if (v - v == 0.0) lo[k] = v < lo[k] ? v : lo[k], hi[k] = v > hi[k] ? v : hi[k];
      }
    sga_choose_origin(lo, hi, origin);
    std::vector<float> xyz(pts.size() * 3);
    for (size_t i = 0; i < pts.size(); i++)
      for (int k = 0; k < 3; k++) xyz[3 * i + k] = static_cast<float>(static_cast<double>(pts[i][k]) - origin[k]);
    check(sga_cloud_create_f32_origin(ctx, xyz.data(), nullptr, nullptr, pts.size(), origin, &h), "sga_cloud_create_f32_origin");
  }
  /// xyz n*3 floats [+ normals n*3] [+ cov6 n*6 (xx,xy,xz,yy,yz,zz)]
  PointCloud(const float* xyz, const float* normals, const float* cov6, size_t n, int device = 0) : ctx(default_context(device)) { check(sga_cloud_create_f32(ctx, xyz, normals, cov6, n, &h), "sga_cloud_create_f32"); }
  explicit PointCloud(sga_cloud* handle, int device = 0) : ctx(default_context(device)), h(handle) {}
  /// On a context of the caller's (sga_context_create: its own HIP stream) instead of the device's default one — the stages of a
  /// pipeline that run side by side take one each (examples/odometry_benchmark_flow.cpp).
  PointCloud(const float* xyz, const float* normals, const float* cov6, size_t n, sga_context* context) : ctx(context) { check(sga_cloud_create_f32(ctx, xyz, normals, cov6, n, &h), "sga_cloud_create_f32"); }
  PointCloud(sga_cloud* handle, sga_context* context) : ctx(context), h(handle) {}
  PointCloud(const PointCloud&) = delete;
  PointCloud& operator=(const PointCloud&) = delete;
  ~PointCloud() { sga_cloud_destroy(h); }

  size_t size() const {
    size_t n = 0;
    sga_cloud_size(h, &n);
    return n;
  }
  bool empty() const { return size() == 0; }
  bool has_normals() const {
    int a = 0, b = 0;
    sga_cloud_has(h, &a, &b);
    return a != 0;
  }
  bool has_covs() const {
    int a = 0, b = 0;
    sga_cloud_has(h, &a, &b);
    return b != 0;
  }
  /// (x, y, z, 1) of point i
  std::array<double, 4> point(size_t i) const {
    sync_host();
    return {host_xyz[3 * i], host_xyz[3 * i + 1], host_xyz[3 * i + 2], 1.0};
  }
  std::array<double, 4> normal(size_t i) const {
    sync_host();
    if (host_nrm.empty()) return {0, 0, 0, 0};
    return {host_nrm[3 * i], host_nrm[3 * i + 1], host_nrm[3 * i + 2], 0.0};
  }
  /// 4x4 covariance (3x3 block + zero padding), column-major
  std::array<double, 16> cov(size_t i) const {
    sync_host();
    std::array<double, 16> c{};
    if (host_cov.empty()) return c;
    const float* s = &host_cov[6 * i];
    c[0] = s[0], c[1] = s[1], c[2] = s[2], c[4] = s[1], c[5] = s[3], c[6] = s[4], c[8] = s[2], c[9] = s[4], c[10] = s[5];
    return c;
  }
  void invalidate_host() const { host_valid = false; }

  sga_context* ctx = default_context();
  sga_cloud* h = nullptr;

private:
  void sync_host() const {
    if (host_valid) return;
    const size_t n = size();
    host_xyz.assign(3 * n, 0.0);
    host_nrm.assign(has_normals() ? 3 * n : 0, 0.f);
    host_cov.assign(has_covs() ? 6 * n : 0, 0.f);
    check(sga_cloud_download_f64(ctx, h, host_xyz.data(), host_nrm.empty() ? nullptr : host_nrm.data(), host_cov.empty() ? nullptr : host_cov.data()), "sga_cloud_download_f64");  // device record + origin, in double
    host_valid = true;
  }
  mutable bool host_valid = false;
  mutable std::vector<double> host_xyz;
  mutable std::vector<float> host_nrm, host_cov;
};

/// ann/kdtree.hpp:248-291 KdTree<PointCloud>: exact nearest-neighbour index over a cloud (GPU kd-tree).
struct KdTree {
  using Ptr = std::shared_ptr<KdTree>;
  explicit KdTree(std::shared_ptr<const PointCloud> pts) : points(std::move(pts)) { check(sga_index_build_kdtree(points->ctx, points->h, &h), "sga_index_build_kdtree"); }
  KdTree(const KdTree&) = delete;
  KdTree& operator=(const KdTree&) = delete;
  ~KdTree() { sga_index_destroy(h); }
  /// traits::knn_search (ann/traits.hpp:22-25) for one query (x, y, z[, 1]); returns the number of neighbours found
  size_t knn_search(const double* pt, size_t k, size_t* k_indices, double* k_sq_dists) const {
    const float q[3] = {static_cast<float>(pt[0]), static_cast<float>(pt[1]), static_cast<float>(pt[2])};
    std::vector<int64_t> idx(k);
    std::vector<float> d2(k);
    check(sga_index_knn(points->ctx, h, q, 1, static_cast<int>(k), -1.0, idx.data(), d2.data()), "sga_index_knn");
    size_t found = 0;
    for (size_t j = 0; j < k; j++)
      if (idx[j] >= 0) {
        k_indices[found] = static_cast<size_t>(idx[j]);
        k_sq_dists[found] = d2[j];
        found++;
      }
    return found;
  }
  size_t nearest_neighbor_search(const double* pt, size_t* k_index, double* k_sq_dist) const { return knn_search(pt, 1, k_index, k_sq_dist); }

  std::shared_ptr<const PointCloud> points;
  sga_index* h = nullptr;
};

/// traits::knn_search / nearest_neighbor_search of a voxel map (ann/incremental_voxelmap.hpp:99-149) for one query: global indices
/// (voxel_id << 32) | point_id, squared distances ascending; returns the number found
inline size_t voxelmap_knn_search(sga_context* ctx, const sga_index* h, const double* pt, size_t k, size_t* k_indices, double* k_sq_dists) {
  if (!h) return 0;
  std::vector<int64_t> idx(k);
  std::vector<double> d2(k);
  check(sga_index_knn_f64(ctx, h, pt, 1, static_cast<int>(k), -1.0, idx.data(), d2.data()), "sga_index_knn_f64");
  size_t found = 0;
  for (size_t j = 0; j < k; j++)
    if (idx[j] >= 0) {
      k_indices[found] = static_cast<size_t>(idx[j]);
      k_sq_dists[found] = d2[j];
      found++;
    }
  return found;
}

/// ann/gaussian_voxelmap.hpp + incremental_voxelmap.hpp: one-shot Gaussian voxel map (VGICP target).
struct GaussianVoxelMap {
  using Ptr = std::shared_ptr<GaussianVoxelMap>;
  explicit GaussianVoxelMap(double leaf_size) : leaf(leaf_size) {}
  GaussianVoxelMap(const GaussianVoxelMap&) = delete;
  GaussianVoxelMap& operator=(const GaussianVoxelMap&) = delete;
  ~GaussianVoxelMap() { sga_index_destroy(h); }
  // ann/incremental_voxelmap.hpp:55-92: any number of inserts, each with a pose; LRU removal of voxels not touched recently
  void insert(const PointCloud& points, const Isometry3d& T = Isometry3d::Identity()) {
    if (!h) {
      ctx = points.ctx;
      check(sga_voxelmap_create(ctx, leaf, &h), "sga_voxelmap_create");
      check(sga_voxelmap_set_lru(h, static_cast<uint32_t>(lru_horizon), static_cast<uint32_t>(lru_clear_cycle)), "sga_voxelmap_set_lru");
    }
    check(sga_voxelmap_insert(ctx, h, points.h, T.data()), "sga_voxelmap_insert");
  }
  size_t lru_horizon = 100, lru_clear_cycle = 10;  // set before the first insert (incremental_voxelmap.hpp:46)
  size_t size() const {
    size_t n = 0;
    if (h) sga_index_size(h, &n);
    return n;
  }
  size_t knn_search(const double* pt, size_t k, size_t* k_indices, double* k_sq_dists) const { return voxelmap_knn_search(ctx, h, pt, k, k_indices, k_sq_dists); }
  size_t nearest_neighbor_search(const double* pt, size_t* k_index, double* k_sq_dist) const { return knn_search(pt, 1, k_index, k_sq_dist); }
  static size_t voxel_id(size_t i) { return i >> 32; }          // incremental_voxelmap.hpp:153-154
  static size_t point_id(size_t i) { return i & 0xffffffffull; }
  double leaf;
  sga_context* ctx = nullptr;
  sga_index* h = nullptr;
};

// ann/flat_container.hpp + ann/incremental_voxelmap.hpp: IncrementalVoxelMap<FlatContainerCov>, the scan-to-model GICP target
struct FlatContainerCov {
  struct Setting {
    double min_sq_dist_in_cell = 0.1 * 0.1;
    size_t max_num_points_in_cell = 10;
  };
};
template <typename VoxelContents>
struct IncrementalVoxelMap;
template <>
struct IncrementalVoxelMap<FlatContainerCov> {
  using Ptr = std::shared_ptr<IncrementalVoxelMap>;
  explicit IncrementalVoxelMap(double leaf_size) : leaf(leaf_size) {}
  IncrementalVoxelMap(const IncrementalVoxelMap&) = delete;
  IncrementalVoxelMap& operator=(const IncrementalVoxelMap&) = delete;
  ~IncrementalVoxelMap() { sga_index_destroy(h); }
  void insert(const PointCloud& points, const Isometry3d& T = Isometry3d::Identity()) {
    if (!h) {
      ctx = points.ctx;
      check(sga_flatmap_create(ctx, leaf, &h), "sga_flatmap_create");
      check(sga_flatmap_set_setting(h, voxel_setting.min_sq_dist_in_cell, static_cast<uint32_t>(voxel_setting.max_num_points_in_cell)), "sga_flatmap_set_setting");
      check(sga_voxelmap_set_lru(h, static_cast<uint32_t>(lru_horizon), static_cast<uint32_t>(lru_clear_cycle)), "sga_voxelmap_set_lru");
      check(sga_voxelmap_set_search_offsets(h, num_search_offsets), "sga_voxelmap_set_search_offsets");
    }
    check(sga_voxelmap_insert(ctx, h, points.h, T.data()), "sga_voxelmap_insert");
  }
  void set_search_offsets(int num_offsets) {
    num_search_offsets = num_offsets;
    if (h) check(sga_voxelmap_set_search_offsets(h, num_offsets), "sga_voxelmap_set_search_offsets");
  }
  size_t size() const {
    size_t n = 0;
    if (h) sga_index_size(h, &n);
    return n;
  }
  size_t knn_search(const double* pt, size_t k, size_t* k_indices, double* k_sq_dists) const { return voxelmap_knn_search(ctx, h, pt, k, k_indices, k_sq_dists); }
  size_t nearest_neighbor_search(const double* pt, size_t* k_index, double* k_sq_dist) const { return knn_search(pt, 1, k_index, k_sq_dist); }
  static size_t voxel_id(size_t i) { return i >> 32; }          // incremental_voxelmap.hpp:153-154
  static size_t point_id(size_t i) { return i & 0xffffffffull; }
  double leaf;
  size_t lru_horizon = 100, lru_clear_cycle = 10;  // set before the first insert
  FlatContainerCov::Setting voxel_setting;          // set before the first insert
  int num_search_offsets = 1;
  sga_context* ctx = nullptr;
  sga_index* h = nullptr;
};

// ---- util/downsampling.hpp, util/normal_estimation.hpp ------------------------------------------------------------------------
inline PointCloud::Ptr voxelgrid_sampling(const PointCloud& points, double leaf_size) {
  sga_cloud* out = nullptr;
  check(sga_voxelgrid_sampling(points.ctx, points.h, leaf_size, &out), "sga_voxelgrid_sampling");
  return std::make_shared<PointCloud>(out, points.ctx);
}
inline void estimate_normals(PointCloud& cloud, int num_neighbors = 20) {
  check(sga_estimate_normals_covariances(cloud.ctx, cloud.h, nullptr, num_neighbors, 1), "estimate_normals");
  cloud.invalidate_host();
}
inline void estimate_covariances(PointCloud& cloud, int num_neighbors = 20) {
  check(sga_estimate_normals_covariances(cloud.ctx, cloud.h, nullptr, num_neighbors, 2), "estimate_covariances");
  cloud.invalidate_host();
}
inline void estimate_normals_covariances(PointCloud& cloud, int num_neighbors = 20) {
  check(sga_estimate_normals_covariances(cloud.ctx, cloud.h, nullptr, num_neighbors, 3), "estimate_normals_covariances");
  cloud.invalidate_host();
}
inline void estimate_normals_covariances(PointCloud& cloud, KdTree& tree, int num_neighbors = 20) {
  check(sga_estimate_normals_covariances(cloud.ctx, cloud.h, tree.h, num_neighbors, 3), "estimate_normals_covariances");
  cloud.invalidate_host();
}
/// util/normal_estimation_omp.hpp estimate_covariances_omp(points, tree, k, threads): with the cloud's own tree (which then also holds
/// the covariances in its kd order: ready to be a registration target, or a source by its index)
inline void estimate_covariances(PointCloud& cloud, KdTree& tree, int num_neighbors = 20) {
  check(sga_estimate_normals_covariances(cloud.ctx, cloud.h, tree.h, num_neighbors, 2), "estimate_covariances");
  cloud.invalidate_host();
}
inline void estimate_normals(PointCloud& cloud, KdTree& tree, int num_neighbors = 20) {
  check(sga_estimate_normals_covariances(cloud.ctx, cloud.h, tree.h, num_neighbors, 1), "estimate_normals");
  cloud.invalidate_host();
}

// ---- factors (factors/*.hpp): tag types, the per-point state lives on the device ----------------------------------------------------
struct ICPFactor {
  struct Setting {};
  static constexpr int kind = SGA_ICP;
  static void fill(const Setting&, sga_factor_params&) {}
};
struct PointToPlaneICPFactor {
  struct Setting {};
  static constexpr int kind = SGA_PLANE_ICP;
  static void fill(const Setting&, sga_factor_params&) {}
};
struct GICPFactor {
  struct Setting {};
  static constexpr int kind = SGA_GICP;
  static void fill(const Setting&, sga_factor_params&) {}
};
struct Huber {
  struct Setting {
    double c = 1.0;
  };
  static constexpr int kind = SGA_ROBUST_HUBER;
};
struct Cauchy {
  struct Setting {
    double c = 1.0;
  };
  static constexpr int kind = SGA_ROBUST_CAUCHY;
};
template <typename Kernel, typename Factor>
struct RobustFactor {
  struct Setting {
    typename Kernel::Setting robust_kernel;
    typename Factor::Setting factor;
  };
  static constexpr int kind = Factor::kind;
  static void fill(const Setting& s, sga_factor_params& p) {
    p.robust_kind = Kernel::kind;
    p.robust_c = s.robust_kernel.c;
    Factor::fill(s.factor, p);
  }
};

// ---- registration/rejector.hpp, termination_criteria.hpp, factors/general_factor.hpp, optimizer.hpp ---------------------------------
struct NullRejector {
  double max_dist_sq_or_negative() const { return -1.0; }
};
struct DistanceRejector {
  double max_dist_sq = 1.0;
  double max_dist_sq_or_negative() const { return max_dist_sq; }
};
struct TerminationCriteria {
  double translation_eps = 1e-3;
  double rotation_eps = 0.1 * M_PI / 180.0;
};
struct NullFactor {
  void fill(sga_registration_setting&) const {}
};
struct RestrictDoFFactor {
  double lambda = 1e9;
  std::array<double, 6> mask{{1, 1, 1, 1, 1, 1}};  // (rx, ry, rz, tx, ty, tz): 1 = active, 0 = inactive
  void set_rotation_mask(double x, double y, double z) { mask[0] = x, mask[1] = y, mask[2] = z; }
  void set_translation_mask(double x, double y, double z) { mask[3] = x, mask[4] = y, mask[5] = z; }
  void fill(sga_registration_setting& s) const {
    s.restrict_dof_lambda = lambda;
    for (int i = 0; i < 6; i++) s.restrict_dof_mask[i] = mask[i];
  }
};
struct LevenbergMarquardtOptimizer {
  bool verbose = false;
  int max_iterations = 20;
  int max_inner_iterations = 10;
  double init_lambda = 1e-3;
  double lambda_factor = 10.0;
  void fill(sga_registration_setting& s) const {
    s.optimizer = SGA_LEVENBERG_MARQUARDT;
    s.verbose = verbose;
    s.max_iterations = max_iterations;
    s.max_inner_iterations = max_inner_iterations;
    s.init_lambda = init_lambda;
    s.lambda_factor = lambda_factor;
  }
};
struct GaussNewtonOptimizer {
  bool verbose = false;
  int max_iterations = 20;
  double lambda = 1e-6;
  void fill(sga_registration_setting& s) const {
    s.optimizer = SGA_GAUSS_NEWTON;
    s.verbose = verbose;
    s.max_iterations = max_iterations;
    s.gn_lambda = lambda;
  }
};

/// The reduction policy of this library: the slot of SerialReduction / ParallelReductionOMP / ParallelReductionTBB
/// (registration/reduction*.hpp).  linearize / error run as HIP kernels on `device`.
struct ParallelReductionHIP {
  int device = 0;
  int math_mode = SGA_MATH_FP32;
  sga_context* context = nullptr;  // the context (HIP stream) the passes run on; null: the source's.  Registrations on different contexts run side by side
};

/// registration/registration_result.hpp:11-30
struct RegistrationResult {
  explicit RegistrationResult(const Isometry3d& T = Isometry3d::Identity()) : T_target_source(T) {}
  Isometry3d T_target_source;
  bool converged = false;
  size_t iterations = 0;
  size_t num_inliers = 0;
  std::array<double, 36> H{};  // 6x6, symmetric
  std::array<double, 6> b{};
  double error = 0.0;
};

inline RegistrationResult to_result(const sga_result& r) {
  RegistrationResult out;
  std::memcpy(out.T_target_source.m.data(), r.T_target_source, sizeof(double) * 16);
  out.converged = r.converged != 0;
  out.iterations = r.iterations;
  out.num_inliers = r.num_inliers;
  std::memcpy(out.H.data(), r.H, sizeof(double) * 36);
  std::memcpy(out.b.data(), r.b, sizeof(double) * 6);
  out.error = r.error;
  return out;
}

/// registration/registration.hpp:17-54
template <typename PointFactor, typename Reduction = ParallelReductionHIP, typename GeneralFactor = NullFactor, typename CorrespondenceRejector = DistanceRejector, typename Optimizer = LevenbergMarquardtOptimizer>
struct Registration {
  using PointFactorSetting = typename PointFactor::Setting;

  sga_registration_setting make_setting() const {
    sga_registration_setting s;
    sga_registration_setting_default(&s);
    s.factor.factor_kind = PointFactor::kind;
    s.factor.max_dist_sq = rejector.max_dist_sq_or_negative();
    s.factor.math_mode = reduction.math_mode;
    PointFactor::fill(point_factor, s.factor);
    s.translation_eps = criteria.translation_eps;
    s.rotation_eps = criteria.rotation_eps;
    optimizer.fill(s);
    general_factor.fill(s);
    return s;
  }

  /// target_tree: KdTree (ICP / PLANE_ICP / GICP).  `target` is accepted for signature parity; the tree holds the target's data.
  RegistrationResult align(const PointCloud& target, const PointCloud& source, const KdTree& target_tree, const Isometry3d& init_T = Isometry3d::Identity()) const {
    (void)target;
    check(sga_index_refresh_attributes(source.ctx, target_tree.h, target_tree.points->h), "sga_index_refresh_attributes");
    return run(target_tree.h, source, init_T);
  }
  /// The same with the SOURCE given by its own KdTree (an addition to the reference's signature for the odometry loop,
  /// odometry_benchmark_small_gicp_omp.cpp:22-38, where every scan is indexed anyway): sga_problem_create_from_index takes the index's
  /// kd-ordered points and covariances as they are — no spatial sort, no copy.  Both trees must hold the attributes the factor needs
  /// (estimate them with the tree, or build the tree afterwards).
  RegistrationResult align(const PointCloud& target, const KdTree& source_tree, const KdTree& target_tree, const Isometry3d& init_T = Isometry3d::Identity()) const {
    (void)target;
    static_assert(std::is_same<Reduction, ParallelReductionHIP>::value, "this library provides the ParallelReductionHIP reduction only");
    sga_context* ctx = reduction.context ? reduction.context : source_tree.points->ctx;
    const sga_registration_setting s = make_setting();
    sga_problem* pb = nullptr;
    check(sga_problem_create_from_index(ctx, target_tree.h, source_tree.h, init_T.data(), &pb), "sga_problem_create_from_index");
    sga_result r;
    const int rc = sga_align_problem(ctx, pb, init_T.data(), &s, &r);
    sga_problem_destroy(pb);
    check(rc, "sga_align_problem");
    return to_result(r);
  }
  /// VGICP form (registration_helper.cpp:136: the voxel map is both target cloud and search structure)
  RegistrationResult align(const GaussianVoxelMap& target, const PointCloud& source, const GaussianVoxelMap& target_tree, const Isometry3d& init_T = Isometry3d::Identity()) const {
    (void)target_tree;
    return run(target.h, source, init_T);
  }
  /// scan-to-model GICP form (odometry_benchmark_small_gicp_model_omp.cpp:33-36)
  RegistrationResult align(
    const IncrementalVoxelMap<FlatContainerCov>& target, const PointCloud& source, const IncrementalVoxelMap<FlatContainerCov>& target_tree, const Isometry3d& init_T = Isometry3d::Identity()) const {
    (void)target_tree;
    return run(target.h, source, init_T);
  }

  TerminationCriteria criteria;
  CorrespondenceRejector rejector;
  PointFactorSetting point_factor;
  GeneralFactor general_factor;
  Reduction reduction;
  Optimizer optimizer;

private:
  RegistrationResult run(const sga_index* index, const PointCloud& source, const Isometry3d& init_T) const {
    static_assert(std::is_same<Reduction, ParallelReductionHIP>::value, "this library provides the ParallelReductionHIP reduction only");
    const sga_registration_setting s = make_setting();
    sga_result r;
    check(sga_align(reduction.context ? reduction.context : source.ctx, index, source.h, init_T.data(), &s, &r), "sga_align");
    return to_result(r);
  }
};

// ---- registration_helper.hpp:15-90 ------------------------------------------------------------------------------------------------
struct RegistrationSetting {
  enum RegistrationType { ICP, PLANE_ICP, GICP, VGICP };
  RegistrationType type = GICP;
  double voxel_resolution = 1.0;
  double downsampling_resolution = 0.25;
  double max_correspondence_distance = 1.0;
  double rotation_eps = 0.1 * M_PI / 180.0;
  double translation_eps = 1e-3;
  int num_threads = 4;  // kept for signature parity; the GPU path ignores it
  int max_iterations = 20;
  bool verbose = false;
};

/// registration_helper.cpp:22-34: downsample -> kd-tree -> normals + covariances
inline std::pair<PointCloud::Ptr, std::shared_ptr<KdTree>> preprocess_points(const PointCloud& points, double downsampling_resolution, int num_neighbors = 10, int num_threads = 4) {
  (void)num_threads;
  auto down = voxelgrid_sampling(points, downsampling_resolution);
  auto tree = std::make_shared<KdTree>(down);
  estimate_normals_covariances(*down, *tree, num_neighbors);
  return {down, tree};
}
template <typename T, size_t D>
std::pair<PointCloud::Ptr, std::shared_ptr<KdTree>> preprocess_points(const std::vector<std::array<T, D>>& points, double downsampling_resolution, int num_neighbors = 10, int num_threads = 4) {
  return preprocess_points(PointCloud(points), downsampling_resolution, num_neighbors, num_threads);
}
/// registration_helper.cpp:50-54
inline GaussianVoxelMap::Ptr create_gaussian_voxelmap(const PointCloud& points, double voxel_resolution) {
  auto vm = std::make_shared<GaussianVoxelMap>(voxel_resolution);
  vm->insert(points);
  return vm;
}

template <typename Reg>
void copy_setting(Reg& reg, const RegistrationSetting& setting) {
  reg.criteria.rotation_eps = setting.rotation_eps;
  reg.criteria.translation_eps = setting.translation_eps;
  reg.optimizer.max_iterations = setting.max_iterations;
  reg.optimizer.verbose = setting.verbose;
}

/// registration_helper.cpp:81-122
inline RegistrationResult align(const PointCloud& target, const PointCloud& source, const KdTree& target_tree, const Isometry3d& init_T = Isometry3d::Identity(), const RegistrationSetting& setting = RegistrationSetting()) {
  const double md2 = setting.max_correspondence_distance * setting.max_correspondence_distance;
  switch (setting.type) {
    case RegistrationSetting::ICP: {
      Registration<ICPFactor, ParallelReductionHIP> reg;
      reg.rejector.max_dist_sq = md2;
      copy_setting(reg, setting);
      return reg.align(target, source, target_tree, init_T);
    }
    case RegistrationSetting::PLANE_ICP: {
      Registration<PointToPlaneICPFactor, ParallelReductionHIP> reg;
      reg.rejector.max_dist_sq = md2;
      copy_setting(reg, setting);
      return reg.align(target, source, target_tree, init_T);
    }
    case RegistrationSetting::GICP: {
      Registration<GICPFactor, ParallelReductionHIP> reg;
      reg.rejector.max_dist_sq = md2;
      copy_setting(reg, setting);
      return reg.align(target, source, target_tree, init_T);
    }
    default:
      std::fprintf(stderr, "error: use align(const GaussianVoxelMap&, const PointCloud&, ...) for VGICP\n");
      return RegistrationResult(Isometry3d::Identity());
  }
}
/// registration_helper.cpp:125-137 (the rejector stays at its 1.0 m^2 default like the reference)
inline RegistrationResult align(const GaussianVoxelMap& target, const PointCloud& source, const Isometry3d& init_T = Isometry3d::Identity(), const RegistrationSetting& setting = RegistrationSetting()) {
  if (setting.type != RegistrationSetting::VGICP) std::fprintf(stderr, "invalid registration type for GaussianVoxelMap\n");
  Registration<GICPFactor, ParallelReductionHIP> reg;
  copy_setting(reg, setting);
  return reg.align(target, source, target, init_T);
}
/// registration_helper.cpp:57-69: raw points in, everything on the GPU
template <typename T, size_t D>
RegistrationResult align(const std::vector<std::array<T, D>>& target, const std::vector<std::array<T, D>>& source, const Isometry3d& init_T = Isometry3d::Identity(), const RegistrationSetting& setting = RegistrationSetting()) {
  auto [target_points, target_tree] = preprocess_points(PointCloud(target), setting.downsampling_resolution, 10, setting.num_threads);
  auto [source_points, source_tree] = preprocess_points(PointCloud(source), setting.downsampling_resolution, 10, setting.num_threads);
  if (setting.type == RegistrationSetting::VGICP) {
    auto voxelmap = create_gaussian_voxelmap(*target_points, setting.voxel_resolution);
    return align(*voxelmap, *source_points, init_T, setting);
  }
  return align(*target_points, *source_points, *target_tree, init_T, setting);
}

}  // namespace small_gicp_amd
