// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
// C entry points over the UNMODIFIED reference (koide3/small_gicp v1.0.1): its headers under /root/reference/include and its one
// compiled source, src/small_gicp/registration/registration_helper.cpp, are compiled where they lie (nothing is copied into this
// repository) against the home-made Eigen stand-in in eigen_shim/.  Output: oracle/_ref/libsmall_gicp_ref.so (git-ignored).
// Used to validate the CPU restatement in oracle/ (tests/test_oracle_vs_reference.py) and, optionally, as the timed CPU baseline.
#include <chrono>
#include <cstdint>
#include <cstring>

#include <small_gicp/ann/kdtree.hpp>
#include <small_gicp/ann/kdtree_omp.hpp>
#include <small_gicp/ann/gaussian_voxelmap.hpp>
#include <small_gicp/ann/flat_container.hpp>
#include <small_gicp/ann/incremental_voxelmap.hpp>
#include <small_gicp/factors/general_factor.hpp>
#include <small_gicp/factors/gicp_factor.hpp>
#include <small_gicp/factors/icp_factor.hpp>
#include <small_gicp/factors/plane_icp_factor.hpp>
#include <small_gicp/factors/robust_kernel.hpp>
#include <small_gicp/points/point_cloud.hpp>
#include <small_gicp/registration/reduction.hpp>
#include <small_gicp/registration/reduction_omp.hpp>
#include <small_gicp/registration/registration.hpp>
#include <small_gicp/registration/registration_helper.hpp>
#include <small_gicp/util/downsampling.hpp>
#include <small_gicp/util/normal_estimation.hpp>
#include <small_gicp/util/normal_estimation_omp.hpp>

using namespace small_gicp;

namespace {
struct RefCloud {
  std::shared_ptr<PointCloud> cloud;
  std::shared_ptr<KdTree<PointCloud>> tree;
};
Eigen::Isometry3d to_iso(const double* T16) {
  Eigen::Isometry3d T = Eigen::Isometry3d::Identity();
  std::memcpy(T.matrix().data(), T16, sizeof(double) * 16);
  return T;
}
}  // namespace

// One linearization at T with the reference's own factors and reduction (robust: 0 none, 1 Huber, 2 Cauchy with width c)
template <typename Factor>
static void linearize_with(const RefCloud& t, const RefCloud& s, double max_dist_sq, int num_threads, const Eigen::Isometry3d& T, const typename Factor::Setting& fs, double* H36, double* b6, double* e, double* e_again,
                           std::uint64_t* inliers) {
  std::vector<Factor> factors(s.cloud->size(), Factor(fs));
  DistanceRejector rejector;
  rejector.max_dist_sq = max_dist_sq;
  Eigen::Matrix<double, 6, 6> H;
  Eigen::Matrix<double, 6, 1> b;
  double err;
  if (num_threads > 1) {
    ParallelReductionOMP red;
    red.num_threads = num_threads;
    std::tie(H, b, err) = red.linearize(*t.cloud, *s.cloud, *t.tree, rejector, T, factors);
    *e_again = red.error(*t.cloud, *s.cloud, T, factors);
  } else {
    SerialReduction red;
    std::tie(H, b, err) = red.linearize(*t.cloud, *s.cloud, *t.tree, rejector, T, factors);
    *e_again = red.error(*t.cloud, *s.cloud, T, factors);
  }
  for (int i = 0; i < 6; i++) {
    b6[i] = b(i);
    for (int j = 0; j < 6; j++) H36[6 * i + j] = H(i, j);
  }
  *e = err;
  *inliers = std::count_if(factors.begin(), factors.end(), [](const Factor& f) { return f.inlier(); });
}


// (templates live outside the C linkage block)
// Registration<GICPFactor, ParallelReductionOMP, RestrictDoFFactor | NullFactor, DistanceRejector, LM | GN> of the reference
// (registration/registration.hpp:17-54, factors/general_factor.hpp:41-75, optimizer.hpp:24-149).
template <typename General, typename Optimizer>
static RegistrationResult align_general(const RefCloud& t, const RefCloud& s, double max_corr_dist, int num_threads, int max_iterations, double rotation_eps, double translation_eps, const Eigen::Isometry3d& init_T, const General& general) {
  Registration<GICPFactor, ParallelReductionOMP, General, DistanceRejector, Optimizer> reg;
  reg.reduction.num_threads = num_threads;
  reg.rejector.max_dist_sq = max_corr_dist * max_corr_dist;
  reg.criteria.rotation_eps = rotation_eps;
  reg.criteria.translation_eps = translation_eps;
  reg.optimizer.max_iterations = max_iterations;
  reg.general_factor = general;
  return reg.align(*t.cloud, *s.cloud, *t.tree, init_T);
}


// IncrementalVoxelMap::knn_search (ann/incremental_voxelmap.hpp:127-149) for m queries: idx / sqd m*k, unfilled entries -1 / inf;
// the indices are the map's global indices (voxel_id << 32) | point_id
template <typename Map>
static void map_knn(const Map& map, const double* q, size_t m, int k, std::int64_t* idx, double* sqd) {
  std::vector<size_t> ki(k);
  std::vector<double> kd(k);
  for (size_t i = 0; i < m; i++) {
    const Eigen::Vector4d pt(q[3 * i], q[3 * i + 1], q[3 * i + 2], 1.0);
    const size_t n = map.knn_search(pt, static_cast<size_t>(k), ki.data(), kd.data());
    for (int j = 0; j < k; j++) {
      idx[i * k + j] = static_cast<size_t>(j) < n ? static_cast<std::int64_t>(ki[j]) : -1;
      sqd[i * k + j] = static_cast<size_t>(j) < n ? kd[j] : std::numeric_limits<double>::infinity();
    }
  }
}

extern "C" {

struct ref_result {
  double T[16];
  int converged;
  std::uint64_t iterations, num_inliers;
  double H[36];
  double b[6];
  double error;
};

static void fill_result(const RegistrationResult& r, ref_result* out, double seconds, double* elapsed) {
  std::memcpy(out->T, r.T_target_source.matrix().data(), sizeof(double) * 16);
  out->converged = r.converged;
  out->iterations = r.iterations;
  out->num_inliers = r.num_inliers;
  for (int i = 0; i < 6; i++) {
    out->b[i] = r.b(i);
    for (int j = 0; j < 6; j++) out->H[6 * i + j] = r.H(i, j);
  }
  out->error = r.error;
  if (elapsed) *elapsed = seconds;
}

// points n*3 doubles; normals n*3 / covs n*9 (row-major 3x3) optional
void* ref_cloud_create(const double* pts, const double* normals, const double* covs, size_t n, int build_tree, int tree_threads) {
  auto* rc = new RefCloud;
  rc->cloud = std::make_shared<PointCloud>();
  rc->cloud->resize(n);
  for (size_t i = 0; i < n; i++) {
    rc->cloud->point(i) = Eigen::Vector4d(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], 1.0);
    rc->cloud->normal(i) = normals ? Eigen::Vector4d(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2], 0.0) : Eigen::Vector4d::Zero();
    Eigen::Matrix4d c = Eigen::Matrix4d::Zero();
    if (covs)
      for (int r = 0; r < 3; r++)
        for (int k = 0; k < 3; k++) c(r, k) = covs[9 * i + 3 * r + k];
    rc->cloud->cov(i) = c;
  }
  if (build_tree) {
    if (tree_threads > 1)
      rc->tree = std::make_shared<KdTree<PointCloud>>(rc->cloud, KdTreeBuilderOMP(tree_threads));
    else
      rc->tree = std::make_shared<KdTree<PointCloud>>(rc->cloud);
  }
  return rc;
}
void ref_cloud_destroy(void* h) { delete static_cast<RefCloud*>(h); }
size_t ref_cloud_size(void* h) { return static_cast<RefCloud*>(h)->cloud->size(); }
void ref_cloud_get(void* h, double* pts, double* normals, double* covs) {
  const PointCloud& c = *static_cast<RefCloud*>(h)->cloud;
  for (size_t i = 0; i < c.size(); i++) {
    for (int k = 0; k < 3; k++) {
      if (pts) pts[3 * i + k] = c.point(i)(k);
      if (normals) normals[3 * i + k] = c.normal(i)(k);
    }
    if (covs)
      for (int r = 0; r < 3; r++)
        for (int k = 0; k < 3; k++) covs[9 * i + 3 * r + k] = c.cov(i)(r, k);
  }
}

// util/downsampling.hpp voxelgrid_sampling (serial)
void* ref_voxelgrid_sampling(void* h, double leaf) {
  auto* rc = new RefCloud;
  rc->cloud = voxelgrid_sampling(*static_cast<RefCloud*>(h)->cloud, leaf);
  return rc;
}

void ref_estimate_normals_covariances(void* h, int k, int num_threads) {
  auto* rc = static_cast<RefCloud*>(h);
  if (!rc->tree) rc->tree = std::make_shared<KdTree<PointCloud>>(rc->cloud);
  if (num_threads > 1)
    estimate_normals_covariances_omp(*rc->cloud, *rc->tree, k, num_threads);
  else
    estimate_normals_covariances(*rc->cloud, *rc->tree, k);
}

// util/normal_estimation_omp.hpp estimate_covariances_omp (the preprocessing of the odometry benchmark, odometry_benchmark_small_gicp_omp.cpp:24)
void ref_estimate_covariances(void* h, int k, int num_threads) {
  auto* rc = static_cast<RefCloud*>(h);
  if (!rc->tree) rc->tree = std::make_shared<KdTree<PointCloud>>(rc->cloud, KdTreeBuilderOMP(num_threads > 1 ? num_threads : 1));
  if (num_threads > 1)
    estimate_covariances_omp(*rc->cloud, *rc->tree, k, num_threads);
  else
    estimate_covariances(*rc->cloud, *rc->tree, k);
}

// KdTree::nearest_neighbor_search (ann/kdtree.hpp:193-205) for m queries, OpenMP over the queries: idx = -1 where the tree is empty
void ref_nearest(void* h, const double* q, size_t m, int num_threads, std::int64_t* idx, double* sqd) {
  auto* rc = static_cast<RefCloud*>(h);
#pragma omp parallel for num_threads(num_threads) schedule(guided, 64)
  for (std::int64_t i = 0; i < static_cast<std::int64_t>(m); i++) {
    const Eigen::Vector4d pt(q[3 * i], q[3 * i + 1], q[3 * i + 2], 1.0);
    size_t ki = 0;
    double kd = std::numeric_limits<double>::infinity();
    const size_t n = rc->cloud->size() ? rc->tree->nearest_neighbor_search(pt, &ki, &kd) : 0;
    idx[i] = n ? static_cast<std::int64_t>(ki) : -1;
    sqd[i] = n ? kd : std::numeric_limits<double>::infinity();
  }
}

size_t ref_knn(void* h, const double* q, size_t m, int k, std::int64_t* idx, double* sqd) {
  auto* rc = static_cast<RefCloud*>(h);
  size_t total = 0;
  std::vector<size_t> ki(k);
  std::vector<double> kd(k);
  for (size_t i = 0; i < m; i++) {
    const Eigen::Vector4d pt(q[3 * i], q[3 * i + 1], q[3 * i + 2], 1.0);
    const size_t n = rc->cloud->size() ? rc->tree->knn_search(pt, k, ki.data(), kd.data()) : 0;
    for (int j = 0; j < k; j++) {
      idx[i * k + j] = static_cast<size_t>(j) < n ? static_cast<std::int64_t>(ki[j]) : -1;
      sqd[i * k + j] = static_cast<size_t>(j) < n ? kd[j] : std::numeric_limits<double>::infinity();
    }
    total += n;
  }
  return total;
}

// registration_helper.cpp align(PointCloud, PointCloud, KdTree, init_T, setting) / align(GaussianVoxelMap, PointCloud, ...)
// type: 0 ICP, 1 PLANE_ICP, 2 GICP, 3 VGICP
int ref_align(void* target_h, void* source_h, int type, double voxel_resolution, double max_corr_dist, int num_threads, int max_iterations, double rotation_eps, double translation_eps, const double* init_T16, ref_result* out, double* elapsed_sec) {
  auto* t = static_cast<RefCloud*>(target_h);
  auto* s = static_cast<RefCloud*>(source_h);
  RegistrationSetting setting;
  setting.type = static_cast<RegistrationSetting::RegistrationType>(type);
  setting.voxel_resolution = voxel_resolution;
  setting.max_correspondence_distance = max_corr_dist;
  setting.num_threads = num_threads;
  setting.max_iterations = max_iterations;
  setting.rotation_eps = rotation_eps;
  setting.translation_eps = translation_eps;
  const Eigen::Isometry3d init_T = to_iso(init_T16);
  RegistrationResult r;
  std::chrono::steady_clock::time_point t0, t1;
  if (type == 3) {
    auto voxelmap = create_gaussian_voxelmap(*t->cloud, voxel_resolution);
    t0 = std::chrono::steady_clock::now();
    r = align(*voxelmap, *s->cloud, init_T, setting);
    t1 = std::chrono::steady_clock::now();
  } else {
    if (!t->tree) return -1;
    t0 = std::chrono::steady_clock::now();
    r = align(*t->cloud, *s->cloud, *t->tree, init_T, setting);
    t1 = std::chrono::steady_clock::now();
  }
  fill_result(r, out, std::chrono::duration<double>(t1 - t0).count(), elapsed_sec);
  return 0;
}

// optimizer: 0 LM, 1 GN; restrict_lambda > 0: RestrictDoFFactor with that lambda and mask (rx ry rz tx ty tz; 1 = free)
int ref_align_general(void* target_h, void* source_h, int optimizer, double restrict_lambda, const double* mask6, double max_corr_dist, int num_threads, int max_iterations, double rotation_eps, double translation_eps,
                      const double* init_T16, ref_result* out) {
  auto* t = static_cast<RefCloud*>(target_h);
  auto* s = static_cast<RefCloud*>(source_h);
  if (!t->tree) return -1;
  const Eigen::Isometry3d init_T = to_iso(init_T16);
  RegistrationResult r;
  if (restrict_lambda > 0) {
    RestrictDoFFactor g;
    g.lambda = restrict_lambda;
    g.set_rotation_mask(Eigen::Array3d(mask6[0], mask6[1], mask6[2]));
    g.set_translation_mask(Eigen::Array3d(mask6[3], mask6[4], mask6[5]));
    r = optimizer == 1 ? align_general<RestrictDoFFactor, GaussNewtonOptimizer>(*t, *s, max_corr_dist, num_threads, max_iterations, rotation_eps, translation_eps, init_T, g)
                       : align_general<RestrictDoFFactor, LevenbergMarquardtOptimizer>(*t, *s, max_corr_dist, num_threads, max_iterations, rotation_eps, translation_eps, init_T, g);
  } else {
    r = optimizer == 1 ? align_general<NullFactor, GaussNewtonOptimizer>(*t, *s, max_corr_dist, num_threads, max_iterations, rotation_eps, translation_eps, init_T, NullFactor())
                       : align_general<NullFactor, LevenbergMarquardtOptimizer>(*t, *s, max_corr_dist, num_threads, max_iterations, rotation_eps, translation_eps, init_T, NullFactor());
  }
  double el = 0;
  fill_result(r, out, 0.0, &el);
  return 0;
}

int ref_linearize(void* target_h, void* source_h, int type, int robust, double robust_c, double max_dist_sq, int num_threads, const double* T16, double* H36, double* b6, double* e, double* e_again, std::uint64_t* inliers) {
  const auto& t = *static_cast<RefCloud*>(target_h);
  const auto& s = *static_cast<RefCloud*>(source_h);
  if (!t.tree) return -1;
  const Eigen::Isometry3d T = to_iso(T16);
  if (robust == 0) {
    if (type == 0) linearize_with<ICPFactor>(t, s, max_dist_sq, num_threads, T, {}, H36, b6, e, e_again, inliers);
    else if (type == 1) linearize_with<PointToPlaneICPFactor>(t, s, max_dist_sq, num_threads, T, {}, H36, b6, e, e_again, inliers);
    else linearize_with<GICPFactor>(t, s, max_dist_sq, num_threads, T, {}, H36, b6, e, e_again, inliers);
  } else if (robust == 1) {
    RobustFactor<Huber, GICPFactor>::Setting fs;
    fs.robust_kernel.c = robust_c;
    linearize_with<RobustFactor<Huber, GICPFactor>>(t, s, max_dist_sq, num_threads, T, fs, H36, b6, e, e_again, inliers);
  } else {
    RobustFactor<Cauchy, GICPFactor>::Setting fs;
    fs.robust_kernel.c = robust_c;
    linearize_with<RobustFactor<Cauchy, GICPFactor>>(t, s, max_dist_sq, num_threads, T, fs, H36, b6, e, e_again, inliers);
  }
  return 0;
}

size_t ref_voxelmap_size(void* cloud_h, double leaf) { return create_gaussian_voxelmap(*static_cast<RefCloud*>(cloud_h)->cloud, leaf)->size(); }


// ---- incremental GaussianVoxelMap of the reference (ann/incremental_voxelmap.hpp, ann/gaussian_voxelmap.hpp), unmodified ----
struct RefVoxelMap {
  GaussianVoxelMap map;
  explicit RefVoxelMap(double leaf) : map(leaf) {}
};
void* ref_ivm_create(double leaf) { return new RefVoxelMap(leaf); }
void ref_ivm_destroy(void* h) { delete static_cast<RefVoxelMap*>(h); }
void ref_ivm_set_lru(void* h, size_t horizon, size_t clear_cycle) {
  static_cast<RefVoxelMap*>(h)->map.lru_horizon = horizon;
  static_cast<RefVoxelMap*>(h)->map.lru_clear_cycle = clear_cycle;
}
void ref_ivm_insert(void* h, void* cloud_h, const double* T16) {
  static_cast<RefVoxelMap*>(h)->map.insert(*static_cast<RefCloud*>(cloud_h)->cloud, T16 ? to_iso(T16) : Eigen::Isometry3d::Identity());
}
size_t ref_ivm_size(void* h) { return static_cast<RefVoxelMap*>(h)->map.size(); }
void ref_ivm_get(void* h, int* coords, double* means, double* covs, std::uint64_t* counts) {
  const auto& fv = static_cast<RefVoxelMap*>(h)->map.flat_voxels;
  for (size_t i = 0; i < fv.size(); i++) {
    for (int k = 0; k < 3; k++) {
      coords[3 * i + k] = fv[i]->first.coord[k];
      means[3 * i + k] = fv[i]->second.mean[k];
      for (int c = 0; c < 3; c++) covs[9 * i + 3 * k + c] = fv[i]->second.cov(k, c);
    }
    counts[i] = fv[i]->second.num_points;
  }
}


// ---- IncrementalVoxelMap<FlatContainerCov> of the reference (ann/flat_container.hpp, ann/incremental_voxelmap.hpp), unmodified ----
struct RefFlatMap {
  IncrementalVoxelMap<FlatContainerCov> map;
  explicit RefFlatMap(double leaf) : map(leaf) {}
};
void* ref_fvm_create(double leaf) { return new RefFlatMap(leaf); }
void ref_fvm_destroy(void* h) { delete static_cast<RefFlatMap*>(h); }
void ref_fvm_set_lru(void* h, size_t horizon, size_t clear_cycle) {
  static_cast<RefFlatMap*>(h)->map.lru_horizon = horizon;
  static_cast<RefFlatMap*>(h)->map.lru_clear_cycle = clear_cycle;
}
void ref_fvm_set_setting(void* h, double min_sq_dist_in_cell, size_t max_num_points_in_cell) {
  static_cast<RefFlatMap*>(h)->map.voxel_setting.min_sq_dist_in_cell = min_sq_dist_in_cell;
  static_cast<RefFlatMap*>(h)->map.voxel_setting.max_num_points_in_cell = max_num_points_in_cell;
}
void ref_fvm_set_search_offsets(void* h, int n) { static_cast<RefFlatMap*>(h)->map.set_search_offsets(n); }
void ref_fvm_insert(void* h, void* cloud_h, const double* T16) {
  static_cast<RefFlatMap*>(h)->map.insert(*static_cast<RefCloud*>(cloud_h)->cloud, T16 ? to_iso(T16) : Eigen::Isometry3d::Identity());
}
size_t ref_fvm_size(void* h) { return static_cast<RefFlatMap*>(h)->map.size(); }
void ref_fvm_knn(void* h, const double* q, size_t m, int k, std::int64_t* idx, double* sqd) { map_knn(static_cast<RefFlatMap*>(h)->map, q, m, k, idx, sqd); }
void ref_ivm_knn(void* h, const double* q, size_t m, int k, std::int64_t* idx, double* sqd) { map_knn(static_cast<RefVoxelMap*>(h)->map, q, m, k, idx, sqd); }
size_t ref_fvm_total_points(void* h) {
  size_t n = 0;
  for (const auto& v : static_cast<RefFlatMap*>(h)->map.flat_voxels) n += v->second.size();
  return n;
}
void ref_fvm_get(void* h, int* coords, std::uint64_t* counts, double* points, double* covs) {
  const auto& fv = static_cast<RefFlatMap*>(h)->map.flat_voxels;
  size_t o = 0;
  for (size_t i = 0; i < fv.size(); i++) {
    for (int k = 0; k < 3; k++) coords[3 * i + k] = fv[i]->first.coord[k];
    counts[i] = fv[i]->second.size();
    for (size_t j = 0; j < fv[i]->second.size(); j++, o++) {
      for (int k = 0; k < 3; k++) {
        points[3 * o + k] = fv[i]->second.points[j][k];
        for (int c = 0; c < 3; c++) covs[9 * o + 3 * k + c] = fv[i]->second.covs[j](k, c);
      }
    }
  }
}
// odometry_benchmark_small_gicp_model_omp.cpp:33-36: Registration<GICPFactor, ParallelReductionOMP>::align(voxelmap, points, voxelmap, T)
int ref_fvm_align(void* h, void* source_h, int num_threads, const double* init_T16, ref_result* out) {
  auto& vm = static_cast<RefFlatMap*>(h)->map;
  auto* s = static_cast<RefCloud*>(source_h);
  Registration<GICPFactor, ParallelReductionOMP> registration;
  registration.reduction.num_threads = num_threads;
  const RegistrationResult r = registration.align(vm, *s->cloud, vm, to_iso(init_T16));
  fill_result(r, out, 0.0, nullptr);
  return 0;
}

}  // extern "C"
