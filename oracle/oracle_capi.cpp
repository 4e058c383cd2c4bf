// TEST INFRASTRUCTURE — NOT PRODUCT CODE.  C entry points over oracle.hpp so pytest (ctypes) and bench.py's
// cpu_baseline leg can drive the CPU oracle.  See oracle.hpp for the reference citations and the pinning statement.
#include <chrono>
#include <cstdint>

#include "oracle.hpp"

using namespace orc;

namespace {
struct OrcCloud {
  PointCloud cloud;
  KdTree tree;
  bool has_tree = false;
};
struct OrcVoxelMap {
  GaussianVoxelMap map;   // flat == false
  FlatVoxelMap fmap;      // flat == true: IncrementalVoxelMap<FlatContainerCov>
  bool flat = false;
  explicit OrcVoxelMap(double leaf, bool flat_ = false) : map(leaf), fmap(leaf), flat(flat_) {}
};
struct OrcFactors {
  std::vector<Factor> factors;
};

void fill_cloud(PointCloud& c, const double* pts, const double* normals, const double* covs, size_t n) {
  c.resize(n);
  for (size_t i = 0; i < n; i++) {
    c.points[i] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    if (normals) c.normals[i] = {normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]};
    if (covs)
      for (int r = 0; r < 3; r++)
        for (int k = 0; k < 3; k++) c.covs[i](r, k) = covs[9 * i + 3 * r + k];
  }
}
}  // namespace

extern "C" {

struct orc_setting {
  int factor_kind;  // 0 ICP, 1 PLANE_ICP, 2 GICP
  int robust_kind;  // 0 none, 1 Huber, 2 Cauchy
  double robust_c;
  double max_dist_sq;
  int num_threads;
  int optimizer_type;  // 0 LM, 1 GN
  int max_iterations;
  int max_inner_iterations;
  double init_lambda;
  double lambda_factor;
  double gn_lambda;
  double translation_eps;
  double rotation_eps;
  int verbose;
  double restrict_lambda;   // RestrictDoFFactor (general_factor.hpp:41-75); 0 = NullFactor
  double restrict_mask[6];
};

struct orc_result {
  double T[16];  // column-major 4x4
  int converged;
  std::uint64_t iterations;
  std::uint64_t num_inliers;
  double H[36];  // row-major (symmetric)
  double b[6];
  double error;
};

void orc_default_setting(orc_setting* s) {
  s->factor_kind = FACTOR_GICP;
  s->robust_kind = ROBUST_NONE;
  s->robust_c = 1.0;
  s->max_dist_sq = 1.0;
  s->num_threads = 4;
  s->optimizer_type = 0;
  s->max_iterations = 20;
  s->max_inner_iterations = 10;
  s->init_lambda = 1e-3;
  s->lambda_factor = 10.0;
  s->gn_lambda = 1e-6;
  s->translation_eps = 1e-3;
  s->rotation_eps = 0.1 * M_PI / 180.0;
  s->verbose = 0;
  s->restrict_lambda = 0.0;
  for (int k = 0; k < 6; k++) s->restrict_mask[k] = 1.0;
}

int orc_fast_floor(double x) { return fast_floor(x); }

// returns number of output points; out must hold n*3 doubles
size_t orc_voxelgrid_sampling(const double* pts, size_t n, double leaf, double* out) {
  std::vector<Vec3> in(n), res;
  for (size_t i = 0; i < n; i++) in[i] = {pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
  voxelgrid_sampling(in, leaf, res);
  for (size_t i = 0; i < res.size(); i++) {
    out[3 * i] = res[i][0];
    out[3 * i + 1] = res[i][1];
    out[3 * i + 2] = res[i][2];
  }
  return res.size();
}

void* orc_cloud_create(const double* pts, const double* normals, const double* covs, size_t n, int build_tree) {
  auto* c = new OrcCloud;
  fill_cloud(c->cloud, pts, normals, covs, n);
  if (build_tree) {
    c->tree.build(c->cloud.points);
    c->has_tree = true;
  }
  return c;
}
void orc_cloud_destroy(void* h) { delete static_cast<OrcCloud*>(h); }
size_t orc_cloud_size(void* h) { return static_cast<OrcCloud*>(h)->cloud.size(); }
void orc_cloud_get(void* h, double* pts, double* normals, double* covs) {
  const PointCloud& c = static_cast<OrcCloud*>(h)->cloud;
  for (size_t i = 0; i < c.size(); i++) {
    for (int k = 0; k < 3; k++) {
      if (pts) pts[3 * i + k] = c.points[i][k];
      if (normals) normals[3 * i + k] = c.normals[i][k];
    }
    if (covs)
      for (int r = 0; r < 3; r++)
        for (int k = 0; k < 3; k++) covs[9 * i + 3 * r + k] = c.covs[i](r, k);
  }
}

// batch kNN; idx is m*k int64 (-1 = not found), sqd m*k doubles. returns total found.
size_t orc_knn(void* h, const double* q, size_t m, int k, std::int64_t* idx, double* sqd, int num_threads) {
  auto* c = static_cast<OrcCloud*>(h);
  size_t total = 0;
#pragma omp parallel for num_threads(std::max(1, num_threads)) schedule(dynamic, 64) reduction(+ : total)
  for (std::int64_t i = 0; i < static_cast<std::int64_t>(m); i++) {
    std::vector<size_t> ki(k);
    std::vector<double> kd(k);
    const size_t n = c->tree.knn_search(Vec3{q[3 * i], q[3 * i + 1], q[3 * i + 2]}, k, ki.data(), kd.data());
    for (int j = 0; j < k; j++) {
      idx[i * k + j] = (static_cast<size_t>(j) < n) ? static_cast<std::int64_t>(ki[j]) : -1;
      sqd[i * k + j] = (static_cast<size_t>(j) < n) ? kd[j] : std::numeric_limits<double>::infinity();
    }
    total += n;
  }
  return total;
}

void orc_estimate_normals_covariances(void* h, int num_neighbors, int num_threads) {
  auto* c = static_cast<OrcCloud*>(h);
  if (!c->has_tree) {
    c->tree.build(c->cloud.points);
    c->has_tree = true;
  }
  estimate_normals_covariances(c->cloud, c->tree, num_neighbors, std::max(1, num_threads));
}

// eigen-decomposition helpers exposed for the oracle's own unit tests (method 0 = Eigen computeDirect restatement, 1 = Jacobi)
void orc_eigen_sym3(const double* m9, int method, double* eivals3, double* eivecs9) {
  Mat3 m, v;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) m(r, c) = m9[3 * r + c];
  if (method == 0)
    eigen_sym3_direct(m, eivals3, v);
  else
    eigen_sym3_jacobi(m, eivals3, v);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) eivecs9[3 * r + c] = v(r, c);
}

void orc_se3_exp(const double* twist6, double* T16) {
  Vec6 a;
  for (int i = 0; i < 6; i++) a[i] = twist6[i];
  se3_to_colmajor16(se3_exp(a), T16);
}

void orc_ldlt_solve(const double* A36, const double* rhs6, double* x6) {
  Mat6 A;
  Vec6 r;
  for (int i = 0; i < 6; i++) {
    r[i] = rhs6[i];
    for (int j = 0; j < 6; j++) A(i, j) = A36[6 * i + j];
  }
  const Vec6 x = ldlt_solve(A, r);
  for (int i = 0; i < 6; i++) x6[i] = x[i];
}

void* orc_voxelmap_create(void* cloud_h, double leaf) {
  auto* vm = new OrcVoxelMap(leaf);
  vm->map.insert(static_cast<OrcCloud*>(cloud_h)->cloud);
  return vm;
}
// incremental use: an empty map, then insert(cloud, T) any number of times (incremental_voxelmap.hpp:55-92)
void* orc_voxelmap_new(double leaf) { return new OrcVoxelMap(leaf); }
void* orc_flatmap_new(double leaf) { return new OrcVoxelMap(leaf, true); }
void orc_voxelmap_insert(void* h, void* cloud_h, const double* T16) {
  auto* vm = static_cast<OrcVoxelMap*>(h);
  const SE3 T = T16 ? se3_from_colmajor16(T16) : SE3::identity();
  if (vm->flat)
    vm->fmap.insert(static_cast<OrcCloud*>(cloud_h)->cloud, T);
  else
    vm->map.insert(static_cast<OrcCloud*>(cloud_h)->cloud, T);
}
void orc_voxelmap_set_lru(void* h, size_t horizon, size_t clear_cycle) {
  auto* vm = static_cast<OrcVoxelMap*>(h);
  vm->map.lru_horizon = vm->fmap.lru_horizon = horizon;
  vm->map.lru_clear_cycle = vm->fmap.lru_clear_cycle = clear_cycle;
}
void orc_flatmap_set_setting(void* h, double min_sq_dist_in_cell, size_t max_num_points_in_cell) {
  auto& m = static_cast<OrcVoxelMap*>(h)->fmap;
  m.min_sq_dist_in_cell = min_sq_dist_in_cell;
  m.max_num_points_in_cell = max_num_points_in_cell;
}
// per voxel: coords (3 ints), number of points; points / covs packed voxel after voxel (total = sum of counts)
size_t orc_flatmap_total_points(void* h) {
  size_t n = 0;
  for (const auto& v : static_cast<OrcVoxelMap*>(h)->fmap.flat_voxels) n += v.points.size();
  return n;
}
void orc_flatmap_get(void* h, int* coords, std::uint64_t* counts, double* points, double* covs) {
  const auto& fv = static_cast<OrcVoxelMap*>(h)->fmap.flat_voxels;
  size_t o = 0;
  for (size_t i = 0; i < fv.size(); i++) {
    for (int k = 0; k < 3; k++) coords[3 * i + k] = fv[i].coord[k];
    counts[i] = fv[i].points.size();
    for (size_t j = 0; j < fv[i].points.size(); j++, o++) {
      for (int k = 0; k < 3; k++) points[3 * o + k] = fv[i].points[j][k];
      for (int r = 0; r < 3; r++)
        for (int k = 0; k < 3; k++) covs[9 * o + 3 * r + k] = fv[i].covs[j](r, k);
    }
  }
}
void orc_voxelmap_destroy(void* h) { delete static_cast<OrcVoxelMap*>(h); }
size_t orc_voxelmap_size(void* h) {
  auto* vm = static_cast<OrcVoxelMap*>(h);
  return vm->flat ? vm->fmap.size() : vm->map.size();
}
void orc_voxelmap_set_search_offsets(void* h, int n) {
  static_cast<OrcVoxelMap*>(h)->map.num_search_offsets = n;
  static_cast<OrcVoxelMap*>(h)->fmap.set_search_offsets(n);
}
void orc_voxelmap_get(void* h, int* coords, double* means, double* covs, std::uint64_t* counts) {
  const auto& fv = static_cast<OrcVoxelMap*>(h)->map.flat_voxels;
  for (size_t i = 0; i < fv.size(); i++) {
    for (int k = 0; k < 3; k++) {
      if (coords) coords[3 * i + k] = fv[i].coord[k];
      if (means) means[3 * i + k] = fv[i].mean[k];
    }
    if (covs)
      for (int r = 0; r < 3; r++)
        for (int k = 0; k < 3; k++) covs[9 * i + 3 * r + k] = fv[i].cov(r, k);
    if (counts) counts[i] = fv[i].num_points;
  }
}

void* orc_factors_create(size_t n) {
  auto* f = new OrcFactors;
  f->factors.resize(n);
  return f;
}
void orc_factors_destroy(void* h) { delete static_cast<OrcFactors*>(h); }
// target_index as int64 (-1 = outlier; voxelmap targets report the voxel id, i.e. index >> 32), mahalanobis 9 doubles row-major
void orc_factors_get(void* h, int is_voxelmap, std::int64_t* target_index, double* mahalanobis) {
  const auto& fs = static_cast<OrcFactors*>(h)->factors;
  for (size_t i = 0; i < fs.size(); i++) {
    if (target_index) target_index[i] = fs[i].inlier() ? static_cast<std::int64_t>(is_voxelmap == 1 ? (fs[i].target_index >> 32) : fs[i].target_index) : -1;  // 2 = flat map: (voxel << 32) | point
    if (mahalanobis)
      for (int r = 0; r < 3; r++)
        for (int k = 0; k < 3; k++) mahalanobis[9 * i + 3 * r + k] = fs[i].mahalanobis(r, k);
  }
}

static Target make_target(void* target_cloud, void* target_voxelmap) {
  Target t;
  if (target_voxelmap) {
    auto* vm = static_cast<OrcVoxelMap*>(target_voxelmap);
    if (vm->flat)
      t.flatmap = &vm->fmap;
    else
      t.voxelmap = &vm->map;
  } else {
    auto* c = static_cast<OrcCloud*>(target_cloud);
    t.cloud = &c->cloud;
    t.tree = &c->tree;
  }
  return t;
}

// Exactly one of target_cloud (with tree) / target_voxelmap must be non-null.
int orc_linearize(void* target_cloud, void* target_voxelmap, void* source_h, const orc_setting* s, const double* T16, void* factors_h, double* H36, double* b6, double* e, std::uint64_t* num_inliers) {
  const Target target = make_target(target_cloud, target_voxelmap);
  const PointCloud& source = static_cast<OrcCloud*>(source_h)->cloud;
  auto& factors = static_cast<OrcFactors*>(factors_h)->factors;
  if (factors.size() != source.size()) return -1;
  FactorSetting fs{s->factor_kind, s->robust_kind, s->robust_c};
  Reduction red;
  red.num_threads = s->num_threads;
  auto [H, b, err] = red.linearize(fs, target, source, s->max_dist_sq, se3_from_colmajor16(T16), factors);
  for (int i = 0; i < 6; i++) {
    b6[i] = b[i];
    for (int j = 0; j < 6; j++) H36[6 * i + j] = H(i, j);
  }
  *e = err;
  if (num_inliers) *num_inliers = std::count_if(factors.begin(), factors.end(), [](const Factor& f) { return f.inlier(); });
  return 0;
}

int orc_error(void* target_cloud, void* target_voxelmap, void* source_h, const orc_setting* s, const double* T16, void* factors_h, double* e) {
  const Target target = make_target(target_cloud, target_voxelmap);
  const PointCloud& source = static_cast<OrcCloud*>(source_h)->cloud;
  const auto& factors = static_cast<OrcFactors*>(factors_h)->factors;
  if (factors.size() != source.size()) return -1;
  FactorSetting fs{s->factor_kind, s->robust_kind, s->robust_c};
  Reduction red;
  red.num_threads = s->num_threads;
  *e = red.error(fs, target, source, se3_from_colmajor16(T16), factors);
  return 0;
}

// Full Registration<>::align.  trace_e / trace_new_e (optional, capacity trace_cap) receive the per-iteration errors.
// elapsed_sec (optional) = wall time of the optimizer loop only.
int orc_align(
  void* target_cloud,
  void* target_voxelmap,
  void* source_h,
  const orc_setting* s,
  const double* init_T16,
  orc_result* out,
  double* trace_e,
  double* trace_new_e,
  int trace_cap,
  int* trace_len,
  double* elapsed_sec) {
  const Target target = make_target(target_cloud, target_voxelmap);
  const PointCloud& source = static_cast<OrcCloud*>(source_h)->cloud;
  FactorSetting fs{s->factor_kind, s->robust_kind, s->robust_c};
  Reduction red;
  red.num_threads = s->num_threads;
  OptimizerSetting opt;
  opt.type = s->optimizer_type;
  opt.max_iterations = s->max_iterations;
  opt.max_inner_iterations = s->max_inner_iterations;
  opt.init_lambda = s->init_lambda;
  opt.lambda_factor = s->lambda_factor;
  opt.gn_lambda = s->gn_lambda;
  opt.verbose = s->verbose != 0;
  opt.restrict_lambda = s->restrict_lambda;
  for (int k = 0; k < 6; k++) opt.restrict_mask[k] = s->restrict_mask[k];
  TerminationCriteria crit;
  crit.translation_eps = s->translation_eps;
  crit.rotation_eps = s->rotation_eps;
  IterationTrace trace;
  const auto t0 = std::chrono::steady_clock::now();
  const RegistrationResult r = registration_align(opt, crit, red, fs, target, source, s->max_dist_sq, se3_from_colmajor16(init_T16), &trace);
  const auto t1 = std::chrono::steady_clock::now();
  if (elapsed_sec) *elapsed_sec = std::chrono::duration<double>(t1 - t0).count();
  se3_to_colmajor16(r.T_target_source, out->T);
  out->converged = r.converged;
  out->iterations = r.iterations;
  out->num_inliers = r.num_inliers;
  for (int i = 0; i < 6; i++) {
    out->b[i] = r.b[i];
    for (int j = 0; j < 6; j++) out->H[6 * i + j] = r.H(i, j);
  }
  out->error = r.error;
  if (trace_len) *trace_len = static_cast<int>(trace.e.size());
  for (int i = 0; i < trace_cap; i++) {
    if (trace_e && i < static_cast<int>(trace.e.size())) trace_e[i] = trace.e[i];
    if (trace_new_e && i < static_cast<int>(trace.new_e.size())) trace_new_e[i] = trace.new_e[i];
  }
  return 0;
}

int orc_max_threads() { return omp_get_max_threads(); }

}  // extern "C"
