// The reference's compiled helper library (L5: registration/registration_helper.hpp; its implementation is the 139 lines of
// src/small_gicp/registration/registration_helper.cpp) with the MI355X path behind the SAME exported functions: a build of
// koide3/small_gicp that compiles THIS file in place of registration_helper.cpp — nothing else changes, no caller is edited — runs
// preprocess_points / create_gaussian_voxelmap / align() x3 on the GPU through libsmall_gicp_amd.so.
//
//   preprocess_points       registration_helper.cpp:22-47   voxel grid + normals + covariances on the device (sga_voxelgrid_sampling,
//                                                            sga_estimate_normals_covariances); the returned KdTree<PointCloud> is the reference's
//                                                            own (host) tree, built with its OpenMP builder, because the return type promises one
//   create_gaussian_voxelmap registration_helper.cpp:50-54  the reference's GaussianVoxelMap (host object, as promised by the return type)
//   align (points)          registration_helper.cpp:57-79   preprocess both clouds, then one of the two below
//   align (clouds + tree)   registration_helper.cpp:81-122  Registration<Factor, ParallelReductionHIP> by setting.type; the tree argument is not
//                                                            read (the device searches its own exact index over `target`)
//   align (voxel map)       registration_helper.cpp:125-137 VGICP: the voxel map as target and tree through the same policy
// setting.num_threads keeps its meaning for the host parts (tree build); the reduction itself runs on `SGA_HELPER_GPUS` devices (default 1).
//
// Compiled and run by this repository's tests against the unmodified reference headers (oracle/ref/Makefile -> oracle/_ref/test_helper_hip).
#include <small_gicp/registration/registration_helper.hpp>

#include <cstdlib>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <vector>

#include <small_gicp/ann/kdtree_omp.hpp>
#include <small_gicp/factors/gicp_factor.hpp>
#include <small_gicp/factors/icp_factor.hpp>
#include <small_gicp/factors/plane_icp_factor.hpp>
#include <small_gicp/registration/reduction_hip.hpp>

namespace small_gicp {

namespace {

void must(int rc, const char* what) {
  if (rc != SGA_OK) throw std::runtime_error(std::string("small_gicp_amd: ") + what + ": " + sga_last_error());
}

// one device context per calling thread, for the preprocessing calls (the registrations keep their own inside the policy)
sga_context* thread_context() {
  struct Holder {
    sga_context* ctx = nullptr;
    ~Holder() {
      if (ctx) sga_context_destroy(ctx);
    }
  };
  thread_local Holder h;
  if (!h.ctx) must(sga_context_create(0, &h.ctx), "sga_context_create");
  return h.ctx;
}

struct CloudHandle {
  sga_cloud* c = nullptr;
  ~CloudHandle() {
    if (c) sga_cloud_destroy(c);
  }
};
struct IndexHandle {
  sga_index* i = nullptr;
  ~IndexHandle() {
    if (i) sga_index_destroy(i);
  }
};

int helper_gpus() {
  const char* e = std::getenv("SGA_HELPER_GPUS");
  const int g = e ? std::atoi(e) : 1;
  return g < 1 ? 1 : g;
}

// what registration_helper.cpp:89-120 sets on every Registration<> it creates
template <typename Factor, typename Target, typename Tree>
RegistrationResult run(const Target& target, const PointCloud& source, const Tree& tree, const Eigen::Isometry3d& init_T, const RegistrationSetting& setting, bool set_rejector) {
  Registration<Factor, ParallelReductionHIP> registration;
  registration.reduction.num_gpus = helper_gpus();
  if (set_rejector) registration.rejector.max_dist_sq = setting.max_correspondence_distance * setting.max_correspondence_distance;
  registration.criteria.rotation_eps = setting.rotation_eps;
  registration.criteria.translation_eps = setting.translation_eps;
  registration.optimizer.max_iterations = setting.max_iterations;
  registration.optimizer.verbose = setting.verbose;
  return registration.align(target, source, tree, init_T);
}

}  // namespace

std::pair<PointCloud::Ptr, std::shared_ptr<KdTree<PointCloud>>> preprocess_points(const PointCloud& points, double downsampling_resolution, int num_neighbors, int num_threads) {
  sga_context* ctx = thread_context();
  const size_t n_in = points.size();
  std::vector<float> xyz(3 * n_in);
  for (size_t i = 0; i < n_in; i++)
    for (int k = 0; k < 3; k++) xyz[3 * i + k] = static_cast<float>(points.point(i)[k]);
  CloudHandle raw, down;
  must(sga_cloud_create_f32(ctx, xyz.data(), nullptr, nullptr, n_in, &raw.c), "sga_cloud_create_f32");
  must(sga_voxelgrid_sampling(ctx, raw.c, downsampling_resolution, &down.c), "sga_voxelgrid_sampling");                 // util/downsampling.hpp:23-78
  must(sga_estimate_normals_covariances(ctx, down.c, nullptr, num_neighbors, 3), "sga_estimate_normals_covariances");  // util/normal_estimation.hpp:65-92
  size_t n = 0;
  must(sga_cloud_size(down.c, &n), "sga_cloud_size");
  std::vector<float> p(3 * n), nr(3 * n), c6(6 * n);
  must(sga_cloud_download(ctx, down.c, p.data(), nr.data(), c6.data()), "sga_cloud_download");
  auto out = std::make_shared<PointCloud>();
  out->resize(n);
  for (size_t i = 0; i < n; i++) {
    out->point(i) = Eigen::Vector4d(p[3 * i], p[3 * i + 1], p[3 * i + 2], 1.0);
    out->normal(i) = Eigen::Vector4d(nr[3 * i], nr[3 * i + 1], nr[3 * i + 2], 0.0);
    Eigen::Matrix4d C = Eigen::Matrix4d::Zero();
    const float* m = &c6[6 * i];
    C(0, 0) = m[0], C(0, 1) = C(1, 0) = m[1], C(0, 2) = C(2, 0) = m[2], C(1, 1) = m[3], C(1, 2) = C(2, 1) = m[4], C(2, 2) = m[5];
    out->cov(i) = C;
  }
  // the tree the return type promises (callers may search it themselves); the registrations below do not read it
  auto tree = num_threads == 1 ? std::make_shared<KdTree<PointCloud>>(out) : std::make_shared<KdTree<PointCloud>>(out, KdTreeBuilderOMP(num_threads));
  return {out, tree};
}

template <typename T, int D>
std::pair<PointCloud::Ptr, std::shared_ptr<KdTree<PointCloud>>>
preprocess_points(const std::vector<Eigen::Matrix<T, D, 1>>& points, double downsampling_resolution, int num_neighbors, int num_threads) {
  return preprocess_points(*std::make_shared<PointCloud>(points), downsampling_resolution, num_neighbors, num_threads);
}

template std::pair<PointCloud::Ptr, std::shared_ptr<KdTree<PointCloud>>> preprocess_points(const std::vector<Eigen::Matrix<float, 3, 1>>&, double, int, int);
template std::pair<PointCloud::Ptr, std::shared_ptr<KdTree<PointCloud>>> preprocess_points(const std::vector<Eigen::Matrix<float, 4, 1>>&, double, int, int);
template std::pair<PointCloud::Ptr, std::shared_ptr<KdTree<PointCloud>>> preprocess_points(const std::vector<Eigen::Matrix<double, 3, 1>>&, double, int, int);
template std::pair<PointCloud::Ptr, std::shared_ptr<KdTree<PointCloud>>> preprocess_points(const std::vector<Eigen::Matrix<double, 4, 1>>&, double, int, int);

GaussianVoxelMap::Ptr create_gaussian_voxelmap(const PointCloud& points, double voxel_resolution) {
  auto voxelmap = std::make_shared<GaussianVoxelMap>(voxel_resolution);
  voxelmap->insert(points);
  return voxelmap;
}

template <typename T, int D>
RegistrationResult
align(const std::vector<Eigen::Matrix<T, D, 1>>& target, const std::vector<Eigen::Matrix<T, D, 1>>& source, const Eigen::Isometry3d& init_T, const RegistrationSetting& setting) {
  const auto tgt = preprocess_points(target, setting.downsampling_resolution, 10, setting.num_threads);
  const auto src = preprocess_points(source, setting.downsampling_resolution, 10, setting.num_threads);
  if (setting.type == RegistrationSetting::VGICP) return align(*create_gaussian_voxelmap(*tgt.first, setting.voxel_resolution), *src.first, init_T, setting);
  return align(*tgt.first, *src.first, *tgt.second, init_T, setting);
}

template RegistrationResult
align(const std::vector<Eigen::Matrix<float, 3, 1>>&, const std::vector<Eigen::Matrix<float, 3, 1>>&, const Eigen::Isometry3d&, const RegistrationSetting&);
template RegistrationResult
align(const std::vector<Eigen::Matrix<float, 4, 1>>&, const std::vector<Eigen::Matrix<float, 4, 1>>&, const Eigen::Isometry3d&, const RegistrationSetting&);
template RegistrationResult
align(const std::vector<Eigen::Matrix<double, 3, 1>>&, const std::vector<Eigen::Matrix<double, 3, 1>>&, const Eigen::Isometry3d&, const RegistrationSetting&);
template RegistrationResult
align(const std::vector<Eigen::Matrix<double, 4, 1>>&, const std::vector<Eigen::Matrix<double, 4, 1>>&, const Eigen::Isometry3d&, const RegistrationSetting&);

RegistrationResult
align(const PointCloud& target, const PointCloud& source, const KdTree<PointCloud>& target_tree, const Eigen::Isometry3d& init_T, const RegistrationSetting& setting) {
  switch (setting.type) {
    case RegistrationSetting::ICP:
      return run<ICPFactor>(target, source, target_tree, init_T, setting, true);
    case RegistrationSetting::PLANE_ICP:
      return run<PointToPlaneICPFactor>(target, source, target_tree, init_T, setting, true);
    case RegistrationSetting::GICP:
      return run<GICPFactor>(target, source, target_tree, init_T, setting, true);
    case RegistrationSetting::VGICP:  // registration_helper.cpp:116-119: a message and an identity result
      std::cerr << "error: use align(const GaussianVoxelMap&, const GaussianVoxelMap&, const Eigen::Isometry3d&, const RegistrationSetting&) for VGICP" << std::endl;
      return RegistrationResult(Eigen::Isometry3d::Identity());
  }
  std::cerr << "invalid registration type" << std::endl;  // registration_helper.cpp:84-86
  abort();
}

RegistrationResult align(const GaussianVoxelMap& target, const PointCloud& source, const Eigen::Isometry3d& init_T, const RegistrationSetting& setting) {
  if (setting.type != RegistrationSetting::VGICP) std::cerr << "invalid registration type for GaussianVoxelMap" << std::endl;
  return run<GICPFactor>(target, source, target, init_T, setting, false);  // (the reference leaves the rejector at its default here, registration_helper.cpp:130-135)
}

}  // namespace small_gicp
