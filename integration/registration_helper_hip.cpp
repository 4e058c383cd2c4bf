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
// Device residency (round 5).  A cloud is uploaded ONCE: preprocess_points keeps what it built on the device — the downsampled cloud with its
// normals / covariances and the exact search index the covariance estimation needed anyway — in a side table keyed by the PointCloud it
// returns (weak ownership: the entry dies with the cloud; a cloud edited afterwards is recognised by its content hash and goes through the
// general policy path).  align(clouds + tree) of two such clouds then runs on those device objects (sga_problem / sga_align: no upload, no
// second tree); align(points ...) never leaves the device between the raw points and the result: no download, no host tree.
//
// Compiled and run by this repository's tests against the unmodified reference headers (oracle/ref/Makefile -> oracle/_ref/test_helper_hip).
#include <small_gicp/registration/registration_helper.hpp>

#include <cstdlib>
#include <iostream>
#include <map>
#include <memory>
#include <mutex>
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

// ---- what preprocess_points leaves on the device ------------------------------------------------------------------------------
struct DeviceCloud {
  CloudHandle cloud;  // downsampled, with normals and covariances
  IndexHandle index;  // exact nearest-neighbour index over it (attributes in kd order)
  size_t n = 0;
};

// raw points (the reference's layout: Vector4d, contiguous) -> voxel grid -> index -> normals + covariances; nothing comes back to the host
std::shared_ptr<DeviceCloud> device_preprocess(const PointCloud& points, double downsampling_resolution, int num_neighbors) {
  sga_context* ctx = thread_context();
  auto dc = std::make_shared<DeviceCloud>();
  CloudHandle raw;
  static_assert(sizeof(Eigen::Vector4d) == 4 * sizeof(double), "PointCloud::points is n x 4 doubles (points/point_cloud.hpp:69)");
  must(sga_cloud_create_f64(ctx, points.size() ? &points.points[0][0] : nullptr, nullptr, nullptr, points.size(), &raw.c), "sga_cloud_create_f64");  // recentred in double (device frames)
  must(sga_voxelgrid_sampling(ctx, raw.c, downsampling_resolution, &dc->cloud.c), "sga_voxelgrid_sampling");  // util/downsampling.hpp:23-78
  must(sga_index_build_kdtree(ctx, dc->cloud.c, &dc->index.i), "sga_index_build_kdtree");                     // KdTree<PointCloud>(points), registration_helper.cpp:30-31
  must(sga_estimate_normals_covariances(ctx, dc->cloud.c, dc->index.i, num_neighbors, 3), "sga_estimate_normals_covariances");  // util/normal_estimation.hpp:65-92 (fills the index's copies too)
  must(sga_cloud_size(dc->cloud.c, &dc->n), "sga_cloud_size");
  return dc;
}

// side table: host cloud returned by preprocess_points -> its device twin
struct Resident {
  std::weak_ptr<PointCloud> host;
  std::shared_ptr<DeviceCloud> device;
  std::uint64_t fingerprint = 0;
};
std::mutex g_resident_mutex;
std::map<const PointCloud*, Resident> g_resident;

void remember(const PointCloud::Ptr& host, const std::shared_ptr<DeviceCloud>& device) {
  std::lock_guard<std::mutex> lock(g_resident_mutex);
  for (auto it = g_resident.begin(); it != g_resident.end();) it = it->second.host.expired() ? g_resident.erase(it) : std::next(it);  // clouds that are gone take their device twins along
  g_resident[host.get()] = Resident{host, device, hip_detail::fingerprint(*host)};
}
// The device twin of a cloud preprocess_points returned, if it is still what the caller holds.  (twins of clouds that are gone
// are dropped here as well, not only by the next preprocess_points; the content hash of the cloud — a pass over its points — runs OUTSIDE
// the table's lock, so concurrent align() calls do not serialise on it.)  A twin is created on the preprocessing thread's context and may
// be consumed by any thread's: every entry point that takes a cloud or an index waits for the producer's event first (common.hpp: Ready),
// and the objects live on the device, not in a context.
std::shared_ptr<DeviceCloud> resident(const PointCloud& cloud) {
  std::shared_ptr<DeviceCloud> dev;
  std::uint64_t want = 0;
  std::vector<std::shared_ptr<DeviceCloud>> gone;  // released after the lock (freeing device memory is not a thing to do under it)
  {
    std::lock_guard<std::mutex> lock(g_resident_mutex);
    for (auto it = g_resident.begin(); it != g_resident.end();) {
      if (it->second.host.expired()) {
        gone.push_back(std::move(it->second.device));
        it = g_resident.erase(it);
      } else {
        ++it;
      }
    }
    const auto it = g_resident.find(&cloud);
    if (it == g_resident.end() || it->second.device->n != cloud.size()) return nullptr;
    dev = it->second.device;
    want = it->second.fingerprint;
  }
  gone.clear();
  if (want != hip_detail::fingerprint(cloud)) return nullptr;  // edited since: the general path uploads what the caller holds now
  return dev;
}

RegistrationResult to_result(const sga_result& r) {
  Eigen::Isometry3d T = Eigen::Isometry3d::Identity();
  for (int c = 0; c < 4; c++)
    for (int rr = 0; rr < 3; rr++) T.matrix()(rr, c) = r.T_target_source[4 * c + rr];
  RegistrationResult out(T);
  out.converged = r.converged != 0;
  out.iterations = r.iterations;
  out.num_inliers = r.num_inliers;
  for (int i = 0; i < 6; i++) {
    out.b(i) = r.b[i];
    for (int j = 0; j < 6; j++) out.H(i, j) = r.H[6 * i + j];
  }
  out.error = r.error;
  return out;
}

// Registration<Factor, ...>::align on objects that already live on the device (registration_helper.cpp:89-120 for the settings)
RegistrationResult device_align(const sga_index* target, const sga_cloud* source, size_t n_target, size_t n_source, int factor_kind, bool set_rejector, const Eigen::Isometry3d& init_T, const RegistrationSetting& setting) {
  if (n_target <= 10) std::cerr << "warning: target point cloud is too small. |target|=" << n_target << std::endl;  // registration.hpp:34-39
  if (n_source <= 10) std::cerr << "warning: source point cloud is too small. |source|=" << n_source << std::endl;
  sga_registration_setting st;
  sga_registration_setting_default(&st);
  st.factor.factor_kind = factor_kind;
  st.factor.max_dist_sq = set_rejector ? setting.max_correspondence_distance * setting.max_correspondence_distance : 1.0;  // DistanceRejector's default, rejector.hpp:20
  st.optimizer = SGA_LEVENBERG_MARQUARDT;
  st.max_iterations = setting.max_iterations;
  st.rotation_eps = setting.rotation_eps;
  st.translation_eps = setting.translation_eps;
  st.verbose = setting.verbose ? 1 : 0;
  double T16[16];
  for (int c = 0; c < 4; c++)
    for (int r = 0; r < 4; r++) T16[4 * c + r] = init_T.matrix()(r, c);
  sga_result res;
  must(sga_align(thread_context(), target, source, T16, &st, &res), "sga_align");
  return to_result(res);
}

}  // namespace

std::pair<PointCloud::Ptr, std::shared_ptr<KdTree<PointCloud>>> preprocess_points(const PointCloud& points, double downsampling_resolution, int num_neighbors, int num_threads) {
  sga_context* ctx = thread_context();
  const std::shared_ptr<DeviceCloud> dc = device_preprocess(points, downsampling_resolution, num_neighbors);
  const size_t n = dc->n;
  std::vector<double> p(3 * n);
  std::vector<float> nr(3 * n), c6(6 * n);
  must(sga_cloud_download_f64(ctx, dc->cloud.c, p.data(), nr.data(), c6.data()), "sga_cloud_download_f64");
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
  remember(out, dc);  // align(*out, ...) finds the cloud and its index on the device
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
  // registration_helper.cpp:57-79 without leaving the device: raw points up, result down — no host copy of the preprocessed clouds, no host tree
  const std::shared_ptr<DeviceCloud> tgt = device_preprocess(PointCloud(target), setting.downsampling_resolution, 10);
  const std::shared_ptr<DeviceCloud> src = device_preprocess(PointCloud(source), setting.downsampling_resolution, 10);
  if (setting.type == RegistrationSetting::VGICP) {
    IndexHandle voxelmap;  // create_gaussian_voxelmap, registration_helper.cpp:50-54
    must(sga_index_build_gaussian_voxelmap(thread_context(), tgt->cloud.c, setting.voxel_resolution, &voxelmap.i), "sga_index_build_gaussian_voxelmap");
    size_t voxels = 0;
    must(sga_index_size(voxelmap.i, &voxels), "sga_index_size");
    return device_align(voxelmap.i, src->cloud.c, voxels, src->n, SGA_GICP, false, init_T, setting);  // (the rejector stays at its default there, registration_helper.cpp:130-135)
  }
  const int kind = setting.type == RegistrationSetting::ICP ? SGA_ICP : (setting.type == RegistrationSetting::PLANE_ICP ? SGA_PLANE_ICP : SGA_GICP);
  return device_align(tgt->index.i, src->cloud.c, tgt->n, src->n, kind, true, init_T, setting);
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
  if (setting.type != RegistrationSetting::VGICP && helper_gpus() == 1) {
    // both clouds came out of preprocess_points and were not edited since: their device twins (cloud, index) are still there
    const std::shared_ptr<DeviceCloud> t = resident(target), sdev = t ? resident(source) : nullptr;
    if (t && sdev) {
      const int kind = setting.type == RegistrationSetting::ICP ? SGA_ICP : (setting.type == RegistrationSetting::PLANE_ICP ? SGA_PLANE_ICP : SGA_GICP);
      return device_align(t->index.i, sdev->cloud.c, t->n, sdev->n, kind, true, init_T, setting);
    }
  }
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
