// Host-side pieces of the reference-side policy (include/small_gicp/registration/reduction_hip.hpp) that need no GPU: how clouds and the
// reference's voxel maps are repacked for the C ABI, what the content hash notices, and the reference behaviour the packing relies on
// (set_search_offsets(27) APPENDS: incremental_voxelmap.hpp:46,174-182).  Compiled against the UNMODIFIED reference headers by
// oracle/ref/Makefile (-> oracle/_ref/test_policy_host), run by tests/test_integration_policy.py in the CPU suite.
#include <cmath>
#include <cstdio>
#include <random>

#include <small_gicp/ann/flat_container.hpp>
#include <small_gicp/ann/gaussian_voxelmap.hpp>
#include <small_gicp/points/point_cloud.hpp>
#include <small_gicp/registration/reduction_hip.hpp>

using namespace small_gicp;

static int failures = 0;
#define CHECK(cond)                                                  \
  do {                                                               \
    if (!(cond)) {                                                   \
      std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); \
      failures++;                                                    \
    }                                                                \
  } while (0)

static PointCloud make_cloud(size_t n, unsigned seed, bool normals, bool covs) {
  std::mt19937 rng(seed);
  std::uniform_real_distribution<double> u(-10.0, 10.0);
  PointCloud c;
  c.resize(n);
  if (!normals) c.normals.clear();
  if (!covs) c.covs.clear();
  for (size_t i = 0; i < n; i++) {
    c.point(i) = Eigen::Vector4d(u(rng), u(rng), u(rng), 1.0);
    if (normals) c.normal(i) = Eigen::Vector4d(u(rng), u(rng), u(rng), 0.0);
    if (covs) {
      Eigen::Matrix4d m = Eigen::Matrix4d::Zero();
      const double a = u(rng), b = u(rng), cc = u(rng), d = u(rng), e = u(rng), f = u(rng);
      m(0, 0) = a, m(0, 1) = m(1, 0) = b, m(0, 2) = m(2, 0) = cc, m(1, 1) = d, m(1, 2) = m(2, 1) = e, m(2, 2) = f;
      c.cov(i) = m;
    }
  }
  return c;
}

int main() {
  // ---- pack(): traits::point / normal / cov -> fp32 xyz, normals, xx xy xz yy yz zz
  {
    const PointCloud c = make_cloud(1000, 1, true, true);
    const hip_detail::PackedCloud p = hip_detail::pack(c);
    CHECK(p.n == 1000 && p.p.size() == 3000 && p.nr.size() == 3000 && p.cv.size() == 6000);
    bool same = true;
    for (size_t i = 0; i < 1000; i++) {
      for (int k = 0; k < 3; k++) same = same && p.p[3 * i + k] == static_cast<float>(c.point(i)[k]) && p.nr[3 * i + k] == static_cast<float>(c.normal(i)[k]);
      const Eigen::Matrix4d& m = c.cov(i);
      const double want[6] = {m(0, 0), m(0, 1), m(0, 2), m(1, 1), m(1, 2), m(2, 2)};
      for (int k = 0; k < 6; k++) same = same && p.cv[6 * i + k] == static_cast<float>(want[k]);
    }
    CHECK(same);
    const hip_detail::PackedCloud q = hip_detail::pack(make_cloud(10, 2, false, false));
    CHECK(q.n == 10 && q.nr.empty() && q.cv.empty());
    CHECK(hip_detail::pack(PointCloud()).n == 0);
  }
  // ---- the content hash: equal for equal content, different after ANY single entry changed (a sampled hash would miss these)
  {
    PointCloud a = make_cloud(100000, 3, true, true);
    const PointCloud b = a;
    const auto h = hip_detail::fingerprint(a);
    CHECK(h == hip_detail::fingerprint(b));
    a.point(77777)[1] += 1e-9;
    CHECK(hip_detail::fingerprint(a) != h);
    a = b;
    a.cov(5)(1, 2) += 1e-12;
    CHECK(hip_detail::fingerprint(a) != h);
    a = b;
    a.normal(99999)[2] = -a.normal(99999)[2];
    CHECK(hip_detail::fingerprint(a) != h);
    a = b;
    std::swap(a.point(10), a.point(11));  // the same multiset of points in another order
    CHECK(hip_detail::fingerprint(a) != h);
    a = b;
    a.resize(99999);
    CHECK(hip_detail::fingerprint(a) != h);
    PointCloud no_covs = b;
    no_covs.covs.clear();
    CHECK(hip_detail::fingerprint(no_covs) != h);
  }
  // ---- GaussianVoxelMap -> coords / means / cov6 in flat order
  {
    const PointCloud c = make_cloud(5000, 4, false, true);
    GaussianVoxelMap vm(1.0);
    vm.insert(c);
    const hip_detail::PackedVoxels v = hip_detail::pack_voxels(vm);
    CHECK(v.n == vm.flat_voxels.size() && v.n > 100);
    bool same = true;
    for (size_t i = 0; i < v.n; i++) {
      const auto& vox = *vm.flat_voxels[i];
      for (int k = 0; k < 3; k++) same = same && v.coord[3 * i + k] == vox.first.coord[k] && v.mean[3 * i + k] == vox.second.mean[k];
      same = same && v.cov6[6 * i + 1] == vox.second.cov(0, 1) && v.cov6[6 * i + 5] == vox.second.cov(2, 2);
    }
    CHECK(same);
    const auto h = hip_detail::fingerprint(vm);
    vm.insert(make_cloud(100, 5, false, true));
    CHECK(hip_detail::fingerprint(vm) != h);  // a map that grew is another map
  }
  // ---- IncrementalVoxelMap<FlatContainerCov> -> 16 slots per voxel; the reference's search-offset quirk
  {
    const PointCloud c = make_cloud(5000, 6, false, true);
    IncrementalVoxelMap<FlatContainerCov> vm(1.0);
    CHECK(vm.search_offsets.size() == 1);
    vm.insert(c);
    const hip_detail::PackedFlatVoxels v = hip_detail::pack_voxels(vm);
    CHECK(v.n == vm.flat_voxels.size() && v.pts.size() == 3 * 16 * v.n && v.cov6.size() == 6 * 16 * v.n);
    bool same = true;
    size_t total = 0;
    for (size_t i = 0; i < v.n; i++) {
      const auto& vox = *vm.flat_voxels[i];
      same = same && v.count[i] == vox.second.points.size() && v.count[i] <= 10;
      total += v.count[i];
      for (size_t j = 0; j < vox.second.points.size(); j++) same = same && v.pts[3 * (16 * i + j) + 2] == vox.second.points[j][2] && v.cov6[6 * (16 * i + j) + 4] == vox.second.covs[j](1, 2);
    }
    CHECK(same && total > 1000);
    vm.set_search_offsets(7);
    CHECK(vm.search_offsets.size() == 7 && hip_detail::device_search_offsets(vm.search_offsets.size()) == 7);
    IncrementalVoxelMap<FlatContainerCov> fresh(1.0);
    fresh.set_search_offsets(27);  // appends to the offset of the constructor: 28 entries, the query's own voxel first
    CHECK(fresh.search_offsets.size() == 28 && fresh.search_offsets[0] == Eigen::Vector3i(0, 0, 0));
    CHECK(hip_detail::device_search_offsets(28) == 27 && hip_detail::device_search_offsets(1) == 1);
    bool threw = false;
    try {
      hip_detail::device_search_offsets(27);
    } catch (const std::exception&) {
      threw = true;
    }
    CHECK(threw);
    IncrementalVoxelMap<FlatContainerPoints> crowded(100.0);
    crowded.voxel_setting.max_num_points_in_cell = 40;
    crowded.voxel_setting.min_sq_dist_in_cell = 0.0;
    crowded.insert(c);
    threw = false;
    try {
      hip_detail::pack_voxels(crowded);
    } catch (const std::exception&) {
      threw = true;
    }
    CHECK(threw);  // more than 16 points in a voxel: refused, not truncated
  }
  std::printf("DONE failures=%d\n", failures);
  return failures == 0 ? 0 : 1;
}
