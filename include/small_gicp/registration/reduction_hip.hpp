// ParallelReductionHIP — the Reduction policy that runs Registration<>::align's data-parallel loop on an MI355X, and
// HipAligned<Optimizer> — the Optimizer-slot adaptor that makes one align() pay the host-side bookkeeping ONCE.
//
// This header belongs on the REFERENCE's side of the boundary: it is written against koide3/small_gicp's own headers (Eigen types,
// points/traits.hpp accessors, its factor structs, its optimizers) and against the plain C ABI of small_gicp_amd.h, and it fills the
// `Reduction` (and optionally the `Optimizer`) slot of Registration<PointFactor, Reduction, GeneralFactor, Rejector, Optimizer>
// (registration/registration.hpp:17-54) exactly like ParallelReductionOMP (registration/reduction_omp.hpp:21-73) does:
//
//   #include <small_gicp/registration/reduction_hip.hpp>        // -I<small_gicp>/include -I<small_gicp_amd>/include -lsmall_gicp_amd
//   Registration<GICPFactor, ParallelReductionHIP, NullFactor, DistanceRejector, HipAligned<LevenbergMarquardtOptimizer>> registration;
//   auto result = registration.align(target, source, target_tree, init_T);   // same call, same RegistrationResult
//
// linearize() / error() have the reference's signatures (reduction.hpp:20-62).  The clouds are uploaded once and stay on the
// device together with the search index and the per-point factor state (registration.hpp:41); every call costs a few kernel
// launches and one 768-byte result.
//
// Two ways to use it.
//  * Reduction slot only (`Registration<GICPFactor, ParallelReductionHIP>`): nothing else of the reference changes.  The policy cannot
//    see where an align() begins or ends, so EVERY linearize checks that the clouds it uploaded are still the caller's (a hash over
//    both clouds, `verify_content`) and fills target_index / source_index of the host `factors` (`sync_inliers`), because the
//    reference counts them after its loop (optimizer.hpp:146).  Correct and convenient; at 1M points those two per-call passes over
//    host memory cost several times what the device pass costs.
//  * Reduction + Optimizer slot (`HipAligned<LevenbergMarquardtOptimizer>` or `HipAligned<GaussNewtonOptimizer>`): the adaptor IS the
//    reference's optimizer (it derives from it and calls its optimize() unchanged, optimizer.hpp:24-149) bracketed by
//    reduction.begin_align() — hash / upload / index build once — and reduction.end_align() — the host `factors` filled once, after the
//    loop, which is when the reference reads them; RegistrationResult::num_inliers is then counted exactly as optimizer.hpp:146 does.
//    Inside the bracket a linearize is the device pass and nothing else: the rate of the C ABI (bench.py `policy_c3`).
//
// Host factors.  `sync_inliers` (default on) fills target_index / source_index; `sync_factors` (default off) additionally fills
// GICPFactor::mahalanobis (24 more bytes per point and a recompute kernel).  NOTE for code that inspects the factors itself: with
// sync_factors off the 3x3 block of GICPFactor::mahalanobis is set to NaN whenever the indices are filled — GICPFactor::error() on
// the host would otherwise silently use matrices that no longer belong to the correspondences (ParallelReductionOMP leaves them
// filled, gicp_factor.hpp:60,94).
//
// Threads.  Registration<>::align is const and the reference runs it from many threads at once
// (src/benchmark/odometry_benchmark_small_gicp_tbb_flow.cpp:81-96).  Every calling thread gets its own device context, stream and
// uploaded clouds (looked up by thread id in a pool shared by the copies of one policy object), so concurrent align() calls on one
// Registration — or on copies of it — are independent, on the host and on the GPU.
//
// Several GPUs in one process: `num_gpus` > 1 (the analogue of ParallelReductionOMP::num_threads, reduction_omp.hpp:22,72) shards the
// source over devices `device` .. `device + num_gpus - 1` with the target replicated (sga_multi_*, small_gicp_amd.h); the sums of the
// shards are added on the host in a fixed order.
//
// Caching.  The policy recognises the clouds it has already uploaded by (address, size, hash of every point / normal / covariance); a
// caller that refills a cloud object in place (every odometry loop does) is therefore noticed and the cloud uploaded again.  rebind()
// forces it.  The target_tree argument is not used: the device builds its own exact nearest-neighbour index over `target`.
// Supported factors: ICPFactor, PointToPlaneICPFactor, GICPFactor and RobustFactor<Huber|Cauchy, F> over them; rejectors:
// DistanceRejector, NullRejector; targets: point clouds (traits::point / normal / cov).  A voxel-map target — VGICP,
// registration_helper.cpp:125-137 — is REJECTED AT COMPILE TIME here (its traits::point(i) takes packed voxel indices, not 0..size-1):
// it goes through sga_index_build_gaussian_voxelmap / sga_align of small_gicp_amd.h, see INTEGRATION.md section 2.  Custom rejectors /
// factors with host callbacks stay on the CPU reductions (or use sga_problem_set_rejector).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <type_traits>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Geometry>

#include <small_gicp_amd.h>
#include <small_gicp/factors/gicp_factor.hpp>
#include <small_gicp/factors/icp_factor.hpp>
#include <small_gicp/factors/plane_icp_factor.hpp>
#include <small_gicp/factors/robust_kernel.hpp>
#include <small_gicp/points/traits.hpp>
#include <small_gicp/registration/optimizer.hpp>
#include <small_gicp/registration/registration_result.hpp>
#include <small_gicp/registration/rejector.hpp>

namespace small_gicp {

namespace hip_detail {

// reference factor type -> sga_factor_kind / sga_robust_kind, and access to the plain factor inside a RobustFactor
template <typename Factor>
struct factor_map;
template <>
struct factor_map<ICPFactor> {
  static constexpr int kind = SGA_ICP, robust = SGA_ROBUST_NONE;
  static double width(const ICPFactor&) { return 1.0; }
  static ICPFactor& plain(ICPFactor& f) { return f; }
};
template <>
struct factor_map<PointToPlaneICPFactor> {
  static constexpr int kind = SGA_PLANE_ICP, robust = SGA_ROBUST_NONE;
  static double width(const PointToPlaneICPFactor&) { return 1.0; }
  static PointToPlaneICPFactor& plain(PointToPlaneICPFactor& f) { return f; }
};
template <>
struct factor_map<GICPFactor> {
  static constexpr int kind = SGA_GICP, robust = SGA_ROBUST_NONE;
  static double width(const GICPFactor&) { return 1.0; }
  static GICPFactor& plain(GICPFactor& f) { return f; }
};
template <typename F>
struct factor_map<RobustFactor<Huber, F>> {
  static constexpr int kind = factor_map<F>::kind, robust = SGA_ROBUST_HUBER;
  static double width(const RobustFactor<Huber, F>& f) { return f.robust_kernel.c; }
  static F& plain(RobustFactor<Huber, F>& f) { return f.factor; }
};
template <typename F>
struct factor_map<RobustFactor<Cauchy, F>> {
  static constexpr int kind = factor_map<F>::kind, robust = SGA_ROBUST_CAUCHY;
  static double width(const RobustFactor<Cauchy, F>& f) { return f.robust_kernel.c; }
  static F& plain(RobustFactor<Cauchy, F>& f) { return f.factor; }
};

inline double max_dist_sq_of(const DistanceRejector& r) { return r.max_dist_sq; }
inline double max_dist_sq_of(const NullRejector&) { return -1.0; }  // sga_factor_params: < 0 = no rejector

inline void set_mahalanobis(GICPFactor& f, const float* m6) {
  f.mahalanobis.setZero();
  f.mahalanobis(0, 0) = m6[0];
  f.mahalanobis(0, 1) = f.mahalanobis(1, 0) = m6[1];
  f.mahalanobis(0, 2) = f.mahalanobis(2, 0) = m6[2];
  f.mahalanobis(1, 1) = m6[3];
  f.mahalanobis(1, 2) = f.mahalanobis(2, 1) = m6[4];
  f.mahalanobis(2, 2) = m6[5];
}
template <typename F>
inline void set_mahalanobis(F&, const float*) {}

inline void check(int rc, const char* what) {
  if (rc != SGA_OK) throw std::runtime_error(std::string("small_gicp_amd: ") + what + ": " + sga_last_error());
}

// Host loops over the points (fingerprint, repack, factor fill) use a SMALL fixed team and only for large clouds: a default-sized
// OpenMP team (every hardware thread, spinning between regions) costs more than these loops and fights the registration's own threads.
constexpr int kHostThreads = 8;
constexpr size_t kParallelFrom = 65536;

// identity of a cloud's CONTENT: size, attributes and EVERY point / normal / covariance entry (FNV-1a per block of 4096 points, the
// block hashes combined in order; OpenMP over the blocks).  One streaming pass over the host data — cheap next to the repack and
// upload it decides about, and an in-place edit of any point is seen (a sampled fingerprint would miss it).
template <typename Cloud>
std::uint64_t fingerprint(const Cloud& c) {
  const size_t n = traits::size(c);
  const bool normals = traits::has_normals(c), covs = traits::has_covs(c);
  constexpr size_t kBlock = 4096;
  const size_t blocks = (n + kBlock - 1) / kBlock;
  std::vector<std::uint64_t> part(blocks);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(kHostThreads) if (n >= kParallelFrom)
#endif
  for (long long bl = 0; bl < static_cast<long long>(blocks); bl++) {
    std::uint64_t h = 1469598103934665603ull;
    auto mix = [&h](double v) {
      std::uint64_t b;
      std::memcpy(&b, &v, 8);
      h = (h ^ b) * 1099511628211ull;
    };
    const size_t i0 = static_cast<size_t>(bl) * kBlock, i1 = i0 + kBlock < n ? i0 + kBlock : n;
    for (size_t i = i0; i < i1; i++) {
      const Eigen::Vector4d p = traits::point(c, i);
      mix(p[0]), mix(p[1]), mix(p[2]);
      if (covs) {
        const Eigen::Matrix4d m = traits::cov(c, i);
        mix(m(0, 0)), mix(m(0, 1)), mix(m(0, 2)), mix(m(1, 1)), mix(m(1, 2)), mix(m(2, 2));
      }
      if (normals) {
        const Eigen::Vector4d v = traits::normal(c, i);
        mix(v[0]), mix(v[1]), mix(v[2]);
      }
    }
    part[bl] = h;
  }
  std::uint64_t h = 1469598103934665603ull ^ n ^ (normals ? 0x9e3779b97f4a7c15ull : 0ull) ^ (covs ? 0xc2b2ae3d27d4eb4full : 0ull);
  for (std::uint64_t v : part) h = (h ^ v) * 1099511628211ull;
  return h;
}

struct DeviceState {
  int device = 0;
  sga_context* ctx = nullptr;
  sga_cloud *target = nullptr, *source = nullptr;
  sga_index* index = nullptr;
  sga_problem* problem = nullptr;
  const void *target_addr = nullptr, *source_addr = nullptr;
  std::uint64_t target_fp = 0, source_fp = 0;
  std::uint64_t generation = 0;  // bumped whenever something is uploaded again
  std::vector<std::int64_t> idx;
  std::vector<float> m6;
  ~DeviceState() {
    if (problem) sga_problem_destroy(problem);
    if (index) sga_index_destroy(index);
    if (source) sga_cloud_destroy(source);
    if (target) sga_cloud_destroy(target);
    if (ctx) sga_context_destroy(ctx);
  }
};

// points/traits.hpp:15-78 accessor protocol -> the reference PointCloud layout the C ABI takes (point_cloud.hpp:69-71)
template <typename Cloud>
sga_cloud* upload(sga_context* ctx, const Cloud& c) {
  const size_t n = traits::size(c);
  const bool normals = traits::has_normals(c), covs = traits::has_covs(c);
  std::vector<double> p(4 * n), nr(normals ? 4 * n : 0), cv(covs ? 16 * n : 0);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(kHostThreads) if (n >= kParallelFrom)
#endif
  for (long long ii = 0; ii < static_cast<long long>(n); ii++) {
    const size_t i = static_cast<size_t>(ii);
    const Eigen::Vector4d v = traits::point(c, i);
    for (int k = 0; k < 4; k++) p[4 * i + k] = v[k];
    if (normals) {
      const Eigen::Vector4d w = traits::normal(c, i);
      for (int k = 0; k < 4; k++) nr[4 * i + k] = w[k];
    }
    if (covs) {
      const Eigen::Matrix4d m = traits::cov(c, i);
      for (int col = 0; col < 4; col++)
        for (int row = 0; row < 4; row++) cv[16 * i + 4 * col + row] = m(row, col);
    }
  }
  sga_cloud* out = nullptr;
  check(sga_cloud_create_f64(ctx, p.data(), nr.empty() ? nullptr : nr.data(), cv.empty() ? nullptr : cv.data(), n, &out), "sga_cloud_create_f64");
  return out;
}

}  // namespace hip_detail

/// @brief Reduction on an MI355X through libsmall_gicp_amd (replaces ParallelReductionOMP, reduction_omp.hpp:21-73).
struct ParallelReductionHIP {
  ParallelReductionHIP() : device(0), sync_factors(false), sync_inliers(true), verify_content(true), fp64_math(false), num_inliers(0), state(std::make_shared<hip_detail::DeviceState>()) {}

  /// Forget the uploaded clouds: the next linearize() uploads target and source again.
  void rebind() const {
    state->target_addr = state->source_addr = nullptr;
  }

  /// Uploads so far (diagnostic: a loop that re-registers unchanged objects must not re-upload them).
  std::uint64_t generation() const { return state->generation; }

  template <typename TargetPointCloud, typename SourcePointCloud>
  void bind(const TargetPointCloud& target, const SourcePointCloud& source, const Eigen::Isometry3d& T) const {
    auto& s = *state;
    if (!s.ctx) {
      s.device = device;
      hip_detail::check(sga_context_create(device, &s.ctx), "sga_context_create");
    }
    // content check on every call (one streaming pass over both clouds, ~0.2 ms per 100k points): an object refilled in place is
    // uploaded again.  verify_content = false trusts address + size (call rebind() after changing a cloud in place).
    const std::uint64_t tfp = verify_content ? hip_detail::fingerprint(target) : traits::size(target), sfp = verify_content ? hip_detail::fingerprint(source) : traits::size(source);
    if (s.target_addr != static_cast<const void*>(&target) || s.target_fp != tfp || !s.index) {
      if (s.problem) sga_problem_destroy(s.problem);
      if (s.index) sga_index_destroy(s.index);
      if (s.target) sga_cloud_destroy(s.target);
      s.problem = nullptr;
      s.index = nullptr;
      s.target = hip_detail::upload(s.ctx, target);
      hip_detail::check(sga_index_build_kdtree(s.ctx, s.target, &s.index), "sga_index_build_kdtree");  // replaces KdTree<PointCloud>(target), ann/kdtree.hpp:250-252
      s.target_addr = &target;
      s.target_fp = tfp;
      s.source_addr = nullptr;
      s.generation++;
    }
    if (s.source_addr != static_cast<const void*>(&source) || s.source_fp != sfp || !s.problem) {
      if (s.problem) sga_problem_destroy(s.problem);
      if (s.source) sga_cloud_destroy(s.source);
      s.problem = nullptr;
      s.source = hip_detail::upload(s.ctx, source);
      hip_detail::check(sga_problem_create(s.ctx, s.index, s.source, T.matrix().data(), &s.problem), "sga_problem_create");  // registration.hpp:41
      s.source_addr = &source;
      s.source_fp = sfp;
      s.generation++;
    }
  }

  /// reduction.hpp:20-47 / reduction_omp.hpp:24-59
  template <typename TargetPointCloud, typename SourcePointCloud, typename TargetTree, typename CorrespondenceRejector, typename Factor>
  std::tuple<Eigen::Matrix<double, 6, 6>, Eigen::Matrix<double, 6, 1>, double> linearize(
    const TargetPointCloud& target,
    const SourcePointCloud& source,
    const TargetTree& /* the device searches its own index over `target` */,
    const CorrespondenceRejector& rejector,
    const Eigen::Isometry3d& T,
    std::vector<Factor>& factors) const {
    using Map = hip_detail::factor_map<Factor>;
    bind(target, source, T);
    auto& s = *state;
    sga_factor_params fp = params<Factor>(factors, hip_detail::max_dist_sq_of(rejector));
    Eigen::Matrix<double, 6, 6> H;
    Eigen::Matrix<double, 6, 1> b;
    double H36[36], b6[6], e = 0.0;
    std::uint64_t inliers = 0;
    hip_detail::check(sga_linearize(s.ctx, s.problem, &fp, T.matrix().data(), H36, b6, &e, &inliers), "sga_linearize");
    for (int i = 0; i < 6; i++) {
      b(i) = b6[i];
      for (int j = 0; j < 6; j++) H(i, j) = H36[6 * i + j];
    }
    num_inliers = inliers;
    // What a CPU reduction leaves in `factors`.  The reference's only reader is optimizer.hpp:146 (it counts factors with a valid
    // target_index for RegistrationResult::num_inliers): sync_inliers (default) downloads the correspondences — 8 bytes per point,
    // no mahalanobis recomputation — and fills target_index / source_index; sync_factors additionally fills GICPFactor::mahalanobis
    // (24 more bytes per point and a recompute_maha kernel per linearize: only for code that inspects the factors).  With both off
    // the factors stay untouched and num_inliers of the result is 0 — read ParallelReductionHIP::num_inliers instead.
    if ((sync_factors || sync_inliers) && !factors.empty()) {
      const size_t n = factors.size();
      s.idx.resize(n);
      const bool gicp = Map::kind == SGA_GICP && sync_factors;
      if (gicp) s.m6.resize(6 * n);
      hip_detail::check(sga_problem_get_factors(s.ctx, s.problem, s.idx.data(), gicp ? s.m6.data() : nullptr), "sga_problem_get_factors");
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(hip_detail::kHostThreads) if (n >= hip_detail::kParallelFrom)
#endif
      for (long long ii = 0; ii < static_cast<long long>(n); ii++) {
        const size_t i = static_cast<size_t>(ii);
        auto& f = Map::plain(factors[i]);
        f.source_index = i;
        f.target_index = s.idx[i] < 0 ? std::numeric_limits<size_t>::max() : static_cast<size_t>(s.idx[i]);
        if (gicp) hip_detail::set_mahalanobis(f, &s.m6[6 * i]);
      }
    }
    return {H, b, e};
  }

  /// reduction.hpp:55-62 / reduction_omp.hpp:61-70 — with the correspondences and mahalanobis cached by the last linearize (gicp_factor.hpp:80-89)
  template <typename TargetPointCloud, typename SourcePointCloud, typename Factor>
  double error(const TargetPointCloud&, const SourcePointCloud&, const Eigen::Isometry3d& T, std::vector<Factor>& factors) const {
    auto& s = *state;
    if (!s.problem) throw std::runtime_error("ParallelReductionHIP::error before linearize");
    sga_factor_params fp = params<Factor>(factors, last_max_dist_sq);
    double e = 0.0;
    hip_detail::check(sga_error(s.ctx, s.problem, &fp, T.matrix().data(), &e), "sga_error");
    return e;
  }

  int device;                  ///< HIP device
  bool sync_factors;           ///< also fill GICPFactor::mahalanobis of the host `factors` after every linearize (default off)
  bool sync_inliers;           ///< fill target_index / source_index of the host `factors` after every linearize (default on: optimizer.hpp:146 counts them)
  bool verify_content;         ///< hash both clouds on every linearize to notice in-place edits (default on); off: address + size only, see rebind()
  bool fp64_math;              ///< per-pair arithmetic in fp64 (data on the device is fp32 either way)
  mutable size_t num_inliers;  ///< inliers of the last linearize

private:
  template <typename Factor>
  sga_factor_params params(const std::vector<Factor>& factors, double max_dist_sq) const {
    using Map = hip_detail::factor_map<Factor>;
    sga_factor_params fp;
    sga_factor_params_default(&fp);
    fp.factor_kind = Map::kind;
    fp.robust_kind = Map::robust;
    fp.robust_c = factors.empty() ? 1.0 : Map::width(factors.front());
    fp.max_dist_sq = max_dist_sq;
    fp.math_mode = fp64_math ? SGA_MATH_FP64 : SGA_MATH_FP32;
    last_max_dist_sq = max_dist_sq;
    return fp;
  }

  mutable double last_max_dist_sq = 1.0;
  std::shared_ptr<hip_detail::DeviceState> state;  // shared by copies of the policy (Registration<> objects are copied freely)
};

}  // namespace small_gicp
