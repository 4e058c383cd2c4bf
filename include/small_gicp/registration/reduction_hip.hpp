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
// How to use it.
//  * Swap the Reduction type, nothing else: `Registration<GICPFactor, ParallelReductionHIP>`.  This header specialises Registration<> for
//    its own policy type (bottom of the file): align() is the reference's (registration.hpp:33-43) with the optimizer of the Optimizer
//    slot — the reference's LevenbergMarquardtOptimizer / GaussNewtonOptimizer, unchanged, all its settings — run between
//    reduction.begin_align() (hash / upload / index build: once per align) and reduction.end_align() (the host `factors` filled once,
//    after the loop, which is when the reference reads them, optimizer.hpp:146).  Inside the bracket a linearize is the device pass and
//    nothing else.  The vector of per-point host factors (registration.hpp:41: 144 bytes per source point, 35 ms per call at 1M points) is
//    a local of align() nobody can see: it is not created, RegistrationResult::num_inliers comes from the device.  With
//    `reduction.verify_content = false` a whole align() runs at the rate of the C ABI (bench.py `policy_c3`: 8.2 k iterations/s at 1M <-> 1M).
//  * `HipAligned<Optimizer>` in the Optimizer slot is the same bracket as a type of its own (for code that calls optimizer.optimize()
//    itself, or Registration<> types written before the specialisation existed).
//  * reduction.linearize() / error() called directly, outside any align(): the policy cannot see where an align() begins or ends, so EVERY
//    linearize checks that the clouds it uploaded are still the caller's (a hash over both clouds, `verify_content`) and fills
//    target_index / source_index of the host `factors` (`sync_inliers`).  Correct and convenient; at 1M points those two per-call passes
//    over host memory cost several times what the device pass costs.
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
// Registration — or on copies of it — are independent, on the host and on the GPU.  A thread that ends takes its state along (device
// memory does not grow with the number of short-lived threads that ever called align()).
//
// Several GPUs in one process: `num_gpus` > 1 (the analogue of ParallelReductionOMP::num_threads, reduction_omp.hpp:22,72) shards the
// source over devices `device` .. `device + num_gpus - 1` with the target replicated (sga_multi_*, small_gicp_amd.h); the sums of the
// shards are added on the host in a fixed order.
//
// Caching.  The policy recognises the clouds it has already uploaded by (address, size, hash of every point / normal / covariance); a
// caller that refills a cloud object in place (every odometry loop does) is therefore noticed and the cloud uploaded again.  rebind()
// forces it.  The target_tree argument is not used: the device builds its own exact nearest-neighbour index over `target`.
// Supported factors: ICPFactor, PointToPlaneICPFactor, GICPFactor and RobustFactor<Huber|Cauchy, F> over them; rejectors:
// DistanceRejector, NullRejector; targets: point clouds (traits::point / normal / cov) and GaussianVoxelMap — VGICP,
// registration_helper.cpp:125-137: `registration.align(voxelmap, source, voxelmap, init_T)` with GICPFactor.  The map's voxels go to
// the device in flat order (sga_multi_set_target_voxels), so the voxel ids of target_index (voxel << 32, incremental_voxelmap.hpp:153)
// are the caller's; search_offsets 1 / 7 / 27.  IncrementalVoxelMap<FlatContainer<...>> (linear iVox, the scan-to-model target)
// works the same way (sga_multi_set_target_flat_voxels: <= 16 points per voxel, search offsets 1 / 7 / 27); other contents are rejected at compile time.
// Any other rejector type (registration/rejector.hpp:11-28 is a duck-typed slot) decides on the host through the device's batch callback
// (sga_multi_set_rejector: nearest neighbours down, verdicts up, once per linearization — correct, not fast; point-cloud targets).
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <iostream>
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
#include <small_gicp/ann/flat_container.hpp>
#include <small_gicp/ann/gaussian_voxelmap.hpp>
#include <small_gicp/ann/incremental_voxelmap.hpp>
#include <small_gicp/factors/gicp_factor.hpp>
#include <small_gicp/factors/icp_factor.hpp>
#include <small_gicp/factors/plane_icp_factor.hpp>
#include <small_gicp/factors/robust_kernel.hpp>
#include <small_gicp/points/traits.hpp>
#include <small_gicp/registration/optimizer.hpp>
#include <small_gicp/registration/registration.hpp>
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
template <typename Rejector>
inline double max_dist_sq_of(const Rejector&) { return -1.0; }  // any other rejector decides on the host (batch callback): the device searches without a bound
template <typename Rejector>
struct is_builtin_rejector : std::integral_constant<bool, std::is_same<Rejector, DistanceRejector>::value || std::is_same<Rejector, NullRejector>::value> {};

// A user-defined CorrespondenceRejector (registration/rejector.hpp:11-28 is a duck-typed slot: bool operator()(target, source, T,
// target_index, source_index, sq_dist), true = reject) decides on the host: once per linearization and shard the device hands over the
// nearest neighbour of every source point and gets the verdicts back (sga_multi_set_rejector).  Correct, not fast.
template <typename Target, typename Source, typename Rejector>
struct RejectorCall {
  const Target* target;
  const Source* source;
  const Rejector* rejector;
  static int invoke(void* user, const double T16[16], size_t first, size_t n, const std::int64_t* target_index, const float* sq_dist, unsigned char* reject) {
    const auto* c = static_cast<const RejectorCall*>(user);
    Eigen::Isometry3d T = Eigen::Isometry3d::Identity();
    for (int col = 0; col < 4; col++)
      for (int r = 0; r < 3; r++) T.matrix()(r, col) = T16[4 * col + r];
    for (size_t i = 0; i < n; i++)
      reject[i] = target_index[i] < 0 ? 1 : ((*c->rejector)(*c->target, *c->source, T, static_cast<size_t>(target_index[i]), first + i, static_cast<double>(sq_dist[i])) ? 1 : 0);
    return 0;
  }
};

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
// the content hash streams 160 bytes per point (points + 4 x 4 covariances): bound by memory bandwidth, which one socket's 8 threads do not saturate
inline int hash_threads(size_t n) { return n >= 8 * kParallelFrom ? 2 * kHostThreads : kHostThreads; }  // (target and source are hashed side by side during an align)

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
#pragma omp parallel for schedule(static) num_threads(hash_threads(n)) if (n >= kParallelFrom)
#endif
  for (long long bl = 0; bl < static_cast<long long>(blocks); bl++) {
    // four interleaved FNV chains (point i feeds chain i & 3): one chain is bound by the latency of its multiply, four keep the core busy
    std::uint64_t hh[4] = {1469598103934665603ull, 1469598103934665603ull ^ 1u, 1469598103934665603ull ^ 2u, 1469598103934665603ull ^ 3u};
    const size_t i0 = static_cast<size_t>(bl) * kBlock, i1 = i0 + kBlock < n ? i0 + kBlock : n;
    for (size_t i = i0; i < i1; i++) {
      std::uint64_t& h = hh[i & 3];
      auto mix = [&h](double v) {
        std::uint64_t b;
        std::memcpy(&b, &v, 8);
        h = (h ^ b) * 1099511628211ull;
      };
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
    part[bl] = (((hh[0] * 1099511628211ull ^ hh[1]) * 1099511628211ull ^ hh[2]) * 1099511628211ull) ^ hh[3];
  }
  std::uint64_t h = 1469598103934665603ull ^ n ^ (normals ? 0x9e3779b97f4a7c15ull : 0ull) ^ (covs ? 0xc2b2ae3d27d4eb4full : 0ull);
  for (std::uint64_t v : part) h = (h ^ v) * 1099511628211ull;
  return h;
}

// traits::point / normal / cov (points/traits.hpp:15-78) -> what the device keeps: fp32 xyz, normals, the six distinct covariance entries
// (the layout of sga_cloud_create_f32).  Converted here, while the cloud is repacked anyway and on the policy's threads: 48 bytes written
// per point instead of 192, and no second pass over them inside the library.
// The points are stored RELATIVE to `origin` (the centre of the cloud's bounding box, rounded by the library's own rule,
// sga_choose_origin): the reference's clouds are double (points/point_cloud.hpp:69-71) and may lie kilometres from the origin — the
// subtraction happens here, in double, before the cast, so that what fp32 holds is the cloud's shape (small_gicp_amd.h: device frames).
struct PackedCloud {
  std::vector<float> p, nr, cv;  // n x 3, n x 3, n x 6 (xx xy xz yy yz zz)
  double origin[3] = {0, 0, 0};
  size_t n = 0;
};
template <typename Cloud>
PackedCloud pack(const Cloud& c) {
  PackedCloud out;
  const size_t n = out.n = traits::size(c);
  const bool normals = traits::has_normals(c), covs = traits::has_covs(c);
  {
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(kHostThreads) if (n >= kParallelFrom) reduction(min : lo[:3]) reduction(max : hi[:3])
#endif
    for (long long ii = 0; ii < static_cast<long long>(n); ii++) {
      const Eigen::Vector4d v = traits::point(c, static_cast<size_t>(ii));
      for (int k = 0; k < 3; k++)
        if (std::isfinite(v[k])) lo[k] = v[k] < lo[k] ? v[k] : lo[k], hi[k] = v[k] > hi[k] ? v[k] : hi[k];
    }
    sga_choose_origin(lo, hi, out.origin);
  }
  const double o0 = out.origin[0], o1 = out.origin[1], o2 = out.origin[2];
  out.p.resize(3 * n);
  out.nr.resize(normals ? 3 * n : 0);
  out.cv.resize(covs ? 6 * n : 0);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(kHostThreads) if (n >= kParallelFrom)
#endif
  for (long long ii = 0; ii < static_cast<long long>(n); ii++) {
    const size_t i = static_cast<size_t>(ii);
    const Eigen::Vector4d v = traits::point(c, i);
    out.p[3 * i] = static_cast<float>(v[0] - o0), out.p[3 * i + 1] = static_cast<float>(v[1] - o1), out.p[3 * i + 2] = static_cast<float>(v[2] - o2);
    if (normals) {
      const Eigen::Vector4d w = traits::normal(c, i);
      for (int k = 0; k < 3; k++) out.nr[3 * i + k] = static_cast<float>(w[k]);
    }
    if (covs) {
      const Eigen::Matrix4d m = traits::cov(c, i);
      float* o = &out.cv[6 * i];
      o[0] = static_cast<float>(m(0, 0)), o[1] = static_cast<float>(m(1, 0)), o[2] = static_cast<float>(m(2, 0));  // (the entries sga_cloud_create_f64 reads: column-major m[0], m[1], m[2], m[5], m[6], m[10])
      o[3] = static_cast<float>(m(1, 1)), o[4] = static_cast<float>(m(2, 1)), o[5] = static_cast<float>(m(2, 2));
    }
  }
  return out;
}

// One helper thread per device state: hashes the caller's clouds WHILE the registration already runs on the cached upload (begin_align).
// A persistent thread, because the OpenMP team of a thread lives with it: a fresh thread per align would create a fresh team every time.
class HashWorker {
public:
  ~HashWorker() {
    {
      std::lock_guard<std::mutex> lock(m_);
      quit_ = true;
    }
    cv_.notify_all();
    if (worker_.joinable()) worker_.join();
  }
  void start(std::function<int()> job) {
    std::lock_guard<std::mutex> lock(m_);
    if (!worker_.joinable()) worker_ = std::thread([this] { run(); });
    job_ = std::move(job);
    pending_ = true;
    done_ = false;
    cv_.notify_all();
  }
  bool pending() const { return pending_; }
  int wait() {  // the job's verdict; 1 also when it threw
    std::unique_lock<std::mutex> lock(m_);
    cv_.wait(lock, [this] { return done_; });
    pending_ = false;
    return result_;
  }

private:
  void run() {
    std::unique_lock<std::mutex> lock(m_);
    for (;;) {
      cv_.wait(lock, [this] { return quit_ || (pending_ && !done_ && job_); });
      if (quit_) return;
      std::function<int()> job = std::move(job_);
      job_ = nullptr;
      lock.unlock();
      int r = 1;
      try {
        r = job();
      } catch (...) {
        r = 1;
      }
      lock.lock();
      result_ = r;
      done_ = true;
      cv_.notify_all();
    }
  }
  std::mutex m_;
  std::condition_variable cv_;
  std::thread worker_;
  std::function<int()> job_;
  bool pending_ = false, done_ = false, quit_ = false;
  int result_ = 0;
};
// target and source are hashed side by side (two teams): bit 0 = the target changed, bit 1 = the source
class Verifier {
public:
  void start(std::function<int()> target_changed, std::function<int()> source_changed) {
    t_.start(std::move(target_changed));
    s_.start(std::move(source_changed));
  }
  bool pending() const { return t_.pending() || s_.pending(); }
  int wait() { return (t_.wait() ? 1 : 0) | (s_.wait() ? 2 : 0); }

private:
  HashWorker t_, s_;
};

// What one calling thread keeps on the device(s): contexts + streams, the uploaded clouds, the search index, the factor state.
struct DeviceState {
  int device = 0, num_gpus = 1;
  sga_multi* multi = nullptr;  // one registration over num_gpus devices (sga_multi_*; num_gpus = 1: one context, one stream)
  const void *target_addr = nullptr, *source_addr = nullptr;
  std::uint64_t target_fp = 0, source_fp = 0;
  bool has_target = false, has_source = false;
  size_t n_source = 0;  // points of the uploaded source (the host `factors` are filled only if there is one per point)
  size_t n_target = 0;
  double last_max_dist_sq = 1.0;  // of this thread's last linearize (error() evaluates with the same rejector)
  bool custom_rejector = false;   // a host rejector is installed on the device problems
  Verifier verifier;              // begin_align: the content hash of cached clouds runs beside the registration
  size_t num_inliers = 0;  // of this thread's last linearize
  bool voxel_target = false;  // Gaussian voxel map: target_index of the host factors = voxel id << 32 (incremental_voxelmap.hpp:153)
  std::uint64_t generation = 0;  // bumped whenever something is uploaded again
  bool in_align = false;         // between begin_align() and end_align(): linearize() is the device pass and nothing else
  std::vector<std::int64_t> idx;
  std::vector<float> m6;
  // timings of the last begin_align / end_align bracket (seconds)
  double bind_s = 0.0, fill_s = 0.0, loop_s = 0.0, calls_s = 0.0;  // calls_s: inside linearize() / error() during the bracket
  ~DeviceState() {
    if (multi) sga_multi_destroy(multi);
  }
};

// The states of all threads that use (copies of) one policy object.  A thread's state is created on its first call and lives as long
// as the pool (TBB / OpenMP worker threads are persistent; a thread that is gone leaves its state behind until the last copy of the
// policy dies).
struct StatePool;
// A thread that ends takes its states along (contexts, streams, uploaded clouds, indices: ~0.3 GB at 1M <-> 1M): callers that run align()
// from short-lived threads (std::async) would otherwise grow device memory without bound.
struct ThreadStates {
  std::vector<std::weak_ptr<StatePool>> pools;
  ~ThreadStates();
};
inline ThreadStates& thread_states() {
  static thread_local ThreadStates t;
  return t;
}
struct StatePool : std::enable_shared_from_this<StatePool> {
  std::mutex mutex;
  std::map<std::thread::id, std::unique_ptr<DeviceState>> per_thread;
  std::uint64_t generation_total = 0;  // uploads of all threads (under the mutex)
  DeviceState& mine() {
    std::lock_guard<std::mutex> lock(mutex);
    auto& slot = per_thread[std::this_thread::get_id()];
    if (!slot) {
      slot.reset(new DeviceState);
      thread_states().pools.push_back(weak_from_this());  // dropped again when this thread ends
    }
    return *slot;
  }
  size_t states() {
    std::lock_guard<std::mutex> lock(mutex);
    return per_thread.size();
  }
  void uploaded() {
    std::lock_guard<std::mutex> lock(mutex);
    generation_total++;
  }
};

inline ThreadStates::~ThreadStates() {
  for (auto& w : pools)
    if (auto p = w.lock()) {
      std::unique_ptr<DeviceState> mine;
      {
        std::lock_guard<std::mutex> lock(p->mutex);
        auto it = p->per_thread.find(std::this_thread::get_id());
        if (it != p->per_thread.end()) {
          mine = std::move(it->second);
          p->per_thread.erase(it);
        }
      }
    }  // (the state is destroyed outside the pool's lock)
}

// A voxel map in the target slot (VGICP, registration_helper.cpp:125-137) is not a point cloud: traits::point(i) takes packed
// (voxel, point) indices (incremental_voxelmap.hpp:153-155).  A GaussianVoxelMap goes to the device as it is — its voxels in flat
// order, so that voxel ids mean the same on both sides; other contents (FlatContainer: several points per voxel) are refused.
template <typename T>
struct is_voxelmap : std::false_type {};
template <typename Contents>
struct is_voxelmap<IncrementalVoxelMap<Contents>> : std::true_type {};
template <typename T>
struct is_gaussian_voxelmap : std::false_type {};
template <>
struct is_gaussian_voxelmap<IncrementalVoxelMap<GaussianVoxel>> : std::true_type {};

struct PackedVoxels {
  std::vector<std::int32_t> coord;
  std::vector<double> mean, cov6;
  size_t n = 0;
};
inline PackedVoxels pack_voxels(const IncrementalVoxelMap<GaussianVoxel>& vm) {
  PackedVoxels out;
  const size_t n = out.n = vm.flat_voxels.size();
  out.coord.resize(3 * n);
  out.mean.resize(3 * n);
  out.cov6.resize(6 * n);
  for (size_t i = 0; i < n; i++) {
    const auto& v = *vm.flat_voxels[i];
    for (int k = 0; k < 3; k++) out.coord[3 * i + k] = v.first.coord[k], out.mean[3 * i + k] = v.second.mean[k];
    const Eigen::Matrix4d& m = v.second.cov;
    double* c = &out.cov6[6 * i];
    c[0] = m(0, 0), c[1] = m(0, 1), c[2] = m(0, 2), c[3] = m(1, 1), c[4] = m(1, 2), c[5] = m(2, 2);
  }
  return out;
}
inline std::uint64_t fingerprint(const IncrementalVoxelMap<GaussianVoxel>& vm) {
  // blocks of 1024 voxels hashed in parallel (the voxels hang off shared_ptrs: one thread chasing 165 000 of them took 1 ms), combined in order
  const size_t n = vm.flat_voxels.size();
  constexpr size_t kBlock = 1024;
  const size_t blocks = (n + kBlock - 1) / kBlock;
  std::vector<std::uint64_t> part(blocks);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(kHostThreads) if (n >= 16 * kBlock)
#endif
  for (long long bl = 0; bl < static_cast<long long>(blocks); bl++) {
    std::uint64_t h = 1469598103934665603ull ^ static_cast<std::uint64_t>(bl);
    auto mix = [&h](double v) {
      std::uint64_t b;
      std::memcpy(&b, &v, 8);
      h = (h ^ b) * 1099511628211ull;
    };
    const size_t i0 = static_cast<size_t>(bl) * kBlock, i1 = i0 + kBlock < n ? i0 + kBlock : n;
    for (size_t i = i0; i < i1; i++) {
      const auto& v = *vm.flat_voxels[i];
      h = (h ^ static_cast<std::uint32_t>(v.first.coord[0])) * 1099511628211ull;
      h = (h ^ static_cast<std::uint32_t>(v.first.coord[1])) * 1099511628211ull;
      h = (h ^ static_cast<std::uint32_t>(v.first.coord[2])) * 1099511628211ull;
      mix(v.second.mean[0]), mix(v.second.mean[1]), mix(v.second.mean[2]);
      const Eigen::Matrix4d& m = v.second.cov;
      mix(m(0, 0)), mix(m(0, 1)), mix(m(0, 2)), mix(m(1, 1)), mix(m(1, 2)), mix(m(2, 2));
    }
    part[bl] = h;
  }
  std::uint64_t h = 1469598103934665603ull ^ n ^ (static_cast<std::uint64_t>(vm.search_offsets.size()) << 40);
  {
    std::uint64_t b;
    const double il = vm.inv_leaf_size;
    std::memcpy(&b, &il, 8);
    h = (h ^ b) * 1099511628211ull;
  }
  for (std::uint64_t v : part) h = (h ^ v) * 1099511628211ull;
  return h;
}
// IncrementalVoxelMap<FlatContainer<N, C>> (linear iVox, the scan-to-model target of odometry_benchmark_small_gicp_model_omp.cpp): every voxel
// keeps up to max_num_points_in_cell points (flat_container.hpp:21-58); the device holds 16 slots per voxel.
template <typename T>
struct is_flat_voxelmap : std::false_type {};
template <bool N, bool C>
struct is_flat_voxelmap<IncrementalVoxelMap<FlatContainer<N, C>>> : std::true_type {};
constexpr size_t kFlatSlots = 16;
// set_search_offsets(27) APPENDS its 27 offsets to the one the constructor set (incremental_voxelmap.hpp:46,174-182): 28 entries, the
// query's own voxel first and once more inside the cube — which is the order the device's 27-voxel search follows.
inline int device_search_offsets(size_t n) {
  if (n == 1 || n == 7) return static_cast<int>(n);
  if (n == 28) return 27;
  throw std::runtime_error("ParallelReductionHIP: the voxel map's search_offsets must be what set_search_offsets(1 | 7 | 27) leaves");
}

struct PackedFlatVoxels {
  std::vector<std::int32_t> coord;
  std::vector<std::uint32_t> count;
  std::vector<double> pts, cov6;  // n x 16 x 3, n x 16 x 6 (empty without covariances)
  size_t n = 0;
};
template <bool N, bool C>
PackedFlatVoxels pack_voxels(const IncrementalVoxelMap<FlatContainer<N, C>>& vm) {
  PackedFlatVoxels out;
  const size_t n = out.n = vm.flat_voxels.size();
  out.coord.resize(3 * n);
  out.count.resize(n);
  out.pts.assign(3 * kFlatSlots * n, 0.0);
  if (C) out.cov6.assign(6 * kFlatSlots * n, 0.0);
  for (size_t i = 0; i < n; i++) {
    const auto& v = *vm.flat_voxels[i];
    const size_t m = v.second.points.size();
    if (m > kFlatSlots) throw std::runtime_error("ParallelReductionHIP: a voxel of the FlatContainer map holds more than 16 points (max_num_points_in_cell): not supported on the device");
    for (int k = 0; k < 3; k++) out.coord[3 * i + k] = v.first.coord[k];
    out.count[i] = static_cast<std::uint32_t>(m);
    for (size_t j = 0; j < m; j++) {
      for (int k = 0; k < 3; k++) out.pts[3 * (kFlatSlots * i + j) + k] = v.second.points[j][k];
      if constexpr (C) {
        const Eigen::Matrix4d& c = v.second.covs[j];
        double* o = &out.cov6[6 * (kFlatSlots * i + j)];
        o[0] = c(0, 0), o[1] = c(0, 1), o[2] = c(0, 2), o[3] = c(1, 1), o[4] = c(1, 2), o[5] = c(2, 2);
      }
    }
  }
  return out;
}
template <bool N, bool C>
std::uint64_t fingerprint(const IncrementalVoxelMap<FlatContainer<N, C>>& vm) {
  std::uint64_t h = 1469598103934665603ull ^ vm.flat_voxels.size() ^ (static_cast<std::uint64_t>(vm.search_offsets.size()) << 40);
  auto mix = [&h](double v) {
    std::uint64_t b;
    std::memcpy(&b, &v, 8);
    h = (h ^ b) * 1099511628211ull;
  };
  mix(vm.inv_leaf_size);
  for (const auto& pv : vm.flat_voxels) {
    const auto& v = *pv;
    for (int k = 0; k < 3; k++) h = (h ^ static_cast<std::uint32_t>(v.first.coord[k])) * 1099511628211ull;
    h = (h ^ v.second.points.size()) * 1099511628211ull;
    for (size_t j = 0; j < v.second.points.size(); j++) {
      mix(v.second.points[j][0]), mix(v.second.points[j][1]), mix(v.second.points[j][2]);
      if constexpr (C) {
        const Eigen::Matrix4d& m = v.second.covs[j];
        mix(m(0, 0)), mix(m(0, 1)), mix(m(0, 2)), mix(m(1, 1)), mix(m(1, 2)), mix(m(2, 2));
      }
    }
  }
  return h;
}

template <typename Target>
size_t target_size(const Target& t) {
  if constexpr (is_voxelmap<Target>::value) {
    return t.flat_voxels.size();
  } else {
    return traits::size(t);
  }
}

inline bool& warned_about_shared_devices() {
  static bool warned = false;
  return warned;
}
inline double seconds_since(const std::chrono::steady_clock::time_point& t0) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }

}  // namespace hip_detail

/// @brief Reduction on an MI355X through libsmall_gicp_amd (replaces ParallelReductionOMP, reduction_omp.hpp:21-73).
struct ParallelReductionHIP {
  ParallelReductionHIP()
  : device(0),
    num_gpus(1),
    sync_factors(false),
    sync_inliers(true),
    verify_content(true),
    fp64_math(false),
    num_inliers(0),
    pool(std::make_shared<hip_detail::StatePool>()) {}

  /// Forget the clouds this thread uploaded: its next linearize() / begin_align() uploads target and source again.
  void rebind() const {
    auto& s = pool->mine();
    s.target_addr = s.source_addr = nullptr;
  }

  /// Device states alive (one per thread that has used this policy or a copy of it and has not ended yet).
  size_t device_states() const { return pool->states(); }

  /// Uploads so far, over all threads (diagnostic: a loop that re-registers unchanged objects must not re-upload them).
  std::uint64_t generation() const {
    std::lock_guard<std::mutex> lock(pool->mutex);
    return pool->generation_total;
  }

  /// Hash / upload / index what is not on the device(s) yet.  Called by every linearize() outside an align bracket.
  template <typename TargetPointCloud, typename SourcePointCloud>
  void bind(const TargetPointCloud& target, const SourcePointCloud& source, const Eigen::Isometry3d& T, bool defer_verification = false) const {
    constexpr bool voxel_target = hip_detail::is_voxelmap<TargetPointCloud>::value;
    constexpr bool flat_target = hip_detail::is_flat_voxelmap<TargetPointCloud>::value;
    static_assert(
      !voxel_target || flat_target || hip_detail::is_gaussian_voxelmap<TargetPointCloud>::value,
      "ParallelReductionHIP: of the voxel maps only GaussianVoxelMap (one Gaussian per voxel: VGICP, registration_helper.cpp:125-137) and "
      "IncrementalVoxelMap<FlatContainer<...>> (linear iVox: scan-to-model ICP / GICP) are targets the device knows.");
    auto& s = pool->mine();
    if (s.verifier.pending()) {  // a bracket that was never closed (begin_align twice): its check first — the helper threads take one job at a time
      const int changed = s.verifier.wait();
      if (changed & 1) s.target_addr = nullptr;
      if (changed & 2) s.source_addr = nullptr;
    }
    if (s.multi && (s.device != device || s.num_gpus != num_gpus)) {
      sga_multi_destroy(s.multi);
      s.multi = nullptr;
      s.has_target = s.has_source = false;
    }
    if (!s.multi) {
      s.device = device;
      s.num_gpus = num_gpus < 1 ? 1 : num_gpus;
      std::vector<int> devices(static_cast<size_t>(s.num_gpus));
      int visible = sga_device_count();
      for (int g = 0; g < s.num_gpus; g++) devices[g] = visible > 0 ? (device + g) % visible : device + g;  // more shards than devices: logical shards share a device
      if (visible > 0 && device + s.num_gpus > visible) {
        bool& warned = hip_detail::warned_about_shared_devices();
        if (!warned) std::cerr << "warning: ParallelReductionHIP: devices " << device << " .. " << device + s.num_gpus - 1 << " requested, " << visible << " visible: the shards share devices (logical shards)" << std::endl;
        warned = true;
      }
      hip_detail::check(sga_multi_create(devices.data(), s.num_gpus, &s.multi), "sga_multi_create");
    }
    // content check (one streaming pass over both clouds, ~0.2 ms per 100k points): an object refilled in place is uploaded again.
    // verify_content = false trusts address + size (call rebind() after changing a cloud in place).
    // Inside an align bracket, for large clouds whose objects (address, size) are the ones already on the device: the registration starts
    // on the cached upload at once and the hash runs beside it on the state's helper thread; end of the bracket: equal -> done, different ->
    // upload what the caller holds now and register again (HipAligned::optimize).  A refilled cloud costs one wasted registration,
    // an unchanged one no longer waits 1.7 - 2.3 ms (2 x 1M points) for a hash before ten 0.12 ms iterations.
    const size_t nt = hip_detail::target_size(target), ns = traits::size(source);
    const bool same_objects = s.has_target && s.has_source && s.target_addr == static_cast<const void*>(&target) && s.source_addr == static_cast<const void*>(&source) && s.n_target == nt && s.n_source == ns;
    if (verify_content && defer_verification && same_objects && nt + ns >= hip_detail::kParallelFrom) {
      const std::uint64_t want_t = s.target_fp, want_s = s.source_fp;
      const TargetPointCloud* tp = &target;
      const SourcePointCloud* sp = &source;
      s.verifier.start([tp, want_t] { return hip_detail::fingerprint(*tp) != want_t ? 1 : 0; }, [sp, want_s] { return hip_detail::fingerprint(*sp) != want_s ? 1 : 0; });
      return;
    }
    const std::uint64_t tfp = verify_content ? hip_detail::fingerprint(target) : nt, sfp = verify_content ? hip_detail::fingerprint(source) : ns;
    if (s.target_addr != static_cast<const void*>(&target) || s.target_fp != tfp || !s.has_target) {
      if constexpr (flat_target) {
        // linear iVox: the voxels' points (and covariances) in flat order, 16 slots per voxel; searched over the map's 1 / 7 / 27 offsets
        const hip_detail::PackedFlatVoxels v = hip_detail::pack_voxels(target);
        hip_detail::check(
          sga_multi_set_target_flat_voxels(s.multi, 1.0 / target.inv_leaf_size, v.coord.data(), v.count.data(), v.pts.data(), v.cov6.empty() ? nullptr : v.cov6.data(), hip_detail::device_search_offsets(target.search_offsets.size()), v.n),
          "sga_multi_set_target_flat_voxels");
      } else if constexpr (voxel_target) {
        // the voxel map IS the search structure (incremental_voxelmap.hpp:99-119): its voxels in flat order + a hash of their coordinates per device
        const hip_detail::PackedVoxels v = hip_detail::pack_voxels(target);
        hip_detail::check(sga_multi_set_target_voxels(s.multi, 1.0 / target.inv_leaf_size, v.coord.data(), v.mean.data(), v.cov6.data(), v.n), "sga_multi_set_target_voxels");
        // incremental_voxelmap.hpp:99-119 visits the voxels of search_offsets around the query's own and keeps the nearest mean: 1 / 7 / 27 on the device too
        hip_detail::check(sga_multi_set_search_offsets(s.multi, hip_detail::device_search_offsets(target.search_offsets.size())), "sga_multi_set_search_offsets");
      } else {
        const hip_detail::PackedCloud c = hip_detail::pack(target);
        // replaces KdTree<PointCloud>(target), ann/kdtree.hpp:250-252: every device builds its own exact index over its copy
        hip_detail::check(sga_multi_set_target_f32_origin(s.multi, c.p.data(), c.nr.empty() ? nullptr : c.nr.data(), c.cv.empty() ? nullptr : c.cv.data(), c.n, c.origin), "sga_multi_set_target_f32_origin");
      }
      s.voxel_target = voxel_target && !flat_target;  // (a flat map's indices come packed from the device: (voxel << 32) | point)
      s.target_addr = &target;
      s.target_fp = tfp;
      s.n_target = nt;
      s.has_target = true;
      s.has_source = false;
      s.generation++;
      pool->uploaded();
    }
    if (s.source_addr != static_cast<const void*>(&source) || s.source_fp != sfp || !s.has_source) {
      const hip_detail::PackedCloud c = hip_detail::pack(source);
      hip_detail::check(sga_multi_set_source_f32_origin(s.multi, c.p.data(), c.nr.empty() ? nullptr : c.nr.data(), c.cv.empty() ? nullptr : c.cv.data(), c.n, c.origin, T.matrix().data()), "sga_multi_set_source_f32_origin");  // registration.hpp:41
      s.source_addr = &source;
      s.source_fp = sfp;
      s.n_source = c.n;
      s.has_source = true;
      s.generation++;
      pool->uploaded();
    }
  }

  /// The bracket HipAligned<> puts around the reference's optimizer: everything that has to happen once per align() happens here.
  template <typename TargetPointCloud, typename SourcePointCloud>
  void begin_align(const TargetPointCloud& target, const SourcePointCloud& source, const Eigen::Isometry3d& init_T) const {
    const auto t0 = std::chrono::steady_clock::now();
    bind(target, source, init_T, /*defer_verification=*/true);
    auto& s = pool->mine();
    hip_detail::check(sga_multi_reset_search_state(s.multi), "sga_multi_reset_search_state");  // a registration's result must not depend on earlier ones
    s.in_align = true;
    s.calls_s = 0.0;
    s.bind_s = hip_detail::seconds_since(t0);
  }
  /// The verdict of the content check begin_align() started beside the registration (true when none was pending).  False: the clouds were
  /// edited in place since their upload — the caller (HipAligned::optimize) forgets them and registers again.
  bool verified() const {
    auto& s = pool->mine();
    if (!s.verifier.pending()) return true;
    const int changed = s.verifier.wait();
    if (changed & 1) s.target_addr = nullptr;
    if (changed & 2) s.source_addr = nullptr;
    return changed == 0;
  }
  /// End of the bracket: the host `factors` as a CPU reduction would have left them after the last linearize (what optimizer.hpp:146 counts).
  /// Returns whether the host factors were filled (one per source point and sync_inliers / sync_factors on).
  template <typename Factor>
  bool end_align(std::vector<Factor>& factors) const {
    auto& s = pool->mine();
    s.in_align = false;
    const auto t0 = std::chrono::steady_clock::now();
    const bool filled = fill_factors(s, factors);
    s.fill_s = hip_detail::seconds_since(t0);
    return filled;
  }
  /// Seconds the last bracket of this thread took: begin_align (hash + upload + index build, if any), the optimizer between the
  /// two (the policy's linearize / error calls AND the reference's own host code: 6x6 solves, and its count over the host factors,
  /// optimizer.hpp:146, a pass over 144 bytes per source point), end_align (factor fill); last: the share of the optimizer's time spent
  /// inside this policy's linearize() / error().
  std::tuple<double, double, double, double> last_bracket_seconds() const {
    auto& s = pool->mine();
    return {s.bind_s, s.loop_s, s.fill_s, s.calls_s};
  }
  void note_loop_seconds(double v) const { pool->mine().loop_s = v; }
  /// Inliers of the calling thread's last linearize (the member `num_inliers` is the last write of ANY thread).
  size_t thread_num_inliers() const { return pool->mine().num_inliers; }

  /// reduction.hpp:20-47 / reduction_omp.hpp:24-59
  template <typename TargetPointCloud, typename SourcePointCloud, typename TargetTree, typename CorrespondenceRejector, typename Factor>
  std::tuple<Eigen::Matrix<double, 6, 6>, Eigen::Matrix<double, 6, 1>, double> linearize(
    const TargetPointCloud& target,
    const SourcePointCloud& source,
    const TargetTree& /* the device searches its own index over `target` */,
    const CorrespondenceRejector& rejector,
    const Eigen::Isometry3d& T,
    std::vector<Factor>& factors) const {
    auto& s = pool->mine();
    const auto call_t0 = std::chrono::steady_clock::now();
    if (!s.in_align) bind(target, source, T);
    sga_factor_params fp = params<Factor>(factors, hip_detail::max_dist_sq_of(rejector));
    s.last_max_dist_sq = fp.max_dist_sq;
    // any rejector type other than the two the device evaluates itself decides on the host, through the batch callback
    hip_detail::RejectorCall<TargetPointCloud, SourcePointCloud, CorrespondenceRejector> call{&target, &source, &rejector};
    if constexpr (!hip_detail::is_builtin_rejector<CorrespondenceRejector>::value) {
      static_assert(!hip_detail::is_voxelmap<TargetPointCloud>::value, "ParallelReductionHIP: a user-defined rejector needs a point-cloud target (the device reports nearest neighbours of a kd-tree)");
      hip_detail::check(sga_multi_set_rejector(s.multi, &decltype(call)::invoke, &call), "sga_multi_set_rejector");
      s.custom_rejector = true;
    } else if (s.custom_rejector) {
      hip_detail::check(sga_multi_set_rejector(s.multi, nullptr, nullptr), "sga_multi_set_rejector");
      s.custom_rejector = false;
    }
    Eigen::Matrix<double, 6, 6> H;
    Eigen::Matrix<double, 6, 1> b;
    double H36[36], b6[6], e = 0.0;
    std::uint64_t inliers = 0;
    hip_detail::check(sga_multi_linearize(s.multi, &fp, T.matrix().data(), H36, b6, &e, &inliers), "sga_multi_linearize");
    if constexpr (!hip_detail::is_builtin_rejector<CorrespondenceRejector>::value)
      hip_detail::check(sga_multi_set_rejector(s.multi, nullptr, nullptr), "sga_multi_set_rejector"), s.custom_rejector = false;  // `call` dies with this frame
    for (int i = 0; i < 6; i++) {
      b(i) = b6[i];
      for (int j = 0; j < 6; j++) H(i, j) = H36[6 * i + j];
    }
    num_inliers = inliers;
    s.num_inliers = inliers;
    // Outside an align bracket the policy cannot know which linearize is the last one: the host factors are filled after each
    // (sync_inliers: indices, 8 bytes per point; sync_factors: the GICP mahalanobis too).  Inside a bracket end_align() does it once.
    if (!s.in_align) fill_factors(s, factors);
    s.calls_s += hip_detail::seconds_since(call_t0);
    return {H, b, e};
  }

  /// reduction.hpp:55-62 / reduction_omp.hpp:61-70 — with the correspondences and mahalanobis cached by the last linearize (gicp_factor.hpp:80-89)
  template <typename TargetPointCloud, typename SourcePointCloud, typename Factor>
  double error(const TargetPointCloud&, const SourcePointCloud&, const Eigen::Isometry3d& T, std::vector<Factor>& factors) const {
    auto& s = pool->mine();
    if (!s.multi || !s.has_source) throw std::runtime_error("ParallelReductionHIP::error before linearize");
    const auto call_t0 = std::chrono::steady_clock::now();
    sga_factor_params fp = params<Factor>(factors, s.last_max_dist_sq);
    double e = 0.0;
    hip_detail::check(sga_multi_error(s.multi, &fp, T.matrix().data(), &e), "sga_multi_error");
    s.calls_s += hip_detail::seconds_since(call_t0);
    return e;
  }

  int device;                  ///< first HIP device
  int num_gpus;                ///< devices device .. device + num_gpus - 1 share every registration: source sharded, target replicated (the num_threads of reduction_omp.hpp:22)
  bool sync_factors;           ///< also fill GICPFactor::mahalanobis of the host `factors` (default off: then its 3x3 block is set to NaN, see the header comment)
  bool sync_inliers;           ///< fill target_index / source_index of the host `factors` (default on: optimizer.hpp:146 counts them)
  bool verify_content;         ///< hash both clouds on every bind to notice in-place edits (default on); off: address + size only, see rebind()
  bool fp64_math;              ///< per-pair arithmetic in fp64 (data on the device is fp32 either way)
  mutable size_t num_inliers;  ///< inliers of the last linearize (of the thread that wrote last)

private:
  template <typename Factor>
  bool fill_factors(hip_detail::DeviceState& s, std::vector<Factor>& factors) const {
    using Map = hip_detail::factor_map<Factor>;
    if (!(sync_factors || sync_inliers) || factors.empty() || factors.size() != s.n_source) return false;  // (the Registration<> specialisation below hands over a one-element stub)
    const size_t n = factors.size();
    s.idx.resize(n);
    constexpr bool is_gicp = Map::kind == SGA_GICP;
    const bool maha = is_gicp && sync_factors;
    if (maha) s.m6.resize(6 * n);
    hip_detail::check(sga_multi_get_factors(s.multi, s.idx.data(), maha ? s.m6.data() : nullptr), "sga_multi_get_factors");
    const float nan6[6] = {std::nanf(""), std::nanf(""), std::nanf(""), std::nanf(""), std::nanf(""), std::nanf("")};
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(hip_detail::kHostThreads) if (n >= hip_detail::kParallelFrom)
#endif
    for (long long ii = 0; ii < static_cast<long long>(n); ii++) {
      const size_t i = static_cast<size_t>(ii);
      auto& f = Map::plain(factors[i]);
      f.source_index = i;
      f.target_index = s.idx[i] < 0 ? std::numeric_limits<size_t>::max() : (s.voxel_target ? static_cast<size_t>(s.idx[i]) << 32 : static_cast<size_t>(s.idx[i]));
      if (is_gicp) hip_detail::set_mahalanobis(f, maha ? &s.m6[6 * i] : nan6);  // not synced: poisoned rather than stale
    }
    return true;
  }

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
    return fp;
  }

  std::shared_ptr<hip_detail::StatePool> pool;  // shared by copies of the policy (Registration<> objects are copied freely); one state per calling thread
};

/// @brief Optimizer-slot adaptor: the reference's optimizer, unchanged, between ParallelReductionHIP::begin_align and ::end_align.
/// `HipAligned<LevenbergMarquardtOptimizer>` / `HipAligned<GaussNewtonOptimizer>` (optimizer.hpp:13-58, :62-149) keep every setting of the
/// optimizer they derive from (max_iterations, init_lambda, ...).  With any other Reduction the adaptor is the plain optimizer.
template <typename Optimizer = LevenbergMarquardtOptimizer>
struct HipAligned : public Optimizer {
  template <
    typename TargetPointCloud,
    typename SourcePointCloud,
    typename TargetTree,
    typename CorrespondenceRejector,
    typename TerminationCriteria,
    typename Reduction,
    typename Factor,
    typename GeneralFactor>
  RegistrationResult optimize(
    const TargetPointCloud& target,
    const SourcePointCloud& source,
    const TargetTree& target_tree,
    const CorrespondenceRejector& rejector,
    const TerminationCriteria& criteria,
    Reduction& reduction,
    const Eigen::Isometry3d& init_T,
    std::vector<Factor>& factors,
    GeneralFactor& general_factor) const {
    if constexpr (std::is_same<typename std::remove_cv<Reduction>::type, ParallelReductionHIP>::value) {
      reduction.begin_align(target, source, init_T);  // hash / upload / index: once (the hash of clouds already on the device: beside the registration)
      RegistrationResult result(init_T);
      const auto t0 = std::chrono::steady_clock::now();
      try {
        result = Optimizer::optimize(target, source, target_tree, rejector, criteria, reduction, init_T, factors, general_factor);  // registration/optimizer.hpp, as it is
        if (!reduction.verified()) {
          // the caller refilled a cloud in place since its upload (every odometry loop does): what was registered is the old content.
          // Upload what the caller holds now and register again; the wasted pass is what the unchanged case no longer waits for.
          std::vector<Factor> none;
          reduction.end_align(none);
          reduction.begin_align(target, source, init_T);  // (addresses forgotten by verified(): hashes, uploads, no deferral)
          result = Optimizer::optimize(target, source, target_tree, rejector, criteria, reduction, init_T, factors, general_factor);
          (void)reduction.verified();
        }
        reduction.note_loop_seconds(hip_detail::seconds_since(t0));
      } catch (...) {
        (void)reduction.verified();  // (never leave a check running over clouds that may go away)
        std::vector<Factor> none;
        reduction.end_align(none);
        throw;
      }
      const bool filled = reduction.end_align(factors);  // the host factors: once, when the reference reads them
      if (filled)
        result.num_inliers = std::count_if(factors.begin(), factors.end(), [](const auto& factor) { return factor.inlier(); });  // optimizer.hpp:146, after the fill
      else
        result.num_inliers = reduction.thread_num_inliers();  // host factors untouched (on request, or a stub vector): the count of THIS thread's last linearization, from the device
      return result;
    } else {
      return Optimizer::optimize(target, source, target_tree, rejector, criteria, reduction, init_T, factors, general_factor);
    }
  }
};

namespace hip_detail {
template <typename T>
struct is_hip_aligned : std::false_type {};
template <typename O>
struct is_hip_aligned<HipAligned<O>> : std::true_type {};
}  // namespace hip_detail

/// @brief Registration<> with ParallelReductionHIP in the Reduction slot (a partial specialisation of registration/registration.hpp:17-54 for
/// this repository's own policy type): the reference's align(), with what has to happen once per align() happening once.  Swapping the
/// Reduction type is then ALL a user does — `Registration<GICPFactor, ParallelReductionHIP>` — and gets
///  * the optimizer of the Optimizer slot, unchanged, between reduction.begin_align() and reduction.end_align() (what HipAligned<> does;
///    an Optimizer that already is a HipAligned<> is used as it is);
///  * no vector of per-point host factors (registration.hpp:41: 144 bytes per source point for GICP, allocated and initialised per call:
///    35 ms at 1M points, thirty times the device's share of the align): it is a local of align() that no caller can see; the one
///    thing read of it, RegistrationResult::num_inliers (optimizer.hpp:146), comes from the device's count of the last linearization.
/// Same members, same call, same RegistrationResult.
template <typename PointFactor, typename GeneralFactor, typename CorrespondenceRejector, typename Optimizer>
struct Registration<PointFactor, ParallelReductionHIP, GeneralFactor, CorrespondenceRejector, Optimizer> {
public:
  template <typename TargetPointCloud, typename SourcePointCloud, typename TargetTree>
  RegistrationResult
  align(const TargetPointCloud& target, const SourcePointCloud& source, const TargetTree& target_tree, const Eigen::Isometry3d& init_T = Eigen::Isometry3d::Identity()) const {
    if (traits::size(target) <= 10) {
      std::cerr << "warning: target point cloud is too small. |target|=" << traits::size(target) << std::endl;
    }
    if (traits::size(source) <= 10) {
      std::cerr << "warning: source point cloud is too small. |source|=" << traits::size(source) << std::endl;
    }
    // The reference creates one factor per source point here (registration.hpp:41) — a local of this function that no caller ever sees;
    // what is read of it is RegistrationResult::num_inliers (optimizer.hpp:146), and that count comes from the device.  A one-element
    // stub (the factor settings) takes its place.
    std::vector<PointFactor> factors(1, PointFactor(point_factor));
    if constexpr (hip_detail::is_hip_aligned<Optimizer>::value) {
      return optimizer.optimize(target, source, target_tree, rejector, criteria, reduction, init_T, factors, general_factor);
    } else {
      HipAligned<Optimizer> bracketed;
      static_cast<Optimizer&>(bracketed) = optimizer;  // every setting of the caller's optimizer
      return bracketed.optimize(target, source, target_tree, rejector, criteria, reduction, init_T, factors, general_factor);
    }
  }

public:
  using PointFactorSetting = typename PointFactor::Setting;

  TerminationCriteria criteria;     ///< Termination criteria
  CorrespondenceRejector rejector;  ///< Correspondence rejector
  PointFactorSetting point_factor;  ///< Factor setting
  GeneralFactor general_factor;     ///< General factor
  ParallelReductionHIP reduction;   ///< Reduction
  Optimizer optimizer;              ///< Optimizer
};

}  // namespace small_gicp
