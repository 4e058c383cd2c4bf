// The reference-side binding, compiled and run: Registration<Factor, ParallelReductionHIP> (include/small_gicp/registration/
// reduction_hip.hpp) instantiated from the UNMODIFIED reference headers (/root/reference/include, over the Eigen stand-in of
// oracle/ref/eigen_shim) and linked against libsmall_gicp_amd.so, next to the reference's own Registration<Factor,
// ParallelReductionOMP> on the same clouds.  Built by oracle/ref/Makefile into oracle/_ref/test_reduction_hip (git-ignored, travels
// to the GPU box); driven by tests/test_integration_policy.py.
//
//   test_reduction_hip <target.bin> <source.bin>      (raw float32 xyz triples)
// Prints one "CASE {json}" line per case and exits non-zero if any check fails.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <random>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include <small_gicp/ann/flat_container.hpp>
#include <small_gicp/ann/gaussian_voxelmap.hpp>
#include <small_gicp/ann/kdtree.hpp>
#include <small_gicp/ann/kdtree_omp.hpp>
#include <small_gicp/factors/gicp_factor.hpp>
#include <small_gicp/factors/icp_factor.hpp>
#include <small_gicp/factors/plane_icp_factor.hpp>
#include <small_gicp/factors/robust_kernel.hpp>
#include <small_gicp/points/point_cloud.hpp>
#include <small_gicp/registration/reduction_omp.hpp>
#include <small_gicp/registration/registration.hpp>
#include <small_gicp/util/downsampling.hpp>
#include <small_gicp/util/normal_estimation_omp.hpp>

#include <small_gicp/registration/reduction_hip.hpp>

using namespace small_gicp;

static std::vector<Eigen::Vector4d> read_xyz(const char* path) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) {
    std::fprintf(stderr, "cannot read %s\n", path);
    std::exit(2);
  }
  const size_t bytes = static_cast<size_t>(f.tellg());
  f.seekg(0);
  std::vector<float> raw(bytes / 4);
  f.read(reinterpret_cast<char*>(raw.data()), bytes);
  std::vector<Eigen::Vector4d> pts(raw.size() / 3);
  for (size_t i = 0; i < pts.size(); i++) pts[i] = Eigen::Vector4d(raw[3 * i], raw[3 * i + 1], raw[3 * i + 2], 1.0);
  return pts;
}

static std::shared_ptr<PointCloud> preprocess(const std::vector<Eigen::Vector4d>& pts) {
  auto cloud = std::make_shared<PointCloud>(pts);
  auto down = voxelgrid_sampling(*cloud, 0.25);  // serial: deterministic order (util/downsampling.hpp:23-78)
  auto tree = std::make_shared<KdTree<PointCloud>>(down, KdTreeBuilderOMP(4));
  estimate_normals_covariances_omp(*down, *tree, 10, 4);
  // the device stores fp32: round the host copy the same way so that both reductions see identical inputs
  for (size_t i = 0; i < down->size(); i++) {
    for (int k = 0; k < 3; k++) down->point(i)[k] = static_cast<float>(down->point(i)[k]);
    for (int k = 0; k < 3; k++) down->normal(i)[k] = static_cast<float>(down->normal(i)[k]);
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) down->cov(i)(r, c) = static_cast<float>(down->cov(i)(r, c));
  }
  return down;
}

static void pose_error(const Eigen::Isometry3d& A, const Eigen::Isometry3d& B, double* dt, double* dr) {
  const Eigen::Isometry3d E = A.inverse() * B;
  *dt = E.translation().norm();
  const double c = (E.linear()(0, 0) + E.linear()(1, 1) + E.linear()(2, 2) - 1.0) / 2.0;
  *dr = std::acos(std::min(1.0, std::max(-1.0, c)));
}

static int failures = 0;

template <typename Factor, typename HipOptimizer>
static void run_case(const char* name, const PointCloud& target, const PointCloud& source, const KdTree<PointCloud>& tree, Registration<Factor, ParallelReductionHIP, NullFactor, DistanceRejector, HipOptimizer>& hip,
                     const Eigen::Isometry3d& init) {
  // the CPU side runs the optimizer the adaptor derives from (HipAligned<X> -> X)
  using CpuOptimizer = typename std::conditional<std::is_base_of<GaussNewtonOptimizer, HipOptimizer>::value, GaussNewtonOptimizer, LevenbergMarquardtOptimizer>::type;
  Registration<Factor, ParallelReductionOMP, NullFactor, DistanceRejector, CpuOptimizer> cpu;
  cpu.reduction.num_threads = 4;
  cpu.rejector.max_dist_sq = hip.rejector.max_dist_sq;
  cpu.point_factor = hip.point_factor;
  const RegistrationResult rc = cpu.align(target, source, tree, init);
  const RegistrationResult rh = hip.align(target, source, tree, init);
  double dt, dr;
  pose_error(rc.T_target_source, rh.T_target_source, &dt, &dr);
  double dH = 0.0, mH = 0.0;
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      dH = std::max(dH, std::abs(rc.H(i, j) - rh.H(i, j)));
      mH = std::max(mH, std::abs(rc.H(i, j)));
    }
  const double relH = dH / mH;
  const bool ok = dt < 1e-4 && dr < 1e-4 && rc.iterations == rh.iterations && rc.converged == rh.converged && std::llabs(static_cast<long long>(rc.num_inliers) - static_cast<long long>(rh.num_inliers)) <= 2 &&
                  rh.num_inliers == hip.reduction.num_inliers && relH < 1e-4;
  std::printf(
    "CASE {\"name\": \"%s\", \"ok\": %s, \"dt\": %.3e, \"dr\": %.3e, \"iterations\": [%zu, %zu], \"num_inliers\": [%zu, %zu], \"reduction_num_inliers\": %zu, \"rel_err_H\": %.3e, \"uploads\": %llu}\n", name, ok ? "true" : "false", dt, dr,
    rh.iterations, rc.iterations, rh.num_inliers, rc.num_inliers, hip.reduction.num_inliers, relH, static_cast<unsigned long long>(hip.reduction.generation()));
  if (!ok) failures++;
}

// iterations/s THROUGH the policy (what a small_gicp user who swaps the Reduction — and, for HipAligned, the Optimizer — gets): whole
// align() calls with a fixed number of LM iterations like bench.py, the reference's OpenMP reduction on 32 threads beside it
template <typename Reg>
static void rate(const char* name, const PointCloud& tgt, const PointCloud& src, const KdTree<PointCloud>& tr, int reps, bool lean = false) {
  const Eigen::Isometry3d I = Eigen::Isometry3d::Identity();
  Reg reg;
  if (lean) reg.reduction.verify_content = reg.reduction.sync_inliers = false;  // nothing per call but the device passes
  reg.criteria.rotation_eps = 0.0;
  reg.criteria.translation_eps = 0.0;
  reg.optimizer.max_iterations = 10;
  reg.align(tgt, src, tr, I);  // upload + index build + warm-up
  size_t iters = 0;
  double loop_s = 0.0;
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; r++) {
    iters += reg.align(tgt, src, tr, I).iterations + 1;
    loop_s += std::get<1>(reg.reduction.last_bracket_seconds());
  }
  const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  Registration<GICPFactor, ParallelReductionOMP> cpu;
  cpu.criteria = reg.criteria;
  cpu.optimizer.max_iterations = 10;
  cpu.reduction.num_threads = 32;
  const auto c0 = std::chrono::steady_clock::now();
  const size_t citers = cpu.align(tgt, src, tr, I).iterations + 1;
  const double cel = std::chrono::duration<double>(std::chrono::steady_clock::now() - c0).count();
  std::printf("RATE {\"name\": \"%s\", \"points\": [%zu, %zu], \"hip_policy_iterations_per_s\": %.1f, \"inside_the_optimizer_iterations_per_s\": %.1f, \"omp32_iterations_per_s\": %.1f, \"uploads\": %llu}\n", name, tgt.size(),
              src.size(), iters / el, loop_s > 0 ? iters / loop_s : 0.0, citers / cel, static_cast<unsigned long long>(reg.reduction.generation()));
}

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s target.bin source.bin\n", argv[0]);
    return 2;
  }
  auto target = preprocess(read_xyz(argv[1]));
  auto source = preprocess(read_xyz(argv[2]));
  KdTree<PointCloud> tree(target, KdTreeBuilderOMP(4));
  const Eigen::Isometry3d I = Eigen::Isometry3d::Identity();

  {
    Registration<GICPFactor, ParallelReductionHIP> reg;
    run_case("GICP", *target, *source, tree, reg, I);
    const auto uploads = reg.reduction.generation();
    run_case("GICP again (cached uploads)", *target, *source, tree, reg, I);
    if (reg.reduction.generation() != uploads) {
      std::printf("CASE {\"name\": \"unchanged clouds were uploaded again\", \"ok\": false}\n");
      failures++;
    }
    // an odometry loop refills the same objects: move the source in place — the policy must notice and upload it again
    Eigen::Isometry3d M = Eigen::Isometry3d::Identity();
    M.matrix()(0, 0) = std::cos(0.01);  // 0.01 rad about z
    M.matrix()(0, 1) = -std::sin(0.01);
    M.matrix()(1, 0) = std::sin(0.01);
    M.matrix()(1, 1) = std::cos(0.01);
    M.matrix()(0, 3) = 0.05;
    M.matrix()(1, 3) = -0.03;
    M.matrix()(2, 3) = 0.01;
    for (size_t i = 0; i < source->size(); i++) {
      Eigen::Vector4d p = M * source->point(i);
      for (int k = 0; k < 3; k++) p[k] = static_cast<float>(p[k]);
      source->point(i) = p;
      Eigen::Matrix4d c = M.matrix() * source->cov(i) * M.matrix().transpose();
      for (int r = 0; r < 4; r++)
        for (int cc = 0; cc < 4; cc++) c(r, cc) = static_cast<float>(c(r, cc));
      source->cov(i) = c;
      Eigen::Vector4d nn = M.matrix() * source->normal(i);
      for (int k = 0; k < 3; k++) nn[k] = static_cast<float>(nn[k]);
      source->normal(i) = nn;
    }
    run_case("GICP after refilling the source object in place", *target, *source, tree, reg, I);
    {
      // a PARTIAL in-place edit (a filter that touches a few points between two registrations): every point is part of the
      // fingerprint, so the stale device copy must be replaced
      const auto before = reg.reduction.generation();
      for (size_t i = 1; i < source->size(); i += 97) source->point(i)[2] = static_cast<float>(source->point(i)[2] + 0.02);
      run_case("GICP after editing a subset of the source points in place", *target, *source, tree, reg, I);
      const bool ok = reg.reduction.generation() == before + 1;
      std::printf("CASE {\"name\": \"partial in-place edit was uploaded again\", \"ok\": %s, \"uploads\": [%llu, %llu]}\n", ok ? "true" : "false", static_cast<unsigned long long>(before),
                  static_cast<unsigned long long>(reg.reduction.generation()));
      if (!ok) failures++;
    }
    {
      // full factor state on request: GICPFactor::mahalanobis filled like a CPU reduction leaves it
      reg.reduction.sync_factors = true;
      run_case("GICP with sync_factors (mahalanobis on the host)", *target, *source, tree, reg, I);
      reg.reduction.sync_factors = false;
    }
    reg.reduction.sync_inliers = false;  // fastest path: the host factors stay untouched, the count comes from the reduction
    Registration<GICPFactor, ParallelReductionOMP> cpu;
    cpu.reduction.num_threads = 4;
    const auto rc = cpu.align(*target, *source, tree, I);
    auto rh = reg.align(*target, *source, tree, I);
    // (the Registration<> specialisation puts the device's count into the result; without it the reference's count over untouched factors was 0)
    const bool ok = rh.num_inliers == reg.reduction.num_inliers && std::llabs(static_cast<long long>(reg.reduction.num_inliers) - static_cast<long long>(rc.num_inliers)) <= 2;
    std::printf("CASE {\"name\": \"sync_inliers = false: count from reduction.num_inliers\", \"ok\": %s, \"reduction_num_inliers\": %zu, \"cpu\": %zu}\n", ok ? "true" : "false", reg.reduction.num_inliers, rc.num_inliers);
    if (!ok) failures++;
  }
  {
    Registration<PointToPlaneICPFactor, ParallelReductionHIP> reg;
    run_case("PLANE_ICP", *target, *source, tree, reg, I);
  }
  {
    Registration<ICPFactor, ParallelReductionHIP> reg;
    reg.rejector.max_dist_sq = 0.25;
    run_case("ICP max_dist 0.5", *target, *source, tree, reg, I);
  }
  {
    Registration<RobustFactor<Huber, GICPFactor>, ParallelReductionHIP> reg;
    reg.point_factor.robust_kernel.c = 0.5;
    run_case("Huber(0.5) GICP", *target, *source, tree, reg, I);
  }
  {
    Registration<RobustFactor<Cauchy, GICPFactor>, ParallelReductionHIP, NullFactor, NullRejector> reg;
    Registration<RobustFactor<Cauchy, GICPFactor>, ParallelReductionOMP, NullFactor, NullRejector> cpu;
    cpu.reduction.num_threads = 4;
    const auto rc = cpu.align(*target, *source, tree, I);
    const auto rh = reg.align(*target, *source, tree, I);
    double dt, dr;
    pose_error(rc.T_target_source, rh.T_target_source, &dt, &dr);
    const bool ok = dt < 1e-4 && dr < 1e-4 && rc.iterations == rh.iterations && rc.num_inliers == rh.num_inliers;
    std::printf("CASE {\"name\": \"Cauchy GICP, NullRejector\", \"ok\": %s, \"dt\": %.3e, \"dr\": %.3e, \"iterations\": [%zu, %zu], \"num_inliers\": [%zu, %zu]}\n", ok ? "true" : "false", dt, dr, rh.iterations, rc.iterations, rh.num_inliers,
                rc.num_inliers);
    if (!ok) failures++;
  }
  // ---- the Optimizer-slot adaptor: the reference's optimizer between begin_align() and end_align()
  {
    Registration<GICPFactor, ParallelReductionHIP, NullFactor, DistanceRejector, HipAligned<LevenbergMarquardtOptimizer>> reg;
    run_case("HipAligned<LM> GICP", *target, *source, tree, reg, I);
    const auto uploads = reg.reduction.generation();
    run_case("HipAligned<LM> GICP again (cached uploads)", *target, *source, tree, reg, I);
    if (reg.reduction.generation() != uploads) {
      std::printf("CASE {\"name\": \"HipAligned: unchanged clouds were uploaded again\", \"ok\": false}\n");
      failures++;
    }
    // sync_factors: the mahalanobis of the host factors after the bracket; without it the block is NaN (never stale)
    {
      std::vector<GICPFactor> factors(source->size());
      TerminationCriteria crit;
      NullFactor gf;
      reg.reduction.sync_factors = true;
      reg.optimizer.optimize(*target, *source, tree, reg.rejector, crit, reg.reduction, I, factors, gf);
      size_t filled = 0, inl = 0;
      for (const auto& f : factors)
        if (f.inlier()) {
          inl++;
          filled += std::isfinite(f.mahalanobis(0, 0)) && f.mahalanobis(0, 0) > 0.0;
        }
      reg.reduction.sync_factors = false;
      std::vector<GICPFactor> factors2(source->size());
      reg.optimizer.optimize(*target, *source, tree, reg.rejector, crit, reg.reduction, I, factors2, gf);
      size_t poisoned = 0;
      for (const auto& f : factors2) poisoned += std::isnan(f.mahalanobis(0, 0));
      const bool ok = inl > 5000 && filled == inl && poisoned == factors2.size();
      std::printf("CASE {\"name\": \"HipAligned: mahalanobis filled with sync_factors, NaN without\", \"ok\": %s, \"inliers\": %zu, \"filled\": %zu, \"poisoned\": %zu}\n", ok ? "true" : "false", inl, filled, poisoned);
      if (!ok) failures++;
    }
  }
  {
    Registration<GICPFactor, ParallelReductionHIP, NullFactor, DistanceRejector, HipAligned<GaussNewtonOptimizer>> reg;
    run_case("HipAligned<GN> GICP", *target, *source, tree, reg, I);
  }
  {
    Registration<PointToPlaneICPFactor, ParallelReductionHIP, NullFactor, DistanceRejector, HipAligned<LevenbergMarquardtOptimizer>> reg;
    run_case("HipAligned<LM> PLANE_ICP", *target, *source, tree, reg, I);
  }
  {
    // source sharded over two (logical) devices inside this process: the same registration
    Registration<GICPFactor, ParallelReductionHIP, NullFactor, DistanceRejector, HipAligned<LevenbergMarquardtOptimizer>> reg;
    reg.reduction.num_gpus = 2;
    run_case("HipAligned<LM> GICP, num_gpus = 2 (two shards)", *target, *source, tree, reg, I);
    Registration<GICPFactor, ParallelReductionHIP> plain;
    plain.reduction.num_gpus = 3;
    run_case("GICP, num_gpus = 3, Reduction slot only", *target, *source, tree, plain, I);
  }
  {
    // Registration<>::align is const and the reference runs it from many threads at once (odometry_benchmark_small_gicp_tbb_flow.cpp:
    // 81-96): four threads, ONE registration object, every thread its own target / source pair (the source displaced differently)
    Registration<GICPFactor, ParallelReductionHIP, NullFactor, DistanceRejector, HipAligned<LevenbergMarquardtOptimizer>> reg;
    Registration<GICPFactor, ParallelReductionOMP> cpu;
    cpu.reduction.num_threads = 4;
    constexpr int kThreads = 4;
    std::vector<std::shared_ptr<PointCloud>> sources(kThreads);
    std::vector<RegistrationResult> want(kThreads, RegistrationResult(I)), got(kThreads, RegistrationResult(I));
    for (int t = 0; t < kThreads; t++) {
      Eigen::Isometry3d M = Eigen::Isometry3d::Identity();
      M.matrix()(0, 3) = 0.02 * t;
      M.matrix()(1, 3) = -0.01 * t;
      sources[t] = std::make_shared<PointCloud>(*source);
      for (size_t i = 0; i < sources[t]->size(); i++) {
        Eigen::Vector4d q = M * sources[t]->point(i);
        for (int k = 0; k < 3; k++) q[k] = static_cast<float>(q[k]);
        sources[t]->point(i) = q;
      }
      want[t] = cpu.align(*target, *sources[t], tree, I);
    }
    std::vector<std::thread> pool;
    std::vector<std::string> errors(kThreads);
    for (int t = 0; t < kThreads; t++)
      pool.emplace_back([&, t] {
        try {
          for (int rep = 0; rep < 3; rep++) got[t] = reg.align(*target, *sources[t], tree, I);
        } catch (const std::exception& ex) {
          errors[t] = ex.what();
        }
      });
    for (auto& th : pool) th.join();
    bool ok = true;
    double worst = 0.0;
    for (int t = 0; t < kThreads; t++) {
      double dt, dr;
      pose_error(want[t].T_target_source, got[t].T_target_source, &dt, &dr);
      worst = std::max(worst, std::max(dt, dr));
      ok = ok && errors[t].empty() && dt < 1e-4 && dr < 1e-4 && want[t].iterations == got[t].iterations && std::llabs(static_cast<long long>(want[t].num_inliers) - static_cast<long long>(got[t].num_inliers)) <= 2;
      if (!errors[t].empty()) std::fprintf(stderr, "thread %d: %s\n", t, errors[t].c_str());
    }
    std::printf("CASE {\"name\": \"4 threads align concurrently through one Registration object\", \"ok\": %s, \"worst_pose_error\": %.3e, \"uploads\": %llu}\n", ok ? "true" : "false", worst,
                static_cast<unsigned long long>(reg.reduction.generation()));
    if (!ok) failures++;
  }
  // ---- the Registration<> specialisation: swapping the Reduction type alone brackets the optimizer; no host factors when nobody reads them ----
  {
    Registration<GICPFactor, ParallelReductionOMP> cpu;
    cpu.reduction.num_threads = 4;
    const RegistrationResult rc = cpu.align(*target, *source, tree, I);
    Registration<GICPFactor, ParallelReductionHIP> lean;  // Reduction slot only
    lean.reduction.sync_inliers = false;                  // -> a one-element stub instead of the vector of per-point factors
    const RegistrationResult r1 = lean.align(*target, *source, tree, I);
    const auto uploads = lean.reduction.generation();
    const RegistrationResult r2 = lean.align(*target, *source, tree, I);
    double dt, dr;
    pose_error(rc.T_target_source, r2.T_target_source, &dt, &dr);
    bool ok = dt < 1e-4 && dr < 1e-4 && rc.iterations == r2.iterations && std::llabs(static_cast<long long>(rc.num_inliers) - static_cast<long long>(r2.num_inliers)) <= 2 && r1.num_inliers == r2.num_inliers &&
              lean.reduction.generation() == uploads && uploads == 2;
    // the caller's optimizer settings reach the bracketed optimizer
    Registration<GICPFactor, ParallelReductionHIP, NullFactor, DistanceRejector, GaussNewtonOptimizer> gn;
    gn.optimizer.max_iterations = 2;
    const RegistrationResult r3 = gn.align(*target, *source, tree, I);
    ok = ok && r3.iterations == 1 && !r3.converged && r3.num_inliers > 5000;
    std::printf("CASE {\"name\": \"Registration<GICPFactor, ParallelReductionHIP>: bracketed by the specialisation, stub factors with sync_inliers = false\", \"ok\": %s, \"dt\": %.3e, \"num_inliers\": [%zu, %zu], \"uploads\": %llu, \"gn_iterations\": %zu}\n",
                ok ? "true" : "false", dt, r2.num_inliers, rc.num_inliers, static_cast<unsigned long long>(uploads), r3.iterations);
    if (!ok) failures++;
  }
  // ---- VGICP (registration_helper.cpp:125-137): a GaussianVoxelMap as target AND as tree, through the policy ----
  {
    auto voxelmap = std::make_shared<GaussianVoxelMap>(1.0);
    voxelmap->insert(*target);
    auto run_vgicp = [&](const char* name, auto& hip, const Eigen::Isometry3d& init) {
      Registration<GICPFactor, ParallelReductionOMP> cpu;
      cpu.reduction.num_threads = 4;
      cpu.rejector.max_dist_sq = hip.rejector.max_dist_sq;
      const RegistrationResult rc = cpu.align(*voxelmap, *source, *voxelmap, init);
      const RegistrationResult rh = hip.align(*voxelmap, *source, *voxelmap, init);
      double dt, dr;
      pose_error(rc.T_target_source, rh.T_target_source, &dt, &dr);
      double dH = 0.0, mH = 0.0;
      for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) dH = std::max(dH, std::abs(rc.H(i, j) - rh.H(i, j))), mH = std::max(mH, std::abs(rc.H(i, j)));
      // a query within float rounding of a voxel face may fall into the other voxel (the device floors in fp32): a handful of correspondences
      const long long dinl = std::llabs(static_cast<long long>(rc.num_inliers) - static_cast<long long>(rh.num_inliers));
      const bool ok = dt < 2e-4 && dr < 2e-4 && rc.converged == rh.converged && std::llabs(static_cast<long long>(rc.iterations) - static_cast<long long>(rh.iterations)) <= 1 && dinl <= 4 && dH / mH < 1e-3;
      std::printf("CASE {\"name\": \"%s\", \"ok\": %s, \"voxels\": %zu, \"dt\": %.3e, \"dr\": %.3e, \"iterations\": [%zu, %zu], \"num_inliers\": [%zu, %zu], \"rel_err_H\": %.3e}\n", name, ok ? "true" : "false", voxelmap->size(), dt,
                  dr, rh.iterations, rc.iterations, rh.num_inliers, rc.num_inliers, dH / mH);
      if (!ok) failures++;
    };
    Eigen::Isometry3d near = Eigen::Isometry3d::Identity();
    near.matrix()(0, 3) = 0.1, near.matrix()(1, 3) = -0.05;
    {
      Registration<GICPFactor, ParallelReductionHIP> hip;
      run_vgicp("VGICP: GaussianVoxelMap target, Reduction slot only", hip, I);
      // the host factors carry the reference's packed indices: voxel id << 32 (incremental_voxelmap.hpp:153), a voxel of the caller's map
      std::vector<GICPFactor> factors(source->size());
      hip.reduction.linearize(*voxelmap, *source, *voxelmap, hip.rejector, I, factors);
      std::vector<GICPFactor> want(source->size());
      ParallelReductionOMP omp;
      omp.linearize(*voxelmap, *source, *voxelmap, hip.rejector, I, want);
      size_t differ = 0, inl = 0;
      for (size_t i = 0; i < factors.size(); i++) {
        differ += factors[i].target_index != want[i].target_index;
        inl += want[i].target_index != std::numeric_limits<size_t>::max();
      }
      const bool ok = differ <= 2 && inl > source->size() / 2;
      std::printf("CASE {\"name\": \"VGICP: packed voxel indices of the host factors\", \"ok\": %s, \"differ\": %zu, \"inliers\": %zu}\n", ok ? "true" : "false", differ, inl);
      if (!ok) failures++;
    }
    {
      Registration<GICPFactor, ParallelReductionHIP, NullFactor, DistanceRejector, HipAligned<LevenbergMarquardtOptimizer>> hip;
      run_vgicp("VGICP: HipAligned<LM>, displaced start", hip, near);
      // a map that grows (the odometry use, odometry_benchmark_small_vgicp.cpp:41-43): the policy sees the new voxels and uploads the map again
      const auto before = hip.reduction.generation();
      voxelmap->insert(*source, Eigen::Isometry3d::Identity());
      run_vgicp("VGICP: after inserting the source into the map", hip, near);
      const bool ok = hip.reduction.generation() > before;
      std::printf("CASE {\"name\": \"VGICP: a grown map was uploaded again\", \"ok\": %s}\n", ok ? "true" : "false");
      if (!ok) failures++;
    }
    {
      Registration<GICPFactor, ParallelReductionHIP> hip;
      hip.reduction.num_gpus = 2;
      run_vgicp("VGICP: num_gpus = 2", hip, I);
      // incremental_voxelmap.hpp:99-119,157-186: the Gaussians of 7 / 27 voxels around the query's own compete, the nearest mean wins
      for (const int offsets : {7, 27}) {
        voxelmap->set_search_offsets(1);  // (27 APPENDS to what is there, incremental_voxelmap.hpp:176-184: from the default that makes the 28 entries the device's 27-voxel search follows)
        voxelmap->set_search_offsets(offsets);
        Registration<GICPFactor, ParallelReductionHIP> wide;
        run_vgicp(offsets == 7 ? "VGICP: search_offsets = 7" : "VGICP: search_offsets = 27", wide, near);
      }
      voxelmap->set_search_offsets(1);
    }
  }
  // ---- scan-to-model GICP (odometry_benchmark_small_gicp_model_omp.cpp:20-47): IncrementalVoxelMap<FlatContainerCov> (linear iVox) as target AND tree ----
  {
    using FlatMap = IncrementalVoxelMap<FlatContainerCov>;
    for (int offsets : {1, 7, 27}) {
      auto model = std::make_shared<FlatMap>(1.0);
      model->set_search_offsets(offsets);
      model->insert(*target);
      Registration<GICPFactor, ParallelReductionOMP> cpu;
      cpu.reduction.num_threads = 4;
      Registration<GICPFactor, ParallelReductionHIP, NullFactor, DistanceRejector, HipAligned<LevenbergMarquardtOptimizer>> hip;
      hip.rejector.max_dist_sq = cpu.rejector.max_dist_sq;
      const RegistrationResult rc = cpu.align(*model, *source, *model, I);
      const RegistrationResult rh = hip.align(*model, *source, *model, I);
      double dt, dr;
      pose_error(rc.T_target_source, rh.T_target_source, &dt, &dr);
      // the packed (voxel, point) indices of the host factors against the CPU reduction's, at the start pose
      std::vector<GICPFactor> got(source->size()), want(source->size());
      Registration<GICPFactor, ParallelReductionHIP> plain;
      plain.reduction.linearize(*model, *source, *model, plain.rejector, I, got);
      ParallelReductionOMP omp;
      omp.linearize(*model, *source, *model, plain.rejector, I, want);
      size_t differ = 0, inl = 0;
      for (size_t i = 0; i < got.size(); i++) differ += got[i].target_index != want[i].target_index, inl += want[i].target_index != std::numeric_limits<size_t>::max();
      const long long dinl = std::llabs(static_cast<long long>(rc.num_inliers) - static_cast<long long>(rh.num_inliers));
      const bool ok = dt < 2e-4 && dr < 2e-4 && rc.converged == rh.converged && std::llabs(static_cast<long long>(rc.iterations) - static_cast<long long>(rh.iterations)) <= 1 && dinl <= 4 && differ <= 4 && inl > source->size() / 2;
      std::printf("CASE {\"name\": \"scan-to-model GICP: IncrementalVoxelMap<FlatContainerCov>, search_offsets %d\", \"ok\": %s, \"voxels\": %zu, \"dt\": %.3e, \"dr\": %.3e, \"iterations\": [%zu, %zu], \"num_inliers\": [%zu, %zu], \"indices_differing\": %zu}\n",
                  offsets, ok ? "true" : "false", model->size(), dt, dr, rh.iterations, rc.iterations, rh.num_inliers, rc.num_inliers, differ);
      if (!ok) failures++;
    }
    {  // ICP against a map of bare points (FlatContainerPoints), and a map whose voxels hold too many points for the device
      auto model = std::make_shared<IncrementalVoxelMap<FlatContainerPoints>>(1.0);
      model->insert(*target);
      Registration<ICPFactor, ParallelReductionOMP> cpu;
      cpu.reduction.num_threads = 4;
      cpu.rejector.max_dist_sq = 0.25;
      Registration<ICPFactor, ParallelReductionHIP> hip;
      hip.rejector.max_dist_sq = 0.25;
      const RegistrationResult rc = cpu.align(*model, *source, *model, I);
      const RegistrationResult rh = hip.align(*model, *source, *model, I);
      double dt, dr;
      pose_error(rc.T_target_source, rh.T_target_source, &dt, &dr);
      const bool ok = dt < 2e-4 && dr < 2e-4 && std::llabs(static_cast<long long>(rc.num_inliers) - static_cast<long long>(rh.num_inliers)) <= 4;
      std::printf("CASE {\"name\": \"scan-to-model ICP: IncrementalVoxelMap<FlatContainerPoints>\", \"ok\": %s, \"dt\": %.3e, \"dr\": %.3e, \"num_inliers\": [%zu, %zu]}\n", ok ? "true" : "false", dt, dr, rh.num_inliers, rc.num_inliers);
      if (!ok) failures++;
      auto crowded = std::make_shared<IncrementalVoxelMap<FlatContainerPoints>>(2.0);
      crowded->voxel_setting.max_num_points_in_cell = 40;
      crowded->voxel_setting.min_sq_dist_in_cell = 1e-6;
      crowded->insert(*target);
      bool threw = false;
      try {
        hip.align(*crowded, *source, *crowded, I);
      } catch (const std::exception&) {
        threw = true;
      }
      std::printf("CASE {\"name\": \"a FlatContainer voxel with more than 16 points is refused\", \"ok\": %s}\n", threw ? "true" : "false");
      if (!threw) failures++;
    }
  }
  // ---- a user-defined CorrespondenceRejector (rejector.hpp:11-28 is a duck-typed slot; example 03_registration_template.cpp): decides on the host
  {
    struct EveryOtherRejector {
      double max_dist_sq = 0.5;
      bool operator()(const PointCloud&, const PointCloud&, const Eigen::Isometry3d&, size_t target_index, size_t source_index, double sq_dist) const {
        return sq_dist > max_dist_sq || ((target_index + source_index) % 3 == 0);
      }
    };
    Registration<GICPFactor, ParallelReductionOMP, NullFactor, EveryOtherRejector> cpu;
    cpu.reduction.num_threads = 4;
    Registration<GICPFactor, ParallelReductionHIP, NullFactor, EveryOtherRejector> hip;
    const RegistrationResult rc = cpu.align(*target, *source, tree, I);
    const RegistrationResult rh = hip.align(*target, *source, tree, I);
    double dt, dr;
    pose_error(rc.T_target_source, rh.T_target_source, &dt, &dr);
    const bool ok = dt < 1e-4 && dr < 1e-4 && rc.iterations == rh.iterations && std::llabs(static_cast<long long>(rc.num_inliers) - static_cast<long long>(rh.num_inliers)) <= 2 && rh.num_inliers < source->size() * 3 / 4;
    std::printf("CASE {\"name\": \"user-defined rejector type through the batch callback\", \"ok\": %s, \"dt\": %.3e, \"dr\": %.3e, \"iterations\": [%zu, %zu], \"num_inliers\": [%zu, %zu]}\n", ok ? "true" : "false", dt, dr, rh.iterations,
                rc.iterations, rh.num_inliers, rc.num_inliers);
    if (!ok) failures++;
    // the built-in rejector afterwards: the callback is gone again
    Registration<GICPFactor, ParallelReductionHIP> plain;
    run_case("GICP after a custom rejector (built-in rejector restored)", *target, *source, tree, plain, I);
  }
  // ---- short-lived threads: a thread that ends takes its device state along
  {
    Registration<GICPFactor, ParallelReductionHIP> reg;
    reg.align(*target, *source, tree, I);
    const size_t before = reg.reduction.device_states();
    for (int r = 0; r < 3; r++) {
      std::thread t([&] { reg.align(*target, *source, tree, I); });
      t.join();
    }
    const bool ok = before == 1 && reg.reduction.device_states() == 1;
    std::printf("CASE {\"name\": \"device states of ended threads are released\", \"ok\": %s, \"states\": [%zu, %zu]}\n", ok ? "true" : "false", before, reg.reduction.device_states());
    if (!ok) failures++;
  }
  // ---- the reference's coordinate range (points/point_cloud.hpp:69-71: Vector4d): C1 moved kilometres from the origin.  The CPU reduction
  // works on the doubles; the policy subtracts the cloud's origin in double while it repacks (hip_detail::pack) and the device keeps fp32
  // records relative to it.  Error measured AT THE DATA (displacement of the source's centre, angle of the relative rotation): in the
  // translation column of T a rotation error appears multiplied by the distance from the origin.
  for (int far = 0; far < 2; far++) {
    const Eigen::Vector4d shift = far ? Eigen::Vector4d(1e5, 2e5, 300.0, 0.0) : Eigen::Vector4d(1e4, -1e4, 50.0, 0.0);
    auto tgt2 = std::make_shared<PointCloud>(*target);
    auto src2 = std::make_shared<PointCloud>(*source);
    Eigen::Vector4d centre = Eigen::Vector4d::Zero();
    for (size_t i = 0; i < tgt2->size(); i++) tgt2->point(i) += shift;
    for (size_t i = 0; i < src2->size(); i++) {
      src2->point(i) += shift;
      centre += src2->point(i);
    }
    centre /= static_cast<double>(src2->size());
    centre[3] = 1.0;
    KdTree<PointCloud> tree2(tgt2, KdTreeBuilderOMP(4));
    Registration<GICPFactor, ParallelReductionOMP> cpu;
    cpu.reduction.num_threads = 1;  // (a fixed summation order: at these condition numbers the order of the per-thread sums changes the LM path)
  for (int math64 = 0; math64 <= far; math64++) {  // the far cloud also with fp64 per-pair arithmetic
    Registration<GICPFactor, ParallelReductionHIP> hip;
    hip.reduction.fp64_math = math64 != 0;
    const RegistrationResult rc = cpu.align(*tgt2, *src2, tree2, I);
    const RegistrationResult rh = hip.align(*tgt2, *src2, tree2, I);
    const double dt = ((rc.T_target_source * centre) - (rh.T_target_source * centre)).norm();
    const Eigen::Matrix3d R = rc.T_target_source.linear().transpose() * rh.T_target_source.linear();
    const double dr = std::asin(std::min(1.0, 0.5 * std::sqrt((R(2, 1) - R(1, 2)) * (R(2, 1) - R(1, 2)) + (R(0, 2) - R(2, 0)) * (R(0, 2) - R(2, 0)) + (R(1, 0) - R(0, 1)) * (R(1, 0) - R(0, 1)))));
    double lt, lr;
    pose_error(rc.T_target_source, rh.T_target_source, &lt, &lr);
    // iteration counts: equal while the caller-frame normal equations are still well enough conditioned for the reference's own LM to be
    // reproducible (1.4e4 m); at 2.2e5 m its steps carry millimetres of solve noise and the count depends on it (tests/test_coordinate_range.py)
    // With fp32 per-pair arithmetic at 2.2e5 m the noise of a step (1e-7 of H, times the 2.2e5 m lever arm of the caller-frame twist) is
    // several millimetres — ABOVE translation_eps: whether and when a step passes the termination test is luck (round 5: 9 iterations,
    // round 6: 19 without the flag, on the same poses to 2e-6 m), so only the pose at the data and the inliers are asserted there; the
    // fp64 arithmetic (1e-8 rad of solve noise, like the reference's own) must stop within 3 iterations of the reference and agree on the flag.
    const long long dit = std::llabs(static_cast<long long>(rc.iterations) - static_cast<long long>(rh.iterations));
    // (far + fp64: the reference's LM ends either by the termination test or by an outer iteration whose trial steps all fail to lower the
    // error — at this condition number which of the two happens first is a matter of the last bits too: the count, not the flag)
    const bool counts = far ? (math64 ? dit <= 3 : true) : (rc.converged == rh.converged && dit == 0);
    const bool ok = dt < 1e-4 && dr < 1e-4 && counts && std::llabs(static_cast<long long>(rc.num_inliers) - static_cast<long long>(rh.num_inliers)) <= 2;
    std::printf("CASE {\"name\": \"GICP, clouds moved by (%g, %g, %g)%s\", \"ok\": %s, \"dt_at_the_data\": %.3e, \"dr\": %.3e, \"dt_of_T\": %.3e, \"iterations\": [%zu, %zu], \"num_inliers\": [%zu, %zu], \"converged\": [%d, %d]}\n", shift[0], shift[1], shift[2],
                math64 ? ", fp64 arithmetic" : "", ok ? "true" : "false", dt, dr, lt, rh.iterations, rc.iterations, rh.num_inliers, rc.num_inliers, rh.converged ? 1 : 0, rc.converged ? 1 : 0);
    if (!ok) failures++;
  }
  }
  // ---- iterations/s THROUGH the policy (what a small_gicp user who swaps the Reduction gets), default settings of the policy ----
  using Plain = Registration<GICPFactor, ParallelReductionHIP>;
  using Aligned = Registration<GICPFactor, ParallelReductionHIP, NullFactor, DistanceRejector, HipAligned<LevenbergMarquardtOptimizer>>;
  rate<Plain>("C1 (downsampled 0.25 m)", *target, *source, tree, 50);
  rate<Aligned>("C1, HipAligned<LM>", *target, *source, tree, 50);
  {
    // ~100k-point clouds: a ground plane and two walls, sampled twice; the source displaced by a small rigid motion
    auto make = [&](unsigned seed, const Eigen::Isometry3d& M) {
      std::mt19937 rng(seed);
      std::uniform_real_distribution<double> u(0.0, 1.0);
      std::normal_distribution<double> noise(0.0, 0.01);
      std::vector<Eigen::Vector4d> pts;
      for (int i = 0; i < 260000; i++) {
        const double a = u(rng), b = u(rng);
        Eigen::Vector4d p;
        if (i % 10 < 6) p = Eigen::Vector4d(60 * a - 30, 60 * b - 30, 0, 1);
        else if (i % 10 < 8) p = Eigen::Vector4d(60 * a - 30, 12, 8 * b, 1);
        else p = Eigen::Vector4d(-15, 60 * a - 30, 8 * b, 1);
        for (int k = 0; k < 3; k++) p[k] += noise(rng);
        pts.push_back(M * p);
      }
      return preprocess(pts);
    };
    Eigen::Isometry3d M = Eigen::Isometry3d::Identity();
    M.matrix()(0, 0) = std::cos(0.01), M.matrix()(0, 1) = -std::sin(0.01), M.matrix()(1, 0) = std::sin(0.01), M.matrix()(1, 1) = std::cos(0.01);
    M.matrix()(0, 3) = 0.1, M.matrix()(1, 3) = -0.05;
    auto big_t = make(1, Eigen::Isometry3d::Identity());
    auto big_s = make(2, M.inverse());
    KdTree<PointCloud> big_tree(big_t, KdTreeBuilderOMP(8));
    {
      // clouds large enough for the DEFERRED content check (the hash runs beside the registration on the cached upload): unchanged objects are
      // not uploaded again; a source refilled in place is noticed at the end of the bracket and registered again from what the caller holds now
      Registration<GICPFactor, ParallelReductionHIP> hip;
      Registration<GICPFactor, ParallelReductionOMP> cpu;
      cpu.reduction.num_threads = 16;
      const RegistrationResult h1 = hip.align(*big_t, *big_s, big_tree, I);
      const auto uploads = hip.reduction.generation();
      const RegistrationResult h2 = hip.align(*big_t, *big_s, big_tree, I);
      bool ok = hip.reduction.generation() == uploads && (h1.T_target_source.matrix() - h2.T_target_source.matrix()).norm() == 0.0;
      for (size_t i = 0; i < big_s->size(); i++) big_s->point(i)[0] = static_cast<float>(big_s->point(i)[0] + 0.05);
      const RegistrationResult h3 = hip.align(*big_t, *big_s, big_tree, I);
      const RegistrationResult c3 = cpu.align(*big_t, *big_s, big_tree, I);
      double dt, dr;
      pose_error(c3.T_target_source, h3.T_target_source, &dt, &dr);
      ok = ok && hip.reduction.generation() == uploads + 1 && dt < 1e-4 && dr < 1e-4 && c3.iterations == h3.iterations && std::abs(h3.T_target_source.translation()[0] - h1.T_target_source.translation()[0] + 0.05) < 5e-3;
      for (size_t i = 0; i < big_s->size(); i++) big_s->point(i)[0] = static_cast<float>(big_s->point(i)[0] - 0.05);
      std::printf("CASE {\"name\": \"deferred content check: unchanged clouds stay, a refilled source is registered again\", \"ok\": %s, \"dt\": %.3e, \"dr\": %.3e, \"iterations\": [%zu, %zu], \"uploads\": [%llu, %llu]}\n", ok ? "true" : "false", dt, dr,
                  h3.iterations, c3.iterations, static_cast<unsigned long long>(uploads), static_cast<unsigned long long>(hip.reduction.generation()));
      if (!ok) failures++;
    }
    rate<Plain>("synthetic planes (~100k after 0.25 m voxel grid)", *big_t, *big_s, big_tree, 20);
    rate<Plain>("synthetic planes, verify_content = sync_inliers = false", *big_t, *big_s, big_tree, 20, true);
    rate<Aligned>("synthetic planes, HipAligned<LM>", *big_t, *big_s, big_tree, 20);
  }
  std::printf("DONE failures=%d\n", failures);
  return failures == 0 ? 0 : 1;
}
