// bench.py's `policy_c3` / `policy_c2` legs: the benchmark clouds through the REFERENCE's own Registration<>::align with the MI355X
// policy in its Reduction slot and HipAligned<> in its Optimizer slot (include/small_gicp/registration/reduction_hip.hpp), compiled
// from the UNMODIFIED reference headers over the Eigen stand-in of oracle/ref/eigen_shim (oracle/ref/Makefile -> oracle/_ref/policy_bench;
// the binary travels to the GPU box, /root/reference does not).
//
//   policy_bench <kind: GICP|PLANE_ICP|VGICP> <target.f32> <source.f32> <target_attr.f32> <source_attr.f32> <reps> [num_gpus]
// points: n x 3 float32; attr: n x 6 float32 covariances (xx xy xz yy yz zz) for GICP / VGICP, n x 3 normals for PLANE_ICP.
// VGICP (registration_helper.cpp:125-137): the target is the reference's own GaussianVoxelMap (0.5 m voxels, built here on the host by the
// reference's insert()) in the target AND the tree slot.
// Prints ONE json line: whole-align and inside-the-optimizer iteration rates, what the bracket costs, the pose of the last align.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include <small_gicp/ann/gaussian_voxelmap.hpp>
#include <small_gicp/factors/gicp_factor.hpp>
#include <small_gicp/factors/plane_icp_factor.hpp>
#include <small_gicp/points/point_cloud.hpp>
#include <small_gicp/registration/registration.hpp>

#include <small_gicp/registration/reduction_hip.hpp>

using namespace small_gicp;

static std::vector<float> read_f32(const char* path) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) {
    std::fprintf(stderr, "cannot read %s\n", path);
    std::exit(2);
  }
  const size_t bytes = static_cast<size_t>(f.tellg());
  f.seekg(0);
  std::vector<float> raw(bytes / 4);
  f.read(reinterpret_cast<char*>(raw.data()), static_cast<std::streamsize>(bytes));
  return raw;
}

static PointCloud make_cloud(const std::vector<float>& xyz, const std::vector<float>& attr, bool covs) {
  const size_t n = xyz.size() / 3;
  PointCloud c;
  c.resize(n);
  for (size_t i = 0; i < n; i++) {
    c.point(i) = Eigen::Vector4d(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], 1.0);
    if (covs) {
      const float* m = &attr[6 * i];
      Eigen::Matrix4d C = Eigen::Matrix4d::Zero();
      C(0, 0) = m[0], C(0, 1) = C(1, 0) = m[1], C(0, 2) = C(2, 0) = m[2], C(1, 1) = m[3], C(1, 2) = C(2, 1) = m[4], C(2, 2) = m[5];
      c.cov(i) = C;
    } else {
      c.normal(i) = Eigen::Vector4d(attr[3 * i], attr[3 * i + 1], attr[3 * i + 2], 0.0);
    }
  }
  return c;
}

struct NoTree {};  // the policy searches its own index on the device; align() only forwards the tree argument

template <typename Factor, typename Target, typename Tree>
static int run(const Target& target, const PointCloud& source, const Tree& tree, int reps, int num_gpus, const char* kind) {
  using Aligned = Registration<Factor, ParallelReductionHIP, NullFactor, DistanceRejector, HipAligned<LevenbergMarquardtOptimizer>>;
  const Eigen::Isometry3d I = Eigen::Isometry3d::Identity();
  Aligned reg;
  reg.reduction.num_gpus = num_gpus;
  reg.rejector.max_dist_sq = 1.0;
  reg.criteria.rotation_eps = 0.0;  // fixed number of LM iterations, like bench.py's headline
  reg.criteria.translation_eps = 0.0;
  reg.optimizer.max_iterations = 10;
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
  // first align: hash + repack + upload + index build + problem (the once-per-cloud cost), then a warm-up
  auto t0 = now();
  RegistrationResult res = reg.align(target, source, tree, I);
  const double first_s = secs(t0, now());
  const double first_bind_s = std::get<0>(reg.reduction.last_bracket_seconds());
  reg.align(target, source, tree, I);
  size_t iters = 0;
  double bind_s = 0.0, loop_s = 0.0, fill_s = 0.0, calls_s = 0.0;
  t0 = now();
  std::vector<double> each_ms;
  for (int r = 0; r < reps; r++) {
    const auto a0 = now();
    res = reg.align(target, source, tree, I);
    each_ms.push_back(1e3 * secs(a0, now()));
    iters += res.iterations + 1;
    const auto b = reg.reduction.last_bracket_seconds();
    bind_s += std::get<0>(b), loop_s += std::get<1>(b), fill_s += std::get<2>(b), calls_s += std::get<3>(b);
  }
  const double total_s = secs(t0, now());
  std::vector<double> sorted_ms = each_ms;
  std::sort(sorted_ms.begin(), sorted_ms.end());
  const double median_ms = sorted_ms[sorted_ms.size() / 2];
  // what the reference's align() itself spends per call before the optimizer runs: std::vector<Factor>(n) (registration.hpp:41)
  t0 = now();
  {
    std::vector<Factor> factors(source.size(), Factor(typename Factor::Setting()));
    asm volatile("" ::"r"(factors.data()) : "memory");
  }
  const double factors_s = secs(t0, now());
  // the lean bracket: no content check (the caller promises not to edit clouds in place), no host factors (num_inliers from the device;
  // the Registration<> specialisation then does not even create the vector of per-point factors)
  Aligned lean;
  lean.reduction.num_gpus = num_gpus;
  lean.reduction.verify_content = lean.reduction.sync_inliers = false;
  lean.rejector.max_dist_sq = 1.0;
  lean.criteria = reg.criteria;
  lean.optimizer.max_iterations = 10;
  lean.align(target, source, tree, I);
  lean.align(target, source, tree, I);
  size_t liters = 0;
  double lean_loop_s = 0.0, lean_calls_s = 0.0;
  t0 = now();
  for (int r = 0; r < reps; r++) {
    const RegistrationResult lr = lean.align(target, source, tree, I);
    liters += lr.iterations + 1;
    lean_loop_s += std::get<1>(lean.reduction.last_bracket_seconds());
    lean_calls_s += std::get<3>(lean.reduction.last_bracket_seconds());
  }
  const double lean_total_s = secs(t0, now());
  // the same clouds with the policy in the Reduction slot only: Registration<Factor, ParallelReductionHIP> (the specialisation in reduction_hip.hpp brackets it like HipAligned<>)
  Registration<Factor, ParallelReductionHIP> plain;
  plain.reduction.num_gpus = num_gpus;
  plain.rejector.max_dist_sq = 1.0;
  plain.criteria = reg.criteria;
  plain.optimizer.max_iterations = 10;
  plain.align(target, source, tree, I);
  t0 = now();
  const size_t piters = plain.align(target, source, tree, I).iterations + 1;
  const double plain_s = secs(t0, now());
  std::printf(
    "POLICY {\"kind\": \"%s\", \"points\": [%zu, %zu], \"num_gpus\": %d, \"reps\": %d, \"iterations\": %zu, \"first_align_s\": %.4f, \"first_bind_s\": %.4f, "
    "\"whole_align_iterations_per_s\": %.1f, \"inside_the_optimizer_iterations_per_s\": %.1f, \"policy_calls_iterations_per_s\": %.1f, \"per_align_ms\": {\"total\": %.3f, \"content_check\": %.3f, \"optimizer\": %.3f, "
    "\"optimizer_policy_calls\": %.3f, \"factor_fill\": %.3f, \"reference_factor_vector\": %.3f}, \"lean\": {\"whole_align_iterations_per_s\": %.1f, \"inside_the_optimizer_iterations_per_s\": %.1f, "
    "\"policy_calls_iterations_per_s\": %.1f}, \"reduction_slot_only_iterations_per_s\": %.1f, \"num_inliers\": %zu, \"converged\": %d, \"whole_align_median_iterations_per_s\": %.1f, ",
    kind, target.size(), source.size(), num_gpus, reps, iters, first_s, first_bind_s, iters / total_s, iters / loop_s, iters / calls_s, 1e3 * total_s / reps, 1e3 * bind_s / reps, 1e3 * loop_s / reps, 1e3 * calls_s / reps,
    1e3 * fill_s / reps, 1e3 * factors_s, liters / lean_total_s, liters / lean_loop_s, liters / lean_calls_s, piters / plain_s, res.num_inliers, res.converged ? 1 : 0,
    (static_cast<double>(iters) / reps) / (1e-3 * median_ms));
  std::printf("\"align_ms\": [");
  for (size_t k = 0; k < each_ms.size(); k++) std::printf("%s%.3f", k ? ", " : "", each_ms[k]);
  std::printf("], \"T\": [");
  for (int k = 0; k < 16; k++) std::printf("%s%.12g", k ? ", " : "", res.T_target_source.matrix().data()[k]);
  std::printf("]}\n");
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 7) {
    std::fprintf(stderr, "usage: %s GICP|PLANE_ICP target.f32 source.f32 target_attr.f32 source_attr.f32 reps [num_gpus]\n", argv[0]);
    return 2;
  }
  const std::string kind = argv[1];
  const bool gicp = kind == "GICP" || kind == "VGICP";
  const PointCloud target = make_cloud(read_f32(argv[2]), read_f32(argv[4]), gicp);
  const PointCloud source = make_cloud(read_f32(argv[3]), read_f32(argv[5]), gicp);
  const int reps = std::atoi(argv[6]);
  const int num_gpus = argc > 7 ? std::atoi(argv[7]) : 1;
  try {
    const NoTree tree;
    if (kind == "VGICP") {
      GaussianVoxelMap voxelmap(0.5);
      voxelmap.insert(target);
      return run<GICPFactor>(voxelmap, source, voxelmap, reps, num_gpus, "VGICP");
    }
    return gicp ? run<GICPFactor>(target, source, tree, reps, num_gpus, "GICP") : run<PointToPlaneICPFactor>(target, source, tree, reps, num_gpus, "PLANE_ICP");
  } catch (const std::exception& e) {
    std::fprintf(stderr, "policy_bench: %s\n", e.what());
    return 1;
  }
}
