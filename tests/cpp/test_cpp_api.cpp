// Exercises the header-only C++ layer (include/small_gicp_amd.hpp) the way the reference's own C++ users do
// (src/example/01_basic_registration.cpp, 03_registration_template.cpp; src/test/helper_test.cpp:94-157).
// usage: test_cpp_api target.f32 source.f32     (raw float32 xyz triples)   -> one "CASE ..." line per registration on stdout
#include <cstdio>
#include <fstream>
#include <vector>

#include "small_gicp_amd.hpp"

using namespace small_gicp_amd;

static std::vector<std::array<float, 3>> read_f32(const char* path) {
  std::ifstream ifs(path, std::ios::binary | std::ios::ate);
  if (!ifs) throw std::runtime_error(std::string("cannot open ") + path);
  const size_t bytes = ifs.tellg();
  std::vector<std::array<float, 3>> pts(bytes / 12);
  ifs.seekg(0);
  ifs.read(reinterpret_cast<char*>(pts.data()), pts.size() * 12);
  return pts;
}

static void report(const char* name, const RegistrationResult& r) {
  std::printf("CASE %s %zu %zu %d %.10g", name, r.iterations, r.num_inliers, r.converged ? 1 : 0, r.error);
  for (int i = 0; i < 16; i++) std::printf(" %.12g", r.T_target_source.m[i]);
  std::printf("\n");
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  try {
    const auto target_raw = read_f32(argv[1]);
    const auto source_raw = read_f32(argv[2]);

    // 1. helper: raw points in (registration_helper.cpp:57-69)
    RegistrationSetting setting;
    setting.type = RegistrationSetting::GICP;
    report("HELPER_GICP", align(target_raw, source_raw, Isometry3d::Identity(), setting));
    setting.type = RegistrationSetting::VGICP;
    report("HELPER_VGICP", align(target_raw, source_raw, Isometry3d::Identity(), setting));

    // 2. explicit pipeline + Registration<> template (03_registration_template.cpp)
    auto [target, target_tree] = preprocess_points(target_raw, 0.25, 10);
    auto [source, source_tree] = preprocess_points(source_raw, 0.25, 10);
    std::printf("SIZES %zu %zu\n", target->size(), source->size());
    {
      Registration<GICPFactor, ParallelReductionHIP> reg;
      report("GICP", reg.align(*target, *source, *target_tree));
    }
    {
      Registration<PointToPlaneICPFactor, ParallelReductionHIP> reg;
      report("PLANE_ICP", reg.align(*target, *source, *target_tree));
    }
    {
      Registration<ICPFactor, ParallelReductionHIP> reg;
      report("ICP", reg.align(*target, *source, *target_tree));
    }
    {
      Registration<RobustFactor<Huber, GICPFactor>, ParallelReductionHIP> reg;
      reg.point_factor.robust_kernel.c = 1.0;
      report("HUBER_GICP", reg.align(*target, *source, *target_tree));
    }
    {
      Registration<RobustFactor<Cauchy, GICPFactor>, ParallelReductionHIP> reg;
      report("CAUCHY_GICP", reg.align(*target, *source, *target_tree));
    }
    {
      Registration<GICPFactor, ParallelReductionHIP, NullFactor, DistanceRejector, GaussNewtonOptimizer> reg;
      report("GN_GICP", reg.align(*target, *source, *target_tree));
    }
    {
      Registration<GICPFactor, ParallelReductionHIP, RestrictDoFFactor> reg;
      reg.general_factor.set_rotation_mask(0.0, 0.0, 1.0);  // yaw only
      reg.general_factor.set_translation_mask(1.0, 1.0, 0.0);  // no z
      report("RESTRICT_GICP", reg.align(*target, *source, *target_tree));
    }
    {
      auto voxelmap = create_gaussian_voxelmap(*target, 1.0);
      std::printf("VOXELS %zu\n", voxelmap->size());
      RegistrationSetting s;
      s.type = RegistrationSetting::VGICP;
      report("VGICP", align(*voxelmap, *source, Isometry3d::Identity(), s));
    }
    {
      // scan-to-model GICP target (odometry_benchmark_small_gicp_model_omp.cpp:24-40)
      IncrementalVoxelMap<FlatContainerCov> model(1.0);
      model.set_search_offsets(7);
      model.insert(*target);
      std::printf("MODEL_VOXELS %zu\n", model.size());
      Registration<GICPFactor, ParallelReductionHIP> reg;
      report("MODEL_GICP", reg.align(model, *source, model));
    }
    // 3. accessors
    const auto p0 = target->point(0);
    const auto c0 = target->cov(0);
    size_t idx;
    double d2;
    const size_t found = target_tree->nearest_neighbor_search(p0.data(), &idx, &d2);
    std::printf("ACCESS %zu %zu %.6g %.6g %.6g\n", found, idx, d2, p0[3], c0[0] + c0[5] + c0[10]);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
