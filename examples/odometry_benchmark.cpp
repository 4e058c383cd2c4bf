// Scan-to-scan GICP odometry over the C++ layer (include/small_gicp_amd.hpp) — the protocol of the reference's benchmark driver and its
// small_gicp_omp engine (src/benchmark/odometry_benchmark.cpp, src/benchmark/odometry_benchmark_small_gicp_omp.cpp:16-49,
// include/small_gicp/benchmark/benchmark_odom.hpp:49-82), with Registration<GICPFactor, ParallelReductionHIP> in the place of
// Registration<GICPFactor, ParallelReductionOMP>:
//
//   all scans are read into host memory first (KittiDataset, benchmark.hpp:96-115: the *.bin files of a directory in name order);
//   per scan:  voxelgrid_sampling(downsampling_resolution)                       -> counted in the total throughput only
//              KdTree + estimate_covariances(num_neighbors)                      -> registration time
//              align(previous scan, this scan, previous tree, Identity); T_world_lidar *= T_target_source   -> registration time
//   output: one pose per scan, 12 numbers per line (the first three rows of T_world_lidar), "%.6f".
//
// usage: odometry_benchmark <dataset_path> <output_path> [--num_neighbors 20] [--downsampling_resolution 0.25]
//                           [--max_correspondence_distance 1.0] [--max_frames N]
// Build:  g++ -O2 -std=c++17 -Iinclude examples/odometry_benchmark.cpp -o odometry_benchmark -Lsmall_gicp_amd/lib -lsmall_gicp_amd
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "small_gicp_amd.hpp"

namespace {

struct Params {
  int num_neighbors = 20;
  double downsampling_resolution = 0.25;
  double max_correspondence_distance = 1.0;
  size_t max_frames = 1000000;
};

// mean +- standard deviation of a series of milliseconds (the reference prints its Summarizer the same way)
struct Series {
  std::vector<double> v;
  void push(double x) { v.push_back(x); }
  double mean(size_t skip = 0) const {
    if (v.size() <= skip) return NAN;
    double s = 0;
    for (size_t i = skip; i < v.size(); i++) s += v[i];
    return s / static_cast<double>(v.size() - skip);
  }
  double stddev(size_t skip = 0) const {
    if (v.size() <= skip + 1) return 0.0;
    const double m = mean(skip);
    double s = 0;
    for (size_t i = skip; i < v.size(); i++) s += (v[i] - m) * (v[i] - m);
    return std::sqrt(s / static_cast<double>(v.size() - skip - 1));
  }
};

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// KITTI velodyne file: float32 x, y, z, reflectance per point -> xyz
std::vector<float> read_scan(const std::string& path) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) throw std::runtime_error("cannot open " + path);
  const std::streamsize bytes = f.tellg();
  f.seekg(0);
  std::vector<float> raw(static_cast<size_t>(bytes) / sizeof(float));
  f.read(reinterpret_cast<char*>(raw.data()), static_cast<std::streamsize>(raw.size() * sizeof(float)));
  const size_t n = raw.size() / 4;
  std::vector<float> xyz(3 * n);
  for (size_t i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) xyz[3 * i + k] = raw[4 * i + k];
  return xyz;
}

// the small_gicp_omp engine, on the GPU
class OnlineOdometryEstimationHIP {
public:
  explicit OnlineOdometryEstimationHIP(const Params& p) : params(p), T_world_lidar(small_gicp_amd::Isometry3d::Identity()) {
    // every step of a scan runs on one context: no host wait between the index build, the covariances and the registration
    small_gicp_amd::check(sga_context_set_stream_ordered(small_gicp_amd::default_context(), 1), "sga_context_set_stream_ordered");
    registration.rejector.max_dist_sq = params.max_correspondence_distance * params.max_correspondence_distance;
  }
  // the mode changes what "returned" means for every user of the default context: restored when the estimator goes away
  ~OnlineOdometryEstimationHIP() {
    target_points.reset();
    target_tree.reset();
    (void)sga_context_set_stream_ordered(small_gicp_amd::default_context(), 0);
  }

  // `points` is already downsampled (odometry_benchmark_small_gicp_omp.cpp:20-21)
  small_gicp_amd::Isometry3d estimate(const small_gicp_amd::PointCloud::Ptr& points) {
    const double t0 = now_ms();
    auto tree = std::make_shared<small_gicp_amd::KdTree>(points);
    small_gicp_amd::estimate_covariances(*points, *tree, params.num_neighbors);
    if (target_points == nullptr) {  // the very first frame
      target_points = points;
      target_tree = tree;
      small_gicp_amd::check(sga_context_synchronize(points->ctx), "sga_context_synchronize");
      return T_world_lidar;
    }
    // the scan enters the registration by its own index (its kd order is spatially coherent: no sort, no copy)
    const auto result = registration.align(*target_points, *tree, *target_tree, small_gicp_amd::Isometry3d::Identity());
    T_world_lidar = T_world_lidar * result.T_target_source;
    iterations.push(static_cast<double>(result.iterations + 1));
    target_points = points;
    target_tree = tree;
    reg_times.push(now_ms() - t0);
    return T_world_lidar;
  }

  Series reg_times, iterations;

private:
  const Params params;
  small_gicp_amd::Registration<small_gicp_amd::GICPFactor, small_gicp_amd::ParallelReductionHIP> registration;
  small_gicp_amd::PointCloud::Ptr target_points;
  std::shared_ptr<small_gicp_amd::KdTree> target_tree;
  small_gicp_amd::Isometry3d T_world_lidar;
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) {
    std::cout << "USAGE: odometry_benchmark <dataset_path> <output_path> [options]\nOPTIONS:\n  --num_neighbors <value> (default: 20)\n  --downsampling_resolution <value> (default: 0.25)\n"
                 "  --max_correspondence_distance <value> (default: 1.0)\n  --max_frames <value>\n";
    return 0;
  }
  const std::string dataset_path = argv[1], output_path = argv[2];
  Params params;
  for (int i = 3; i < argc; i++) {
    const std::string arg = argv[i];
    const bool has_value = i + 1 < argc;
    if (arg == "--num_neighbors" && has_value) params.num_neighbors = std::stoi(argv[++i]);
    else if (arg == "--downsampling_resolution" && has_value) params.downsampling_resolution = std::stod(argv[++i]);
    else if (arg == "--max_correspondence_distance" && has_value) params.max_correspondence_distance = std::stod(argv[++i]);
    else if (arg == "--max_frames" && has_value) params.max_frames = static_cast<size_t>(std::stoll(argv[++i]));
    else {
      std::cerr << "unknown option: " << arg << std::endl;
      return 1;
    }
  }
  try {
    std::vector<std::string> filenames;
    for (const auto& e : std::filesystem::directory_iterator(dataset_path))
      if (e.path().extension() == ".bin") filenames.push_back(e.path().string());
    std::sort(filenames.begin(), filenames.end());
    if (filenames.size() > params.max_frames) filenames.resize(params.max_frames);
    std::vector<std::vector<float>> scans;
    size_t total_points = 0;
    for (const auto& f : filenames) {
      scans.push_back(read_scan(f));
      total_points += scans.back().size() / 3;
    }
    std::cout << "dataset_path=" << dataset_path << "\nnum_frames=" << scans.size() << "\nnum_points=" << (scans.empty() ? 0 : total_points / scans.size()) << " [points/scan, mean]\nnum_neighbors=" << params.num_neighbors
              << "\ndownsampling_resolution=" << params.downsampling_resolution << std::endl;

    OnlineOdometryEstimationHIP odom(params);
    Series total_times;
    std::vector<small_gicp_amd::Isometry3d> traj;
    for (auto& xyz : scans) {
      const double t0 = now_ms();
      small_gicp_amd::PointCloud raw(xyz.data(), nullptr, nullptr, xyz.size() / 3);
      auto downsampled = small_gicp_amd::voxelgrid_sampling(raw, params.downsampling_resolution);
      traj.push_back(odom.estimate(downsampled));
      small_gicp_amd::check(sga_context_synchronize(raw.ctx), "sga_context_synchronize");
      total_times.push(now_ms() - t0);
      std::vector<float>().swap(xyz);  // like the reference: the raw scan is released once it has been used
    }
    const size_t skip = scans.size() > 4 ? 2 : 0;  // start-up (first launches, allocator warm-up) left out of the means, stated in the output
    std::printf("registration_time_stats=%.4f +- %.4f [msec/scan]  total_throughput=%.4f +- %.4f [msec/scan]  mean_iterations=%.2f  (means without the first %zu scans)\n", odom.reg_times.mean(skip),
                odom.reg_times.stddev(skip), total_times.mean(skip + 1), total_times.stddev(skip + 1), odom.iterations.mean(), skip);
    std::ofstream ofs(output_path);
    char buf[64];
    for (const auto& T : traj) {
      for (int i = 0; i < 3; i++)
        for (int j = 0; j < 4; j++) {
          std::snprintf(buf, sizeof(buf), "%s%.6f", (i || j) ? " " : "", T(i, j));
          ofs << buf;
        }
      ofs << "\n";
    }
  } catch (const std::exception& e) {
    std::cerr << "error: " << e.what() << std::endl;
    return 2;
  }
  return 0;
}
