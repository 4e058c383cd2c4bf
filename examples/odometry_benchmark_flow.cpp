// Scan-to-scan GICP odometry as a FLOW of stages over HIP streams — the throughput protocol of the reference's TBB flow-graph engine
// (src/benchmark/odometry_benchmark_small_gicp_tbb_flow.cpp:50-141; its report is the "total throughput" of BENCHMARK.md), over the C++
// layer (include/small_gicp_amd.hpp):
//
//   preprocess_node    (:61-67, unlimited concurrency)  -> `--preprocess_workers` threads, a context (HIP stream) each, stream-ordered:
//                       upload, voxelgrid_sampling, KdTree, estimate_covariances of frame i are enqueued without a host wait
//   sequencer + pairing_node (:70-78)                    -> frame i-1 is the target of frame i
//   registration_node  (:81-97, unlimited concurrency)  -> `--registration_workers` threads, a context each: every pair is registered
//                       from the identity, so the pairs are independent; Registration<GICPFactor, ParallelReductionHIP> on the worker's context
//   sequencer + output_node (:100-110)                   -> the relative poses multiplied up in frame order
//
// A frame's chain is ~30 dependent launches that leave most of the GPU idle; chains on different streams fill it.  The poses are the
// ones the sequential driver (examples/odometry_benchmark.cpp) writes: same kernels, same order of operations per frame and per pair.
//
// usage: odometry_benchmark_flow <dataset_path> <output_path> [--num_neighbors 20] [--downsampling_resolution 0.25]
//          [--max_correspondence_distance 1.0] [--max_frames N] [--preprocess_workers 2] [--registration_workers 2] [--depth D]
//          [--pinned] [--repeat K]
//   --pinned: the scans are read into pinned host memory (sga_host_alloc): the upload kernel reads them in place, no staging pass
//   --repeat: run the whole sequence K times and report the last run (the first carries code-object loads and first allocations)
// Build:  g++ -O2 -std=c++17 -pthread -Iinclude examples/odometry_benchmark_flow.cpp -o odometry_benchmark_flow -Lsmall_gicp_amd/lib -lsmall_gicp_amd
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <exception>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "small_gicp_amd.hpp"

namespace {

struct Params {
  int num_neighbors = 20;
  double downsampling_resolution = 0.25;
  double max_correspondence_distance = 1.0;
  size_t max_frames = 1000000;
  int preprocess_workers = 2;
  int registration_workers = 2;
  int depth = 0;  // frames the preprocessing may run ahead of the registrations (0: workers + 4)
  bool pinned = false;
  int repeat = 1;
};

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// a scan in host memory: xyz, pageable or pinned
struct Scan {
  float* xyz = nullptr;
  size_t n = 0;
  bool pinned = false;
  Scan() = default;
  Scan(const Scan&) = delete;
  Scan& operator=(const Scan&) = delete;
  Scan(Scan&& o) noexcept : xyz(o.xyz), n(o.n), pinned(o.pinned) { o.xyz = nullptr; }
  ~Scan() {
    if (!xyz) return;
    if (pinned) sga_host_free(xyz);
    else delete[] xyz;
  }
};

// KITTI velodyne file: float32 x, y, z, reflectance per point -> xyz
Scan read_scan(const std::string& path, bool pinned) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) throw std::runtime_error("cannot open " + path);
  const std::streamsize bytes = f.tellg();
  f.seekg(0);
  std::vector<float> raw(static_cast<size_t>(bytes) / sizeof(float));
  f.read(reinterpret_cast<char*>(raw.data()), static_cast<std::streamsize>(raw.size() * sizeof(float)));
  Scan s;
  s.n = raw.size() / 4;
  s.pinned = pinned;
  if (pinned) {
    void* p = nullptr;
    small_gicp_amd::check(sga_host_alloc(std::max<size_t>(1, 3 * s.n) * sizeof(float), &p), "sga_host_alloc");
    s.xyz = static_cast<float*>(p);
  } else {
    s.xyz = new float[std::max<size_t>(1, 3 * s.n)];
  }
  for (size_t i = 0; i < s.n; i++)
    for (int k = 0; k < 3; k++) s.xyz[3 * i + k] = raw[4 * i + k];
  return s;
}

// InputFrame of the reference's flow graph (:18-27)
struct Frame {
  small_gicp_amd::PointCloud::Ptr points;
  std::shared_ptr<small_gicp_amd::KdTree> kdtree;
  small_gicp_amd::Isometry3d T_last_current = small_gicp_amd::Isometry3d::Identity();
  size_t iterations = 0;
  double t_start = 0.0, t_done = 0.0;
};

class FlowOdometry {
public:
  explicit FlowOdometry(const Params& p) : params(p) {
    for (int w = 0; w < std::max(1, p.preprocess_workers); w++) {
      sga_context* c = nullptr;
      small_gicp_amd::check(sga_context_create(0, &c), "sga_context_create");
      small_gicp_amd::check(sga_context_set_stream_ordered(c, 1), "sga_context_set_stream_ordered");  // a frame's chain is enqueued, not waited for
      pre_ctx.push_back(c);
    }
    for (int r = 0; r < std::max(1, p.registration_workers); r++) {
      sga_context* c = nullptr;
      small_gicp_amd::check(sga_context_create(0, &c), "sga_context_create");
      reg_ctx.push_back(c);
    }
    depth = p.depth > 0 ? p.depth : static_cast<int>(pre_ctx.size() + reg_ctx.size()) + 4;
    depth = std::max<int>(depth, static_cast<int>(reg_ctx.size()) + 1);
  }
  ~FlowOdometry() {
    for (sga_context* c : pre_ctx) sga_context_destroy(c);
    for (sga_context* c : reg_ctx) sga_context_destroy(c);
  }

  // One pass over the sequence; returns the trajectory.  wall_ms: first preprocessing call to last pose.
  std::vector<small_gicp_amd::Isometry3d> estimate(const std::vector<Scan>& scans, double& wall_ms, double& mean_iterations, double& mean_latency_ms) {
    const size_t n = scans.size();
    frames.assign(n, nullptr);
    std::vector<Frame> results(n);
    next_frame = 0;
    next_pair_to_take = 1;
    pairs_done_below = 1;
    done.assign(n + 1, 0);
    failure = nullptr;
    const double t0 = now_ms();
    std::vector<std::thread> threads;
    for (size_t w = 0; w < pre_ctx.size(); w++) threads.emplace_back([&, w] { guarded([&] { preprocess_loop(scans, pre_ctx[w]); }); });
    for (size_t r = 0; r < reg_ctx.size(); r++) threads.emplace_back([&, r] { guarded([&] { registration_loop(n, reg_ctx[r], results); }); });
    for (auto& t : threads) t.join();
    if (failure) std::rethrow_exception(failure);
    // output node: products in frame order
    std::vector<small_gicp_amd::Isometry3d> traj;
    traj.reserve(n);
    double it = 0.0, lat = 0.0;
    for (size_t i = 0; i < n; i++) {
      if (traj.empty()) traj.push_back(small_gicp_amd::Isometry3d::Identity());
      else traj.push_back(traj.back() * results[i].T_last_current);
      if (i > 0) it += static_cast<double>(results[i].iterations), lat += results[i].t_done - results[i].t_start;
    }
    wall_ms = now_ms() - t0;
    mean_iterations = n > 1 ? it / static_cast<double>(n - 1) : 0.0;
    mean_latency_ms = n > 1 ? lat / static_cast<double>(n - 1) : 0.0;
    frames.clear();
    return traj;
  }

private:
  template <typename F>
  void guarded(F&& f) {
    try {
      f();
    } catch (...) {
      std::lock_guard<std::mutex> lk(mu);
      if (!failure) failure = std::current_exception();
      cv.notify_all();
    }
  }

  void preprocess_loop(const std::vector<Scan>& scans, sga_context* ctx) {
    for (;;) {
      size_t i;
      {
        std::unique_lock<std::mutex> lk(mu);
        i = next_frame;
        if (i >= scans.size() || failure) return;
        next_frame++;
        cv.wait(lk, [&] { return i < pairs_done_below + static_cast<size_t>(depth) || failure; });  // frames in flight are bounded (device memory)
        if (failure) return;
      }
      auto f = std::make_shared<Frame>();
      f->t_start = now_ms();
      small_gicp_amd::PointCloud raw(scans[i].xyz, nullptr, nullptr, scans[i].n, ctx);
      f->points = small_gicp_amd::voxelgrid_sampling(raw, params.downsampling_resolution);  // Downsampling
      f->kdtree = std::make_shared<small_gicp_amd::KdTree>(f->points);                       // KdTree construction
      small_gicp_amd::estimate_covariances(*f->points, *f->kdtree, params.num_neighbors);    // Covariance estimation
      {
        std::lock_guard<std::mutex> lk(mu);
        frames[i] = std::move(f);
      }
      cv.notify_all();
    }
  }

  void registration_loop(size_t n, sga_context* ctx, std::vector<Frame>& results) {
    small_gicp_amd::Registration<small_gicp_amd::GICPFactor, small_gicp_amd::ParallelReductionHIP> registration;
    registration.rejector.max_dist_sq = params.max_correspondence_distance * params.max_correspondence_distance;
    registration.reduction.context = ctx;  // this worker's stream; it waits for the events behind the producers' work (csrc/common.hpp: Ready)
    for (;;) {
      size_t i;
      std::shared_ptr<Frame> target, source;
      {
        std::unique_lock<std::mutex> lk(mu);
        i = next_pair_to_take;
        if (i >= n || failure) return;
        next_pair_to_take++;
        cv.wait(lk, [&] { return (frames[i - 1] && frames[i]) || failure; });
        if (failure) return;
        target = frames[i - 1];
        source = frames[i];
      }
      const auto result = registration.align(*target->points, *source->kdtree, *target->kdtree, small_gicp_amd::Isometry3d::Identity());
      results[i].T_last_current = result.T_target_source;
      results[i].iterations = result.iterations + 1;
      results[i].t_start = source->t_start;
      results[i].t_done = now_ms();
      target.reset();
      source.reset();
      {
        std::lock_guard<std::mutex> lk(mu);
        done[i] = 1;
        while (pairs_done_below < n && done[pairs_done_below]) {  // frame j has served as the source of pair j and the target of pair j + 1
          frames[pairs_done_below - 1].reset();
          pairs_done_below++;
        }
        if (pairs_done_below == n) frames[n - 1].reset();
      }
      cv.notify_all();
    }
  }

  const Params params;
  std::vector<sga_context*> pre_ctx, reg_ctx;
  int depth = 8;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::shared_ptr<Frame>> frames;
  std::vector<char> done;
  size_t next_frame = 0, next_pair_to_take = 1, pairs_done_below = 1;
  std::exception_ptr failure;
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) {
    std::cout << "USAGE: odometry_benchmark_flow <dataset_path> <output_path> [options]\nOPTIONS:\n  --num_neighbors <value> (default: 20)\n  --downsampling_resolution <value> (default: 0.25)\n"
                 "  --max_correspondence_distance <value> (default: 1.0)\n  --max_frames <value>\n  --preprocess_workers <value> (default: 2)\n  --registration_workers <value> (default: 2)\n"
                 "  --depth <value>\n  --pinned\n  --repeat <value> (default: 1)\n";
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
    else if (arg == "--preprocess_workers" && has_value) params.preprocess_workers = std::stoi(argv[++i]);
    else if (arg == "--registration_workers" && has_value) params.registration_workers = std::stoi(argv[++i]);
    else if (arg == "--depth" && has_value) params.depth = std::stoi(argv[++i]);
    else if (arg == "--repeat" && has_value) params.repeat = std::max(1, std::stoi(argv[++i]));
    else if (arg == "--pinned") params.pinned = true;
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
    std::vector<Scan> scans;
    size_t total_points = 0;
    for (const auto& f : filenames) {
      scans.push_back(read_scan(f, params.pinned));
      total_points += scans.back().n;
    }
    std::cout << "dataset_path=" << dataset_path << "\nnum_frames=" << scans.size() << "\nnum_points=" << (scans.empty() ? 0 : total_points / scans.size()) << " [points/scan, mean]\nnum_neighbors=" << params.num_neighbors
              << "\ndownsampling_resolution=" << params.downsampling_resolution << "\npreprocess_workers=" << params.preprocess_workers << "\nregistration_workers=" << params.registration_workers
              << "\nscans_in_pinned_host_memory=" << (params.pinned ? 1 : 0) << std::endl;

    FlowOdometry odom(params);
    std::vector<small_gicp_amd::Isometry3d> traj;
    for (int rep = 0; rep < params.repeat; rep++) {
      double wall_ms = 0.0, mean_iterations = 0.0, latency_ms = 0.0;
      const double t_begin = now_ms();
      traj = odom.estimate(scans, wall_ms, mean_iterations, latency_ms);
      // the reference's flow engine reports elapsed / frames (:113-114)
      uint64_t al[5];
      sga_allocator_stats(al);  // cumulative: hipMalloc calls / served by the stream's own list / the shared pool / completed deferred frees / frees deferred
      std::printf("run=%d total_throughput=%.4f [msec/scan]  frame_latency=%.4f [msec]  mean_iterations=%.2f  window_ns=%.0f,%.0f  allocator=%llu,%llu,%llu,%llu,%llu\n", rep,
                  scans.empty() ? 0.0 : wall_ms / static_cast<double>(scans.size()), latency_ms, mean_iterations, 1e6 * t_begin, 1e6 * (t_begin + wall_ms),  // steady_clock = CLOCK_MONOTONIC: the clock of a rocprofv3 kernel trace
                  static_cast<unsigned long long>(al[0]), static_cast<unsigned long long>(al[1]), static_cast<unsigned long long>(al[2]), static_cast<unsigned long long>(al[3]), static_cast<unsigned long long>(al[4]));
    }
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
