// TEST INFRASTRUCTURE — NOT PRODUCT CODE.
// CPU oracle: a double-precision restatement of the registration hot path of koide3/small_gicp v1.0.1
// (reference tree at /root/reference, citations are relative to it).  It exists to CHECK the HIP path and to be
// timed as the CPU baseline (bench.py cpu_baseline, kind "port").  Nothing in small_gicp_amd/ may call it.
//
// Pinning (both green): (1) tests/test_oracle_pins.py checks this restatement against every known-answer fixture the
// reference's own tests hold for this path (data/T_target_source.txt within the reference tolerances for all factor types, exact
// kNN vs brute force / scipy incl. the synthetic tie/lattice/tiny/huge clouds, fast_floor == floor, H symmetric with
// lambda_min > 10); (2) tests/test_oracle_vs_reference.py checks it stage by stage against oracle/_ref — the UNMODIFIED reference
// headers + registration_helper.cpp compiled in place over a home-made Eigen stand-in (oracle/ref/): same voxels, same kNN
// indices, same inliers and iteration counts, poses equal to 1e-9.  The reference ships no bit-exact numeric goldens.
//
// Homogeneous coordinates: the reference stores (x,y,z,1) / (nx,ny,nz,0) / 4x4 covs with a zero last row+column.
// Every formula below is the 3-D restriction of the reference's 4-D expression; the w terms cancel identically.
#pragma once
#include <omp.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <memory>
#include <numeric>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "orc_math.hpp"

namespace orc {

// ------------------------------------------------------------------------------------------------------------------
// points/point_cloud.hpp:69-71 — AoS double cloud.
struct PointCloud {
  std::vector<Vec3> points;
  std::vector<Vec3> normals;
  std::vector<Mat3> covs;
  size_t size() const { return points.size(); }
  void resize(size_t n) {
    points.resize(n);
    normals.resize(n, Vec3{0, 0, 0});
    covs.resize(n, Mat3::zero());
  }
};

// util/fast_floor.hpp:12-15
inline int fast_floor(double x) {
  const int n = static_cast<int>(x);
  return n - (x < static_cast<double>(n));
}

// ------------------------------------------------------------------------------------------------------------------
// util/downsampling.hpp:23-78 — serial voxel-grid mean.  Output order = ascending packed key.
// (std::sort on (key) only is unstable in the reference; we sort on (key, index) which is one of its valid outcomes.)
inline void voxelgrid_sampling(const std::vector<Vec3>& points, double leaf_size, std::vector<Vec3>& out) {
  out.clear();
  if (points.empty()) return;
  const double inv_leaf_size = 1.0 / leaf_size;
  constexpr std::uint64_t invalid_coord = std::numeric_limits<std::uint64_t>::max();
  constexpr int coord_bit_size = 21;
  constexpr int coord_bit_mask = (1 << 21) - 1;
  constexpr int coord_offset = 1 << (coord_bit_size - 1);

  std::vector<std::pair<std::uint64_t, size_t>> coord_pt(points.size());
  for (size_t i = 0; i < points.size(); i++) {
    int c[3];
    bool bad = false;
    for (int k = 0; k < 3; k++) {
      c[k] = fast_floor(points[i][k] * inv_leaf_size) + coord_offset;
      bad |= (c[k] < 0) || (c[k] > coord_bit_mask);
    }
    if (bad) {
      coord_pt[i] = {invalid_coord, i};
      continue;
    }
    const std::uint64_t bits = (static_cast<std::uint64_t>(c[0] & coord_bit_mask) << (coord_bit_size * 0)) | (static_cast<std::uint64_t>(c[1] & coord_bit_mask) << (coord_bit_size * 1)) |
                               (static_cast<std::uint64_t>(c[2] & coord_bit_mask) << (coord_bit_size * 2));
    coord_pt[i] = {bits, i};
  }
  std::sort(coord_pt.begin(), coord_pt.end());

  out.reserve(points.size());
  // The reference carries w (=1 per point) through the sum and divides by sum.w(): i.e. by the point count.
  Vec3 sum_pt = points[coord_pt.front().second];
  double sum_w = 1.0;
  for (size_t i = 1; i < points.size(); i++) {
    if (coord_pt[i].first == invalid_coord) continue;
    if (coord_pt[i - 1].first != coord_pt[i].first) {
      out.push_back((1.0 / sum_w) * sum_pt);
      sum_pt = {0, 0, 0};
      sum_w = 0.0;
    }
    sum_pt = sum_pt + points[coord_pt[i].second];
    sum_w += 1.0;
  }
  out.push_back((1.0 / sum_w) * sum_pt);
}

// ------------------------------------------------------------------------------------------------------------------
// ann/knn_result.hpp:32-108 — sorted insertion buffer; first pushed wins ties (push ignores distance >= worst).
struct KnnResult {
  int capacity;
  int num_found = 0;
  size_t* indices;
  double* distances;
  KnnResult(size_t* idx, double* dist, int k) : capacity(k), indices(idx), distances(dist) {
    std::fill(indices, indices + k, std::numeric_limits<size_t>::max());
    std::fill(distances, distances + k, std::numeric_limits<double>::max());
  }
  double worst_distance() const { return distances[capacity - 1]; }
  void push(size_t index, double distance) {
    if (distance >= worst_distance()) return;
    int insert_loc = std::min<int>(num_found, capacity - 1);
    for (; insert_loc > 0 && distance < distances[insert_loc - 1]; insert_loc--) {
      indices[insert_loc] = indices[insert_loc - 1];
      distances[insert_loc] = distances[insert_loc - 1];
    }
    indices[insert_loc] = index;
    distances[insert_loc] = distance;
    num_found = std::min<int>(num_found + 1, capacity);
  }
};

// ann/kdtree.hpp:56-71 (node), :96-126 (build), :193-233 (search); ann/projection.hpp:31-50 (axis choice).
struct KdTree {
  struct Node {
    std::uint32_t first = 0, last = 0;  // leaf
    int axis = 0;                       // non-leaf
    double thresh = 0.0;
    std::uint32_t left = 0xffffffffu, right = 0xffffffffu;
  };
  const std::vector<Vec3>* points = nullptr;
  std::vector<size_t> indices;
  std::vector<Node> nodes;
  std::uint32_t root = 0;
  int max_leaf_size = 20;
  int max_scan_count = 128;

  void build(const std::vector<Vec3>& pts) {
    points = &pts;
    indices.resize(pts.size());
    nodes.clear();
    if (pts.empty()) return;
    std::iota(indices.begin(), indices.end(), 0);
    nodes.resize(pts.size());
    size_t node_count = 0;
    root = create_node(node_count, 0, pts.size());
    nodes.resize(node_count);
  }

  int find_axis(size_t first, size_t last) const {
    const size_t N = last - first;
    double sum_pt[3] = {0, 0, 0}, sum_sq[3] = {0, 0, 0};
    double sum_w = 0.0;
    const size_t step = N < static_cast<size_t>(max_scan_count) ? 1 : N / max_scan_count;
    const size_t num_steps = N / step;
    for (size_t i = 0; i < num_steps; i++) {
      const Vec3& pt = (*points)[indices[first + step * i]];
      for (int k = 0; k < 3; k++) {
        sum_pt[k] += pt[k];
        sum_sq[k] += pt[k] * pt[k];
      }
      sum_w += 1.0;
    }
    double var[3];
    for (int k = 0; k < 3; k++) {
      const double mean = sum_pt[k] / sum_w;
      var[k] = sum_sq[k] - mean * sum_pt[k];
    }
    return var[0] > var[1] ? (var[0] > var[2] ? 0 : 2) : (var[1] > var[2] ? 1 : 2);
  }

  std::uint32_t create_node(size_t& node_count, size_t first, size_t last) {
    const size_t N = last - first;
    const std::uint32_t node_index = node_count++;
    if (N <= static_cast<size_t>(max_leaf_size)) {
      nodes[node_index].first = first;
      nodes[node_index].last = last;
      return node_index;
    }
    const int axis = find_axis(first, last);
    const size_t median = first + N / 2;
    std::nth_element(indices.begin() + first, indices.begin() + median, indices.begin() + last, [&](size_t i, size_t j) { return (*points)[i][axis] < (*points)[j][axis]; });
    nodes[node_index].axis = axis;
    nodes[node_index].thresh = (*points)[indices[median]][axis];
    const std::uint32_t l = create_node(node_count, first, median);
    const std::uint32_t r = create_node(node_count, median, last);
    nodes[node_index].left = l;
    nodes[node_index].right = r;
    return node_index;
  }

  // returns false when the search may stop early (KnnSetting::fulfilled, epsilon = 0 -> never before an exact hit)
  bool search(const Vec3& query, std::uint32_t node_index, KnnResult& result) const {
    const Node& node = nodes[node_index];
    if (node.left == 0xffffffffu) {
      for (size_t i = node.first; i < node.last; i++) {
        const double sq_dist = sqnorm((*points)[indices[i]] - query);
        result.push(indices[i], sq_dist);
      }
      return !(result.worst_distance() < 0.0);
    }
    const double diff = query[node.axis] - node.thresh;
    const double cut_sq_dist = diff * diff;
    const std::uint32_t best_child = diff < 0.0 ? node.left : node.right;
    const std::uint32_t other_child = diff < 0.0 ? node.right : node.left;
    if (!search(query, best_child, result)) return false;
    if (result.worst_distance() > cut_sq_dist) return search(query, other_child, result);
    return true;
  }

  size_t knn_search(const Vec3& query, int k, size_t* k_indices, double* k_sq_dists) const {
    KnnResult result(k_indices, k_sq_dists, k);
    if (nodes.empty()) return 0;  // the reference is UB here; we return "not found"
    search(query, root, result);
    return result.num_found;
  }
  size_t nearest_neighbor_search(const Vec3& query, size_t* index, double* sq_dist) const { return knn_search(query, 1, index, sq_dist); }
};

// ------------------------------------------------------------------------------------------------------------------
// util/normal_estimation.hpp:13-92 (+ _omp.hpp:19-26): kNN (incl. self) -> mean/cov (1/n) -> eigvecs ascending ->
// normal = +-v0 (flipped so p.n <= 0), cov = V diag(1e-3,1,1) V^T.  n < 5 -> normal 0, cov I3.
inline void estimate_normals_covariances(PointCloud& cloud, const KdTree& tree, int num_neighbors, int num_threads, bool set_normals = true, bool set_covs = true) {
  cloud.resize(cloud.size());
  const std::int64_t N = cloud.size();
#pragma omp parallel for num_threads(num_threads) schedule(dynamic, 64)
  for (std::int64_t pi = 0; pi < N; pi++) {
    std::vector<size_t> k_indices(num_neighbors);
    std::vector<double> k_sq_dists(num_neighbors);
    const size_t n = tree.knn_search(cloud.points[pi], num_neighbors, k_indices.data(), k_sq_dists.data());
    if (n < 5) {
      if (set_normals) cloud.normals[pi] = {0, 0, 0};
      if (set_covs) cloud.covs[pi] = Mat3::identity();
      continue;
    }
    Vec3 sum_points{0, 0, 0};
    Mat3 sum_cross = Mat3::zero();
    for (size_t i = 0; i < n; i++) {
      const Vec3& pt = cloud.points[k_indices[i]];
      sum_points = sum_points + pt;
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) sum_cross(r, c) += pt[r] * pt[c];
    }
    const Vec3 mean = (1.0 / n) * sum_points;
    Mat3 cov;
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) cov(r, c) = (sum_cross(r, c) - mean[r] * sum_points[c]) / n;
    double eivals[3];
    Mat3 V;
    eigen_sym3_direct(cov, eivals, V);
    if (set_normals) {
      Vec3 nrm = V.col(0);
      nrm = (1.0 / norm(nrm)) * nrm;
      if (dot(cloud.points[pi], nrm) > 0) nrm = -1.0 * nrm;
      cloud.normals[pi] = nrm;
    }
    if (set_covs) {
      const double values[3] = {1e-3, 1.0, 1.0};
      Mat3 C = Mat3::zero();
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++)
          for (int k = 0; k < 3; k++) C(r, c) += V(r, k) * values[k] * V(c, k);
      cloud.covs[pi] = C;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// ann/incremental_voxelmap.hpp:55-119,151-153 + ann/gaussian_voxelmap.hpp:15-60 — one-shot GaussianVoxelMap
// (LRU bookkeeping is a no-op for a single insert: lru_counter 0 -> 1, 1 % 10 != 0).
struct GaussianVoxelMap {
  // ann/gaussian_voxelmap.hpp:15-60 (GaussianVoxel) + the VoxelInfo of ann/incremental_voxelmap.hpp:21-31
  struct Voxel {
    int coord[3];
    size_t lru = 0;
    bool finalized = false;
    size_t num_points = 0;
    Vec3 mean{0, 0, 0};
    Mat3 cov = Mat3::zero();
  };
  double inv_leaf_size;
  size_t lru_horizon = 100, lru_clear_cycle = 10, lru_counter = 0;  // incremental_voxelmap.hpp:46
  std::vector<Voxel> flat_voxels;  // insertion order of the first point of each voxel
  std::unordered_map<std::uint64_t, size_t> voxels;
  int num_search_offsets = 1;

  explicit GaussianVoxelMap(double leaf_size) : inv_leaf_size(1.0 / leaf_size) {}
  static std::uint64_t key(const int c[3]) {
    // 21 bits per axis is ample for the test scenes; semantic = exact equality of integer coords (vector3i_hash.hpp:13-25 is only a hash)
    return (static_cast<std::uint64_t>(static_cast<std::uint32_t>(c[0] + (1 << 20)) & 0x1fffff)) | (static_cast<std::uint64_t>(static_cast<std::uint32_t>(c[1] + (1 << 20)) & 0x1fffff) << 21) |
           (static_cast<std::uint64_t>(static_cast<std::uint32_t>(c[2] + (1 << 20)) & 0x1fffff) << 42);
  }
  size_t size() const { return flat_voxels.size(); }

  // incremental_voxelmap.hpp:55-92: insert (with a pose), LRU sweep every lru_clear_cycle inserts, finalize
  void insert(const PointCloud& points, const SE3& T = SE3::identity()) {
    for (size_t i = 0; i < points.size(); i++) {
      const Vec3 pt = T * points.points[i];
      int c[3] = {fast_floor(pt[0] * inv_leaf_size), fast_floor(pt[1] * inv_leaf_size), fast_floor(pt[2] * inv_leaf_size)};
      const std::uint64_t k = key(c);
      auto found = voxels.find(k);
      if (found == voxels.end()) {
        found = voxels.emplace(k, flat_voxels.size()).first;
        Voxel v;
        v.coord[0] = c[0];
        v.coord[1] = c[1];
        v.coord[2] = c[2];
        v.lru = lru_counter;
        flat_voxels.push_back(v);
      }
      Voxel& v = flat_voxels[found->second];
      v.lru = lru_counter;
      // GaussianVoxel::add (gaussian_voxelmap.hpp:32-42)
      if (v.finalized) {
        v.finalized = false;
        v.mean = static_cast<double>(v.num_points) * v.mean;
        v.cov = static_cast<double>(v.num_points) * v.cov;
      }
      v.num_points++;
      v.mean = v.mean + pt;
      v.cov = v.cov + T.R * points.covs[i] * transpose(T.R);  // T.matrix() * cov4 * T.matrix()^T: the 3x3 block is R C R^T
    }
    if ((++lru_counter) % lru_clear_cycle == 0) {
      // remove the least recently used voxels, keep the order of the others, rehash (:76-88)
      std::vector<Voxel> kept;
      kept.reserve(flat_voxels.size());
      for (const auto& v : flat_voxels)
        if (!(v.lru + lru_horizon < lru_counter)) kept.push_back(v);
      flat_voxels.swap(kept);
      voxels.clear();
      for (size_t i = 0; i < flat_voxels.size(); i++) voxels[key(flat_voxels[i].coord)] = i;
    }
    for (auto& v : flat_voxels) {  // GaussianVoxel::finalize (:45-53)
      if (v.finalized) continue;
      const double np = static_cast<double>(v.num_points);  // `mean /= num_points; cov /= num_points;` element by element
      for (int r = 0; r < 3; r++) {
        v.mean[r] = v.mean[r] / np;
        for (int c = 0; c < 3; c++) v.cov(r, c) = v.cov(r, c) / np;
      }
      v.finalized = true;
    }
  }

  // index = voxel_id << 32 (point_id is always 0 for GaussianVoxel)
  size_t nearest_neighbor_search(const Vec3& pt, size_t* index, double* sq_dist) const {
    const int center[3] = {fast_floor(pt[0] * inv_leaf_size), fast_floor(pt[1] * inv_leaf_size), fast_floor(pt[2] * inv_leaf_size)};
    static const int offs7[7][3] = {{0, 0, 0}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {-1, 0, 0}, {0, -1, 0}, {0, 0, -1}};
    size_t found_n = 0;
    double best = std::numeric_limits<double>::max();
    auto probe = [&](int dx, int dy, int dz) {
      const int c[3] = {center[0] + dx, center[1] + dy, center[2] + dz};
      auto found = voxels.find(key(c));
      if (found == voxels.end()) return;
      const double d = sqnorm(flat_voxels[found->second].mean - pt);
      if (d >= best) return;  // KnnResult<1>::push ignores distance >= worst
      best = d;
      *index = found->second << 32;
      *sq_dist = d;
      found_n = 1;
    };
    if (num_search_offsets == 27) {
      for (int i = -1; i <= 1; i++)
        for (int j = -1; j <= 1; j++)
          for (int k = -1; k <= 1; k++) probe(i, j, k);
    } else if (num_search_offsets == 7) {
      for (auto& o : offs7) probe(o[0], o[1], o[2]);
    } else {
      probe(0, 0, 0);
    }
    return found_n;
  }
};

// ------------------------------------------------------------------------------------------------------------------
// IncrementalVoxelMap<FlatContainerCov>: voxels that keep (a bounded number of) the inserted points themselves with their
// covariances — the scan-to-model GICP target (ann/flat_container.hpp:15-100, ann/incremental_voxelmap.hpp:55-190;
// src/benchmark/odometry_benchmark_small_gicp_model_omp.cpp).
struct FlatVoxelMap {
  struct Voxel {
    int coord[3];
    size_t lru = 0;
    std::vector<Vec3> points;
    std::vector<Mat3> covs;
  };
  double inv_leaf_size;
  size_t lru_horizon = 100, lru_clear_cycle = 10, lru_counter = 0;   // incremental_voxelmap.hpp:46
  double min_sq_dist_in_cell = 0.1 * 0.1;                            // flat_container.hpp:19
  size_t max_num_points_in_cell = 10;                                // flat_container.hpp:20
  std::vector<std::array<int, 3>> search_offsets{{{0, 0, 0}}};
  std::vector<Voxel> flat_voxels;
  std::unordered_map<std::uint64_t, size_t> voxels;

  explicit FlatVoxelMap(double leaf_size) : inv_leaf_size(1.0 / leaf_size) {}
  size_t size() const { return flat_voxels.size(); }

  // incremental_voxelmap.hpp:157-186 (for 27 the reference APPENDS to the offsets set by the constructor: the centre comes first
  // and once more in the middle of the cube — harmless, a second visit of the same voxel only meets ties)
  void set_search_offsets(int n) {
    if (n == 7) {
      search_offsets = {{{0, 0, 0}}, {{1, 0, 0}}, {{0, 1, 0}}, {{0, 0, 1}}, {{-1, 0, 0}}, {{0, -1, 0}}, {{0, 0, -1}}};
    } else if (n == 27) {
      search_offsets = {{{0, 0, 0}}};
      for (int i = -1; i <= 1; i++)
        for (int j = -1; j <= 1; j++)
          for (int k = -1; k <= 1; k++) search_offsets.push_back({{i, j, k}});
    } else {
      search_offsets = {{{0, 0, 0}}};
    }
  }

  void insert(const PointCloud& pts, const SE3& T = SE3::identity()) {
    for (size_t i = 0; i < pts.size(); i++) {
      const Vec3 pt = T * pts.points[i];
      int c[3] = {fast_floor(pt[0] * inv_leaf_size), fast_floor(pt[1] * inv_leaf_size), fast_floor(pt[2] * inv_leaf_size)};
      const std::uint64_t k = GaussianVoxelMap::key(c);
      auto found = voxels.find(k);
      if (found == voxels.end()) {
        found = voxels.emplace(k, flat_voxels.size()).first;
        Voxel v;
        v.coord[0] = c[0];
        v.coord[1] = c[1];
        v.coord[2] = c[2];
        v.lru = lru_counter;
        flat_voxels.push_back(v);
      }
      Voxel& v = flat_voxels[found->second];
      v.lru = lru_counter;
      // FlatContainer::add (flat_container.hpp:33-51)
      bool reject = v.points.size() >= max_num_points_in_cell;
      for (size_t j = 0; j < v.points.size() && !reject; j++) reject = sqnorm(v.points[j] - pt) < min_sq_dist_in_cell;
      if (reject) continue;
      v.points.push_back(pt);
      v.covs.push_back(T.R * pts.covs[i] * transpose(T.R));
    }
    if ((++lru_counter) % lru_clear_cycle == 0) {
      std::vector<Voxel> kept;
      kept.reserve(flat_voxels.size());
      for (auto& v : flat_voxels)
        if (!(v.lru + lru_horizon < lru_counter)) kept.push_back(std::move(v));
      flat_voxels.swap(kept);
      voxels.clear();
      for (size_t i = 0; i < flat_voxels.size(); i++) voxels[GaussianVoxelMap::key(flat_voxels[i].coord)] = i;
    }
  }

  // incremental_voxelmap.hpp:99-119 with FlatContainer::knn_search (flat_container.hpp:84-93) and KnnResult<1>::push (>= keeps the first)
  size_t nearest_neighbor_search(const Vec3& pt, size_t* index, double* sq_dist) const {
    const int center[3] = {fast_floor(pt[0] * inv_leaf_size), fast_floor(pt[1] * inv_leaf_size), fast_floor(pt[2] * inv_leaf_size)};
    double best = std::numeric_limits<double>::max();
    size_t found_n = 0;
    for (const auto& o : search_offsets) {
      const int c[3] = {center[0] + o[0], center[1] + o[1], center[2] + o[2]};
      auto found = voxels.find(GaussianVoxelMap::key(c));
      if (found == voxels.end()) continue;
      const Voxel& v = flat_voxels[found->second];
      for (size_t i = 0; i < v.points.size(); i++) {
        const double d = sqnorm(v.points[i] - pt);
        if (d >= best) continue;
        best = d;
        *index = (found->second << 32) | i;
        *sq_dist = d;
        found_n = 1;
      }
    }
    return found_n;
  }
};

// ------------------------------------------------------------------------------------------------------------------
// Target abstraction: (cloud + kd-tree) or GaussianVoxelMap used as both cloud and tree (registration_helper.cpp:136).
struct Target {
  const PointCloud* cloud = nullptr;
  const KdTree* tree = nullptr;
  const GaussianVoxelMap* voxelmap = nullptr;
  const FlatVoxelMap* flatmap = nullptr;
  size_t nn(const Vec3& q, size_t* idx, double* sqd) const {
    if (flatmap) return flatmap->nearest_neighbor_search(q, idx, sqd);
    return voxelmap ? voxelmap->nearest_neighbor_search(q, idx, sqd) : tree->nearest_neighbor_search(q, idx, sqd);
  }
  const Vec3& point(size_t i) const {
    if (flatmap) return flatmap->flat_voxels[i >> 32].points[i & 0xffffffffu];
    return voxelmap ? voxelmap->flat_voxels[i >> 32].mean : cloud->points[i];
  }
  const Vec3& normal(size_t i) const { return cloud->normals[i]; }
  const Mat3& cov(size_t i) const {
    if (flatmap) return flatmap->flat_voxels[i >> 32].covs[i & 0xffffffffu];
    return voxelmap ? voxelmap->flat_voxels[i >> 32].cov : cloud->covs[i];
  }
};

enum FactorKind { FACTOR_ICP = 0, FACTOR_PLANE_ICP = 1, FACTOR_GICP = 2 };
enum RobustKind { ROBUST_NONE = 0, ROBUST_HUBER = 1, ROBUST_CAUCHY = 2 };

struct FactorSetting {
  int kind = FACTOR_GICP;
  int robust = ROBUST_NONE;
  double robust_c = 1.0;
};

// factors/robust_kernel.hpp:24-27,47
inline double robust_weight(const FactorSetting& s, double x) {
  if (s.robust == ROBUST_HUBER) {
    const double a = std::abs(x);
    return a < s.robust_c ? 1.0 : s.robust_c / a;
  }
  if (s.robust == ROBUST_CAUCHY) return s.robust_c / (s.robust_c + x * x);
  return 1.0;
}

// Per-source-point factor state (gicp_factor.hpp:94-96, icp_factor.hpp:67-68).
struct Factor {
  size_t target_index = std::numeric_limits<size_t>::max();
  Mat3 mahalanobis = Mat3::zero();
  bool inlier() const { return target_index != std::numeric_limits<size_t>::max(); }
};

// H (6x6) = J^T M J, b = J^T M r, e = 0.5 r^T M r with J = [A | B] (3x6).
inline void accumulate_JMJ(const Mat3& A, const Mat3& B, const Mat3& M, const Vec3& r, Mat6* H, Vec6* b, double* e) {
  double J[3][6];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      J[i][j] = A(i, j);
      J[i][j + 3] = B(i, j);
    }
  double MJ[3][6];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 6; j++) MJ[i][j] = M(i, 0) * J[0][j] + M(i, 1) * J[1][j] + M(i, 2) * J[2][j];
  const Vec3 Mr = M * r;
  for (int i = 0; i < 6; i++) {
    for (int j = 0; j < 6; j++) (*H)(i, j) = J[0][i] * MJ[0][j] + J[1][i] * MJ[1][j] + J[2][i] * MJ[2][j];
    (*b)[i] = J[0][i] * Mr[0] + J[1][i] * Mr[1] + J[2][i] * Mr[2];
  }
  *e = 0.5 * dot(r, Mr);
}

// factors/{gicp,plane_icp,icp}_factor.hpp linearize (+ robust_kernel.hpp:70-91), rejector.hpp:19-28 (strict >).
inline bool factor_linearize(const FactorSetting& fs, Factor& f, const Target& target, const PointCloud& source, const SE3& T, size_t i, double max_dist_sq, Mat6* H, Vec6* b, double* e) {
  f.target_index = std::numeric_limits<size_t>::max();
  const Vec3& ps = source.points[i];
  const Vec3 q = T * ps;
  size_t k_index;
  double k_sq_dist;
  if (!target.nn(q, &k_index, &k_sq_dist) || k_sq_dist > max_dist_sq) return false;
  f.target_index = k_index;
  const Vec3 residual = target.point(k_index) - q;
  const Mat3 A = T.R * skew(ps);
  const Mat3 B = -1.0 * T.R;
  if (fs.kind == FACTOR_GICP) {
    const Mat3 RCR = target.cov(k_index) + T.R * source.covs[i] * transpose(T.R);
    f.mahalanobis = inverse(RCR);
    accumulate_JMJ(A, B, f.mahalanobis, residual, H, b, e);
  } else if (fs.kind == FACTOR_PLANE_ICP) {
    const Vec3& n = target.normal(k_index);
    Mat3 D = Mat3::zero();
    for (int k = 0; k < 3; k++) D(k, k) = n[k];
    const Vec3 err{n[0] * residual[0], n[1] * residual[1], n[2] * residual[2]};
    accumulate_JMJ(D * A, D * B, Mat3::identity(), err, H, b, e);
  } else {
    accumulate_JMJ(A, B, Mat3::identity(), residual, H, b, e);
  }
  if (fs.robust != ROBUST_NONE) {
    const double w = robust_weight(fs, std::sqrt(*e));
    for (int r = 0; r < 6; r++) {
      for (int c = 0; c < 6; c++) (*H)(r, c) *= w;
      (*b)[r] *= w;
    }
    *e *= w;
  }
  return true;
}

// factors/*_factor.hpp error(): stale correspondence + stale mahalanobis.
inline double factor_error(const FactorSetting& fs, const Factor& f, const Target& target, const PointCloud& source, const SE3& T, size_t i) {
  if (!f.inlier()) return 0.0;
  const Vec3 residual = target.point(f.target_index) - T * source.points[i];
  double e;
  if (fs.kind == FACTOR_GICP) {
    e = 0.5 * dot(residual, f.mahalanobis * residual);
  } else if (fs.kind == FACTOR_PLANE_ICP) {
    const Vec3& n = target.normal(f.target_index);
    const Vec3 err{n[0] * residual[0], n[1] * residual[1], n[2] * residual[2]};
    e = 0.5 * sqnorm(err);
  } else {
    e = 0.5 * sqnorm(residual);
  }
  if (fs.robust != ROBUST_NONE) e = robust_weight(fs, std::sqrt(e)) * e;
  return e;
}

// registration/reduction.hpp:21-62 (num_threads <= 1) and reduction_omp.hpp:24-70 (per-thread slots, guided,8, serial fold).
struct Reduction {
  int num_threads = 4;
  std::tuple<Mat6, Vec6, double> linearize(const FactorSetting& fs, const Target& target, const PointCloud& source, double max_dist_sq, const SE3& T, std::vector<Factor>& factors) const {
    const int nt = std::max(1, num_threads);
    std::vector<Mat6> Hs(nt, Mat6::zero());
    std::vector<Vec6> bs(nt, Vec6{0, 0, 0, 0, 0, 0});
    std::vector<double> es(nt, 0.0);
    const std::int64_t N = factors.size();
    if (nt == 1) {
      for (std::int64_t i = 0; i < N; i++) {
        Mat6 H;
        Vec6 b;
        double e;
        if (!factor_linearize(fs, factors[i], target, source, T, i, max_dist_sq, &H, &b, &e)) continue;
        Hs[0] += H;
        for (int k = 0; k < 6; k++) bs[0][k] += b[k];
        es[0] += e;
      }
    } else {
#pragma omp parallel for num_threads(nt) schedule(guided, 8)
      for (std::int64_t i = 0; i < N; i++) {
        Mat6 H;
        Vec6 b;
        double e;
        if (!factor_linearize(fs, factors[i], target, source, T, i, max_dist_sq, &H, &b, &e)) continue;
        const int tid = omp_get_thread_num();
        Hs[tid] += H;
        for (int k = 0; k < 6; k++) bs[tid][k] += b[k];
        es[tid] += e;
      }
      for (int t = 1; t < nt; t++) {
        Hs[0] += Hs[t];
        for (int k = 0; k < 6; k++) bs[0][k] += bs[t][k];
        es[0] += es[t];
      }
    }
    return {Hs[0], bs[0], es[0]};
  }
  double error(const FactorSetting& fs, const Target& target, const PointCloud& source, const SE3& T, const std::vector<Factor>& factors) const {
    double sum_e = 0.0;
    const std::int64_t N = factors.size();
    const int nt = std::max(1, num_threads);
    if (nt == 1) {
      for (std::int64_t i = 0; i < N; i++) sum_e += factor_error(fs, factors[i], target, source, T, i);
    } else {
#pragma omp parallel for num_threads(nt) schedule(guided, 8) reduction(+ : sum_e)
      for (std::int64_t i = 0; i < N; i++) sum_e += factor_error(fs, factors[i], target, source, T, i);
    }
    return sum_e;
  }
};

// registration/registration_result.hpp:11-30
struct RegistrationResult {
  SE3 T_target_source = SE3::identity();
  bool converged = false;
  size_t iterations = 0;
  size_t num_inliers = 0;
  Mat6 H = Mat6::zero();
  Vec6 b{0, 0, 0, 0, 0, 0};
  double error = 0.0;
};

// registration/termination_criteria.hpp:11-20
struct TerminationCriteria {
  double translation_eps = 1e-3;
  double rotation_eps = 0.1 * M_PI / 180.0;
  bool converged(const Vec6& d) const { return std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) <= rotation_eps && std::sqrt(d[3] * d[3] + d[4] * d[4] + d[5] * d[5]) <= translation_eps; }
};

struct OptimizerSetting {
  int type = 0;  // 0 = LevenbergMarquardt, 1 = GaussNewton
  int max_iterations = 20;
  int max_inner_iterations = 10;
  double init_lambda = 1e-3;
  double lambda_factor = 10.0;
  double gn_lambda = 1e-6;
  bool verbose = false;
  // general factor: NullFactor (restrict_lambda = 0) or RestrictDoFFactor (factors/general_factor.hpp:41-75):
  // update_linearized_system adds restrict_lambda * |mask - 1| to the diagonal of H; update_error is a no-op (:71)
  double restrict_lambda = 0.0;
  double restrict_mask[6] = {1, 1, 1, 1, 1, 1};  // rx, ry, rz, tx, ty, tz: 1 = free, 0 = (softly) frozen
  void apply_general_factor(Mat6& H) const {
    if (restrict_lambda > 0)
      for (int k = 0; k < 6; k++) H(k, k) += restrict_lambda * std::fabs(restrict_mask[k] - 1.0);
  }
};

struct IterationTrace {
  std::vector<double> e;      // error returned by each linearize
  std::vector<double> new_e;  // accepted error of each outer iteration (LM)
};

// registration/optimizer.hpp:83-149 (LM) and :24-63 (GN); general factor = NullFactor or RestrictDoFFactor (OptimizerSetting).
inline RegistrationResult optimize(
  const OptimizerSetting& opt,
  const TerminationCriteria& criteria,
  const Reduction& reduction,
  const FactorSetting& fs,
  const Target& target,
  const PointCloud& source,
  double max_dist_sq,
  const SE3& init_T,
  std::vector<Factor>& factors,
  IterationTrace* trace = nullptr) {
  RegistrationResult result;
  result.T_target_source = init_T;
  if (opt.type == 1) {
    for (int i = 0; i < opt.max_iterations && !result.converged; i++) {
      auto [H, b, e] = reduction.linearize(fs, target, source, max_dist_sq, result.T_target_source, factors);
      opt.apply_general_factor(H);  // optimizer.hpp:42: general_factor.update_linearized_system
      if (trace) trace->e.push_back(e);
      Mat6 A = H;
      for (int k = 0; k < 6; k++) A(k, k) += opt.gn_lambda;
      Vec6 nb;
      for (int k = 0; k < 6; k++) nb[k] = -b[k];
      const Vec6 delta = ldlt_solve(A, nb);
      result.converged = criteria.converged(delta);
      result.T_target_source = result.T_target_source * se3_exp(delta);
      result.iterations = i;
      result.H = H;
      result.b = b;
      result.error = e;
    }
  } else {
    double lambda = opt.init_lambda;
    for (int i = 0; i < opt.max_iterations && !result.converged; i++) {
      auto [H, b, e] = reduction.linearize(fs, target, source, max_dist_sq, result.T_target_source, factors);
      opt.apply_general_factor(H);  // optimizer.hpp:103
      if (trace) trace->e.push_back(e);
      bool success = false;
      for (int j = 0; j < opt.max_inner_iterations; j++) {
        Mat6 A = H;
        for (int k = 0; k < 6; k++) A(k, k) += lambda;
        Vec6 nb;
        for (int k = 0; k < 6; k++) nb[k] = -b[k];
        const Vec6 delta = ldlt_solve(A, nb);
        const SE3 new_T = result.T_target_source * se3_exp(delta);
        const double new_e = reduction.error(fs, target, source, new_T, factors);
        if (opt.verbose) {
          std::printf("iter=%d inner=%d e=%.9g new_e=%.9g lambda=%g dt=%g dr=%g\n", i, j, e, new_e, lambda, std::sqrt(delta[3] * delta[3] + delta[4] * delta[4] + delta[5] * delta[5]),
                      std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]));
        }
        if (new_e <= e) {
          result.converged = criteria.converged(delta);
          result.T_target_source = new_T;
          lambda /= opt.lambda_factor;
          success = true;
          e = new_e;
          break;
        } else {
          lambda *= opt.lambda_factor;
        }
      }
      if (trace) trace->new_e.push_back(e);
      result.iterations = i;
      result.H = H;
      result.b = b;
      result.error = e;
      if (!success) break;
    }
  }
  result.num_inliers = std::count_if(factors.begin(), factors.end(), [](const Factor& f) { return f.inlier(); });
  return result;
}

// registration/registration.hpp:33-43
inline RegistrationResult registration_align(
  const OptimizerSetting& opt,
  const TerminationCriteria& criteria,
  const Reduction& reduction,
  const FactorSetting& fs,
  const Target& target,
  const PointCloud& source,
  double max_dist_sq,
  const SE3& init_T,
  IterationTrace* trace = nullptr,
  std::vector<Factor>* factors_out = nullptr) {
  std::vector<Factor> factors(source.size());
  RegistrationResult r = optimize(opt, criteria, reduction, fs, target, source, max_dist_sq, init_T, factors, trace);
  if (factors_out) *factors_out = std::move(factors);
  return r;
}

}  // namespace orc
