// Incremental GaussianVoxelMap on the device: repeated insert(points, T) with running per-voxel means and LRU removal — the
// scan-to-model target of the reference (ann/incremental_voxelmap.hpp:55-92 with GaussianVoxel::add / finalize,
// ann/gaussian_voxelmap.hpp:32-53; used by src/benchmark/odometry_benchmark_small_vgicp_model_omp.cpp).  gfx950.
//
// Reference semantics kept:  coord = fast_floor(T p / leaf) in double;  a voxel is created at its first point, voxel ids follow the
// creation order (flat_voxels);  add = un-finalize (mean *= N, cov *= N), N++, mean += T p, cov += T C T^T, point by point in
// insertion order;  finalize = divide by N;  every `clear_cycle` inserts the voxels with lru + horizon < counter are removed
// (order of the others preserved) and the table is rebuilt.
//
// One insert = key generation -> stable radix sort by voxel key (segments = the batch's voxels, points inside a segment in
// insertion order) -> per segment: look the voxel up; new voxels are ranked by their first point = creation order -> one lane per
// segment updates the voxel's fp64 state exactly in the reference's operation order (no floating-point atomics) -> the fp32
// records the VGICP factor kernel reads are refreshed.  The hash table is open addressing on the packed coordinate
// (voxel_hash.hpp), rebuilt when it gets half full or after an LRU sweep.
#include "common.hpp"

#include <cmath>
#include <memory>
#include <rocprim/rocprim.hpp>

#include "device_math.hpp"
#include "voxel_hash.hpp"

namespace sga {

int ensure_temp(sga_context* ctx, size_t bytes);

struct Pose12 {
  double r[9];  // row-major rotation
  double t[3];
};

__global__ void ivm_keys_kernel(const float4* __restrict__ pts, size_t n, Pose12 T, double inv_leaf, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  const double x = T.r[0] * p.x + T.r[1] * p.y + T.r[2] * p.z + T.t[0];
  const double y = T.r[3] * p.x + T.r[4] * p.y + T.r[5] * p.z + T.t[1];
  const double z = T.r[6] * p.x + T.r[7] * p.y + T.r[8] * p.z + T.t[2];
  const int cx = fast_floor_d(x * inv_leaf), cy = fast_floor_d(y * inv_leaf), cz = fast_floor_d(z * inv_leaf);
  const bool bad = abs(cx) >= (1 << 20) || abs(cy) >= (1 << 20) || abs(cz) >= (1 << 20) || !(x == x) || !(y == y) || !(z == z);
  keys[i] = bad ? SGA_HASH_EMPTY : voxel_key(cx, cy, cz);  // out-of-range points sort last and are dropped
  vals[i] = static_cast<uint32_t>(i);
}

__global__ void ivm_heads_kernel(const unsigned long long* __restrict__ keys, size_t n, uint32_t* __restrict__ flags) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  flags[i] = (k != SGA_HASH_EMPTY && (i == 0 || keys[i - 1] != k)) ? 1u : 0u;
}

__global__ void ivm_segments_kernel(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ seg_id, const uint32_t* __restrict__ order, size_t n, uint32_t* __restrict__ seg_start, uint32_t* __restrict__ seg_first) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  if (flags[i]) {
    const uint32_t s = seg_id[i];
    seg_start[s] = static_cast<uint32_t>(i);
    seg_first[s] = order[i];  // stable sort: the first entry of a segment is the earliest inserted point
  }
}

__device__ __forceinline__ uint32_t ivm_find(const unsigned long long* __restrict__ hkeys, const uint32_t* __restrict__ hvals, uint32_t hmask, unsigned long long key) {
  uint32_t slot = voxel_hash(key) & hmask;
  for (;;) {
    const unsigned long long k = hkeys[slot];
    if (k == key) return hvals[slot];
    if (k == SGA_HASH_EMPTY) return 0xffffffffu;
    slot = (slot + 1) & hmask;
  }
}

// existing voxel -> its id; new voxel -> rank key = index of its first point (creation order), counted
__global__ void ivm_lookup_kernel(
  uint32_t nseg, const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ seg_first, const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ hkeys, const uint32_t* __restrict__ hvals,
  uint32_t hmask, uint32_t* __restrict__ seg_vid, uint32_t* __restrict__ rank_key, uint32_t* __restrict__ seg_ids, unsigned int* __restrict__ n_new) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nseg) return;
  const uint32_t v = hmask ? ivm_find(hkeys, hvals, hmask, keys[seg_start[s]]) : 0xffffffffu;
  seg_vid[s] = v;
  rank_key[s] = v == 0xffffffffu ? seg_first[s] : 0xffffffffu;
  seg_ids[s] = s;
  if (v == 0xffffffffu) atomicAdd(n_new, 1u);
}

__global__ void ivm_assign_kernel(uint32_t n_new, const uint32_t* __restrict__ seg_by_rank, uint32_t n_old, uint32_t* __restrict__ seg_vid) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n_new) seg_vid[seg_by_rank[r]] = n_old + r;
}

__device__ __forceinline__ void ivm_hash_insert(unsigned long long* __restrict__ hkeys, uint32_t* __restrict__ hvals, uint32_t hmask, unsigned long long key, uint32_t v) {
  uint32_t slot = voxel_hash(key) & hmask;
  for (;;) {
    const unsigned long long prev = atomicCAS(&hkeys[slot], SGA_HASH_EMPTY, key);
    if (prev == SGA_HASH_EMPTY) {
      hvals[slot] = v;
      return;
    }
    slot = (slot + 1) & hmask;
  }
}

__global__ void ivm_rehash_kernel(uint32_t n, const int* __restrict__ coords, unsigned long long* __restrict__ hkeys, uint32_t* __restrict__ hvals, uint32_t hmask) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  ivm_hash_insert(hkeys, hvals, hmask, voxel_key(coords[3 * v], coords[3 * v + 1], coords[3 * v + 2]), v);
}

// One lane per voxel of the batch: GaussianVoxel::add for its points in insertion order, then finalize (gaussian_voxelmap.hpp:32-53).
__global__ void ivm_update_kernel(
  uint32_t nseg, const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ seg_vid, uint32_t n_valid, const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ order, const float4* __restrict__ pts,
  const Cov8* __restrict__ cov, Pose12 T, uint32_t n_old, uint32_t lru_counter, double* __restrict__ mean64, double* __restrict__ cov64, uint32_t* __restrict__ counts, uint32_t* __restrict__ lru, int* __restrict__ coords,
  unsigned long long* __restrict__ hkeys, uint32_t* __restrict__ hvals, uint32_t hmask) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nseg) return;
  const uint32_t v = seg_vid[s];
  const uint32_t first = seg_start[s];
  const unsigned long long key = keys[first];
  const bool is_new = v >= n_old;
  uint32_t N = is_new ? 0u : counts[v];
  double m[3] = {0, 0, 0}, c[6] = {0, 0, 0, 0, 0, 0};
  if (!is_new) {  // un-finalize: mean *= num_points, cov *= num_points
    for (int k = 0; k < 3; k++) m[k] = mean64[3 * v + k] * static_cast<double>(N);
    for (int k = 0; k < 6; k++) c[k] = cov64[6 * v + k] * static_cast<double>(N);
  }
  for (uint32_t i = first; i < n_valid && keys[i] == key; ++i) {
    const uint32_t src = order[i];
    const float4 p = pts[src];
    const Cov8 q = cov[src];
    m[0] += T.r[0] * p.x + T.r[1] * p.y + T.r[2] * p.z + T.t[0];
    m[1] += T.r[3] * p.x + T.r[4] * p.y + T.r[5] * p.z + T.t[1];
    m[2] += T.r[6] * p.x + T.r[7] * p.y + T.r[8] * p.z + T.t[2];
    // R C R^T (T.matrix() * cov * T.matrix().transpose(): the translation column meets the zero row of the 4x4 covariance)
    const double C[3][3] = {{q.xx, q.xy, q.xz}, {q.xy, q.yy, q.yz}, {q.xz, q.yz, q.zz}};
    double RC[3][3];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) RC[a][b] = T.r[3 * a] * C[0][b] + T.r[3 * a + 1] * C[1][b] + T.r[3 * a + 2] * C[2][b];
    int k = 0;
    for (int a = 0; a < 3; a++)
      for (int b = a; b < 3; b++) c[k++] += RC[a][0] * T.r[3 * b] + RC[a][1] * T.r[3 * b + 1] + RC[a][2] * T.r[3 * b + 2];
    N++;
  }
  for (int k = 0; k < 3; k++) mean64[3 * v + k] = m[k] / static_cast<double>(N);
  for (int k = 0; k < 6; k++) cov64[6 * v + k] = c[k] / static_cast<double>(N);
  counts[v] = N;
  lru[v] = lru_counter;
  if (is_new) {
    coords[3 * v + 0] = static_cast<int>(key & 0x1fffffu) - (1 << 20);
    coords[3 * v + 1] = static_cast<int>((key >> 21) & 0x1fffffu) - (1 << 20);
    coords[3 * v + 2] = static_cast<int>((key >> 42) & 0x1fffffu) - (1 << 20);
    ivm_hash_insert(hkeys, hvals, hmask, key, v);
  }
}

__global__ void ivm_keep_kernel(uint32_t n, const uint32_t* __restrict__ lru, uint32_t horizon, uint32_t counter, uint32_t* __restrict__ keep) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) keep[v] = (static_cast<unsigned long long>(lru[v]) + horizon < counter) ? 0u : 1u;
}

__global__ void ivm_compact_kernel(
  uint32_t n, const uint32_t* __restrict__ keep, const uint32_t* __restrict__ pos, const double* __restrict__ m_in, const double* __restrict__ c_in, const uint32_t* __restrict__ cnt_in, const uint32_t* __restrict__ lru_in,
  const int* __restrict__ co_in, double* __restrict__ m_out, double* __restrict__ c_out, uint32_t* __restrict__ cnt_out, uint32_t* __restrict__ lru_out, int* __restrict__ co_out) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n || !keep[v]) return;
  const uint32_t w = pos[v];
  for (int k = 0; k < 3; k++) {
    m_out[3 * w + k] = m_in[3 * v + k];
    co_out[3 * w + k] = co_in[3 * v + k];
  }
  for (int k = 0; k < 6; k++) c_out[6 * w + k] = c_in[6 * v + k];
  cnt_out[w] = cnt_in[v];
  lru_out[w] = lru_in[v];
}

// (ox, oy, oz): origin of the map's device frame (common.hpp) — the fp64 state is the caller's frame, the fp32 records the kernels read are not
__global__ void ivm_export_kernel(uint32_t n, const double* __restrict__ mean64, const double* __restrict__ cov64, double ox, double oy, double oz, float4* __restrict__ means, Cov8* __restrict__ mcov) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  means[v] = make_float4(static_cast<float>(mean64[3 * v] - ox), static_cast<float>(mean64[3 * v + 1] - oy), static_cast<float>(mean64[3 * v + 2] - oz), __uint_as_float(v));
  Cov8 o;
  o.xx = static_cast<float>(cov64[6 * v]);
  o.xy = static_cast<float>(cov64[6 * v + 1]);
  o.xz = static_cast<float>(cov64[6 * v + 2]);
  o.yy = static_cast<float>(cov64[6 * v + 3]);
  o.yz = static_cast<float>(cov64[6 * v + 4]);
  o.zz = static_cast<float>(cov64[6 * v + 5]);
  o.pad0 = o.pad1 = 0.f;
  mcov[v] = o;
}

// ---- flat maps: IncrementalVoxelMap<FlatContainerCov> ---------------------------------------------------------------------------
// One lane per voxel of the batch: FlatContainer::add for its points in insertion order (flat_container.hpp:33-51): a point is
// kept iff the cell holds fewer than max_points and no kept point lies closer than sqrt(min_sq); kept = T p with covariance R C R^T.
__global__ void fvm_update_kernel(
  uint32_t nseg, const uint32_t* __restrict__ seg_start, const uint32_t* __restrict__ seg_vid, uint32_t n_valid, const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ order, const float4* __restrict__ pts,
  const Cov8* __restrict__ cov, Pose12 T, uint32_t n_old, uint32_t lru_counter, uint32_t max_points, double min_sq, double* __restrict__ fpts64, double* __restrict__ fcov64, uint32_t* __restrict__ counts,
  uint32_t* __restrict__ lru, int* __restrict__ coords, unsigned long long* __restrict__ hkeys, uint32_t* __restrict__ hvals, uint32_t hmask) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nseg) return;
  const uint32_t v = seg_vid[s];
  const uint32_t first = seg_start[s];
  const unsigned long long key = keys[first];
  const bool is_new = v >= n_old;
  uint32_t cnt = is_new ? 0u : counts[v];
  double* P = fpts64 + static_cast<size_t>(v) * kFlatCap * 3;
  double* C6 = fcov64 + static_cast<size_t>(v) * kFlatCap * 6;
  for (uint32_t i = first; i < n_valid && keys[i] == key; ++i) {
    if (cnt >= max_points) break;  // every further point of this batch would be rejected as well
    const uint32_t src = order[i];
    const float4 p = pts[src];
    const double x = T.r[0] * p.x + T.r[1] * p.y + T.r[2] * p.z + T.t[0];
    const double y = T.r[3] * p.x + T.r[4] * p.y + T.r[5] * p.z + T.t[1];
    const double z = T.r[6] * p.x + T.r[7] * p.y + T.r[8] * p.z + T.t[2];
    bool reject = false;
    for (uint32_t j = 0; j < cnt && !reject; j++) {
      const double dx = P[3 * j] - x, dy = P[3 * j + 1] - y, dz = P[3 * j + 2] - z;
      reject = dx * dx + dy * dy + dz * dz < min_sq;
    }
    if (reject) continue;
    P[3 * cnt] = x;
    P[3 * cnt + 1] = y;
    P[3 * cnt + 2] = z;
    const Cov8 q = cov[src];
    const double Cm[3][3] = {{q.xx, q.xy, q.xz}, {q.xy, q.yy, q.yz}, {q.xz, q.yz, q.zz}};
    double RC[3][3];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) RC[a][b] = T.r[3 * a] * Cm[0][b] + T.r[3 * a + 1] * Cm[1][b] + T.r[3 * a + 2] * Cm[2][b];
    int k = 0;
    for (int a = 0; a < 3; a++)
      for (int b = a; b < 3; b++) C6[6 * cnt + (k++)] = RC[a][0] * T.r[3 * b] + RC[a][1] * T.r[3 * b + 1] + RC[a][2] * T.r[3 * b + 2];
    cnt++;
  }
  counts[v] = cnt;
  lru[v] = lru_counter;
  if (is_new) {
    coords[3 * v + 0] = static_cast<int>(key & 0x1fffffu) - (1 << 20);
    coords[3 * v + 1] = static_cast<int>((key >> 21) & 0x1fffffu) - (1 << 20);
    coords[3 * v + 2] = static_cast<int>((key >> 42) & 0x1fffffu) - (1 << 20);
    ivm_hash_insert(hkeys, hvals, hmask, key, v);
  }
}

__global__ void fvm_compact_kernel(
  uint32_t n, const uint32_t* __restrict__ keep, const uint32_t* __restrict__ pos, const double* __restrict__ p_in, const double* __restrict__ c_in, const uint32_t* __restrict__ cnt_in, const uint32_t* __restrict__ lru_in,
  const int* __restrict__ co_in, double* __restrict__ p_out, double* __restrict__ c_out, uint32_t* __restrict__ cnt_out, uint32_t* __restrict__ lru_out, int* __restrict__ co_out) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n || !keep[v]) return;
  const uint32_t w = pos[v];
  for (int k = 0; k < 3; k++) co_out[3 * w + k] = co_in[3 * v + k];
  const uint32_t cnt = cnt_in[v];
  for (uint32_t j = 0; j < cnt; j++) {
    for (int k = 0; k < 3; k++) p_out[(static_cast<size_t>(w) * kFlatCap + j) * 3 + k] = p_in[(static_cast<size_t>(v) * kFlatCap + j) * 3 + k];
    for (int k = 0; k < 6; k++) c_out[(static_cast<size_t>(w) * kFlatCap + j) * 6 + k] = c_in[(static_cast<size_t>(v) * kFlatCap + j) * 6 + k];
  }
  cnt_out[w] = cnt;
  lru_out[w] = lru_in[v];
}

// fp32 records read by the factor kernels: slot = voxel * kFlatCap + i, w = slot
__global__ void fvm_export_kernel(uint32_t n, const uint32_t* __restrict__ counts, const double* __restrict__ fpts64, const double* __restrict__ fcov64, double ox, double oy, double oz, float4* __restrict__ pts, Cov8* __restrict__ cov) {
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t v = t / kFlatCap, j = t % kFlatCap;
  if (v >= n) return;
  if (j >= counts[v]) {
    pts[t] = make_float4(INFINITY, INFINITY, INFINITY, __uint_as_float(t));
    return;
  }
  pts[t] = make_float4(static_cast<float>(fpts64[3 * static_cast<size_t>(t)] - ox), static_cast<float>(fpts64[3 * static_cast<size_t>(t) + 1] - oy), static_cast<float>(fpts64[3 * static_cast<size_t>(t) + 2] - oz), __uint_as_float(t));
  Cov8 o;
  const double* c = fcov64 + 6 * static_cast<size_t>(t);
  o.xx = static_cast<float>(c[0]);
  o.xy = static_cast<float>(c[1]);
  o.xz = static_cast<float>(c[2]);
  o.yy = static_cast<float>(c[3]);
  o.yz = static_cast<float>(c[4]);
  o.zz = static_cast<float>(c[5]);
  o.pad0 = o.pad1 = 0.f;
  cov[t] = o;
}

// origin of a device frame centred on n host points (3 doubles each)
static void host_origin(const double* xyz, size_t n, double origin[3]) {
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (size_t i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) {
      const double x = xyz[3 * i + k];
      if (x - x == 0.0) lo[k] = x < lo[k] ? x : lo[k], hi[k] = x > hi[k] ? x : hi[k];
    }
  choose_origin(lo, hi, origin);
}

template <typename T>
static int grow(sga_context* ctx, DevBuf<T>& buf, size_t used, size_t want) {
  if (buf.n >= want) return SGA_OK;
  DevBuf<T> bigger;
  SGA_TRY(bigger.alloc(want));
  if (used > 0) SGA_HIP(hipMemcpyAsync(bigger.p, buf.p, used * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
  SGA_HIP(hipStreamSynchronize(ctx->stream));  // the old block goes back to the allocator
  buf.swap(bigger);
  return SGA_OK;
}

static int rebuild_hash(sga_context* ctx, sga_index* idx, size_t n_target) {
  uint32_t hsize = 1024;
  while (hsize < 4 * n_target) hsize <<= 1;  // at most 1/4 full right after a rebuild, rebuilt again at 1/2
  if (idx->hkeys.n != hsize) {
    SGA_TRY(idx->hkeys.alloc(hsize));
    SGA_TRY(idx->hvals.alloc(hsize));
  }
  idx->hmask = hsize - 1;
  SGA_HIP(hipMemsetAsync(idx->hkeys.p, 0xff, hsize * sizeof(unsigned long long), ctx->stream));
  SGA_HIP(hipMemsetAsync(idx->hvals.p, 0, hsize * sizeof(uint32_t), ctx->stream));
  if (idx->n > 0) hipLaunchKernelGGL(ivm_rehash_kernel, dim3((idx->n + 255) / 256), dim3(256), 0, ctx->stream, static_cast<uint32_t>(idx->n), idx->vcoords.p, idx->hkeys.p, idx->hvals.p, idx->hmask);
  SGA_HIP(hipGetLastError());
  return SGA_OK;
}

}  // namespace sga

using namespace sga;

extern "C" {

int sga_voxelmap_create(sga_context* ctx, double leaf, sga_index** out) {
  if (!ctx || !out) return fail(SGA_ERR_INVALID, "null argument");
  if (!(leaf > 0)) return fail(SGA_ERR_INVALID, "leaf size must be positive");
  *out = nullptr;
  std::unique_ptr<sga_index> idx(new sga_index);
  idx->kind = SGA_INDEX_VOXELMAP;
  idx->device = ctx->device;
  idx->leaf = leaf;
  idx->has_covs = true;
  idx->incremental = true;
  *out = idx.release();
  return SGA_OK;
}

// A Gaussian voxel map from voxels that already exist on the host: the reference's GaussianVoxelMap object itself (its flat_voxels in
// flat order: coord, mean, cov; ann/incremental_voxelmap.hpp:39-92, gaussian_voxelmap.hpp:15-60) handed over as it is, so that
// Registration<GICPFactor, ParallelReductionHIP>::align(voxelmap, source, voxelmap) (registration_helper.cpp:125-137) searches exactly the
// voxels the caller built, with the caller's voxel ids.
int sga_index_create_voxelmap_from_voxels(sga_context* ctx, double leaf, const int32_t* coords, const double* means3, const double* cov6, size_t n, sga_index** out) {
  if (!ctx || !out || (n > 0 && (!coords || !means3 || !cov6))) return fail(SGA_ERR_INVALID, "null argument");
  if (!(leaf > 0)) return fail(SGA_ERR_INVALID, "leaf size must be positive");
  if (n >= (1ull << 31)) return fail(SGA_ERR_INVALID, "too many voxels");
  *out = nullptr;
  SGA_ENTER(ctx);
  std::unique_ptr<sga_index> idx(new sga_index);
  idx->kind = SGA_INDEX_VOXELMAP;
  idx->device = ctx->device;
  idx->leaf = leaf;
  idx->has_covs = true;
  idx->n = n;
  if (n > 0) {
    DevBuf<double> d_mean, d_cov;
    SGA_TRY(d_mean.alloc(3 * n));
    SGA_TRY(d_cov.alloc(6 * n));
    SGA_TRY(idx->vcoords.alloc(3 * n));
    SGA_TRY(idx->vcounts.alloc(n));
    SGA_TRY(idx->pts.alloc(n));
    SGA_TRY(idx->cov.alloc(n));
    SGA_HIP(hipMemcpyAsync(d_mean.p, means3, 3 * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SGA_HIP(hipMemcpyAsync(d_cov.p, cov6, 6 * n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SGA_HIP(hipMemcpyAsync(idx->vcoords.p, coords, 3 * n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    SGA_HIP(hipMemsetAsync(idx->vcounts.p, 0, n * sizeof(uint32_t), ctx->stream));  // (the number of points behind a voxel is not part of what the registration reads)
    host_origin(means3, n, idx->origin);
    hipLaunchKernelGGL(ivm_export_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, static_cast<uint32_t>(n), d_mean.p, d_cov.p, idx->origin[0], idx->origin[1], idx->origin[2], idx->pts.p, idx->cov.p);
    SGA_HIP(hipGetLastError());
    SGA_TRY(rebuild_hash(ctx, idx.get(), n));
    SGA_HIP(hipStreamSynchronize(ctx->stream));  // the host buffers are the caller's
  } else {
    SGA_TRY(rebuild_hash(ctx, idx.get(), 0));
  }
  *out = idx.release();
  return SGA_OK;
}

// A flat voxel map from voxels that already exist on the host: the reference's IncrementalVoxelMap<FlatContainer*> object as it is
// (flat order; per voxel its points and, for GICP, their covariances: flat_container.hpp:21-58), 16 slots per voxel like
// sga_flatmap_download.  The scan-to-model target of Registration<GICPFactor, ParallelReductionHIP> (odometry_benchmark_small_gicp_model_omp.cpp).
int sga_index_create_flatmap_from_voxels(sga_context* ctx, double leaf, const int32_t* coords, const uint32_t* counts, const double* points3, const double* cov6, int search_offsets, size_t n, sga_index** out) {
  if (!ctx || !out || (n > 0 && (!coords || !counts || !points3))) return fail(SGA_ERR_INVALID, "null argument");
  if (!(leaf > 0)) return fail(SGA_ERR_INVALID, "leaf size must be positive");
  if (search_offsets != 1 && search_offsets != 7 && search_offsets != 27) return fail(SGA_ERR_INVALID, "search offsets must be 1, 7 or 27 (incremental_voxelmap.hpp:157-186)");
  if (n >= (1ull << 27)) return fail(SGA_ERR_INVALID, "too many voxels");
  for (size_t v = 0; v < n; v++)
    if (counts[v] > static_cast<uint32_t>(kFlatCap)) return fail(SGA_ERR_UNSUPPORTED, "voxel %zu holds %u points: at most %d per voxel (max_num_points_in_cell)", v, counts[v], kFlatCap);
  *out = nullptr;
  SGA_ENTER(ctx);
  std::unique_ptr<sga_index> idx(new sga_index);
  idx->kind = SGA_INDEX_FLATMAP;
  idx->device = ctx->device;
  idx->leaf = leaf;
  idx->has_covs = cov6 != nullptr;
  idx->search_offsets = search_offsets;
  idx->n = n;
  if (n > 0) {
    const size_t slots = n * kFlatCap;
    DevBuf<double> d_pts, d_cov;
    SGA_TRY(d_pts.alloc(3 * slots));
    SGA_TRY(d_cov.alloc(6 * slots));
    SGA_TRY(idx->vcoords.alloc(3 * n));
    SGA_TRY(idx->vcounts.alloc(n));
    SGA_TRY(idx->pts.alloc(slots));
    SGA_TRY(idx->cov.alloc(slots));
    SGA_HIP(hipMemcpyAsync(d_pts.p, points3, 3 * slots * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (cov6)
      SGA_HIP(hipMemcpyAsync(d_cov.p, cov6, 6 * slots * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    else
      SGA_HIP(hipMemsetAsync(d_cov.p, 0, 6 * slots * sizeof(double), ctx->stream));
    SGA_HIP(hipMemcpyAsync(idx->vcoords.p, coords, 3 * n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    SGA_HIP(hipMemcpyAsync(idx->vcounts.p, counts, n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
    {  // the device frame: centred on the valid points
      double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
      for (size_t v = 0; v < n; v++)
        for (uint32_t j = 0; j < counts[v]; j++)
          for (int k = 0; k < 3; k++) {
            const double x = points3[3 * (v * kFlatCap + j) + k];
            if (x - x == 0.0) lo[k] = x < lo[k] ? x : lo[k], hi[k] = x > hi[k] ? x : hi[k];
          }
      choose_origin(lo, hi, idx->origin);
    }
    hipLaunchKernelGGL(fvm_export_kernel, dim3((slots + 255) / 256), dim3(256), 0, ctx->stream, static_cast<uint32_t>(n), idx->vcounts.p, d_pts.p, d_cov.p, idx->origin[0], idx->origin[1], idx->origin[2], idx->pts.p, idx->cov.p);
    SGA_HIP(hipGetLastError());
    SGA_TRY(rebuild_hash(ctx, idx.get(), n));
    SGA_HIP(hipStreamSynchronize(ctx->stream));  // the host buffers are the caller's
  } else {
    SGA_TRY(rebuild_hash(ctx, idx.get(), 0));
  }
  *out = idx.release();
  return SGA_OK;
}

int sga_flatmap_create(sga_context* ctx, double leaf, sga_index** out) {
  SGA_TRY(sga_voxelmap_create(ctx, leaf, out));
  (*out)->kind = SGA_INDEX_FLATMAP;
  return SGA_OK;
}

int sga_flatmap_set_setting(sga_index* index, double min_sq_dist_in_cell, uint32_t max_num_points_in_cell) {
  if (!index || index->kind != SGA_INDEX_FLATMAP) return fail(SGA_ERR_INVALID, "not a flat voxel map");
  if (max_num_points_in_cell == 0 || max_num_points_in_cell > static_cast<uint32_t>(kFlatCap)) return fail(SGA_ERR_INVALID, "max_num_points_in_cell must be in [1, %d]", kFlatCap);
  if (index->n > 0) return fail(SGA_ERR_INVALID, "the cell setting must be chosen before the first insert");
  index->flat_min_sq = min_sq_dist_in_cell;
  index->flat_max = max_num_points_in_cell;
  return SGA_OK;
}

int sga_voxelmap_set_search_offsets(sga_index* index, int num_offsets) {
  if (!index || (index->kind != SGA_INDEX_VOXELMAP && index->kind != SGA_INDEX_FLATMAP)) return fail(SGA_ERR_INVALID, "not a voxel map");
  if (num_offsets != 1 && num_offsets != 7 && num_offsets != 27) return fail(SGA_ERR_INVALID, "search offsets must be 1, 7 or 27");
  index->search_offsets = num_offsets;
  return SGA_OK;
}

// per voxel: coords (3 ints) and number of points; points (3 floats) and cov6 (6 floats) for kFlatCap slots per voxel
int sga_flatmap_download(sga_context* ctx, const sga_index* index, int32_t* coords, uint32_t* counts, float* points, float* cov6) {
  if (!ctx || !index) return fail(SGA_ERR_INVALID, "null argument");
  if (index->kind != SGA_INDEX_FLATMAP) return fail(SGA_ERR_INVALID, "not a flat voxel map");
  const size_t n = index->n;
  if (n == 0) return SGA_OK;
  SGA_ENTER(ctx);
  SGA_TRY(wait_ready(ctx, index->ready));
  std::vector<float4> hp;
  std::vector<Cov8> hc;
  if (points) {
    hp.resize(n * kFlatCap);
    SGA_HIP(hipMemcpyAsync(hp.data(), index->pts.p, hp.size() * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
  }
  if (cov6) {
    hc.resize(n * kFlatCap);
    SGA_HIP(hipMemcpyAsync(hc.data(), index->cov.p, hc.size() * sizeof(Cov8), hipMemcpyDeviceToHost, ctx->stream));
  }
  if (coords) SGA_HIP(hipMemcpyAsync(coords, index->vcoords.p, n * 3 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  if (counts) SGA_HIP(hipMemcpyAsync(counts, index->vcounts.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  SGA_HIP(hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < n * kFlatCap; i++) {
    if (points) {  // device frame -> the caller's (empty slots hold +inf and stay +inf)
      points[3 * i] = static_cast<float>(static_cast<double>(hp[i].x) + index->origin[0]);
      points[3 * i + 1] = static_cast<float>(static_cast<double>(hp[i].y) + index->origin[1]);
      points[3 * i + 2] = static_cast<float>(static_cast<double>(hp[i].z) + index->origin[2]);
    }
    if (cov6) {
      cov6[6 * i] = hc[i].xx;
      cov6[6 * i + 1] = hc[i].xy;
      cov6[6 * i + 2] = hc[i].xz;
      cov6[6 * i + 3] = hc[i].yy;
      cov6[6 * i + 4] = hc[i].yz;
      cov6[6 * i + 5] = hc[i].zz;
    }
  }
  return SGA_OK;
}

int sga_voxelmap_set_lru(sga_index* index, uint32_t horizon, uint32_t clear_cycle) {
  if (!index || !index->incremental) return fail(SGA_ERR_INVALID, "not an incremental voxel map");
  if (clear_cycle == 0) return fail(SGA_ERR_INVALID, "clear_cycle must be positive");
  index->lru_horizon = horizon;
  index->lru_clear_cycle = clear_cycle;
  return SGA_OK;
}

int sga_voxelmap_insert(sga_context* ctx, sga_index* idx, const sga_cloud* cloud, const double T16[16]) {
  if (!ctx || !idx || !cloud) return fail(SGA_ERR_INVALID, "null argument");
  if (!idx->incremental) return fail(SGA_ERR_INVALID, "not an incremental voxel map (create it with sga_voxelmap_create)");
  if (cloud->n > 0 && !cloud->has_covs) return fail(SGA_ERR_INVALID, "GaussianVoxelMap needs point covariances");
  if (cloud->device != ctx->device || idx->device != ctx->device) return fail(SGA_ERR_INVALID, "cloud / map live on another device");
  SGA_ENTER(ctx);
  SGA_TRY(wait_ready(ctx, cloud->ready));
  SGA_TRY(wait_ready(ctx, idx->ready));
  // The cloud's records live in its device frame (p' = p - o_c, common.hpp); the map's fp64 state, voxel coordinates and hash keys are the
  // CALLER's frame, exactly the reference's doubles: the kernels move a record by R p' + (R o_c + t).
  Pose12 T;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) T.r[3 * r + c] = T16 ? T16[4 * c + r] : (r == c ? 1.0 : 0.0);
    T.t[r] = T16 ? T16[12 + r] : 0.0;
  }
  for (int r = 0; r < 3; r++) T.t[r] += T.r[3 * r] * cloud->origin[0] + T.r[3 * r + 1] * cloud->origin[1] + T.r[3 * r + 2] * cloud->origin[2];
  // The fp32 records the factor kernels read are re-exported after every insert anyway: their device frame follows the inserted scan (the
  // moved origin of the cloud's frame, quantised), so a map that walks kilometres (scan-to-model odometry) keeps sub-millimetre records.
  // (the new origin is committed only where the records are exported below — an insert that fails on the way leaves the map's
  // fp32 records AND the origin they are relative to as they were)
  double new_origin[3] = {idx->origin[0], idx->origin[1], idx->origin[2]};
  if (cloud->n > 0) {
    const double lo[3] = {T.t[0], T.t[1], T.t[2]};
    choose_origin(lo, lo, new_origin);
  }
  const size_t n = cloud->n;
  const uint32_t n_old = static_cast<uint32_t>(idx->n);
  if (n > 0) {
    DevBuf<unsigned long long> keys, keys_sorted;
    DevBuf<uint32_t> vals, order, flags, seg_id;
    DevBuf<unsigned int> d_new;
    SGA_TRY(keys.alloc(n));
    SGA_TRY(keys_sorted.alloc(n));
    SGA_TRY(vals.alloc(n));
    SGA_TRY(order.alloc(n));
    SGA_TRY(flags.alloc(n));
    SGA_TRY(seg_id.alloc(n));
    SGA_TRY(d_new.alloc(1));
    const dim3 grid((n + 255) / 256), block(256);
    hipLaunchKernelGGL(ivm_keys_kernel, grid, block, 0, ctx->stream, cloud->pts.p, n, T, 1.0 / idx->leaf, keys.p, vals.p);
    size_t tb = 0;
    SGA_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys.p, keys_sorted.p, vals.p, order.p, n, 0, 64, ctx->stream));
    SGA_TRY(ensure_temp(ctx, tb));
    SGA_HIP(rocprim::radix_sort_pairs(ctx->d_temp.p, tb, keys.p, keys_sorted.p, vals.p, order.p, n, 0, 64, ctx->stream));
    hipLaunchKernelGGL(ivm_heads_kernel, grid, block, 0, ctx->stream, keys_sorted.p, n, flags.p);
    size_t tb2 = 0;
    SGA_HIP(rocprim::exclusive_scan(nullptr, tb2, flags.p, seg_id.p, 0u, n, rocprim::plus<uint32_t>(), ctx->stream));
    SGA_TRY(ensure_temp(ctx, tb2));
    SGA_HIP(rocprim::exclusive_scan(ctx->d_temp.p, tb2, flags.p, seg_id.p, 0u, n, rocprim::plus<uint32_t>(), ctx->stream));
    uint32_t last_flag = 0, last_seg = 0;
    SGA_HIP(hipMemcpyAsync(&last_flag, flags.p + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    SGA_HIP(hipMemcpyAsync(&last_seg, seg_id.p + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    SGA_HIP(hipStreamSynchronize(ctx->stream));
    const uint32_t nseg = last_seg + last_flag;
    if (nseg > 0) {
      DevBuf<uint32_t> seg_start, seg_first, seg_vid, rank_key, rank_sorted, seg_ids, seg_by_rank;
      SGA_TRY(seg_start.alloc(nseg));
      SGA_TRY(seg_first.alloc(nseg));
      SGA_TRY(seg_vid.alloc(nseg));
      SGA_TRY(rank_key.alloc(nseg));
      SGA_TRY(rank_sorted.alloc(nseg));
      SGA_TRY(seg_ids.alloc(nseg));
      SGA_TRY(seg_by_rank.alloc(nseg));
      const dim3 sgrid((nseg + 255) / 256);
      hipLaunchKernelGGL(ivm_segments_kernel, grid, block, 0, ctx->stream, flags.p, seg_id.p, order.p, n, seg_start.p, seg_first.p);
      SGA_HIP(hipMemsetAsync(d_new.p, 0, sizeof(unsigned int), ctx->stream));
      hipLaunchKernelGGL(ivm_lookup_kernel, sgrid, block, 0, ctx->stream, nseg, seg_start.p, seg_first.p, keys_sorted.p, idx->hkeys.p, idx->hvals.p, n_old > 0 ? idx->hmask : 0u, seg_vid.p, rank_key.p, seg_ids.p, d_new.p);
      size_t tb3 = 0;
      SGA_HIP(rocprim::radix_sort_pairs(nullptr, tb3, rank_key.p, rank_sorted.p, seg_ids.p, seg_by_rank.p, nseg, 0, 32, ctx->stream));
      SGA_TRY(ensure_temp(ctx, tb3));
      SGA_HIP(rocprim::radix_sort_pairs(ctx->d_temp.p, tb3, rank_key.p, rank_sorted.p, seg_ids.p, seg_by_rank.p, nseg, 0, 32, ctx->stream));
      unsigned int n_new = 0;
      SGA_HIP(hipMemcpyAsync(&n_new, d_new.p, sizeof(n_new), hipMemcpyDeviceToHost, ctx->stream));
      SGA_HIP(hipStreamSynchronize(ctx->stream));
      const size_t n_total = static_cast<size_t>(n_old) + n_new;
      if (n_total >= (1ull << 31)) return fail(SGA_ERR_INVALID, "voxel map too large");
      // capacity of the per-voxel arrays and of the table
      if (n_total > idx->vcap) {
        const size_t cap = std::max<size_t>(2 * n_total, 1024);
        const bool flat = idx->kind == SGA_INDEX_FLATMAP;
        if (flat) {
          SGA_TRY(grow(ctx, idx->fpts64, 3 * kFlatCap * static_cast<size_t>(n_old), 3 * kFlatCap * cap));
          SGA_TRY(grow(ctx, idx->fcov64, 6 * kFlatCap * static_cast<size_t>(n_old), 6 * kFlatCap * cap));
        } else {
          SGA_TRY(grow(ctx, idx->vmean64, 3 * static_cast<size_t>(n_old), 3 * cap));
          SGA_TRY(grow(ctx, idx->vcov64, 6 * static_cast<size_t>(n_old), 6 * cap));
        }
        SGA_TRY(grow(ctx, idx->vcounts, n_old, cap));
        SGA_TRY(grow(ctx, idx->vlru, n_old, cap));
        SGA_TRY(grow(ctx, idx->vcoords, 3 * static_cast<size_t>(n_old), 3 * cap));
        SGA_TRY(grow(ctx, idx->pts, 0, flat ? kFlatCap * cap : cap));
        SGA_TRY(grow(ctx, idx->cov, 0, flat ? kFlatCap * cap : cap));
        idx->vcap = cap;
      }
      if (idx->hkeys.n == 0 || 2 * n_total > idx->hkeys.n) SGA_TRY(rebuild_hash(ctx, idx, n_total));
      if (n_new > 0) hipLaunchKernelGGL(ivm_assign_kernel, dim3((n_new + 255) / 256), block, 0, ctx->stream, n_new, seg_by_rank.p, n_old, seg_vid.p);
      if (idx->kind == SGA_INDEX_FLATMAP)
        hipLaunchKernelGGL(
          fvm_update_kernel, sgrid, block, 0, ctx->stream, nseg, seg_start.p, seg_vid.p, static_cast<uint32_t>(n), keys_sorted.p, order.p, cloud->pts.p, cloud->cov.p, T, n_old, idx->lru_counter, idx->flat_max, idx->flat_min_sq,
          idx->fpts64.p, idx->fcov64.p, idx->vcounts.p, idx->vlru.p, idx->vcoords.p, idx->hkeys.p, idx->hvals.p, idx->hmask);
      else
        hipLaunchKernelGGL(
          ivm_update_kernel, sgrid, block, 0, ctx->stream, nseg, seg_start.p, seg_vid.p, static_cast<uint32_t>(n), keys_sorted.p, order.p, cloud->pts.p, cloud->cov.p, T, n_old, idx->lru_counter, idx->vmean64.p, idx->vcov64.p,
          idx->vcounts.p, idx->vlru.p, idx->vcoords.p, idx->hkeys.p, idx->hvals.p, idx->hmask);
      SGA_HIP(hipGetLastError());
      idx->n = n_total;
    }
    SGA_HIP(hipStreamSynchronize(ctx->stream));  // the scratch buffers of this insert are released below
  }
  // LRU sweep (incremental_voxelmap.hpp:76-88)
  idx->lru_counter++;
  if (idx->lru_counter % idx->lru_clear_cycle == 0 && idx->n > 0) {
    const uint32_t nv = static_cast<uint32_t>(idx->n);
    DevBuf<uint32_t> keep, pos;
    SGA_TRY(keep.alloc(nv));
    SGA_TRY(pos.alloc(nv));
    hipLaunchKernelGGL(ivm_keep_kernel, dim3((nv + 255) / 256), dim3(256), 0, ctx->stream, nv, idx->vlru.p, idx->lru_horizon, idx->lru_counter, keep.p);
    size_t tb = 0;
    SGA_HIP(rocprim::exclusive_scan(nullptr, tb, keep.p, pos.p, 0u, nv, rocprim::plus<uint32_t>(), ctx->stream));
    SGA_TRY(ensure_temp(ctx, tb));
    SGA_HIP(rocprim::exclusive_scan(ctx->d_temp.p, tb, keep.p, pos.p, 0u, nv, rocprim::plus<uint32_t>(), ctx->stream));
    uint32_t last_keep = 0, last_pos = 0;
    SGA_HIP(hipMemcpyAsync(&last_keep, keep.p + (nv - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    SGA_HIP(hipMemcpyAsync(&last_pos, pos.p + (nv - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    SGA_HIP(hipStreamSynchronize(ctx->stream));
    const uint32_t kept = last_pos + last_keep;
    if (kept < nv) {
      DevBuf<double> m2, c2;
      DevBuf<uint32_t> cnt2, lru2;
      DevBuf<int> co2;
      const bool flat = idx->kind == SGA_INDEX_FLATMAP;
      SGA_TRY(m2.alloc((flat ? 3 * kFlatCap : 3) * idx->vcap));
      SGA_TRY(c2.alloc((flat ? 6 * kFlatCap : 6) * idx->vcap));
      SGA_TRY(cnt2.alloc(idx->vcap));
      SGA_TRY(lru2.alloc(idx->vcap));
      SGA_TRY(co2.alloc(3 * idx->vcap));
      if (flat)
        hipLaunchKernelGGL(fvm_compact_kernel, dim3((nv + 255) / 256), dim3(256), 0, ctx->stream, nv, keep.p, pos.p, idx->fpts64.p, idx->fcov64.p, idx->vcounts.p, idx->vlru.p, idx->vcoords.p, m2.p, c2.p, cnt2.p, lru2.p, co2.p);
      else
        hipLaunchKernelGGL(ivm_compact_kernel, dim3((nv + 255) / 256), dim3(256), 0, ctx->stream, nv, keep.p, pos.p, idx->vmean64.p, idx->vcov64.p, idx->vcounts.p, idx->vlru.p, idx->vcoords.p, m2.p, c2.p, cnt2.p, lru2.p, co2.p);
      SGA_HIP(hipGetLastError());
      SGA_HIP(hipStreamSynchronize(ctx->stream));
      if (flat) {
        idx->fpts64.swap(m2);
        idx->fcov64.swap(c2);
      } else {
        idx->vmean64.swap(m2);
        idx->vcov64.swap(c2);
      }
      idx->vcounts.swap(cnt2);
      idx->vlru.swap(lru2);
      idx->vcoords.swap(co2);
      idx->n = kept;
      SGA_TRY(rebuild_hash(ctx, idx, kept));
    }
  }
  for (int k = 0; k < 3; k++) idx->origin[k] = new_origin[k];  // every fallible step is behind us: the records exported now are relative to it
  if (idx->n > 0) {
    if (idx->kind == SGA_INDEX_FLATMAP)
      hipLaunchKernelGGL(fvm_export_kernel, dim3((idx->n * kFlatCap + 255) / 256), dim3(256), 0, ctx->stream, static_cast<uint32_t>(idx->n), idx->vcounts.p, idx->fpts64.p, idx->fcov64.p, idx->origin[0], idx->origin[1], idx->origin[2], idx->pts.p, idx->cov.p);
    else
      hipLaunchKernelGGL(ivm_export_kernel, dim3((idx->n + 255) / 256), dim3(256), 0, ctx->stream, static_cast<uint32_t>(idx->n), idx->vmean64.p, idx->vcov64.p, idx->origin[0], idx->origin[1], idx->origin[2], idx->pts.p, idx->cov.p);
  }
  SGA_HIP(hipGetLastError());
  SGA_HIP(hipStreamSynchronize(ctx->stream));
  return SGA_OK;
}

}  // extern "C"
