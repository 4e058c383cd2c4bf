// The uniform cell grid of a kd-tree index (cell_grid.hpp): build, and the two search kernels of a grid pass.
//
// Replaces, for cold passes near the optimum, the kd walk that stands for KdTree::nearest_neighbor_search (ann/kdtree.hpp:193-233,
// knn_result.hpp:80-100: exact, epsilon = 0) inside ParallelReductionOMP::linearize (registration/reduction_omp.hpp:24-59).  Same
// canonical neighbour, same certificate outputs (nn / nn2 / rex) — the passes that follow cannot tell which search wrote them.
#include "common.hpp"

#include <algorithm>
#include <cmath>
#include <rocprim/rocprim.hpp>

#include "cell_grid.hpp"
#include "device_math.hpp"
#include "kd_search.hpp"
#include "sort_util.hpp"

namespace sga {

int ensure_temp(sga_context* ctx, size_t bytes);
extern int g_grid_mode;              // linearize.hip
extern long long g_grid_min_points;

// ---- build ---------------------------------------------------------------------------------------------------------------------
__global__ void grid_keys_kernel(const float4* __restrict__ kd_pts, uint32_t n, GridView g, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 p = kd_pts[i];
  const int cx = grid_cell(p.x, g.ox, g.inv_h, g.nx), cy = grid_cell(p.y, g.oy, g.inv_h, g.ny), cz = grid_cell(p.z, g.oz, g.inv_h, g.nz);
  keys[i] = static_cast<uint32_t>((cz * g.ny + cy) * g.nx + cx);
  if (vals != nullptr) vals[i] = i;
}

// number of occupied cells of a trial grid without a sort (round 6: the trial loop of build_cell_grid sorted the cell keys of all points
// four times to count the distinct ones — 0.2 ms per trial at 1M): every point sets its cell's bit in a bitmap; the thread that finds
// the bit clear has found a new cell.  Counts are added per workgroup.
__global__ __launch_bounds__(256) void grid_mark_cells_kernel(const float4* __restrict__ kd_pts, uint32_t n, GridView g, uint32_t* __restrict__ bitmap, uint32_t* __restrict__ out) {
  __shared__ uint32_t sh;
  if (threadIdx.x == 0) sh = 0u;
  __syncthreads();
  uint32_t c = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float4 p = kd_pts[i];
    const int cx = grid_cell(p.x, g.ox, g.inv_h, g.nx), cy = grid_cell(p.y, g.oy, g.inv_h, g.ny), cz = grid_cell(p.z, g.oz, g.inv_h, g.nz);
    const uint32_t key = static_cast<uint32_t>((cz * g.ny + cy) * g.nx + cx);
    const uint32_t bit = 1u << (key & 31u);
    if ((bitmap[key >> 5] & bit) == 0u) c += (atomicOr(&bitmap[key >> 5], bit) & bit) == 0u ? 1u : 0u;
  }
  atomicAdd(&sh, c);
  __syncthreads();
  if (threadIdx.x == 0 && sh != 0u) atomicAdd(out, sh);
}

// number of distinct values in a sorted array (= occupied cells)
__global__ __launch_bounds__(256) void grid_count_distinct_kernel(const uint32_t* __restrict__ sorted, uint32_t n, uint32_t* __restrict__ out) {
  __shared__ uint32_t sh;
  if (threadIdx.x == 0) sh = 0u;
  __syncthreads();
  uint32_t c = 0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) c += (i == 0 || sorted[i] != sorted[i - 1]) ? 1u : 0u;
  atomicAdd(&sh, c);
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, sh);
}

__global__ void grid_histogram_kernel(const uint32_t* __restrict__ keys, uint32_t n, uint32_t* __restrict__ counts) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(&counts[keys[i]], 1u);
}

__global__ void grid_gather_kernel(const float4* __restrict__ kd_pts, const uint32_t* __restrict__ order, uint32_t n, float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t j = order[i];
  const float4 p = kd_pts[j];
  out[i] = make_float4(p.x, p.y, p.z, __uint_as_float(j));
}

static void grid_geometry(sga_index* idx, double h) {
  idx->grid_h = static_cast<float>(h);
  double mag = 0.0;
  for (int k = 0; k < 3; k++) {
    idx->grid_org[k] = static_cast<float>(static_cast<double>(idx->bbox_lo[k]) - 1.5 * h);  // the box starts half a cell inside cell 1
    const double ext = static_cast<double>(idx->bbox_hi[k]) - static_cast<double>(idx->grid_org[k]);
    idx->grid_dim[k] = static_cast<int>(std::floor(ext / h)) + 3;  // the last occupied cell is <= n - 2
    mag = std::max({mag, std::fabs(static_cast<double>(idx->bbox_lo[k])), std::fabs(static_cast<double>(idx->bbox_hi[k])), ext + 2.0 * h});
  }
  idx->grid_eps = static_cast<float>(32.0 * 1.1920929e-7 * mag);  // 32 ulps of the largest coordinate involved
}
static double grid_cells(const sga_index* idx) { return static_cast<double>(idx->grid_dim[0]) * idx->grid_dim[1] * idx->grid_dim[2]; }

// Cell edge: the one at which an occupied cell holds `fill` points on average — found by counting the occupied cells at trial edges
// (sort the cell keys, count the distinct ones); on surface-like data the count scales with h^-2, three or four trials settle it.
// SGA_GRID_CELL (metres) overrides; SGA_GRID_FILL sets the target (default 2.2); the dense header array is capped at SGA_GRID_MAX_CELLS
// (default 96 M cells = 384 MB) by enlarging the edge.  Builds nothing for small targets (SGA_GRID_MIN_POINTS, default 65536): their
// passes are launch-bound, not search-bound.
int build_cell_grid(sga_context* ctx, sga_index* idx) {
  const int mode = g_grid_mode;
  const size_t min_points = static_cast<size_t>(std::max(16ll, g_grid_min_points));
  static const double cell_override = getenv("SGA_GRID_CELL") ? atof(getenv("SGA_GRID_CELL")) : 0.0;
  static const double fill = getenv("SGA_GRID_FILL") ? atof(getenv("SGA_GRID_FILL")) : 2.2;
  // the dense header array: at most 48 cells per target point (C3: 25 per point at its natural edge of 0.162 m) and never more than 96 M
  // cells = 384 MB — a sparse cloud in a huge box gets larger cells instead of a header a hundred times its size
  static const double max_cells_env = getenv("SGA_GRID_MAX_CELLS") ? atof(getenv("SGA_GRID_MAX_CELLS")) : 0.0;
  const double max_cells = max_cells_env > 0.0 ? max_cells_env : std::min(96e6, std::max(1.0e6, 48.0 * static_cast<double>(idx->n)));
  idx->grid_h = 0.f;
  const size_t n = idx->n;
  if (mode == 0 || n < min_points || n >= (1ull << 31) || ctx->stream_ordered) return SGA_OK;
  double ext[3], vol = 1.0;
  for (int k = 0; k < 3; k++) {
    ext[k] = std::max(1e-3, static_cast<double>(idx->bbox_hi[k]) - static_cast<double>(idx->bbox_lo[k]));
    vol *= ext[k];
  }
  const double h_floor = std::cbrt(vol / max_cells) * 1.05 + 1e-9;  // smaller edges would exceed the header budget
  const uint32_t un = static_cast<uint32_t>(n);
  const dim3 grid((un + 255) / 256), block(256);
  DevBuf<uint32_t> keys, keys2, vals, vals2, d_count, bitmap;
  SGA_TRY(keys.alloc(n));
  SGA_TRY(keys2.alloc(n));
  SGA_TRY(vals.alloc(n));
  SGA_TRY(vals2.alloc(n));
  SGA_TRY(d_count.alloc(1));
  size_t tb = 0;
  SGA_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys.p, keys2.p, vals.p, vals2.p, n, 0, 32, ctx->stream));
  SGA_TRY(ensure_temp(ctx, tb));
  double h = cell_override > 0.0 ? cell_override : std::max(h_floor, std::sqrt((ext[0] * ext[1] + ext[1] * ext[2] + ext[0] * ext[2]) * fill / static_cast<double>(n)));
  if (cell_override <= 0.0) {
    for (int it = 0; it < 5; it++) {
      grid_geometry(idx, h);
      if (grid_cells(idx) > 2.0e9) {  // keys are 31 bits
        h *= 1.5;
        continue;
      }
      const GridView g = make_grid_view(idx);
      SGA_HIP(hipMemsetAsync(d_count.p, 0, sizeof(uint32_t), ctx->stream));
      const size_t words = (static_cast<size_t>(grid_cells(idx)) + 31) / 32;
      if (words <= (64ull << 20)) {  // a bitmap of at most 256 MB (2^31 cells); C3: 25 M cells = 3 MB
        SGA_TRY(bitmap.reserve(words));
        SGA_HIP(hipMemsetAsync(bitmap.p, 0, words * sizeof(uint32_t), ctx->stream));
        hipLaunchKernelGGL(grid_mark_cells_kernel, dim3(std::min<uint32_t>(2048u, (un + 255u) / 256u)), block, 0, ctx->stream, idx->kd_pts.p, un, g, bitmap.p, d_count.p);
      } else {
        hipLaunchKernelGGL(grid_keys_kernel, grid, block, 0, ctx->stream, idx->kd_pts.p, un, g, keys.p, static_cast<uint32_t*>(nullptr));
        SGA_HIP(rocprim::radix_sort_keys(ctx->d_temp.p, tb, keys.p, keys2.p, n, 0, 32, ctx->stream));
        hipLaunchKernelGGL(grid_count_distinct_kernel, dim3(256), block, 0, ctx->stream, keys2.p, un, d_count.p);
      }
      uint32_t occupied = 0;
      SGA_HIP(hipMemcpyAsync(&occupied, d_count.p, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
      SGA_HIP(hipStreamSynchronize(ctx->stream));
      const double m = static_cast<double>(n) / std::max<uint32_t>(occupied, 1u);
      if (std::fabs(std::log(m / fill)) < 0.08) break;
      const double next = std::max(h_floor, h * std::sqrt(fill / m));
      if (next == h) break;
      h = next;
    }
  }
  h = std::max(h, h_floor);
  grid_geometry(idx, h);
  const double cells_d = grid_cells(idx);
  if (cells_d > 2.0e9 || cells_d > 4.0 * max_cells) {
    idx->grid_h = 0.f;
    return SGA_OK;  // an override that does not fit: no grid, the kd walk serves every pass
  }
  const size_t cells = static_cast<size_t>(cells_d);
  const GridView g0 = make_grid_view(idx);
  SGA_TRY(idx->grid_start.alloc(cells + 1));
  SGA_TRY(idx->grid_pts.alloc(n));
  hipLaunchKernelGGL(grid_keys_kernel, grid, block, 0, ctx->stream, idx->kd_pts.p, un, g0, keys.p, vals.p);
  SGA_TRY(sort_pairs(ctx, keys.p, keys2.p, vals.p, vals2.p, n, 0, 32));  // stable: kd order inside a cell
  hipLaunchKernelGGL(grid_gather_kernel, grid, block, 0, ctx->stream, idx->kd_pts.p, vals2.p, un, idx->grid_pts.p);
  SGA_HIP(hipMemsetAsync(idx->grid_start.p, 0, (cells + 1) * sizeof(uint32_t), ctx->stream));
  hipLaunchKernelGGL(grid_histogram_kernel, grid, block, 0, ctx->stream, keys2.p, un, idx->grid_start.p);
  size_t sb = 0;
  SGA_HIP(rocprim::exclusive_scan(nullptr, sb, idx->grid_start.p, idx->grid_start.p, 0u, cells + 1, rocprim::plus<uint32_t>(), ctx->stream));
  SGA_TRY(ensure_temp(ctx, sb));
  SGA_HIP(rocprim::exclusive_scan(ctx->d_temp.p, sb, idx->grid_start.p, idx->grid_start.p, 0u, cells + 1, rocprim::plus<uint32_t>(), ctx->stream));
  SGA_HIP(hipGetLastError());
  SGA_HIP(hipStreamSynchronize(ctx->stream));
  if (getenv("SGA_GRID_VERBOSE")) fprintf(stderr, "[sga] cell grid: h = %.4f m, %d x %d x %d = %.1f M cells, eps %.2e\n", h, idx->grid_dim[0], idx->grid_dim[1], idx->grid_dim[2], cells_d * 1e-6, idx->grid_eps);
  return SGA_OK;
}

// ---- search ----------------------------------------------------------------------------------------------------------------------
template <typename Real>
struct GridParams {
  const float4* __restrict__ src_pts;
  int n;
  GridView g;
  Rigid<Real> T;
  float bound2;  // the search reach, squared (the rejector's reach + kSearchMargin): a neighbour counts only if kd_dist2 < bound2
  int* __restrict__ nn;
  int* __restrict__ nn2;
  float* __restrict__ rex;  // >= 0: exclusion radius of a settled query; < 0: open after ring 1 (-(distance of the nearest point seen), -inf if none)
  int chunk_tiles;          // finish kernel: tiles of 64 queries per wave
  uint32_t* __restrict__ stats;
};

__device__ __forceinline__ int grid_tile_of_block() {  // XCD-aware tile order (linearize.hip: search_tile_of_block)
  const int nblk = gridDim.x, per_xcd = nblk >> 3, b = blockIdx.x;
  return b < 8 * per_xcd ? (b & 7) * per_xcd + (b >> 3) : b;
}

// Ring 1 for every query: 9 independent header loads, then the candidates of the 9 runs, four at a time, with the first four of the NEXT
// run already in flight while a run's candidates are compared (the scan is bound by the latency of its gathers, not by their bytes).
// One wave per tile of 64 queries.
constexpr int kRing1Waves = 4;  // waves per workgroup: a wave of this kernel lives ~6 us, and 15 625 single-wave workgroups per pass leave the
                                // SIMDs waiting for the dispatcher (measured: 0.9 waves in flight per SIMD); four tiles per workgroup launch
template <typename Real>
__global__ __launch_bounds__(64 * kRing1Waves) void grid_ring1_kernel(const GridParams<Real> p) {
  const int tile = grid_tile_of_block() * kRing1Waves + static_cast<int>(threadIdx.x >> 6);
  const int i = tile * 64 + static_cast<int>(threadIdx.x & 63);
  if (i >= p.n) return;
  const GridView& g = p.g;
  const float4 ps = p.src_pts[i];
  Real x, y, z;
  transform_point<Real>(p.T, ps.x, ps.y, ps.z, x, y, z);
  const float qx = static_cast<float>(x), qy = static_cast<float>(y), qz = static_cast<float>(z);
  int nn, nn2;
  float rex, seen;
  if (grid_ring1_lane(g, qx, qy, qz, p.bound2, nn, nn2, rex, seen)) {
    p.nn[i] = nn;
    p.nn2[i] = nn2;
    p.rex[i] = rex;
  } else {
    p.rex[i] = seen < 3.0e38f ? -fmaxf(seen, 1e-30f) : -INFINITY;
  }
}

// ---- the queries ring 1 left open ---------------------------------------------------------------------------------------------------
// Gathered from a chunk of tiles into full waves (ballot + popcount into an LDS list: a fixed order) and finished in two stages.
//   A, one query per lane: the ring the first result calls for — the block must reach as far as the nearest point seen — if that is ring
//      2 or 3; a query that has seen nothing tries ring 2.  Scanned from scratch (no point is offered twice), row headers eight at a time.
//   B, one query per WAVE: what is still open (isolated points: the nearest neighbour farther than three cells, or none within reach) gets
//      the ring that settles it for certain — up to the whole reach — with the (2r + 1)^2 rows spread over the 64 lanes: every lane looks up
//      and scans its own rows, the lanes' three nearest are merged with wave-wide minima (shuffles on the 64-bit keys).  Hundreds of mostly
//      empty rows cost a handful of independent loads per lane instead of a serial sweep by one lane while 63 wait.
// A block that does not settle its query (a face closer than the arithmetic slack allows) grows by one ring and is scanned again.
constexpr int kGridQueue = 1024;  // chunk_tiles <= 16
constexpr int kGridLaneRings = 3;  // stage A scans rings up to this one

template <typename Real>
__global__ __launch_bounds__(64) void grid_finish_kernel(const GridParams<Real> p) {
  __shared__ int q_idx[kGridQueue];
  const int lane = threadIdx.x;
  const GridView& g = p.g;
  const int num_tiles = (p.n + 63) >> 6;
  const int nchunks = (num_tiles + p.chunk_tiles - 1) / p.chunk_tiles;
  const int per_xcd = nchunks >> 3, b = blockIdx.x;
  const int chunk = b < 8 * per_xcd ? (b & 7) * per_xcd + (b >> 3) : b;
  int total = 0;
  for (int t = 0; t < p.chunk_tiles; t++) {
    const int tile = chunk * p.chunk_tiles + t;
    const int i = tile * 64 + lane;
    const bool open = tile < num_tiles && i < p.n && p.rex[i] < 0.f;
    const unsigned long long m = __ballot(open);
    if (open) q_idx[total + __popcll(m & ((1ull << lane) - 1ull))] = i;
    total += __popcll(m);
  }
  if (total == 0) return;
  __syncthreads();
  const float reach = sqrtf(p.bound2) * 1.00001f;
  unsigned ring_sum = 0;
  for (int k0 = 0; k0 < total; k0 += 64) {
    const bool active = k0 + lane < total;
    const int i = active ? q_idx[k0 + lane] : 0;
    float qx = 0.f, qy = 0.f, qz = 0.f, seen = INFINITY;
    int cx = 1, cy = 1, cz = 1, r = 2;
    if (active) {
      const float4 ps = p.src_pts[i];
      Real x, y, z;
      transform_point<Real>(p.T, ps.x, ps.y, ps.z, x, y, z);
      qx = static_cast<float>(x), qy = static_cast<float>(y), qz = static_cast<float>(z);
      cx = grid_cell(qx, g.ox, g.inv_h, g.nx), cy = grid_cell(qy, g.oy, g.inv_h, g.ny), cz = grid_cell(qz, g.oz, g.inv_h, g.nz);
      seen = -p.rex[i];  // distance of the nearest point ring 1 saw, +inf if none
      if (seen < 3.0e38f) r = grid_ring_for(g, fminf(seen * 1.00001f, reach), qx, qy, qz, cx, cy, cz);
    }
    bool done = !active;
    // ---- stage A: rings 2 .. kGridLaneRings, one query per lane
    for (;;) {
      const bool mine = !done && r <= kGridLaneRings;
      int R = mine ? r : 0;  // wave-uniform maximum
#pragma unroll
      for (int sft = 32; sft >= 1; sft >>= 1) R = max(R, __shfl_xor(R, sft, 64));
      if (R == 0) break;
      if (mine) ring_sum += static_cast<unsigned>(r);
      GridTop3 t = grid_top3();
      const int xlo = max(cx - r, 0), xhi = min(cx + r, g.nx - 1);
      for (int dz = -R; dz <= R; dz++) {
        const int zz = cz + dz;
        const bool zok = mine && dz >= -r && dz <= r && zz >= 0 && zz < g.nz;
        uint32_t s[8], e[8];  // 2 R + 1 <= 7 rows of this dz
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int dy = u - R, yy = cy + dy;
          const bool ok = zok && u <= 2 * R && dy >= -r && dy <= r && yy >= 0 && yy < g.ny;
          s[u] = e[u] = 0u;
          if (ok) {
            const int row = (zz * g.ny + yy) * g.nx;
            s[u] = g.start[row + xlo];
            e[u] = g.start[row + xhi + 1];
          }
        }
#pragma unroll
        for (int u = 0; u < 8; u++) grid_scan_run(g, s[u], e[u], qx, qy, qz, t);
      }
      if (mine) {
        int nn, nn2;
        float rex;
        if (grid_settle(t, grid_rho2(g, qx, qy, qz, cx, cy, cz, r), p.bound2, nn, nn2, rex)) {
          p.nn[i] = nn;
          p.nn2[i] = nn2;
          p.rex[i] = rex;
          done = true;
        } else {
          // something seen (now or before): the ring that reaches it settles the query; nothing seen: one ring farther
          const float d1 = sqrtf(grid_key_dist(t.k1));
          seen = fminf(seen, d1);
          r = seen < 3.0e38f ? max(r + 1, grid_ring_for(g, fminf(seen * 1.00001f, reach), qx, qy, qz, cx, cy, cz)) : r + 1;
        }
      }
    }
    // ---- stage B: what is still open, one query per wave
    unsigned long long open = __ballot(!done);
    while (open != 0ull) {
      const int l = __ffsll(static_cast<long long>(open)) - 1;
      open &= open - 1ull;
      const float ux = __shfl(qx, l, 64), uy = __shfl(qy, l, 64), uz = __shfl(qz, l, 64), useen = __shfl(seen, l, 64);
      const int ui = __shfl(i, l, 64);
      int nn, nn2;
      float rex;
      ring_sum += grid_settle_wave(g, lane, ux, uy, uz, useen, p.bound2, nn, nn2, rex);
      if (lane == 0) {
        p.nn[ui] = nn;
        p.nn2[ui] = nn2;
        p.rex[ui] = rex;
      }
    }
  }
  if (p.stats != nullptr) {
#pragma unroll
    for (int sft = 32; sft >= 1; sft >>= 1) ring_sum += __shfl_xor(ring_sum, sft, 64);
    if (lane == 0) {
      atomicAdd(&p.stats[0], static_cast<uint32_t>(total));
      atomicAdd(&p.stats[1], ring_sum);
    }
  }
}

// One grid pass: nn / nn2 / rex of every source point at pose T.  `reach2` = the squared search reach (finite).
template <typename Real>
int grid_search_pass(sga_context* ctx, const sga_index* idx, const float4* src_pts, int n, const Rigid<Real>& T, float reach2, int* nn, int* nn2, float* rex, uint32_t* stats) {
  static const int chunk_tiles = std::min(16, std::max(1, getenv("SGA_GRID_CHUNK") ? atoi(getenv("SGA_GRID_CHUNK")) : 4));
  GridParams<Real> p{};
  p.src_pts = src_pts;
  p.n = n;
  p.g = make_grid_view(idx);
  p.T = T;
  p.bound2 = reach2;
  p.nn = nn;
  p.nn2 = nn2;
  p.rex = rex;
  p.chunk_tiles = chunk_tiles;
  p.stats = stats;
  const int tiles = (n + 63) / 64;
  if (stats != nullptr) SGA_HIP(hipMemsetAsync(stats, 0, 2 * sizeof(uint32_t), ctx->stream));
  hipLaunchKernelGGL((grid_ring1_kernel<Real>), dim3((tiles + kRing1Waves - 1) / kRing1Waves), dim3(64 * kRing1Waves), 0, ctx->stream, p);
  hipLaunchKernelGGL((grid_finish_kernel<Real>), dim3((tiles + chunk_tiles - 1) / chunk_tiles), dim3(64), 0, ctx->stream, p);
  SGA_HIP(hipGetLastError());
  return SGA_OK;
}
template int grid_search_pass<float>(sga_context*, const sga_index*, const float4*, int, const Rigid<float>&, float, int*, int*, float*, uint32_t*);
template int grid_search_pass<double>(sga_context*, const sga_index*, const float4*, int, const Rigid<double>&, float, int*, int*, float*, uint32_t*);

// rings needed to cover `reach` metres, or -1 if the index has no grid
int grid_rings_for(const sga_index* idx, double reach) {
  if (idx->grid_h <= 0.f) return -1;
  const double rings = std::ceil(reach / idx->grid_h) + 1.0;
  // grid_settle_wave numbers the (2r + 1)^2 rows of a ring block through one integer decoded with a float reciprocal (exact below 2^20):
  // a reach that needs more rows than that is not a grid's job — the caller walks the kd-tree
  if ((2.0 * rings + 1.0) * (2.0 * rings + 1.0) > 1048576.0) return -1;
  return static_cast<int>(rings);
}

}  // namespace sga
