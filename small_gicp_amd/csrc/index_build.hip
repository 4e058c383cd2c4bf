// Search-index construction on the GPU: the implicit balanced kd-tree that replaces KdTreeBuilder::build_tree
// (ann/kdtree.hpp:80-126, reference tree /root/reference) and the one-shot GaussianVoxelMap that replaces
// IncrementalVoxelMap::insert + GaussianVoxel::add/finalize (ann/incremental_voxelmap.hpp:55-92,
// ann/gaussian_voxelmap.hpp:32-53).  Build time is outside the per-iteration hot loop; sorts use rocPRIM.
#include <algorithm>
#include <cmath>

#include "common.hpp"
#include "notes.hpp"
#include "sort_util.hpp"

#include <memory>
#include <rocprim/rocprim.hpp>
#include "kd_search.hpp"
#include "voxel_hash.hpp"

namespace sga {

int ensure_temp(sga_context* ctx, size_t bytes);
int build_cell_grid(sga_context* ctx, sga_index* idx);  // cell_grid.hip

// ---- bounding box --------------------------------------------------------------------------------------------------------
// <= 256 workgroups stream the cloud; wave shuffles + one LDS stage reduce a workgroup to six values, so only six atomics per
// workgroup reach memory (order-preserving int encoding of the floats); the last workgroup to arrive hands the box to the host as a
// note (notes.hpp: box_reduce_publish): no copy commands, no stream synchronisation.
__global__ __launch_bounds__(256) void bbox_note_kernel(const float4* __restrict__ pts, size_t n, int* __restrict__ d_box, unsigned long long* __restrict__ note_slot, unsigned long long seq) {
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x) {
    float4 p = pts[i];
    // a NaN would slip through fminf / fmaxf unseen: every non-finite coordinate counts as +inf, so the box reports it
    p.x = fabsf(p.x) <= 3.4028234e38f ? p.x : INFINITY;
    p.y = fabsf(p.y) <= 3.4028234e38f ? p.y : INFINITY;
    p.z = fabsf(p.z) <= 3.4028234e38f ? p.z : INFINITY;
    lo[0] = fminf(lo[0], p.x);
    lo[1] = fminf(lo[1], p.y);
    lo[2] = fminf(lo[2], p.z);
    hi[0] = fmaxf(hi[0], p.x);
    hi[1] = fmaxf(hi[1], p.y);
    hi[2] = fmaxf(hi[2], p.z);
  }
  box_reduce_publish(lo, hi, d_box, note_slot, seq);
}

// bounding box of a device cloud: enqueue (the box lands in the context's note block when the stream gets there) ...
int cloud_bbox_enqueue(sga_context* ctx, const float4* pts, size_t n, unsigned long long* seq_out) {
  unsigned long long* slot = nullptr;
  *seq_out = note_begin(ctx, &slot);
  hipLaunchKernelGGL(bbox_note_kernel, dim3(std::max<size_t>(1, std::min<size_t>(256, (n + 255) / 256))), dim3(256), 0, ctx->stream, pts, n, ctx->d_box.p, slot, *seq_out);
  SGA_HIP(hipGetLastError());
  return SGA_OK;
}
// ... and wait for it (a spin on mapped memory: the work enqueued behind the box kernel keeps running)
int cloud_bbox_collect(sga_context* ctx, unsigned long long seq, size_t n, float lo[3], float hi[3]) {
  unsigned long long payload[kNoteWords - 1];
  SGA_TRY(note_wait(ctx, seq, payload));
  box_note_decode(payload, lo, hi);
  if (n == 0)
    for (int k = 0; k < 3; k++) lo[k] = hi[k] = 0.f;
  return SGA_OK;
}

// both steps at once
int cloud_bbox(sga_context* ctx, const float4* pts, size_t n, float lo[3], float hi[3]) {
  unsigned long long seq = 0;
  SGA_TRY(cloud_bbox_enqueue(ctx, pts, n, &seq));
  return cloud_bbox_collect(ctx, seq, n, lo, hi);
}

// ---- implicit balanced kd-tree (see kd_search.hpp for the layout) ----------------------------------------------------------------
// Built top-down, one level per pass: per-segment bounding box -> split axis = longest extent -> sort by (segment, coordinate)
// -> threshold = coordinate of the first point of the right half.  (The reference picks the axis of largest sampled
// variance, projection.hpp:31-50; any axis gives an exact search.)
__device__ __forceinline__ uint32_t kd_bound_d(uint32_t n, int d, uint32_t k) { return kd_bound(n, d, k); }

__device__ __forceinline__ uint32_t kd_segment_of(uint32_t i, uint32_t n, int d) {
  uint32_t k = static_cast<uint32_t>((static_cast<unsigned long long>(i) << d) / n);
  while (kd_bound_d(n, d, k + 1) <= i) k++;
  while (kd_bound_d(n, d, k) > i) k--;
  return k;
}

__device__ __forceinline__ int ordered_from_float(float f) {
  const int i = __float_as_int(f);
  return i >= 0 ? i : i ^ 0x7fffffff;
}

// seg_box: 6 ints per segment (min xyz, max xyz) in the order-preserving int encoding.  Only the top levels of the build come here
// (segments of more than kFinishCap points), so a workgroup of 1024 consecutive points nearly always lies inside one segment: it
// reduces in registers + LDS and issues six atomics; mixed workgroups fall back to per-wave / per-lane atomics.
__global__ __launch_bounds__(1024) void kd_segment_box_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ perm, uint32_t n, int d, int* __restrict__ seg_box) {
  __shared__ float sh_lo[16][3], sh_hi[16][3];
  __shared__ uint32_t sh_seg[16];
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const bool valid = i < n;
  const uint32_t seg = kd_segment_of(valid ? i : n - 1, n, d);  // tail lanes join the last segment with neutral values
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  if (valid) {
    const float4 p = pts[perm[i]];
    lo[0] = hi[0] = p.x;
    lo[1] = hi[1] = p.y;
    lo[2] = hi[2] = p.z;
  }
  const uint32_t seg0 = __shfl(seg, 0);
  const bool wave_uniform = __all(seg == seg0);
  if (wave_uniform) {
    for (int k = 0; k < 3; k++)
      for (int off = 32; off > 0; off >>= 1) {
        lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
        hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
      }
  }
  if (lane == 0) {
    sh_seg[wave] = wave_uniform ? seg0 : 0xffffffffu;
    for (int k = 0; k < 3; k++) {
      sh_lo[wave][k] = lo[k];
      sh_hi[wave][k] = hi[k];
    }
  }
  __syncthreads();
  bool block_uniform = true;
  for (int w = 0; w < nwaves; w++) block_uniform = block_uniform && sh_seg[w] == sh_seg[0] && sh_seg[0] != 0xffffffffu;
  if (block_uniform) {
    if (threadIdx.x < 3) {
      const int k = threadIdx.x;
      float l = INFINITY, h = -INFINITY;
      for (int w = 0; w < nwaves; w++) {
        l = fminf(l, sh_lo[w][k]);
        h = fmaxf(h, sh_hi[w][k]);
      }
      if (l <= h) {
        atomicMin(&seg_box[6 * sh_seg[0] + k], ordered_from_float(l));
        atomicMax(&seg_box[6 * sh_seg[0] + 3 + k], ordered_from_float(h));
      }
    }
  } else if (wave_uniform) {
    if (lane == 0 && lo[0] <= hi[0]) {
      for (int k = 0; k < 3; k++) {
        atomicMin(&seg_box[6 * seg0 + k], ordered_from_float(lo[k]));
        atomicMax(&seg_box[6 * seg0 + 3 + k], ordered_from_float(hi[k]));
      }
    }
  } else if (valid) {
    for (int k = 0; k < 3; k++) {
      atomicMin(&seg_box[6 * seg + k], ordered_from_float(lo[k]));
      atomicMax(&seg_box[6 * seg + 3 + k], ordered_from_float(hi[k]));
    }
  }
}

__global__ void kd_init_box_kernel(int* __restrict__ seg_box, uint32_t nseg) {
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nseg) return;
  for (int k = 0; k < 3; k++) {
    seg_box[6 * s + k] = 0x7f800000;                                // +inf
    seg_box[6 * s + 3 + k] = static_cast<int>(0xff800000u) ^ 0x7fffffff;  // -inf
  }
}

__device__ __forceinline__ float float_from_ordered(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// keys for the sort of one level: (segment, coordinate along the segment's split axis)
__device__ __forceinline__ int kd_longest_axis(const int* __restrict__ box6) {
  float v[3];
  for (int a = 0; a < 3; a++) v[a] = float_from_ordered(box6[3 + a]) - float_from_ordered(box6[a]);
  return v[0] >= v[1] ? (v[0] >= v[2] ? 0 : 2) : (v[1] >= v[2] ? 1 : 2);
}

// split axis of every segment = longest extent of its box (each lane derives it from the six box words; the first lane of a
// segment records it for kd_nodes_kernel) + the sort key of every point
__global__ void kd_keys_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ perm, uint32_t n, int d, const int* __restrict__ seg_box, int* __restrict__ axis_of_seg, unsigned long long* __restrict__ keys) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t seg = kd_segment_of(i, n, d);
  const int axis = kd_longest_axis(seg_box + 6 * seg);
  if (i == kd_bound_d(n, d, seg)) axis_of_seg[seg] = axis;
  const float4 p = pts[perm[i]];
  const float c = axis == 0 ? p.x : (axis == 1 ? p.y : p.z);
  const uint32_t oc = static_cast<uint32_t>(ordered_from_float(c)) ^ 0x80000000u;  // unsigned order
  keys[i] = (static_cast<unsigned long long>(seg) << 32) | oc;
}

// thresholds of level d; also resets the boxes of the 2^(d+1) segments of the next level (nobody reads this level's boxes any more)
__global__ void kd_nodes_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ perm, uint32_t n, int d, const int* __restrict__ axis_of_seg, float2* __restrict__ nodes, int* __restrict__ next_box) {
  const uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
  if (seg >= (1u << d)) return;
  const int axis = axis_of_seg[seg];
  const uint32_t first = kd_bound_d(n, d, seg), end = kd_bound_d(n, d, seg + 1);
  const uint32_t m = kd_bound_d(n, d + 1, 2 * seg + 1);  // first point of the right child
  float thr = 0.f;
  if (first < end) {
    const float4 p = pts[perm[min(m, end - 1)]];
    thr = axis == 0 ? p.x : (axis == 1 ? p.y : p.z);
  }
  nodes[(1u << d) + seg] = make_float2(thr, __int_as_float(axis));
  if (next_box != nullptr)
    for (uint32_t c = 2 * seg; c < 2 * seg + 2; c++)
      for (int k = 0; k < 3; k++) {
        next_box[6 * c + k] = 0x7f800000;                                       // +inf
        next_box[6 * c + 3 + k] = static_cast<int>(0xff800000u) ^ 0x7fffffff;  // -inf
      }
}

// ---- bottom levels of the build inside LDS ----------------------------------------------------------------------------------------
// Once a segment holds at most kFinishCap points, ONE workgroup finishes its whole sub-tree: the points' coordinates are loaded
// into LDS once, and every remaining level is (per sub-segment box -> longest extent -> sort by (sub-segment, coordinate) ->
// threshold) without touching global memory or launching anything.  The sort is a rank sort over 64-bit keys
// (sub-segment | ordered coordinate | current position); the position field makes it reproduce the STABLE order of the
// radix sorts of the top levels, so the tree is the same whichever path builds a level.
constexpr int kFinishCap = 2048;      // points per workgroup, large clouds (small ones: 1024, to spread over more CUs)
constexpr int kFinishThreads = 512;
constexpr int kFinishMaxSub = 256;    // sub-segments at the last level: kFinishCap / 8

__device__ __forceinline__ uint32_t ordered_u32(float c) { return static_cast<uint32_t>(ordered_from_float(c)) ^ 0x80000000u; }

// THREADS: workgroup size (a sub-tree of <= 256 points keeps 256 threads busy, not 512: twice the workgroups per CU)
template <int CAP, int THREADS = kFinishThreads>
__global__ __launch_bounds__(THREADS) void kd_finish_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ perm_in, uint32_t* __restrict__ perm_out, uint32_t n, int dA, int D, float2* __restrict__ nodes) {
  __shared__ float cx[CAP], cy[CAP], cz[CAP];
  __shared__ uint32_t gidx[CAP];
  __shared__ unsigned long long key[CAP];
  __shared__ unsigned short ord[CAP], ord2[CAP];
  __shared__ int box[kFinishMaxSub][6];
  __shared__ int axis_of[kFinishMaxSub];
  const uint32_t seg = blockIdx.x;
  const uint32_t B0 = kd_bound(n, dA, seg), B1 = kd_bound(n, dA, seg + 1);
  const uint32_t m = B1 - B0;
  const int tid = threadIdx.x;
  for (uint32_t i = tid; i < m; i += THREADS) {
    const uint32_t g = perm_in[B0 + i];
    const float4 p = pts[g];
    cx[i] = p.x;
    cy[i] = p.y;
    cz[i] = p.z;
    gidx[i] = g;
    ord[i] = static_cast<unsigned short>(i);
  }
  __syncthreads();
  for (int d = dA; d < D; d++) {
    const uint32_t nsub = 1u << (d - dA), sub0 = seg << (d - dA);
    for (uint32_t j = tid; j < nsub * 6; j += THREADS) box[j / 6][j % 6] = (j % 6) < 3 ? 0x7f800000 : (static_cast<int>(0xff800000u) ^ 0x7fffffff);
    __syncthreads();
    for (uint32_t base = 0; base < m; base += THREADS) {  // whole waves take part in the shuffles below
      const uint32_t pos = base + tid;
      const bool valid = pos < m;
      const uint32_t e = ord[valid ? pos : m - 1], j = kd_segment_of(B0 + (valid ? pos : m - 1), n, d) - sub0;
      float lo[3] = {cx[e], cy[e], cz[e]}, hi[3] = {cx[e], cy[e], cz[e]};
      const uint32_t j0 = __shfl(j, 0);
      if (__all(j == j0)) {  // the usual case on the upper levels: one LDS atomic per wave and value instead of 64 on one address
        for (int a = 0; a < 3; a++)
          for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
          }
        if ((tid & 63) == 0)
          for (int a = 0; a < 3; a++) {
            atomicMin(&box[j][a], ordered_from_float(lo[a]));
            atomicMax(&box[j][3 + a], ordered_from_float(hi[a]));
          }
      } else if (valid) {
        for (int a = 0; a < 3; a++) {
          atomicMin(&box[j][a], ordered_from_float(lo[a]));
          atomicMax(&box[j][3 + a], ordered_from_float(hi[a]));
        }
      }
    }
    __syncthreads();
    for (uint32_t j = tid; j < nsub; j += THREADS) {
      float v[3];
      for (int a = 0; a < 3; a++) v[a] = float_from_ordered(box[j][3 + a]) - float_from_ordered(box[j][a]);
      axis_of[j] = v[0] >= v[1] ? (v[0] >= v[2] ? 0 : 2) : (v[1] >= v[2] ? 1 : 2);  // same rule as kd_longest_axis
    }
    __syncthreads();
    for (uint32_t pos = tid; pos < m; pos += THREADS) {
      const uint32_t e = ord[pos], j = kd_segment_of(B0 + pos, n, d) - sub0;
      const int a = axis_of[j];
      const float c = a == 0 ? cx[e] : (a == 1 ? cy[e] : cz[e]);
      key[pos] = (static_cast<unsigned long long>(j) << 43) | (static_cast<unsigned long long>(ordered_u32(c)) << 11) | pos;
    }
    __syncthreads();
    // rank sort inside every sub-segment: the new position of an element is the sub-segment's first position + the number of
    // its keys that are smaller (keys are distinct: they end with the current position).  All lanes of a wave read the same
    // key[i] (LDS broadcast), and the total work halves with every level — far cheaper than a sorting network here.
    for (uint32_t pos = tid; pos < m; pos += THREADS) {
      const unsigned long long mine = key[pos];
      const uint32_t gs = sub0 + static_cast<uint32_t>(mine >> 43);
      const uint32_t first = kd_bound(n, d, gs) - B0, end = kd_bound(n, d, gs + 1) - B0;
      uint32_t smaller = 0;
      uint32_t i = first;
      for (; i + 8 <= end; i += 8) {  // 8 LDS reads in flight
        unsigned long long k8[8];
#pragma unroll
        for (int u = 0; u < 8; u++) k8[u] = key[i + u];
#pragma unroll
        for (int u = 0; u < 8; u++) smaller += k8[u] < mine ? 1u : 0u;
      }
      for (; i < end; i++) smaller += key[i] < mine ? 1u : 0u;
      ord2[first + smaller] = ord[pos];
    }
    __syncthreads();
    for (uint32_t pos = tid; pos < m; pos += THREADS) ord[pos] = ord2[pos];
    __syncthreads();
    for (uint32_t j = tid; j < nsub; j += THREADS) {
      const uint32_t gs = sub0 + j;
      const uint32_t first = kd_bound(n, d, gs), end = kd_bound(n, d, gs + 1), mid = kd_bound(n, d + 1, 2 * gs + 1);
      const int a = axis_of[j];
      float thr = 0.f;
      if (first < end) {
        const uint32_t e = ord[min(mid, end - 1) - B0];
        thr = a == 0 ? cx[e] : (a == 1 ? cy[e] : cz[e]);
      }
      nodes[(1u << d) + gs] = make_float2(thr, __int_as_float(a));
    }
    __syncthreads();
  }
  for (uint32_t pos = tid; pos < m; pos += THREADS) perm_out[B0 + pos] = gidx[ord[pos]];
}

// ---- top levels of SMALL clouds: one launch per level, one workgroup per segment --------------------------------------------------
// A 15k-point scan (the odometry workload) is launch-bound: a level of the sort-based build above is a box kernel, a key kernel, a
// rocPRIM sort (3 - 6 launches at this size) and a node kernel, ~45 us for 15k points that a single workgroup can hold in registers.
// Here one workgroup (1024 threads, or 256 for segments of at most 8192 points: cheaper barriers) owns one segment (<= kSplitKeys
// points per thread) and does the whole level:
// bounding box -> longest axis -> the median by a three-round radix SELECT over the order-preserving keys (LDS histograms of 11 / 11
// / 10 bits) -> a PARTITION around it (elements below the median, then just enough of the equal ones; positions from a block-wide
// scan of per-thread counts, a fixed order) -> threshold.  A kd-tree needs the halves, not a sorted order inside them.  Deterministic.
constexpr uint32_t kSplitMaxPoints = 1024 * 32;                  // 32768: the largest cloud this path builds (1024 threads x 32 keys)
constexpr int kSplitFinish = 256;                                  // segments of at most this many points go to kd_finish_kernel<256>
constexpr int kSplitBins = 2048;

// exclusive prefix sum of one value per thread over the workgroup: wave scan, the wave totals through LDS (`sh_wave`: a slot of its
// own per call site, so no barrier protects its reuse), one barrier
template <int THREADS>
__device__ __forceinline__ uint32_t split_scan_exclusive(uint32_t v, uint32_t* __restrict__ sh_wave) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t t = __shfl_up(inc, off);
    if (lane >= off) inc += t;
  }
  if (lane == 63) sh_wave[wave] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < THREADS / 64; w++) base += w < wave ? sh_wave[w] : 0u;
  return base + inc - v;
}

template <int THREADS, int kSplitKeys>  // kSplitKeys: keys per thread; the launcher picks the smallest THREADS x kSplitKeys that holds a segment
__global__ __launch_bounds__(THREADS) void kd_split_level_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ perm_in /* null: the identity (level 0) */, uint32_t* __restrict__ perm_out, uint32_t n, int d, float2* __restrict__ nodes,
                                                                 unsigned long long* __restrict__ note_slot /* level 0: the cloud's bounding box goes to the host as a note (notes.hpp), or null */, unsigned long long note_seq) {
  constexpr int kWaves = THREADS / 64, kBinsPerThread = kSplitBins / THREADS;
  __shared__ uint32_t hist[3][kSplitBins];
  __shared__ uint32_t sh_wave[4][kWaves];
  __shared__ float sh_lo[kWaves][3], sh_hi[kWaves][3];
  __shared__ uint32_t sh_sel[3][3];  // per round: bucket, keys below it, keys in it
  const uint32_t seg = blockIdx.x, tid = threadIdx.x;
  const uint32_t first = kd_bound(n, d, seg), end = kd_bound(n, d, seg + 1), mid = kd_bound(n, d + 1, 2 * seg + 1);
  const uint32_t len = end - first, m = mid - first;  // the left half gets m elements
  if (len == 0) {  // cannot happen for the clouds this path is used for (segments of more than kSplitFinish points); workgroup-uniform
    if (tid == 0) nodes[(1u << d) + seg] = make_float2(0.f, 0.f);
    return;
  }
  for (int r = 0; r < 3; r++)
    for (int b = 0; b < kBinsPerThread; b++) hist[r][b * THREADS + tid] = 0u;
  // ---- the segment's points: element j of thread t sits at position first + j * THREADS + t.  The loads are UNCONDITIONAL (rows past
  // the end re-read the last element and are ignored) so that all of a stage's loads are in flight together: a row behind a branch
  // costs two dependent memory latencies, and there are up to 32 rows.
  uint32_t src[kSplitKeys];
#pragma unroll
  for (int j = 0; j < kSplitKeys; j++) {
    const uint32_t at = first + min(static_cast<uint32_t>(j * THREADS) + tid, len - 1u);
    src[j] = perm_in != nullptr ? perm_in[at] : at;
  }
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  constexpr int kChunk = kSplitKeys < 8 ? kSplitKeys : 8;
#pragma unroll
  for (int j0 = 0; j0 < kSplitKeys; j0 += kChunk) {
    float4 p[kChunk];
#pragma unroll
    for (int u = 0; u < kChunk; u++) p[u] = pts[src[j0 + u]];
#pragma unroll
    for (int u = 0; u < kChunk; u++) {  // (a repeated last element changes nothing in a box)
      if (note_slot != nullptr) {  // the box the host sees must report non-finite coordinates (fminf / fmaxf let a NaN slip through unseen): they count as +inf
        p[u].x = fabsf(p[u].x) <= 3.4028234e38f ? p[u].x : INFINITY;
        p[u].y = fabsf(p[u].y) <= 3.4028234e38f ? p[u].y : INFINITY;
        p[u].z = fabsf(p[u].z) <= 3.4028234e38f ? p[u].z : INFINITY;
      }
      lo[0] = fminf(lo[0], p[u].x), lo[1] = fminf(lo[1], p[u].y), lo[2] = fminf(lo[2], p[u].z);
      hi[0] = fmaxf(hi[0], p[u].x), hi[1] = fmaxf(hi[1], p[u].y), hi[2] = fmaxf(hi[2], p[u].z);
    }
  }
  for (int a = 0; a < 3; a++)
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
    }
  if ((tid & 63) == 0)
    for (int a = 0; a < 3; a++) {
      sh_lo[tid >> 6][a] = lo[a];
      sh_hi[tid >> 6][a] = hi[a];
    }
  __syncthreads();
  float ext[3];
  for (int a = 0; a < 3; a++) {  // every thread derives the axis itself (broadcast reads): no second barrier
    float l = INFINITY, h = -INFINITY;
    for (int w = 0; w < kWaves; w++) {
      l = fminf(l, sh_lo[w][a]);
      h = fmaxf(h, sh_hi[w][a]);
    }
    ext[a] = h - l;
    if (note_slot != nullptr && tid == 0) note_slot[1 + a] = static_cast<unsigned long long>(static_cast<unsigned>(box_enc(l))) | (static_cast<unsigned long long>(static_cast<unsigned>(box_enc(h))) << 32);
  }
  if (note_slot != nullptr && tid == 0) note_publish(note_slot, note_seq);  // (one workgroup at level 0)
  const int axis = ext[0] >= ext[1] ? (ext[0] >= ext[2] ? 0 : 2) : (ext[1] >= ext[2] ? 1 : 2);  // same rule as kd_longest_axis
  uint32_t key[kSplitKeys];  // the coordinate along the split axis (read again, one word: the lines are in the cache), order-preserving encoding
  const float* __restrict__ coord = reinterpret_cast<const float*>(pts) + axis;
#pragma unroll
  for (int j = 0; j < kSplitKeys; j++) key[j] = ordered_u32(coord[4ull * src[j]]);
  // ---- radix select: the key of rank m (0-based) among the segment's keys, and how many keys are smaller
  uint32_t prefix = 0u, prefix_mask = 0u, rank = min(m, len - 1u), below = 0u;
  const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
#pragma unroll
  for (int round = 0; round < 3; round++) {
    const uint32_t bmask = (1u << bits[round]) - 1u;
#pragma unroll
    for (int j = 0; j < kSplitKeys; j++) {
      const uint32_t pos = j * THREADS + tid;
      if (pos < len && (key[j] & prefix_mask) == prefix) atomicAdd(&hist[round][(key[j] >> shifts[round]) & bmask], 1u);
    }
    __syncthreads();
    uint32_t c[kBinsPerThread], mine = 0u;  // this thread's bins: kBinsPerThread consecutive ones
#pragma unroll
    for (int b = 0; b < kBinsPerThread; b++) {
      c[b] = hist[round][kBinsPerThread * tid + b];
      mine += c[b];
    }
    uint32_t before = split_scan_exclusive<THREADS>(mine, sh_wave[round]);
    if (rank >= before && rank < before + mine) {  // exactly one thread
#pragma unroll
      for (int b = 0; b < kBinsPerThread; b++) {
        if (rank >= before && rank < before + c[b]) {
          sh_sel[round][0] = kBinsPerThread * tid + b;
          sh_sel[round][1] = before;
          sh_sel[round][2] = c[b];
        }
        before += c[b];
      }
    }
    __syncthreads();
    const uint32_t bucket = sh_sel[round][0], under = sh_sel[round][1];
    prefix |= bucket << shifts[round];
    prefix_mask |= bmask << shifts[round];
    below += under;
    rank -= under;
  }
  const uint32_t median = prefix;   // `below` keys are smaller; the first `rank` of the `eq_total` equal ones complete the left half
  const uint32_t eq_total = sh_sel[2][2];
  // ---- partition: [keys < median][`rank` of the equal keys] | [the other equal keys][keys > median].  Order inside the four parts:
  // thread-major (thread t's elements, rows ascending, behind those of threads < t) — fixed, hence deterministic; ONE block-wide scan
  // of the per-thread counts instead of one per row.
  uint32_t my_less = 0u, my_eq = 0u;
#pragma unroll
  for (int j = 0; j < kSplitKeys; j++) {
    const bool valid = j * THREADS + tid < len;
    my_less += (valid && key[j] < median) ? 1u : 0u;
    my_eq += (valid && key[j] == median) ? 1u : 0u;
  }
  const uint32_t packed = split_scan_exclusive<THREADS>(my_less | (my_eq << 16), sh_wave[3]);  // <= 32 per thread, <= 32768 in all: 16 bits each
  uint32_t n_less = packed & 0xffffu, n_eq = packed >> 16;  // elements of the two kinds owned by lower threads
  uint32_t n_all = tid * (len / THREADS) + min(tid, len % THREADS);  // all elements owned by lower threads (thread t owns rows j with j * THREADS + t < len)
#pragma unroll  // static indices: src[] / key[] stay in registers
  for (int j = 0; j < kSplitKeys; j++) {
    if (j * THREADS + tid < len) {
      const bool less = key[j] < median, eq = key[j] == median;
      const uint32_t n_greater = n_all - n_less - n_eq;
      const uint32_t dest = less ? n_less : (eq ? (n_eq < rank ? below + n_eq : m + (n_eq - rank)) : m + (eq_total - rank) + n_greater);
      perm_out[first + dest] = src[j];
      n_less += less ? 1u : 0u;
      n_eq += eq ? 1u : 0u;
      n_all += 1u;
    }
  }
  if (tid == 0) nodes[(1u << d) + seg] = make_float2(float_from_ordered(static_cast<int>(median ^ 0x80000000u)), __int_as_float(axis));
}

// ---- top levels of LARGE clouds: the same select + partition, a segment spread over many workgroups (round 6) -------------------------
// A segment of more than kSplitMaxPoints points does not fit one workgroup's registers.  The sort-based level above orders ALL n (segment,
// coordinate) keys to learn one median per segment: ~260 us per level at 1M points, 47 rocPRIM launches — most of a 1M-point build.  Here the
// level is what kd_split_level_kernel does, with the segment cut into chunks of kTopChunk (4096) points (grid = chunks x segments) and the
// histograms of the three select rounds accumulated in global memory (LDS first, the non-empty bins flushed with atomics):
//   box   -> the segment's box (6 atomics per workgroup)                                           kd_top_box_kernel
//   hist0 -> axis = longest extent; the keys (order-preserving coordinate) are stored once; top 11 bits   kd_top_hist_kernel<0>
//   hist1, hist2 -> every workgroup re-derives the buckets chosen so far from the finished histograms (2048 bins: one scan), then counts
//            the next 11 / 10 bits of the keys inside them                                          kd_top_hist_kernel<1>, <2>
//   count -> the median is known: keys below / equal per chunk; chunk 0 writes the node                kd_top_count_kernel
//   scatter -> [keys < median][`rank` of the equal ones] | [the other equal ones][keys > median], chunks in order, inside a chunk
//            thread-major: a fixed order.  The POINTS move with the permutation (ping-pong copies), so that every pass of the next
//            level streams instead of gathering through the permutation                               kd_top_scatter_kernel
// Six passes over 4 - 20 bytes per point instead of a 64-bit key-value sort of the whole cloud.  The tree is another valid one over the
// same points (like the split path's): the halves are the same SETS as the sort's whenever the median key is unique, the order inside
// them is not the sorted one.
#ifndef SGA_TOP_ITEMS
#define SGA_TOP_ITEMS 4  // items per thread: chunks of 4096 points (measured: 2 / 4 / 8 / 16 items -> 1M-point build 1.29 / 1.23 / 1.26 / 1.42 ms, 400k 0.80 / 0.80 / 0.85 / 0.98)
#endif
constexpr uint32_t kTopItems = SGA_TOP_ITEMS, kTopThreads = 1024, kTopChunk = kTopItems * kTopThreads;
struct TopSel {
  uint32_t median, below, rank, eq_total;  // the median key; keys below it; how many of the equal keys complete the left half; equal keys
};

__global__ __launch_bounds__(kTopThreads) void kd_top_box_kernel(const float4* __restrict__ pts, uint32_t n, int d, int* __restrict__ seg_box) {
  __shared__ float sh_lo[kTopThreads / 64][3], sh_hi[kTopThreads / 64][3];
  const uint32_t seg = blockIdx.y, tid = threadIdx.x;
  const uint32_t first = kd_bound(n, d, seg), end = kd_bound(n, d, seg + 1);
  const uint32_t i0 = first + blockIdx.x * kTopChunk;
  if (i0 >= end) return;  // workgroup-uniform
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  float4 p[kTopItems];
#pragma unroll
  for (uint32_t j = 0; j < kTopItems; j++) p[j] = pts[min(i0 + j * kTopThreads + tid, end - 1u)];  // (a repeated last element changes nothing in a box)
#pragma unroll
  for (uint32_t j = 0; j < kTopItems; j++) {
    lo[0] = fminf(lo[0], p[j].x), lo[1] = fminf(lo[1], p[j].y), lo[2] = fminf(lo[2], p[j].z);
    hi[0] = fmaxf(hi[0], p[j].x), hi[1] = fmaxf(hi[1], p[j].y), hi[2] = fmaxf(hi[2], p[j].z);
  }
  for (int a = 0; a < 3; a++)
    for (int off = 32; off > 0; off >>= 1) {
      lo[a] = fminf(lo[a], __shfl_xor(lo[a], off));
      hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off));
    }
  if ((tid & 63) == 0)
    for (int a = 0; a < 3; a++) sh_lo[tid >> 6][a] = lo[a], sh_hi[tid >> 6][a] = hi[a];
  __syncthreads();
  if (tid < 3) {
    float l = INFINITY, h = -INFINITY;
    for (uint32_t w = 0; w < kTopThreads / 64; w++) l = fminf(l, sh_lo[w][tid]), h = fmaxf(h, sh_hi[w][tid]);
    if (l <= h) {
      atomicMin(&seg_box[6 * seg + tid], ordered_from_float(l));
      atomicMax(&seg_box[6 * seg + 3 + tid], ordered_from_float(h));
    }
  }
}

// the bucket of `rank` in a finished histogram of kSplitBins bins: (bucket, keys in lower buckets, keys in it) -> sh_sel[0..2]; all threads call
__device__ __forceinline__ void top_select(const uint32_t* __restrict__ hist, uint32_t rank, uint32_t* __restrict__ sh_wave, uint32_t* __restrict__ sh_sel) {
  constexpr int kBinsPerThread = kSplitBins / kTopThreads;
  const uint32_t tid = threadIdx.x;
  uint32_t c[kBinsPerThread], mine = 0u;
#pragma unroll
  for (int b = 0; b < kBinsPerThread; b++) {
    c[b] = hist[kBinsPerThread * tid + b];
    mine += c[b];
  }
  uint32_t before = split_scan_exclusive<kTopThreads>(mine, sh_wave);
  if (rank >= before && rank < before + mine) {  // exactly one thread
#pragma unroll
    for (int b = 0; b < kBinsPerThread; b++) {
      if (rank >= before && rank < before + c[b]) {
        sh_sel[0] = kBinsPerThread * tid + b;
        sh_sel[1] = before;
        sh_sel[2] = c[b];
      }
      before += c[b];
    }
  }
  __syncthreads();
}

// hist: [segment][round][kSplitBins].  ROUND 0 also fixes the axis and stores the keys.
template <int ROUND>
__global__ __launch_bounds__(kTopThreads) void kd_top_hist_kernel(const float4* __restrict__ pts, uint32_t* __restrict__ keys, uint32_t n, int d, const int* __restrict__ seg_box, int* __restrict__ axis_of_seg, uint32_t* __restrict__ hist) {
  __shared__ uint32_t sh_hist[kSplitBins];
  __shared__ uint32_t sh_wave[2][kTopThreads / 64];
  __shared__ uint32_t sh_sel[2][3];
  const uint32_t seg = blockIdx.y, tid = threadIdx.x;
  const uint32_t first = kd_bound(n, d, seg), end = kd_bound(n, d, seg + 1), mid = kd_bound(n, d + 1, 2 * seg + 1);
  const uint32_t i0 = first + blockIdx.x * kTopChunk;
  if (i0 >= end) return;  // workgroup-uniform
  const uint32_t len = end - first;
  uint32_t* __restrict__ seg_hist = hist + static_cast<size_t>(seg) * 3 * kSplitBins;
  for (uint32_t b = tid; b < kSplitBins; b += kTopThreads) sh_hist[b] = 0u;
  uint32_t key[kTopItems];
  if constexpr (ROUND == 0) {
    const int axis = kd_longest_axis(seg_box + 6 * seg);
    if (blockIdx.x == 0 && tid == 0) axis_of_seg[seg] = axis;
    const float* __restrict__ coord = reinterpret_cast<const float*>(pts) + axis;
#pragma unroll
    for (uint32_t j = 0; j < kTopItems; j++) {
      const uint32_t i = i0 + j * kTopThreads + tid;
      key[j] = ordered_u32(coord[4ull * min(i, end - 1u)]);
      if (i < end) keys[i] = key[j];
    }
  } else {
#pragma unroll
    for (uint32_t j = 0; j < kTopItems; j++) key[j] = keys[min(i0 + j * kTopThreads + tid, end - 1u)];
  }
  uint32_t prefix = 0u, prefix_mask = 0u, rank = min(mid - first, len - 1u);
  const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
#pragma unroll
  for (int r = 0; r < ROUND; r++) {  // the buckets the finished rounds chose (every workgroup of the segment derives the same)
    top_select(seg_hist + r * kSplitBins, rank, sh_wave[r], sh_sel[r]);
    prefix |= sh_sel[r][0] << shifts[r];
    prefix_mask |= ((1u << bits[r]) - 1u) << shifts[r];
    rank -= sh_sel[r][1];
  }
  __syncthreads();  // (ROUND 0: the cleared histogram)
  const uint32_t bmask = (1u << bits[ROUND]) - 1u;
#pragma unroll
  for (uint32_t j = 0; j < kTopItems; j++)
    if (i0 + j * kTopThreads + tid < end && (key[j] & prefix_mask) == prefix) atomicAdd(&sh_hist[(key[j] >> shifts[ROUND]) & bmask], 1u);
  __syncthreads();
  for (uint32_t b = tid; b < kSplitBins; b += kTopThreads) {
    const uint32_t c = sh_hist[b];
    if (c != 0u) atomicAdd(&seg_hist[ROUND * kSplitBins + b], c);
  }
}

// cnt: [segment][chunk] x {keys below the median, keys equal to it}
__global__ __launch_bounds__(kTopThreads) void kd_top_count_kernel(const uint32_t* __restrict__ keys, uint32_t n, int d, const int* __restrict__ axis_of_seg, const uint32_t* __restrict__ hist, uint32_t chunks, uint2* __restrict__ cnt,
                                                                  TopSel* __restrict__ sel, float2* __restrict__ nodes) {
  __shared__ uint32_t sh_wave[4][kTopThreads / 64];
  __shared__ uint32_t sh_sel[3][3];
  const uint32_t seg = blockIdx.y, tid = threadIdx.x;
  const uint32_t first = kd_bound(n, d, seg), end = kd_bound(n, d, seg + 1), mid = kd_bound(n, d + 1, 2 * seg + 1);
  const uint32_t i0 = first + blockIdx.x * kTopChunk;
  if (i0 >= end) return;  // workgroup-uniform
  const uint32_t len = end - first;
  const uint32_t* __restrict__ seg_hist = hist + static_cast<size_t>(seg) * 3 * kSplitBins;
  uint32_t key[kTopItems];
#pragma unroll
  for (uint32_t j = 0; j < kTopItems; j++) key[j] = keys[min(i0 + j * kTopThreads + tid, end - 1u)];
  uint32_t median = 0u, rank = min(mid - first, len - 1u), below = 0u;
  const int shifts[3] = {21, 10, 0};
#pragma unroll
  for (int r = 0; r < 3; r++) {
    top_select(seg_hist + r * kSplitBins, rank, sh_wave[r], sh_sel[r]);
    median |= sh_sel[r][0] << shifts[r];
    below += sh_sel[r][1];
    rank -= sh_sel[r][1];
  }
  uint32_t my_less = 0u, my_eq = 0u;
#pragma unroll
  for (uint32_t j = 0; j < kTopItems; j++) {
    const bool valid = i0 + j * kTopThreads + tid < end;
    my_less += (valid && key[j] < median) ? 1u : 0u;
    my_eq += (valid && key[j] == median) ? 1u : 0u;
  }
  const uint32_t packed = split_scan_exclusive<kTopThreads>(my_less | (my_eq << 16), sh_wave[3]);  // <= 8 per thread, <= 8192 in all: 16 bits each
  if (tid == kTopThreads - 1) cnt[static_cast<size_t>(seg) * chunks + blockIdx.x] = make_uint2((packed & 0xffffu) + my_less, (packed >> 16) + my_eq);
  if (blockIdx.x == 0 && tid == 0) {
    sel[seg] = TopSel{median, below, rank, sh_sel[2][2]};
    nodes[(1u << d) + seg] = make_float2(float_from_ordered(static_cast<int>(median ^ 0x80000000u)), __int_as_float(axis_of_seg[seg]));
  }
}

__global__ __launch_bounds__(kTopThreads) void kd_top_scatter_kernel(const float4* __restrict__ pts_in, const uint32_t* __restrict__ perm_in /* null: the identity */, const uint32_t* __restrict__ keys, uint32_t n, int d, uint32_t chunks,
                                                                    const uint2* __restrict__ cnt, const TopSel* __restrict__ sel, float4* __restrict__ pts_out, uint32_t* __restrict__ perm_out,
                                                                    int* __restrict__ child_box /* the boxes of the next level's segments (6 ints each, this level's segment s -> 2 s, 2 s + 1), or null */) {
  __shared__ uint32_t sh_wave[3][kTopThreads / 64];
  __shared__ float sh_box[kTopThreads / 64][12];
  const uint32_t seg = blockIdx.y, tid = threadIdx.x, chunk = blockIdx.x;
  const uint32_t first = kd_bound(n, d, seg), end = kd_bound(n, d, seg + 1), mid = kd_bound(n, d + 1, 2 * seg + 1);
  const uint32_t i0 = first + chunk * kTopChunk;
  if (i0 >= end) return;  // workgroup-uniform
  const uint32_t m = mid - first;
  const TopSel s = sel[seg];
  // keys below / equal in the chunks before this one (<= 128 chunks: one value per thread, two block sums)
  uint2 mine = make_uint2(0u, 0u);
  for (uint32_t c = tid; c < chunk; c += kTopThreads) {
    const uint2 v = cnt[static_cast<size_t>(seg) * chunks + c];
    mine.x += v.x, mine.y += v.y;
  }
  uint32_t lt_before = mine.x, eq_before = mine.y;
  for (int off = 32; off > 0; off >>= 1) lt_before += __shfl_xor(lt_before, off), eq_before += __shfl_xor(eq_before, off);
  if ((tid & 63) == 0) sh_wave[0][tid >> 6] = lt_before, sh_wave[1][tid >> 6] = eq_before;
  __syncthreads();
  lt_before = eq_before = 0u;
  for (uint32_t w = 0; w < kTopThreads / 64; w++) lt_before += sh_wave[0][w], eq_before += sh_wave[1][w];
  uint32_t key[kTopItems];
  float4 p[kTopItems];
  uint32_t src[kTopItems];
#pragma unroll
  for (uint32_t j = 0; j < kTopItems; j++) {
    const uint32_t i = min(i0 + j * kTopThreads + tid, end - 1u);
    key[j] = keys[i];
    p[j] = pts_in[i];
    src[j] = perm_in != nullptr ? perm_in[i] : i;
  }
  uint32_t my_less = 0u, my_eq = 0u, my_all = 0u;
#pragma unroll
  for (uint32_t j = 0; j < kTopItems; j++) {
    const bool valid = i0 + j * kTopThreads + tid < end;
    my_less += (valid && key[j] < s.median) ? 1u : 0u;
    my_eq += (valid && key[j] == s.median) ? 1u : 0u;
    my_all += valid ? 1u : 0u;
  }
  const uint32_t packed = split_scan_exclusive<kTopThreads>(my_less | (my_eq << 16), sh_wave[2]);
  __syncthreads();  // sh_wave[0] is reused below
  const uint32_t all_before = split_scan_exclusive<kTopThreads>(my_all, sh_wave[0]);
  uint32_t n_less = lt_before + (packed & 0xffffu), n_eq = eq_before + (packed >> 16);
  uint32_t n_all = (i0 - first) + all_before;  // elements of the segment ahead of this thread's in the fixed order (chunks, then thread-major)
  float bx[12] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY, INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};  // left child's lo, hi; right child's
#pragma unroll
  for (uint32_t j = 0; j < kTopItems; j++) {
    if (i0 + j * kTopThreads + tid < end) {
      const bool less = key[j] < s.median, eq = key[j] == s.median;
      const uint32_t n_greater = n_all - n_less - n_eq;
      const uint32_t off = less ? n_less : (eq ? (n_eq < s.rank ? s.below + n_eq : m + (n_eq - s.rank)) : m + (s.eq_total - s.rank) + n_greater);
      pts_out[first + off] = p[j];
      perm_out[first + off] = src[j];
      n_less += less ? 1u : 0u;
      n_eq += eq ? 1u : 0u;
      n_all += 1u;
      // the box of the child the element goes to: the next level's box pass, taken on the way
      const bool right = off >= m;
      const float c[3] = {p[j].x, p[j].y, p[j].z};
#pragma unroll
      for (int a = 0; a < 3; a++) {
        bx[a] = fminf(bx[a], right ? INFINITY : c[a]);
        bx[3 + a] = fmaxf(bx[3 + a], right ? -INFINITY : c[a]);
        bx[6 + a] = fminf(bx[6 + a], right ? c[a] : INFINITY);
        bx[9 + a] = fmaxf(bx[9 + a], right ? c[a] : -INFINITY);
      }
    }
  }
  if (child_box != nullptr) {  // kernel-uniform
#pragma unroll
    for (int v = 0; v < 12; v++) {
      const bool is_lo = (v % 6) < 3;
      for (int off = 32; off > 0; off >>= 1) bx[v] = is_lo ? fminf(bx[v], __shfl_xor(bx[v], off)) : fmaxf(bx[v], __shfl_xor(bx[v], off));
    }
    if ((tid & 63) == 0)
      for (int v = 0; v < 12; v++) sh_box[tid >> 6][v] = bx[v];
    __syncthreads();
    if (tid < 12) {
      const bool is_lo = (tid % 6) < 3;
      float r = is_lo ? INFINITY : -INFINITY;
      for (uint32_t w = 0; w < kTopThreads / 64; w++) r = is_lo ? fminf(r, sh_box[w][tid]) : fmaxf(r, sh_box[w][tid]);
      int* dst = child_box + 6 * (2 * seg + tid / 6) + tid % 6;
      if (is_lo) {
        if (r < INFINITY) atomicMin(dst, ordered_from_float(r));
      } else {
        if (r > -INFINITY) atomicMax(dst, ordered_from_float(r));
      }
    }
  }
}

// Tight bounding boxes of all nodes, bottom-up (kd_search.hpp: a pending far side is opened only if its box can hold a closer point).
// One launch covers up to 8 levels: every workgroup takes 256 adjacent nodes of depth `base` — their boxes come from the points
// (base = D, leaves) or from the previous launch — and merges them pairwise in LDS up to depth base - 8.
__global__ __launch_bounds__(256) void kd_boxes_kernel(const float4* __restrict__ pts, uint32_t n, int D, int base, float4* __restrict__ boxes) {
  __shared__ float slo[3][256], shi[3][256];
  const uint32_t t = threadIdx.x, k = blockIdx.x * 256u + t;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  if (k < (1u << base)) {
    const uint32_t node = (1u << base) + k;
    if (base == D) {
      const uint32_t first = kd_bound(n, D, k), end = kd_bound(n, D, k + 1);
      for (uint32_t i = first; i < end; i++) {
        const float4 p = pts[i];
        lo[0] = fminf(lo[0], p.x);
        lo[1] = fminf(lo[1], p.y);
        lo[2] = fminf(lo[2], p.z);
        hi[0] = fmaxf(hi[0], p.x);
        hi[1] = fmaxf(hi[1], p.y);
        hi[2] = fmaxf(hi[2], p.z);
      }
      boxes[2 * node] = make_float4(lo[0], lo[1], lo[2], 0.f);
      boxes[2 * node + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
    } else {
      const float4 l = boxes[2 * node], h = boxes[2 * node + 1];
      lo[0] = l.x;
      lo[1] = l.y;
      lo[2] = l.z;
      hi[0] = h.x;
      hi[1] = h.y;
      hi[2] = h.z;
    }
  }
  for (int a = 0; a < 3; a++) {
    slo[a][t] = lo[a];
    shi[a][t] = hi[a];
  }
  __syncthreads();
  for (int l = 1; l <= 8 && l <= base; l++) {
    const uint32_t width = 256u >> l;  // nodes of depth base - l in this workgroup
    if (t < width) {
      for (int a = 0; a < 3; a++) {
        lo[a] = fminf(slo[a][2 * t], slo[a][2 * t + 1]);
        hi[a] = fmaxf(shi[a][2 * t], shi[a][2 * t + 1]);
      }
    }
    __syncthreads();
    if (t < width) {
      for (int a = 0; a < 3; a++) {
        slo[a][t] = lo[a];
        shi[a][t] = hi[a];
      }
      const uint32_t kk = (blockIdx.x * 256u >> l) + t;
      if (kk < (1u << (base - l))) {
        const uint32_t node = (1u << (base - l)) + kk;
        boxes[2 * node] = make_float4(lo[0], lo[1], lo[2], 0.f);
        boxes[2 * node + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
      }
    }
    __syncthreads();
  }
}

// The tail of a build in ONE launch (round 6: a 11.5k-point scan paid six launches for it — gather, leaf boxes, group headers, leaf blocks,
// pair records, upper boxes).  Thread g of the grid owns leaf g and heap node g:
//   * the leaf's points are gathered into kd order (with their attributes); 8 points at infinity follow the last leaf (leaf scans read 8
//     slots unconditionally);
//   * its LEAF BLOCK (kd_search.hpp: the fast leaf scan): the points as structure of arrays in one 128-byte line — x[8], y[8], z[8],
//     original index[8] — so that a lane loads the coordinates of two points into adjacent registers (packed fp32 arithmetic); slots a
//     leaf does not fill lie far away (kKdFar): their distance is huge but finite and never wins;
//   * its tight box; the boxes are merged pairwise in LDS up to 8 levels (kd_boxes_kernel's loop);
//   * GROUP HEADERS (kd_search.hpp: kd_visit_group): for every node of depth D - G (G = min(2, D)) the tight boxes of its 2^G leaves as six
//     float4 {lo.x, lo.y, lo.z, hi.x, hi.y, hi.z} with one lane per leaf, in one 128-byte line; missing leaves get an empty box;
//   * node g's PAIR RECORD for the 1-NN walk (kd_search.hpp): a node of even depth + its two children in one 16-byte record, stored at
//     the node's heap number.
// Trees deeper than 8 levels finish their upper boxes with kd_boxes_kernel(base = D - 8) as before.
__global__ __launch_bounds__(256) void kd_tail_kernel(const uint32_t* __restrict__ order, uint32_t n, int D, const float4* __restrict__ pts, const float4* __restrict__ nrm, const Cov8* __restrict__ cov, const float2* __restrict__ nodes,
                                                      float4* __restrict__ opts, float4* __restrict__ onrm, Cov8* __restrict__ ocov, float4* __restrict__ boxes, float4* __restrict__ groups, float* __restrict__ blocks,
                                                      float4* __restrict__ pairs, uint32_t npairs, unsigned long long* __restrict__ d_spacing, unsigned long long* __restrict__ late_slot, unsigned long long late_seq) {
  __shared__ float slo[3][256], shi[3][256];
  __shared__ long long sh_sum[4];
  __shared__ unsigned sh_cnt[4];
  __shared__ bool sh_last;
  const uint32_t t = threadIdx.x, k = blockIdx.x * 256u + t;
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  if (k < (1u << D)) {
    const uint32_t first = kd_bound(n, D, k), end = kd_bound(n, D, k + 1);
    float4 p[kKdLeafMax];
#pragma unroll
    for (int j = 0; j < kKdLeafMax; j++) {
      const uint32_t pos = first + j;
      const bool valid = pos < end;
      const uint32_t src = order[valid ? pos : first < n ? first : 0u];
      p[j] = pts[src];  // w keeps the original index bits
      if (valid) {
        opts[pos] = p[j];
        if (nrm) onrm[pos] = nrm[src];
        if (cov) ocov[pos] = cov[src];
        lo[0] = fminf(lo[0], p[j].x), lo[1] = fminf(lo[1], p[j].y), lo[2] = fminf(lo[2], p[j].z);
        hi[0] = fmaxf(hi[0], p[j].x), hi[1] = fmaxf(hi[1], p[j].y), hi[2] = fmaxf(hi[2], p[j].z);
      } else {
        p[j] = make_float4(kKdFar, kKdFar, kKdFar, 0.f);
      }
    }
    float* b = blocks + 32ull * k;  // x[8], y[8], z[8], original index[8]: one 128-byte line
#pragma unroll
    for (int j = 0; j < kKdLeafMax; j += 4) {
      *reinterpret_cast<float4*>(b + j) = make_float4(p[j].x, p[j + 1].x, p[j + 2].x, p[j + 3].x);
      *reinterpret_cast<float4*>(b + 8 + j) = make_float4(p[j].y, p[j + 1].y, p[j + 2].y, p[j + 3].y);
      *reinterpret_cast<float4*>(b + 16 + j) = make_float4(p[j].z, p[j + 1].z, p[j + 2].z, p[j + 3].z);
      *reinterpret_cast<float4*>(b + 24 + j) = make_float4(p[j].w, p[j + 1].w, p[j + 2].w, p[j + 3].w);
    }
    const uint32_t node = (1u << D) + k;
    boxes[2 * node] = make_float4(lo[0], lo[1], lo[2], 0.f);
    boxes[2 * node + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
    if (k == (1u << D) - 1u) {
#pragma unroll
      for (int j = 0; j < kKdLeafMax; j++) opts[n + j] = make_float4(INFINITY, INFINITY, INFINITY, __uint_as_float(0xffffffffu));  // padding leaf (kd_search.hpp)
    }
  }
  // The target's own length scale: the geometric mean of the leaf diagonals (a leaf = a neighbourhood of <= 8 points), accumulated as
  // integers (log2 in 2^-20 units: the sum does not depend on the order of the additions) and handed over as a LATE note (notes.hpp) by
  // the last workgroup to arrive.  Scaling the cloud by s scales it by s; the pass routing of linearize.hip measures motions in it.
  if (late_slot != nullptr) {  // grid-uniform
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    const float diag2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
    const bool counted = k < (1u << D) && diag2 > 0.f && diag2 < 3.0e38f;
    long long q = counted ? static_cast<long long>(rintf(0.5f * log2f(diag2) * 1048576.f)) : 0ll;
    unsigned c = counted ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) {
      q += __shfl_xor(q, off);
      c += __shfl_xor(c, off);
    }
    if ((t & 63u) == 0u) sh_sum[t >> 6] = q, sh_cnt[t >> 6] = c;
    __syncthreads();
    if (t == 0) {
      const long long bs = sh_sum[0] + sh_sum[1] + sh_sum[2] + sh_sum[3];
      const unsigned bc = sh_cnt[0] + sh_cnt[1] + sh_cnt[2] + sh_cnt[3];
      atomicAdd(&d_spacing[0], static_cast<unsigned long long>(bs));
      atomicAdd(&d_spacing[1], static_cast<unsigned long long>(bc));
      __threadfence();
      sh_last = __hip_atomic_fetch_add(&d_spacing[2], 1ull, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
      if (sh_last) {
        late_slot[0] = __hip_atomic_load(&d_spacing[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        late_slot[1] = __hip_atomic_load(&d_spacing[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        late_slot[2] = 0ull;
        __hip_atomic_store(&d_spacing[0], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&d_spacing[1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&d_spacing[2], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        note_publish(late_slot + kLateWords - 1, late_seq);
      }
    }
  }
  if (k >= 1u && k < npairs) {  // pair record of heap node k (even depths carry one)
    const int d = 31 - __clz(static_cast<int>(k));
    if ((d & 1) == 0) {
      const float2 a = nodes[k];
      float2 l = make_float2(0.f, 0.f), rr = make_float2(0.f, 0.f);
      if (d + 1 < D) {
        l = nodes[2 * k];
        rr = nodes[2 * k + 1];
      }
      const uint32_t axes = static_cast<uint32_t>(__float_as_int(a.y)) | (static_cast<uint32_t>(__float_as_int(l.y)) << 2) | (static_cast<uint32_t>(__float_as_int(rr.y)) << 4);
      pairs[k] = make_float4(a.x, l.x, rr.x, __uint_as_float(axes));
    }
  }
  for (int a = 0; a < 3; a++) {
    slo[a][t] = lo[a];
    shi[a][t] = hi[a];
  }
  __syncthreads();
  {  // group headers: the leaves under a node of depth D - G, G = min(2, D)
    const int G = D < 2 ? D : 2;
    const uint32_t per = 1u << G;
    if ((t & (per - 1u)) == 0u && k < (1u << D)) {
      float4* h = groups + 8ull * (k >> G);
      for (int a = 0; a < 3; a++) {
        float l4[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, h4[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        for (uint32_t l = 0; l < per; l++) l4[l] = slo[a][t + l], h4[l] = shi[a][t + l];
        h[a] = make_float4(l4[0], l4[1], l4[2], l4[3]);
        h[3 + a] = make_float4(h4[0], h4[1], h4[2], h4[3]);
      }
      h[6] = h[7] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  for (int l = 1; l <= 8 && l <= D; l++) {
    const uint32_t width = 256u >> l;  // nodes of depth D - l in this workgroup
    if (t < width) {
      for (int a = 0; a < 3; a++) {
        lo[a] = fminf(slo[a][2 * t], slo[a][2 * t + 1]);
        hi[a] = fmaxf(shi[a][2 * t], shi[a][2 * t + 1]);
      }
    }
    __syncthreads();
    if (t < width) {
      for (int a = 0; a < 3; a++) {
        slo[a][t] = lo[a];
        shi[a][t] = hi[a];
      }
      const uint32_t kk = (blockIdx.x * 256u >> l) + t;
      if (kk < (1u << (D - l))) {
        const uint32_t node = (1u << (D - l)) + kk;
        boxes[2 * node] = make_float4(lo[0], lo[1], lo[2], 0.f);
        boxes[2 * node + 1] = make_float4(hi[0], hi[1], hi[2], 0.f);
      }
    }
    __syncthreads();
  }
}

__global__ void compose_perm_kernel(const uint32_t* __restrict__ inner, const uint32_t* __restrict__ outer, uint32_t* __restrict__ out, size_t n) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i < n) out[i] = outer[inner[i]];
}

__global__ void iota_kernel(uint32_t* __restrict__ v, size_t n) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i < n) v[i] = static_cast<uint32_t>(i);
}

__global__ void gather_attr_kernel(const float4* __restrict__ sorted_pts, size_t n, const float4* __restrict__ nrm, const Cov8* __restrict__ cov, float4* __restrict__ onrm, Cov8* __restrict__ ocov) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const uint32_t s = __float_as_uint(sorted_pts[i].w);
  if (nrm) onrm[i] = nrm[s];
  if (cov) ocov[i] = cov[s];
}

// box_seq: the note (notes.hpp) that carries the cloud's bounding box to the host — from the first split level when there is one (it takes
// the box of its segment, the whole cloud, anyway), else from bbox_note_kernel
static int build_kdtree(sga_context* ctx, const sga_cloud* cloud, sga_index* idx, unsigned long long* box_seq) {
  const size_t n = cloud->n;
  idx->kd_depth = 0;
  if (n == 0) return SGA_OK;
  int D = 0;
  while (((n + (1ull << D) - 1) >> D) > static_cast<size_t>(kKdLeafMax)) D++;  // ceil(n / 2^D) <= leaf capacity
  if (D > 24) return fail(SGA_ERR_INVALID, "target too large for the kd-tree (%zu points)", n);
  idx->kd_depth = D;
  DevBuf<uint32_t> perm, perm2;
  DevBuf<unsigned long long> keys, keys2;
  DevBuf<int> seg_box;
  SGA_TRY(perm.alloc(n));
  SGA_TRY(perm2.alloc(n));
  SGA_TRY(keys.alloc(n));
  SGA_TRY(keys2.alloc(n));
  SGA_TRY(seg_box.alloc(6ull << (D > 0 ? D - 1 : 0)));
  SGA_TRY(idx->kd_nodes.alloc(1ull << D));
  SGA_TRY(idx->kd_nodes4.alloc(kd_pair_count(D)));
  const dim3 grid((n + 255) / 256), block(256);
  uint32_t* cur = perm.p;
  uint32_t* nxt = perm2.p;
  DevBuf<int> axis_of_seg;
  SGA_TRY(axis_of_seg.alloc(1ull << (D > 0 ? D - 1 : 0)));
  // Three regimes, top down — every level a median select + partition, nothing is sorted (round 6):
  //   [0, dS)   segments of more than 16 384 points (clouds of more than kSplitMaxPoints): the segment spread over many workgroups, six
  //             launches per level, the points moving with the permutation (kd_top_*_kernel);
  //   [dS, dA)  one launch per level, one workgroup per segment (kd_split_level_kernel), gathering from the copy the levels above left;
  //   [dA, D)   the rest of every sub-tree in LDS (kd_finish_kernel<kSplitFinish>).
  // The older paths stay for the tests that compare them: SGA_KD_TOP=0: [0, dS) (then: segments of more than kSplitMaxPoints points) by a
  // box pass, a key pass, a key-value sort of the whole cloud and a node pass per level; SGA_KD_SPLIT=0: no split levels (sort-based down to
  // kFinishCap, then the large LDS finish); SGA_KD_FINISH=0: sort-based throughout.
  const bool lds_finish = !(getenv("SGA_KD_FINISH") && atoi(getenv("SGA_KD_FINISH")) == 0);  // read per build: the tests compare the paths
  const bool split_levels = lds_finish && !(getenv("SGA_KD_SPLIT") && atoi(getenv("SGA_KD_SPLIT")) == 0);
  auto seg_max_at = [&](int d) { return (n + (1ull << d) - 1) >> d; };
  const int cap = n >= 400000 ? kFinishCap : kFinishCap / 2;
  int dS = 0, dA = 0;
  if (split_levels) {
    // the many-workgroup levels (kd_top_*_kernel, ~55 us per level at 1M whatever the number of segments) also take the first levels the split
    // kernel could hold: its 32-key form runs 136 us for 32 segments of 31k points (32 workgroups on 256 CUs), its 16-key form 53 us
    static const size_t top_min = getenv("SGA_KD_TOP_MIN") ? static_cast<size_t>(atoll(getenv("SGA_KD_TOP_MIN"))) : 16384;
    const bool top_on = !(getenv("SGA_KD_TOP") && atoi(getenv("SGA_KD_TOP")) == 0);
    const size_t reach = (top_on && n > kSplitMaxPoints) ? std::min<size_t>(top_min, kSplitMaxPoints) : kSplitMaxPoints;  // (clouds the split kernel holds whole stay with it)
    while (dS < D && seg_max_at(dS) > reach) dS++;
    dA = dS;
    while (dA < D && seg_max_at(dA) > static_cast<size_t>(kSplitFinish)) dA++;
  } else {
    while (dS < D && seg_max_at(dS) > static_cast<size_t>(cap)) dS++;
    if (!lds_finish || D - dS > 8) dS = D;
    dA = dS;
  }
  // SGA_KD_TOP=0: the levels above the split kernel's reach through the key-value sort (rounds 1 - 5); default: select + partition over many workgroups
  const bool top_levels = split_levels && dS > 0 && !(getenv("SGA_KD_TOP") && atoi(getenv("SGA_KD_TOP")) == 0);
  if (dS > 0 || dA == 0) {  // (otherwise the root split level below hands over the box and reads the identity permutation)
    SGA_TRY(cloud_bbox_enqueue(ctx, cloud->pts.p, n, box_seq));
    if (!top_levels) hipLaunchKernelGGL(iota_kernel, grid, block, 0, ctx->stream, perm.p, n);
  }
  DevBuf<float4> top_pts[2];
  DevBuf<uint32_t> top_keys, top_hist, top_perm;
  const float4* base_pts = cloud->pts.p;  // what the levels below gather from: the cloud, or its copy in the order the top levels left (see below)
  DevBuf<uint2> top_cnt;
  DevBuf<TopSel> top_sel;
  if (top_levels) {
    const uint32_t max_seg = 1u << (dS - 1);
    SGA_TRY(top_pts[0].alloc(n));
    SGA_TRY(top_pts[1].alloc(n));
    SGA_TRY(top_keys.alloc(n));
    const uint32_t all_seg = (1u << dS) - 1u;  // the segments of all these levels: level d's start at 2^d - 1
    SGA_TRY(top_hist.alloc(static_cast<size_t>(all_seg) * 3 * kSplitBins));
    SGA_TRY(top_cnt.alloc(static_cast<size_t>(n / kTopChunk) + 2 + 2 * static_cast<size_t>(max_seg)));  // segments x chunks of the longest segment <= n / chunk + 2 x segments
    SGA_TRY(top_sel.alloc(max_seg));
    SGA_TRY(top_perm.alloc(n));
    const float4* pin = cloud->pts.p;
    const uint32_t* permin = nullptr;  // level 0 reads the identity
    // boxes and histograms of ALL these levels are cleared once (level d's segments at offset 2^d - 1); the box pass runs for the root only:
    // every scatter takes the boxes of its segments' children on the way
    hipLaunchKernelGGL(kd_init_box_kernel, dim3((all_seg + 255) / 256), block, 0, ctx->stream, seg_box.p, all_seg);
    SGA_HIP(hipMemsetAsync(top_hist.p, 0, static_cast<size_t>(all_seg) * 3 * kSplitBins * sizeof(uint32_t), ctx->stream));
    for (int d = 0; d < dS; d++) {
      const uint32_t nseg = 1u << d;
      const uint32_t chunks = static_cast<uint32_t>((seg_max_at(d) + kTopChunk - 1) / kTopChunk);
      const dim3 tgrid(chunks, nseg), tblock(kTopThreads);
      float4* pout = top_pts[d & 1].p;
      int* box_d = seg_box.p + 6 * static_cast<size_t>(nseg - 1u);
      uint32_t* hist_d = top_hist.p + static_cast<size_t>(nseg - 1u) * 3 * kSplitBins;
      if (d == 0) hipLaunchKernelGGL(kd_top_box_kernel, tgrid, tblock, 0, ctx->stream, pin, static_cast<uint32_t>(n), d, box_d);
      hipLaunchKernelGGL(kd_top_hist_kernel<0>, tgrid, tblock, 0, ctx->stream, pin, top_keys.p, static_cast<uint32_t>(n), d, box_d, axis_of_seg.p, hist_d);
      hipLaunchKernelGGL(kd_top_hist_kernel<1>, tgrid, tblock, 0, ctx->stream, pin, top_keys.p, static_cast<uint32_t>(n), d, box_d, axis_of_seg.p, hist_d);
      hipLaunchKernelGGL(kd_top_hist_kernel<2>, tgrid, tblock, 0, ctx->stream, pin, top_keys.p, static_cast<uint32_t>(n), d, box_d, axis_of_seg.p, hist_d);
      hipLaunchKernelGGL(kd_top_count_kernel, tgrid, tblock, 0, ctx->stream, top_keys.p, static_cast<uint32_t>(n), d, axis_of_seg.p, hist_d, chunks, top_cnt.p, top_sel.p, idx->kd_nodes.p);
      uint32_t* perm_out = d + 1 == dS ? top_perm.p : nxt;  // the last of these levels keeps its permutation in a buffer of its own
      int* child_box = d + 1 < dS ? seg_box.p + 6 * static_cast<size_t>(2u * nseg - 1u) : static_cast<int*>(nullptr);
      hipLaunchKernelGGL(kd_top_scatter_kernel, tgrid, tblock, 0, ctx->stream, pin, permin, top_keys.p, static_cast<uint32_t>(n), d, chunks, top_cnt.p, top_sel.p, pout, perm_out, child_box);
      SGA_HIP(hipGetLastError());
      pin = pout;
      permin = perm_out;
      if (d + 1 < dS) std::swap(cur, nxt);
    }
    // The levels below work on the MOVED copy of the points with the identity as their input permutation: a segment of <= 16 384
    // points is a contiguous 256 KB of it, so all their gathers stay inside one L2 (through the permutation of the whole cloud they touched a
    // 64-byte sector of a 16 MB array per point).  Their result — positions in the moved copy — is composed with top_perm at the end.
    base_pts = pin;
  }
  for (int d = 0; d < dS && !top_levels; d++) {
    const uint32_t nseg = 1u << d;
    const dim3 sgrid((nseg + 255) / 256);
    const unsigned end_bit = 32 + (d > 0 ? d : 1);
    if (d == 0) hipLaunchKernelGGL(kd_init_box_kernel, sgrid, block, 0, ctx->stream, seg_box.p, nseg);  // later levels: reset by kd_nodes_kernel
    {
      // six atomics per workgroup on a handful of addresses: large workgroups for large clouds (fewer atomics), small ones for
      // small clouds (a 15k-point scan in 1024-thread workgroups would occupy 15 CUs)
      const unsigned bs = n > 200000 ? 1024u : 256u;
      hipLaunchKernelGGL(kd_segment_box_kernel, dim3((n + bs - 1) / bs), dim3(bs), 0, ctx->stream, cloud->pts.p, cur, static_cast<uint32_t>(n), d, seg_box.p);
    }
    hipLaunchKernelGGL(kd_keys_kernel, grid, block, 0, ctx->stream, cloud->pts.p, cur, static_cast<uint32_t>(n), d, seg_box.p, axis_of_seg.p, keys.p);
    SGA_TRY(sort_pairs(ctx, keys.p, keys2.p, cur, nxt, n, 0, end_bit));
    std::swap(cur, nxt);
    hipLaunchKernelGGL(kd_nodes_kernel, sgrid, block, 0, ctx->stream, cloud->pts.p, cur, static_cast<uint32_t>(n), d, axis_of_seg.p, idx->kd_nodes.p, d + 1 < dS ? seg_box.p : static_cast<int*>(nullptr));
  }
  for (int d = dS; d < dA; d++) {
    const size_t seg_max = seg_max_at(d);
    unsigned long long* note_slot = nullptr;
    if (d == 0) {  // the root level reads the identity permutation and hands the cloud's box to the host
      *box_seq = note_begin(ctx, &note_slot);
    }
    const uint32_t* level_in = (d == 0 || (top_levels && d == dS)) ? nullptr : cur;
#define SGA_SPLIT(THREADS, KEYS) hipLaunchKernelGGL((kd_split_level_kernel<THREADS, KEYS>), dim3(1u << d), dim3(THREADS), 0, ctx->stream, base_pts, level_in, nxt, static_cast<uint32_t>(n), d, idx->kd_nodes.p, note_slot, *box_seq)
    if (seg_max <= 256 * 2) SGA_SPLIT(256, 2);
    else if (seg_max <= 256 * 4) SGA_SPLIT(256, 4);
    else if (seg_max <= 256 * 8) SGA_SPLIT(256, 8);
    else if (seg_max <= 1024 * 4) SGA_SPLIT(1024, 4);
    else if (seg_max <= 1024 * 8) SGA_SPLIT(1024, 8);
    else if (seg_max <= 1024 * 16) SGA_SPLIT(1024, 16);
    else SGA_SPLIT(1024, 32);
#undef SGA_SPLIT
    std::swap(cur, nxt);
  }
  if (top_levels && dA == dS) hipLaunchKernelGGL(iota_kernel, grid, block, 0, ctx->stream, cur, n);  // (no split level ran: the finish reads the identity)
  if (dA < D && split_levels) {
    hipLaunchKernelGGL((kd_finish_kernel<kSplitFinish, kSplitFinish>), dim3(1u << dA), dim3(kSplitFinish), 0, ctx->stream, base_pts, cur, nxt, static_cast<uint32_t>(n), dA, D, idx->kd_nodes.p);
    std::swap(cur, nxt);
  } else if (dA < D) {
    if (cap == kFinishCap)
      hipLaunchKernelGGL(kd_finish_kernel<kFinishCap>, dim3(1u << dA), dim3(kFinishThreads), 0, ctx->stream, cloud->pts.p, cur, nxt, static_cast<uint32_t>(n), dA, D, idx->kd_nodes.p);
    else
      hipLaunchKernelGGL(kd_finish_kernel<kFinishCap / 2>, dim3(1u << dA), dim3(kFinishThreads), 0, ctx->stream, cloud->pts.p, cur, nxt, static_cast<uint32_t>(n), dA, D, idx->kd_nodes.p);
    std::swap(cur, nxt);
  }
  if (top_levels) {  // positions in the moved copy -> indices of the cloud
    hipLaunchKernelGGL(compose_perm_kernel, grid, block, 0, ctx->stream, cur, top_perm.p, nxt, n);
    std::swap(cur, nxt);
  }
  SGA_HIP(hipGetLastError());
  SGA_TRY(idx->kd_pts.alloc(n + kKdLeafMax));  // + one leaf of points at infinity: leaf scans read 8 slots unconditionally
  if (cloud->has_normals) SGA_TRY(idx->nrm.alloc(n));
  if (cloud->has_covs) SGA_TRY(idx->cov.alloc(n));
  SGA_TRY(idx->kd_boxes.alloc(4ull << D));
  SGA_TRY(idx->kd_groups.alloc(8ull << (D - (D < 2 ? D : 2))));
  SGA_TRY(idx->kd_leaf.alloc(8ull << D));
  unsigned long long* late_slot = nullptr;
  idx->spacing_seq = late_note_begin(ctx->device, &late_slot);  // the target's length scale arrives whenever the tail kernel has run: nobody waits for it
  idx->spacing = 0.0;
  // gather into kd order, leaf blocks, leaf boxes + 8 levels of boxes, group headers, pair records: one launch (kd_tail_kernel)
  hipLaunchKernelGGL(kd_tail_kernel, dim3(((1u << D) + 255) / 256), block, 0, ctx->stream, cur, static_cast<uint32_t>(n), D, cloud->pts.p, cloud->has_normals ? cloud->nrm.p : nullptr, cloud->has_covs ? cloud->cov.p : nullptr, idx->kd_nodes.p,
                     idx->kd_pts.p, idx->nrm.p, idx->cov.p, idx->kd_boxes.p, idx->kd_groups.p, reinterpret_cast<float*>(idx->kd_leaf.p), idx->kd_nodes4.p, kd_pair_count(D), ctx->d_spacing.p, late_slot, idx->spacing_seq);
  for (int base = D - 8; base > 0; base -= 8) hipLaunchKernelGGL(kd_boxes_kernel, dim3(((1u << base) + 255) / 256), block, 0, ctx->stream, idx->kd_pts.p, static_cast<uint32_t>(n), D, base, idx->kd_boxes.p);
  SGA_HIP(hipGetLastError());
  if (!ctx->stream_ordered) SGA_HIP(hipStreamSynchronize(ctx->stream));
  return SGA_OK;
}

// ---- voxel map kernels -------------------------------------------------------------------------------------------------------
// (ox, oy, oz): origin of the cloud's device frame — voxel coordinates are those of the CALLER's frame (incremental_voxelmap.hpp:60)
__global__ void voxel_keys_kernel(const float4* __restrict__ pts, size_t n, double inv_leaf, double ox, double oy, double oz, unsigned long long* __restrict__ keys, uint32_t* __restrict__ vals) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const float4 p = pts[i];
  const int cx = fast_floor_d((static_cast<double>(p.x) + ox) * inv_leaf), cy = fast_floor_d((static_cast<double>(p.y) + oy) * inv_leaf), cz = fast_floor_d((static_cast<double>(p.z) + oz) * inv_leaf);
  const bool bad = abs(cx) >= (1 << 20) || abs(cy) >= (1 << 20) || abs(cz) >= (1 << 20);
  keys[i] = bad ? SGA_HASH_EMPTY : voxel_key(cx, cy, cz);  // out-of-range points sort last and are dropped
  vals[i] = static_cast<uint32_t>(i);
}

__global__ void segment_heads_kernel(const unsigned long long* __restrict__ keys, size_t n, uint32_t* __restrict__ flags) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  const unsigned long long k = keys[i];
  flags[i] = (k != SGA_HASH_EMPTY && (i == 0 || keys[i - 1] != k)) ? 1u : 0u;
}

__global__ void segment_starts_kernel(const uint32_t* __restrict__ flags, const uint32_t* __restrict__ seg_id, const uint32_t* __restrict__ order, size_t n, uint32_t* __restrict__ seg_start, uint32_t* __restrict__ seg_first_idx, uint32_t* __restrict__ seg_ids) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  if (flags[i]) {
    const uint32_t s = seg_id[i];
    seg_start[s] = static_cast<uint32_t>(i);
    seg_first_idx[s] = order[i];  // stable sort: the first entry of a segment is the earliest inserted point
    seg_ids[s] = s;
  }
}

// One thread per voxel (in voxel-id order): mean of points and mean of covariances, summed in insertion order in fp64.
__global__ void voxel_finalize_kernel(
  const uint32_t* __restrict__ seg_by_rank, uint32_t nvox, const uint32_t* __restrict__ seg_start, uint32_t n_valid, const unsigned long long* __restrict__ keys, const uint32_t* __restrict__ order,
  const float4* __restrict__ pts, const Cov8* __restrict__ cov, float4* __restrict__ means, Cov8* __restrict__ mcov, int* __restrict__ coords, uint32_t* __restrict__ counts,
  unsigned long long* __restrict__ hkeys, uint32_t* __restrict__ hvals, uint32_t hmask) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nvox) return;
  const uint32_t seg = seg_by_rank[v];
  const uint32_t s = seg_start[seg];
  const unsigned long long key = keys[s];
  double m[3] = {0, 0, 0}, c[6] = {0, 0, 0, 0, 0, 0};
  uint32_t cnt = 0;
  for (uint32_t i = s; i < n_valid && keys[i] == key; ++i) {
    const uint32_t src = order[i];
    const float4 p = pts[src];
    const Cov8 q = cov[src];
    m[0] += p.x;
    m[1] += p.y;
    m[2] += p.z;
    c[0] += q.xx;
    c[1] += q.xy;
    c[2] += q.xz;
    c[3] += q.yy;
    c[4] += q.yz;
    c[5] += q.zz;
    cnt++;
  }
  const double inv = 1.0 / cnt;
  means[v] = make_float4(static_cast<float>(m[0] * inv), static_cast<float>(m[1] * inv), static_cast<float>(m[2] * inv), __uint_as_float(v));
  Cov8 o;
  o.xx = static_cast<float>(c[0] * inv);
  o.xy = static_cast<float>(c[1] * inv);
  o.xz = static_cast<float>(c[2] * inv);
  o.yy = static_cast<float>(c[3] * inv);
  o.yz = static_cast<float>(c[4] * inv);
  o.zz = static_cast<float>(c[5] * inv);
  o.pad0 = o.pad1 = 0.f;
  mcov[v] = o;
  coords[3 * v + 0] = static_cast<int>(key & 0x1fffffu) - (1 << 20);
  coords[3 * v + 1] = static_cast<int>((key >> 21) & 0x1fffffu) - (1 << 20);
  coords[3 * v + 2] = static_cast<int>((key >> 42) & 0x1fffffu) - (1 << 20);
  counts[v] = cnt;
  uint32_t slot = voxel_hash(key) & hmask;
  for (;;) {
    const unsigned long long prev = atomicCAS(&hkeys[slot], SGA_HASH_EMPTY, key);
    if (prev == SGA_HASH_EMPTY) {
      hvals[slot] = v;
      break;
    }
    slot = (slot + 1) & hmask;
  }
}

__global__ void count_valid_keys_kernel(const unsigned long long* __restrict__ keys, size_t n, unsigned long long* __restrict__ out) {
  unsigned int local = 0;
  for (size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x) local += keys[i] != SGA_HASH_EMPTY;
  for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off);
  if ((threadIdx.x & 63) == 0 && local) atomicAdd(out, static_cast<unsigned long long>(local));
}

}  // namespace sga

namespace sga {
// the target's length scale (sga_index::spacing), once the late note of its build has arrived; 0 while it is not known
double index_spacing(const sga_index* idx) {
  if (idx->spacing > 0.0 || idx->spacing_seq == 0) return idx->spacing;
  unsigned long long payload[kLateWords - 1];
  const int r = late_note_peek(idx->spacing_seq, payload);
  if (r == 0) return 0.0;  // the build has not got there yet
  idx->spacing_seq = 0;    // read, or lost: never asked for again
  if (r == 1 && payload[1] > 0) idx->spacing = std::exp2(static_cast<double>(static_cast<long long>(payload[0])) / 1048576.0 / static_cast<double>(payload[1]));
  return idx->spacing;
}
}  // namespace sga

using namespace sga;

extern "C" {

// diagnostics: the index's length scale (0: not a kd-tree, or not known yet)
int sga_index_spacing(const sga_index* index, double* spacing) {
  if (!index || !spacing) return fail(SGA_ERR_INVALID, "null argument");
  *spacing = index->kind == SGA_INDEX_KDTREE ? index_spacing(index) : 0.0;
  return SGA_OK;
}

int sga_index_build_kdtree(sga_context* ctx, const sga_cloud* target, sga_index** out) {
  if (!ctx || !target || !out) return fail(SGA_ERR_INVALID, "null argument");
  if (target->device != ctx->device) return fail(SGA_ERR_INVALID, "cloud lives on another device");
  *out = nullptr;
  SGA_ENTER(ctx);
  const size_t n = target->n;
  std::unique_ptr<sga_index> idx(new sga_index);
  idx->kind = SGA_INDEX_KDTREE;
  idx->device = ctx->device;
  idx->n = n;
  for (int k = 0; k < 3; k++) idx->origin[k] = target->origin[k];  // the tree lives in its cloud's device frame (common.hpp)
  idx->has_normals = target->has_normals;
  idx->has_covs = target->has_covs;
  SGA_TRY(wait_ready(ctx, target->ready));  // attributes estimated on another context in stream-ordered mode
  if (n > 0) {
    // the bounding box travels to the host as a note (notes.hpp) while the build behind it is being enqueued
    unsigned long long box_seq = 0;
    SGA_TRY(build_kdtree(ctx, target, idx.get(), &box_seq));
    SGA_TRY(cloud_bbox_collect(ctx, box_seq, n, idx->bbox_lo, idx->bbox_hi));
    for (int k = 0; k < 3; k++)
      if (!std::isfinite(idx->bbox_lo[k]) || !std::isfinite(idx->bbox_hi[k])) return fail(SGA_ERR_INVALID, "target cloud contains non-finite coordinates");
    SGA_TRY(build_cell_grid(ctx, idx.get()));  // large targets: the second search structure (cell_grid.hpp)
    SGA_TRY(mark_ready(ctx, idx->ready));
  }
  *out = idx.release();
  return SGA_OK;
}

int sga_index_build_gaussian_voxelmap(sga_context* ctx, const sga_cloud* cloud, double leaf, sga_index** out) {
  if (!ctx || !cloud || !out) return fail(SGA_ERR_INVALID, "null argument");
  if (!(leaf > 0)) return fail(SGA_ERR_INVALID, "leaf size must be positive");
  if (!cloud->has_covs) return fail(SGA_ERR_INVALID, "GaussianVoxelMap needs point covariances");
  if (cloud->device != ctx->device) return fail(SGA_ERR_INVALID, "cloud lives on another device");
  *out = nullptr;
  SGA_ENTER(ctx);
  const size_t n = cloud->n;
  std::unique_ptr<sga_index> idx(new sga_index);
  idx->kind = SGA_INDEX_VOXELMAP;
  idx->device = ctx->device;
  idx->leaf = leaf;
  idx->has_covs = true;
  idx->has_normals = false;
  for (int k = 0; k < 3; k++) idx->origin[k] = cloud->origin[k];  // the means are averages of the cloud's device-frame records
  SGA_TRY(wait_ready(ctx, cloud->ready));
  uint32_t nvox = 0;
  DevBuf<unsigned long long> keys, keys_sorted;
  DevBuf<uint32_t> vals, order, flags, seg_id, seg_start, seg_first, seg_ids, seg_first_sorted, seg_by_rank;
  DevBuf<unsigned long long> d_count;
  unsigned long long n_valid = 0;
  if (n > 0) {
    SGA_TRY(keys.alloc(n));
    SGA_TRY(keys_sorted.alloc(n));
    SGA_TRY(vals.alloc(n));
    SGA_TRY(order.alloc(n));
    SGA_TRY(flags.alloc(n));
    SGA_TRY(seg_id.alloc(n));
    SGA_TRY(d_count.alloc(1));
    hipLaunchKernelGGL(voxel_keys_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, cloud->pts.p, n, 1.0 / leaf, cloud->origin[0], cloud->origin[1], cloud->origin[2], keys.p, vals.p);
    SGA_TRY(sort_pairs(ctx, keys.p, keys_sorted.p, vals.p, order.p, n, 0, 64));
    hipLaunchKernelGGL(segment_heads_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, keys_sorted.p, n, flags.p);
    size_t tb2 = 0;
    SGA_HIP(rocprim::exclusive_scan(nullptr, tb2, flags.p, seg_id.p, 0u, n, rocprim::plus<uint32_t>(), ctx->stream));
    SGA_TRY(ensure_temp(ctx, tb2));
    SGA_HIP(rocprim::exclusive_scan(ctx->d_temp.p, tb2, flags.p, seg_id.p, 0u, n, rocprim::plus<uint32_t>(), ctx->stream));
    SGA_HIP(hipMemsetAsync(d_count.p, 0, sizeof(unsigned long long), ctx->stream));
    hipLaunchKernelGGL(count_valid_keys_kernel, dim3(256), dim3(256), 0, ctx->stream, keys_sorted.p, n, d_count.p);
    uint32_t last_flag = 0, last_seg = 0;
    SGA_HIP(hipMemcpyAsync(&last_flag, flags.p + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    SGA_HIP(hipMemcpyAsync(&last_seg, seg_id.p + (n - 1), 4, hipMemcpyDeviceToHost, ctx->stream));
    SGA_HIP(hipMemcpyAsync(&n_valid, d_count.p, 8, hipMemcpyDeviceToHost, ctx->stream));
    SGA_HIP(hipStreamSynchronize(ctx->stream));
    nvox = last_seg + last_flag;
  }
  idx->n = nvox;
  uint32_t hsize = 16;
  while (hsize < 2 * static_cast<uint64_t>(nvox)) hsize <<= 1;
  idx->hmask = hsize - 1;
  SGA_TRY(idx->hkeys.alloc(hsize));
  SGA_TRY(idx->hvals.alloc(hsize));
  SGA_HIP(hipMemsetAsync(idx->hkeys.p, 0xff, hsize * sizeof(unsigned long long), ctx->stream));
  SGA_HIP(hipMemsetAsync(idx->hvals.p, 0, hsize * sizeof(uint32_t), ctx->stream));
  if (nvox > 0) {
    SGA_TRY(seg_start.alloc(nvox));
    SGA_TRY(seg_first.alloc(nvox));
    SGA_TRY(seg_ids.alloc(nvox));
    SGA_TRY(seg_first_sorted.alloc(nvox));
    SGA_TRY(seg_by_rank.alloc(nvox));
    hipLaunchKernelGGL(segment_starts_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, flags.p, seg_id.p, order.p, n, seg_start.p, seg_first.p, seg_ids.p);
    // voxel id = rank of the voxel's first inserted point (incremental_voxelmap.hpp:63-69: flat_voxels grows in first-touch order)
    size_t tb = 0;
    SGA_HIP(rocprim::radix_sort_pairs(nullptr, tb, seg_first.p, seg_first_sorted.p, seg_ids.p, seg_by_rank.p, nvox, 0, 32, ctx->stream));
    SGA_TRY(ensure_temp(ctx, tb));
    SGA_HIP(rocprim::radix_sort_pairs(ctx->d_temp.p, tb, seg_first.p, seg_first_sorted.p, seg_ids.p, seg_by_rank.p, nvox, 0, 32, ctx->stream));
    SGA_TRY(idx->pts.alloc(nvox));
    SGA_TRY(idx->cov.alloc(nvox));
    SGA_TRY(idx->vcoords.alloc(static_cast<size_t>(nvox) * 3));
    SGA_TRY(idx->vcounts.alloc(nvox));
    hipLaunchKernelGGL(
      voxel_finalize_kernel, dim3((nvox + 127) / 128), dim3(128), 0, ctx->stream, seg_by_rank.p, nvox, seg_start.p, static_cast<uint32_t>(n_valid), keys_sorted.p, order.p, cloud->pts.p, cloud->cov.p, idx->pts.p, idx->cov.p,
      idx->vcoords.p, idx->vcounts.p, idx->hkeys.p, idx->hvals.p, idx->hmask);
    SGA_HIP(hipGetLastError());
  }
  SGA_HIP(hipStreamSynchronize(ctx->stream));
  *out = idx.release();
  return SGA_OK;
}

// A deep copy of an index on the context's device — which may be another device than the source's (peer copy): how a replicated target
// reaches the other GPUs of a sharded registration without being built G times (a build is ~0.13 s at 1M points, a copy of its ~200 MB a
// few milliseconds over xGMI).  The copy is an independent object (same voxel ids / kd order, same device frame).
}  // extern "C"
namespace {
template <typename T>
int copy_buf(sga_context* ctx, sga::DevBuf<T>& dst, const sga::DevBuf<T>& src, int src_device) {
  if (src.n == 0 || src.p == nullptr) return SGA_OK;
  SGA_TRY(dst.alloc(src.n));
  if (src_device == ctx->device)
    SGA_HIP(hipMemcpyAsync(dst.p, src.p, src.n * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
  else
    SGA_HIP(hipMemcpyPeerAsync(dst.p, ctx->device, src.p, src_device, src.n * sizeof(T), ctx->stream));
  return SGA_OK;
}
}  // namespace
extern "C" {

int sga_index_clone(sga_context* ctx, const sga_index* src, sga_index** out) {
  if (!ctx || !src || !out) return fail(SGA_ERR_INVALID, "null argument");
  *out = nullptr;
  SGA_ENTER(ctx);
  if (src->ready.pending && src->ready.stream != ctx->stream && src->device == ctx->device) SGA_TRY(wait_ready(ctx, src->ready));
  if (src->ready.pending && src->device != ctx->device && src->ready.event) SGA_HIP(hipEventSynchronize(src->ready.event));  // (another device's stream: wait on the host)
  std::unique_ptr<sga_index> idx(new sga_index);
  idx->kind = src->kind;
  idx->device = ctx->device;
  idx->n = src->n;
  for (int k = 0; k < 3; k++) idx->origin[k] = src->origin[k], idx->bbox_lo[k] = src->bbox_lo[k], idx->bbox_hi[k] = src->bbox_hi[k], idx->grid_org[k] = src->grid_org[k], idx->grid_dim[k] = src->grid_dim[k];
  idx->has_normals = src->has_normals;
  idx->has_covs = src->has_covs;
  idx->kd_depth = src->kd_depth;
  idx->spacing = index_spacing(src);
  idx->spacing_seq = 0;
  idx->grid_h = src->grid_h;
  idx->grid_eps = src->grid_eps;
  idx->leaf = src->leaf;
  idx->hmask = src->hmask;
  idx->incremental = src->incremental;
  idx->vcap = src->vcap;
  idx->lru_counter = src->lru_counter;
  idx->lru_horizon = src->lru_horizon;
  idx->lru_clear_cycle = src->lru_clear_cycle;
  idx->flat_max = src->flat_max;
  idx->flat_min_sq = src->flat_min_sq;
  idx->search_offsets = src->search_offsets;
  const int sd = src->device;
  SGA_TRY(copy_buf(ctx, idx->kd_pts, src->kd_pts, sd));
  SGA_TRY(copy_buf(ctx, idx->nrm, src->nrm, sd));
  SGA_TRY(copy_buf(ctx, idx->cov, src->cov, sd));
  SGA_TRY(copy_buf(ctx, idx->kd_nodes, src->kd_nodes, sd));
  SGA_TRY(copy_buf(ctx, idx->kd_nodes4, src->kd_nodes4, sd));
  SGA_TRY(copy_buf(ctx, idx->kd_boxes, src->kd_boxes, sd));
  SGA_TRY(copy_buf(ctx, idx->kd_groups, src->kd_groups, sd));
  SGA_TRY(copy_buf(ctx, idx->kd_leaf, src->kd_leaf, sd));
  SGA_TRY(copy_buf(ctx, idx->grid_pts, src->grid_pts, sd));
  SGA_TRY(copy_buf(ctx, idx->grid_start, src->grid_start, sd));
  SGA_TRY(copy_buf(ctx, idx->pts, src->pts, sd));
  SGA_TRY(copy_buf(ctx, idx->hkeys, src->hkeys, sd));
  SGA_TRY(copy_buf(ctx, idx->hvals, src->hvals, sd));
  SGA_TRY(copy_buf(ctx, idx->vcoords, src->vcoords, sd));
  SGA_TRY(copy_buf(ctx, idx->vcounts, src->vcounts, sd));
  SGA_TRY(copy_buf(ctx, idx->vmean64, src->vmean64, sd));
  SGA_TRY(copy_buf(ctx, idx->vcov64, src->vcov64, sd));
  SGA_TRY(copy_buf(ctx, idx->vlru, src->vlru, sd));
  SGA_TRY(copy_buf(ctx, idx->fpts64, src->fpts64, sd));
  SGA_TRY(copy_buf(ctx, idx->fcov64, src->fcov64, sd));
  if (!ctx->stream_ordered) SGA_HIP(hipStreamSynchronize(ctx->stream));
  SGA_TRY(mark_ready(ctx, idx->ready));
  *out = idx.release();
  return SGA_OK;
}

int sga_index_destroy(sga_index* index) {
  if (index) {
    (void)hipSetDevice(index->device);
    delete index;
  }
  return SGA_OK;
}

int sga_index_size(const sga_index* index, size_t* n) {
  if (!index || !n) return fail(SGA_ERR_INVALID, "null argument");
  *n = index->n;
  return SGA_OK;
}

int sga_index_voxelmap_download(sga_context* ctx, const sga_index* index, int32_t* coords, float* means, float* cov6, uint32_t* counts) {
  if (!ctx || !index) return fail(SGA_ERR_INVALID, "null argument");
  if (index->kind != SGA_INDEX_VOXELMAP) return fail(SGA_ERR_INVALID, "not a voxel map");
  const size_t n = index->n;
  if (n == 0) return SGA_OK;
  SGA_ENTER(ctx);
  std::vector<float4> hp;
  std::vector<Cov8> hc;
  if (means) {
    hp.resize(n);
    SGA_HIP(hipMemcpyAsync(hp.data(), index->pts.p, n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
  }
  if (cov6) {
    hc.resize(n);
    SGA_HIP(hipMemcpyAsync(hc.data(), index->cov.p, n * sizeof(Cov8), hipMemcpyDeviceToHost, ctx->stream));
  }
  if (coords) SGA_HIP(hipMemcpyAsync(coords, index->vcoords.p, n * 3 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  if (counts) SGA_HIP(hipMemcpyAsync(counts, index->vcounts.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  SGA_HIP(hipStreamSynchronize(ctx->stream));
  for (size_t i = 0; i < n; i++) {
    if (means) {  // device frame -> the caller's
      means[3 * i] = static_cast<float>(static_cast<double>(hp[i].x) + index->origin[0]);
      means[3 * i + 1] = static_cast<float>(static_cast<double>(hp[i].y) + index->origin[1]);
      means[3 * i + 2] = static_cast<float>(static_cast<double>(hp[i].z) + index->origin[2]);
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

}  // extern "C"
