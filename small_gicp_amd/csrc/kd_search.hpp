// Exact nearest neighbour over an implicit, perfectly balanced kd-tree (the registration hot path's search; the GPU
// counterpart of ann/kdtree.hpp:80-126 build + :193-233 search in the reference tree /root/reference).  Device code, gfx950.
//
// Layout.  The target points are stored in kd order.  Node (depth d, rank k) owns the contiguous range
// [B(d,k), B(d,k+1)) with B(d,k) = floor(k * n / 2^d); it is split at m = B(d+1, 2k+1): left = [B(d,k), m), right = [m, B(d,k+1)),
// exactly the reference's median split (kdtree.hpp:118-126: left = [first, median), right = [median, last), threshold =
// coordinate of the median element).  Nodes are a 1-based heap (node = 2^d + k, children 2*node and 2*node+1) of
// {threshold, axis}: 8 bytes, no child pointers, no leaf records.  Leaves sit at the fixed depth D with <= 8 points.
//
// Search = the reference's recursion (descend to the near side; visit the far side iff worst > cut^2, kdtree.hpp:207-230)
// unrolled onto an explicit per-lane stack kept in LDS as [level][lane] (conflict-free).  The far side is re-tested against
// the CURRENT best when it is popped, like the recursion does after the near side has returned.
// Ties: the lowest kd position wins (the reference's tie rule is traversal-order dependent, knn_result.hpp:81-83).
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace sga {

#ifndef SGA_KD_LEAF
#define SGA_KD_LEAF 8
#endif
constexpr int kKdLeafMax = SGA_KD_LEAF;  // points per leaf (<=); 8 x 16 B = one 128-byte line
constexpr int kKdMaxDepth = 24;  // stack slots per lane; tree depth D <= 24 (n <= 2^27)
constexpr int kKdTopLevels = 10;  // nodes of depth < 10 (1023 x 8 B) can be mirrored in LDS: no vector-memory access for the top of every descent

struct KdView {
  const float4* __restrict__ pts;    // kd order, w = original index bits
  const float2* __restrict__ nodes;  // [2^D] heap: x = threshold, y = bitcast(axis)
  uint32_t n;
  int depth;  // D; leaves are the 2^D ranges at depth D
  unsigned long long* stats;  // SGA_DEBUG_STATS counters or null
};

inline KdView make_kd_view(const sga_index* idx) {
  KdView k;
  k.pts = idx->kd_pts.p;
  k.nodes = idx->kd_nodes.p;
  k.n = static_cast<uint32_t>(idx->n);
  k.depth = idx->kd_depth;
  k.stats = nullptr;
  if (getenv("SGA_DEBUG_STATS")) {
    static unsigned long long* d_stats = nullptr;
    if (!d_stats) {
      (void)hipMalloc(reinterpret_cast<void**>(&d_stats), 16 * sizeof(unsigned long long));
      (void)hipMemset(d_stats, 0, 16 * sizeof(unsigned long long));
    }
    k.stats = d_stats;
  }
  return k;
}

struct KdBest {
  float d2;
  int idx;  // position in the kd-ordered target, -1 = none
  float x, y, z;
};

__host__ __device__ __forceinline__ uint32_t kd_bound(uint32_t n, int d, uint32_t k) { return static_cast<uint32_t>((static_cast<unsigned long long>(k) * n) >> d); }

__device__ __forceinline__ uint32_t kd_leaf_of(uint32_t i, uint32_t n, int d) {
  uint32_t k = static_cast<uint32_t>((static_cast<unsigned long long>(i) << d) / n);
  while (kd_bound(n, d, k + 1) <= i) k++;
  while (kd_bound(n, d, k) > i) k--;
  return k;
}

// One stack entry per pending far side, 32 bits: [31:5] = cut^2 (float bits >> 4, i.e. rounded toward zero: conservative),
// [4:0] = depth of the far node.  The far node itself is implied: it is the sibling of the depth-`dd` ancestor of the leaf
// the walk currently stands on, so no node index has to be stored.
__device__ __forceinline__ uint32_t kd_pack(float cut, int depth) { return ((__float_as_uint(cut) >> 4) << 5) | static_cast<uint32_t>(depth); }
__device__ __forceinline__ float kd_cut(uint32_t e) { return __uint_as_float((e >> 5) << 4); }

// stack: LDS, kKdMaxDepth * STRIDE words (STRIDE = threads per workgroup); this lane uses stack[level * STRIDE + tid].
// bound2: only points with d2 < bound2 can win (pass max_sq nudged up by one ulp so that d2 == max_sq is still found).
// hint:   position (kd order) of a target point believed to be near the query — normally the neighbour found for this source
//         point in the previous optimizer iteration — or -1.  With a hint the search runs BOTTOM-UP: scan the hint's leaf, then
//         climb its ancestors and descend into a sibling sub-tree only if the query lies on that side of the split or the
//         split plane is closer than the best distance so far.  The result is the same exact nearest neighbour; the hint only
//         removes the root-to-leaf descent and almost all backtracking once the pose has settled.
template <int STRIDE>
__device__ __forceinline__ KdBest kd_nearest(const KdView& t, float qx, float qy, float qz, float bound2, int hint, uint32_t* __restrict__ stack, int tid, bool active, const float2* __restrict__ top = nullptr) {
  KdBest best;
  best.d2 = bound2;
  best.idx = -1;
  best.x = best.y = best.z = 0.f;
  if (!active || t.n == 0) return best;
  const int D = t.depth;
  unsigned int n_int = 0, n_leaf = 0;

  auto leaf_scan = [&](uint32_t leaf_node) {
    n_leaf++;
    const uint32_t k = leaf_node - (1u << D);
    const uint32_t first = kd_bound(t.n, D, k), end = kd_bound(t.n, D, k + 1);
    if (first >= end) return;
    const uint32_t last = end - 1;
    float4 p[kKdLeafMax];
#pragma unroll
    for (int i = 0; i < kKdLeafMax; i++) p[i] = t.pts[min(first + i, last)];
#pragma unroll
    for (int i = 0; i < kKdLeafMax; i++) {
      const float dx = p[i].x - qx, dy = p[i].y - qy, dz = p[i].z - qz;
      const float d2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
      const int id = static_cast<int>(min(first + i, last));
      // canonical tie rule (lowest position wins) so that the answer does not depend on the visit order / the hint
      if (d2 < best.d2 || (d2 == best.d2 && best.idx >= 0 && id < best.idx)) {
        best.d2 = d2;
        best.idx = id;
        best.x = p[i].x;
        best.y = p[i].y;
        best.z = p[i].z;
      }
    }
  };

  // depth-first search of the sub-tree rooted at `start` (depth `sdepth`), nearer side first (kdtree.hpp:207-230)
  auto subtree = [&](uint32_t start, int sdepth) {
    int sp = 0, depth = sdepth;
    uint32_t node = start;
    for (;;) {
      while (depth < D) {
        const float2 nd = (top != nullptr && depth < kKdTopLevels) ? top[node] : t.nodes[node];
        const int axis = __float_as_int(nd.y);
        const float qa = axis == 0 ? qx : (axis == 1 ? qy : qz);
        const float diff = qa - nd.x;
        const float cut = diff * diff;
        depth++;
        if (cut <= best.d2) {  // '<=': an equidistant point on the far side may win the canonical tie
          stack[sp * STRIDE + tid] = kd_pack(cut, depth);
          sp++;
        }
        node = 2 * node + (diff < 0.f ? 0u : 1u);
        n_int++;
      }
      leaf_scan(node);
      bool found = false;
      while (sp > 0) {
        sp--;
        const uint32_t e = stack[sp * STRIDE + tid];
        if (kd_cut(e) <= best.d2) {
          depth = static_cast<int>(e & 31u);
          node = (node >> (D - depth)) ^ 1u;  // sibling of the current leaf's ancestor at that depth
          found = true;
          break;
        }
      }
      if (!found) break;
    }
  };

  if (hint < 0 || static_cast<uint32_t>(hint) >= t.n) {
    subtree(1u, 0);
  } else {
    const uint32_t leaf = (1u << D) + kd_leaf_of(static_cast<uint32_t>(hint), t.n, D);
    leaf_scan(leaf);
    for (int dd = D - 1; dd >= 0; --dd) {
      const uint32_t anc = leaf >> (D - dd);
      const uint32_t sib = (leaf >> (D - dd - 1)) ^ 1u;
      const float2 nd = t.nodes[anc];
      const int axis = __float_as_int(nd.y);
      const float qa = axis == 0 ? qx : (axis == 1 ? qy : qz);
      const float diff = qa - nd.x;
      n_int++;
      // right children (odd) hold coordinates >= threshold, left children <= threshold
      const bool on_sibling_side = (sib & 1u) ? (diff >= 0.f) : (diff <= 0.f);
      const float cut = on_sibling_side ? 0.f : diff * diff;
      if (cut <= best.d2) subtree(sib, dd + 1);
    }
  }
  if (t.stats) {
    atomicAdd(&t.stats[0], static_cast<unsigned long long>(n_int));
    atomicAdd(&t.stats[1], static_cast<unsigned long long>(n_leaf));
    atomicMax(&t.stats[2], static_cast<unsigned long long>(n_int + n_leaf));
    atomicAdd(&t.stats[3], 1ull);
  }
  return best;
}

// ---- k nearest neighbours (traits::knn_search; normal / covariance estimation) -----------------------------------------------------
// Same walk as kd_nearest with the k-th best distance as the pruning bound.  The k-best list lives in LDS as [k][STRIDE]
// (lane-contiguous, conflict-free), sorted ascending by (distance, position) — KnnResult<-1>::push (ann/knn_result.hpp:80-100)
// with a canonical tie rule.  sd / si must be initialised to +inf / -1 by the caller.  bound2: ignore points with d2 >= bound2.
template <int STRIDE>
__device__ __forceinline__ void kd_knn(const KdView& t, float qx, float qy, float qz, int k, float bound2, float* __restrict__ sd, int* __restrict__ si, uint32_t* __restrict__ stack, int tid) {
  if (t.n == 0) return;
  const int D = t.depth;
  float worst = bound2;     // k-th best squared distance so far
  int worst_id = 0x7fffffff;
  auto leaf_scan = [&](uint32_t leaf_node) {
    const uint32_t kk = leaf_node - (1u << D);
    const uint32_t first = kd_bound(t.n, D, kk), end = kd_bound(t.n, D, kk + 1);
    for (uint32_t j0 = first; j0 < end; j0 += 4) {
      float4 p[4];
#pragma unroll
      for (int u = 0; u < 4; u++) p[u] = t.pts[min(j0 + u, end - 1)];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (j0 + u >= end) continue;
        const int id = static_cast<int>(j0 + u);
        const float dx = p[u].x - qx, dy = p[u].y - qy, dz = p[u].z - qz;
        const float d2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
        if (!(d2 < worst || (d2 == worst && id < worst_id))) continue;
        int loc = k - 1;
        for (; loc > 0; loc--) {
          const float pd = sd[(loc - 1) * STRIDE + tid];
          const int pi = si[(loc - 1) * STRIDE + tid];
          if (!(d2 < pd || (d2 == pd && (pi < 0 || id < pi)))) break;
          sd[loc * STRIDE + tid] = pd;
          si[loc * STRIDE + tid] = pi;
        }
        sd[loc * STRIDE + tid] = d2;
        si[loc * STRIDE + tid] = id;
        const int li = si[(k - 1) * STRIDE + tid];
        if (li >= 0) {  // list full: the bound tightens to the k-th best
          worst = sd[(k - 1) * STRIDE + tid];
          worst_id = li;
        }
      }
    }
  };
  int sp = 0, depth = 0;
  uint32_t node = 1;
  for (;;) {
    while (depth < D) {
      const float2 nd = t.nodes[node];
      const int axis = __float_as_int(nd.y);
      const float qa = axis == 0 ? qx : (axis == 1 ? qy : qz);
      const float diff = qa - nd.x;
      const float cut = diff * diff;
      depth++;
      if (cut <= worst) {
        stack[sp * STRIDE + tid] = kd_pack(cut, depth);
        sp++;
      }
      node = 2 * node + (diff < 0.f ? 0u : 1u);
    }
    leaf_scan(node);
    bool found = false;
    while (sp > 0) {
      sp--;
      const uint32_t e = stack[sp * STRIDE + tid];
      if (kd_cut(e) <= worst) {
        depth = static_cast<int>(e & 31u);
        node = (node >> (D - depth)) ^ 1u;
        found = true;
        break;
      }
    }
    if (!found) break;
  }
}

// ---- wave-cooperative search ---------------------------------------------------------------------------------------------------------
// The source cloud is Morton-sorted, so the 64 queries of a wave are neighbours in space and need (almost) the same leaves.
// Instead of 64 divergent tree walks the wave
//   A. seeds every lane with an upper bound on its nearest-neighbour distance: the distance to its hint (the neighbour of the
//      previous optimizer iteration) or, without a hint, to the best point of the leaf a plain root-to-leaf descent ends in;
//   B. takes the box of its queries inflated by the largest bound R and collects, breadth-first and with the whole wave working
//      on the frontier, every leaf whose cell can intersect that box (split-plane tests only);
//   C. stages those leaves' points in LDS, 256 at a time, and lets every lane scan all of them (broadcast ds_read_b128, no
//      divergence).  The true neighbour of lane l lies within its bound <= R of its query, hence inside the box, hence staged.
// Waves whose leaf list would overflow (huge R next to a dense surface) use the per-lane walk above instead.
constexpr int kWaveFrontier = 512;                 // nodes per BFS level / leaves per wave
constexpr int kWaveStage = 256;                    // points staged per pass
constexpr int kWaveLdsWords = 2 * kWaveFrontier + 4 * kWaveStage;  // 2 frontiers (u32) + staged float4s = 2048 words = 8 KB

__device__ __forceinline__ void kd_wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ float kd_wave_max(float v) {
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ float kd_wave_min(float v) {
  for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off));
  return v;
}

// All 64 lanes must call this; lds = this wave's private kWaveLdsWords words (16-byte aligned).
__device__ __forceinline__ KdBest kd_nearest_wave(const KdView& t, uint32_t* __restrict__ lds, float qx, float qy, float qz, float bound2, int hint, bool active, int lane) {
  KdBest best;
  best.d2 = bound2;
  best.idx = -1;
  best.x = best.y = best.z = 0.f;
  if (t.n == 0) return best;
  const int D = t.depth;
  auto consider = [&](const float4 p, int id) {
    const float dx = p.x - qx, dy = p.y - qy, dz = p.z - qz;
    const float d2 = fmaf(dx, dx, fmaf(dy, dy, dz * dz));
    if (d2 < best.d2 || (d2 == best.d2 && best.idx >= 0 && id < best.idx)) {
      best.d2 = d2;
      best.idx = id;
      best.x = p.x;
      best.y = p.y;
      best.z = p.z;
    }
  };
  // ---- A. seed ----
  const bool hinted = active && hint >= 0 && static_cast<uint32_t>(hint) < t.n;
  if (hinted) consider(t.pts[hint], hint);
  if (__any(active && !hinted)) {
    uint32_t node = 1;
    const bool go = active && !hinted;
    for (int d = 0; d < D; d++) {  // lockstep descent, no backtracking
      const float2 nd = t.nodes[go ? node : 1u];
      const int axis = __float_as_int(nd.y);
      const float qa = axis == 0 ? qx : (axis == 1 ? qy : qz);
      node = 2 * node + (qa - nd.x < 0.f ? 0u : 1u);
    }
    if (go) {
      const uint32_t k = node - (1u << D);
      const uint32_t first = kd_bound(t.n, D, k), end = kd_bound(t.n, D, k + 1);
      for (uint32_t i = first; i < end; i++) consider(t.pts[i], static_cast<int>(i));
    }
  }
  // ---- B. the wave's box and radius ----
  const float r2 = active ? best.d2 : 0.f;
  const float R2max = kd_wave_max(r2);
  // A few lanes with a distant neighbour (isolated points, outliers on their way to being rejected) must not inflate the box of
  // the whole wave: the cooperative pass serves the lanes below a robust radius — the smallest step of a x4 ladder that covers
  // at least 7/8 of the active lanes — and the rest finish with the per-lane walk, already bounded by their seed.
  float R2 = R2max;
  {
    const int nact = __popcll(__ballot(active));
    float step = 1.0e-4f;
    for (int it = 0; it < 12 && step < R2max; it++, step *= 4.f) {
      if (__popcll(__ballot(active && r2 <= step)) * 8 >= nact * 7) {
        R2 = step;
        break;
      }
    }
  }
  const bool far_lane = active && r2 > R2;
  bool per_lane = !(R2 < 3.0e38f);  // nothing bounded the search (unbounded query against a tree whose seed leaf was empty)
  int nleaves = 0;
  uint32_t* fa = lds;
  uint32_t* fb = lds + kWaveFrontier;
  if (!per_lane) {
    const float R = sqrtf(R2) * 1.0001f + 1e-6f;
    const float big = 3.0e38f;
    const bool inbox = active && !far_lane;  // far lanes do not stretch the boxes
    // Morton order jumps now and then: cut the wave at its three largest gaps between consecutive lanes into four contiguous
    // clusters and give each its own box, so that one jump does not blow a single box up to the size of the jump.
    int cluster = 0;
    {
      const float nx = __shfl_down(qx, 1), ny = __shfl_down(qy, 1), nz = __shfl_down(qz, 1);
      const bool nin = __shfl_down(inbox ? 1 : 0, 1) != 0;
      float gap = (lane < 63 && inbox && nin) ? (nx - qx) * (nx - qx) + (ny - qy) * (ny - qy) + (nz - qz) * (nz - qz) : 0.f;
      for (int c = 0; c < 3; c++) {
        const float g = kd_wave_max(gap);
        if (!(g > 0.f)) break;
        const int pos = __ffsll(static_cast<long long>(__ballot(gap == g))) - 1;  // cut between lanes pos and pos+1
        if (lane == pos) gap = 0.f;
        if (lane > pos) cluster++;
      }
    }
    float blo[4][3], bhi[4][3];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const bool in = inbox && cluster == c;
      blo[c][0] = kd_wave_min(in ? qx : big) - R;
      blo[c][1] = kd_wave_min(in ? qy : big) - R;
      blo[c][2] = kd_wave_min(in ? qz : big) - R;
      bhi[c][0] = kd_wave_max(in ? qx : -big) + R;
      bhi[c][1] = kd_wave_max(in ? qy : -big) + R;
      bhi[c][2] = kd_wave_max(in ? qz : -big) + R;
    }
    uint32_t alive0 = 0;
#pragma unroll
    for (int c = 0; c < 4; c++)
      if (blo[c][0] <= bhi[c][0]) alive0 |= 1u << c;  // empty clusters have lo = +big, hi = -big
    int nf = alive0 ? 1 : 0;
    if (lane == 0) fa[0] = 1u | (alive0 << 28);  // frontier entry = node | (mask of boxes still intersecting) << 28
    kd_wave_fence();
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int d = 0; d < D && !per_lane; d++) {
      int nout = 0;
      for (int base = 0; base < nf; base += 64) {
        const int i = base + lane;
        uint32_t ml4 = 0, mr4 = 0, node = 0;
        if (i < nf) {
          const uint32_t e = fa[i];
          node = e & 0x0fffffffu;
          const uint32_t alive = e >> 28;
          const float2 nd = t.nodes[node];
          const int axis = __float_as_int(nd.y);
#pragma unroll
          for (int c = 0; c < 4; c++) {
            const float lo = axis == 0 ? blo[c][0] : (axis == 1 ? blo[c][1] : blo[c][2]);
            const float hi = axis == 0 ? bhi[c][0] : (axis == 1 ? bhi[c][1] : bhi[c][2]);
            if ((alive >> c) & 1u) {
              if (lo <= nd.x) ml4 |= 1u << c;  // left cell: coordinates <= threshold
              if (hi >= nd.x) mr4 |= 1u << c;  // right cell: coordinates >= threshold
            }
          }
        }
        const bool gl = ml4 != 0, gr = mr4 != 0;
        const unsigned long long ml = __ballot(gl), mr = __ballot(gr);
        const int cl = __popcll(ml), cr = __popcll(mr);
        if (nout + cl + cr > kWaveFrontier) {
          per_lane = true;
          break;
        }
        if (gl) fb[nout + __popcll(ml & lt_mask)] = (2 * node) | (ml4 << 28);
        if (gr) fb[nout + cl + __popcll(mr & lt_mask)] = (2 * node + 1) | (mr4 << 28);
        nout += cl + cr;
      }
      kd_wave_fence();
      uint32_t* tmp = fa;
      fa = fb;
      fb = tmp;
      nf = nout;
    }
    nleaves = nf;
  }
  if (per_lane) {
    // overflow / unbounded: classic per-lane walk on this wave's LDS slice used as [level][lane] stack
    kd_wave_fence();
    const KdBest pl = kd_nearest<64>(t, qx, qy, qz, active ? best.d2 : bound2, -1, lds, lane, active);
    kd_wave_fence();
    if (active && pl.idx >= 0 && (pl.d2 < best.d2 || (pl.d2 == best.d2 && (best.idx < 0 || pl.idx < best.idx)))) best = pl;
    return best;
  }
  // ---- C. stage the collected leaves and scan them ----
  float4* stage = reinterpret_cast<float4*>(lds + 2 * kWaveFrontier);
  constexpr int kLeavesPerPass = kWaveStage / kKdLeafMax;  // 32
  for (int l0 = 0; l0 < nleaves; l0 += kLeavesPerPass) {
    const int nl = min(kLeavesPerPass, nleaves - l0);
    // lane handles staged slots lane, lane+64, lane+128, lane+192: slot s = leaf (s / 8), point (s % 8)
    float4 v[4];
    int ids[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int sidx = u * 64 + lane;
      const int lf = sidx / kKdLeafMax, pi = sidx % kKdLeafMax;
      ids[u] = -1;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lf < nl) {
        const uint32_t k = (fa[l0 + lf] & 0x0fffffffu) - (1u << D);
        const uint32_t first = kd_bound(t.n, D, k), end = kd_bound(t.n, D, k + 1);
        if (first + pi < end) {
          ids[u] = static_cast<int>(first + pi);
          v[u] = t.pts[first + pi];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      // empty slots are parked far away so that they never win
      stage[u * 64 + lane] = ids[u] >= 0 ? make_float4(v[u].x, v[u].y, v[u].z, __int_as_float(ids[u])) : make_float4(3.0e18f, 3.0e18f, 3.0e18f, __int_as_float(-1));
    }
    kd_wave_fence();
    if (active) {
      const int npts = nl * kKdLeafMax;
      for (int j = 0; j < npts; j += 4) {
        float4 p[4];
#pragma unroll
        for (int u = 0; u < 4; u++) p[u] = stage[j + u];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int id = __float_as_int(p[u].w);
          if (id >= 0) consider(p[u], id);
        }
      }
    }
    kd_wave_fence();
  }
  if (t.stats && lane == 0) {
    atomicAdd(&t.stats[4], static_cast<unsigned long long>(nleaves));
    atomicAdd(&t.stats[5], 1ull);
  }
  if (__any(far_lane)) {
    // the staged points may already have improved a far lane's bound; the walk only has to beat it
    kd_wave_fence();
    const KdBest pl = kd_nearest<64>(t, qx, qy, qz, best.d2, -1, lds, lane, far_lane);
    kd_wave_fence();
    if (far_lane && pl.idx >= 0 && (pl.d2 < best.d2 || (pl.d2 == best.d2 && (best.idx < 0 || pl.idx < best.idx)))) best = pl;
  }
  return best;
}

}  // namespace sga
