// k nearest neighbours of the tree's own points, ONE WAVE PER QUERY — the form for clouds that do not fill the chip.
//
// estimate_local_features (util/normal_estimation.hpp:65-92) asks for the k = 10 / 20 nearest neighbours of every point (itself
// included, ann/kdtree.hpp:172-176).  With one query per lane (kd_knn_own_points) a 11.5k-point LiDAR scan is 180 waves on 1024 SIMDs,
// each running a serial stream of ~60k instructions (every candidate any lane wants costs the whole wave a 20-slot insertion network):
// 100 us whatever the chip could do (profiles/r06_odom_stage_table.txt: the top kernel of a C5 scan).  Here the 64 lanes of a wave
// work on ONE query:
//   * the walk is wave-uniform (node numbers, bounds and the stack are scalars; boxes come through broadcast loads);
//   * a candidate batch is the <= 64 points under a node of depth D - 3 (eight leaves, contiguous in kd order): one coalesced load, one
//     distance per lane;
//   * the k-best list is a sorted list ACROSS the lanes (lane j holds the j-th nearest); the first batch — the query's own — is
//     sorted into it by a bitonic network, later candidates that beat the k-th distance are inserted one at a time (ballot, one
//     shift across the lanes);
//   * the siblings along the root path are tested against the k-th distance all at once (one lane per level, one round trip) and
//     opened nearest level first; inside a sibling the nearer child is followed, the farther one stacked with its box distance.
// A scan becomes 11.5k short waves (~1000 instructions each) that hide each other's latencies.  The neighbour sets are exact: a
// sub-tree is skipped only if its tight box lies farther than the k-th distance (kd_box_dist2 never exceeds the distance of a point
// inside, kd_search.hpp).  Ties at the k-th distance go to the candidate met first (a fixed order: deterministic).  Large clouds keep
// the one-query-per-lane kernel: with every SIMD busy anyway, 64 lanes per query cost more issue slots than they save.
#pragma once
#include "kd_search.hpp"

namespace sga {

constexpr int kKnnWaveBatchLevels = 3;  // a batch = the leaves under a node of depth D - 3: <= 8 x kKdLeafMax = 64 points

// ascending bitonic sort of (d, id) across the 64 lanes of a wave; equal keys keep no particular order (the callers break ties by id)
__device__ __forceinline__ void wave_sort64(float& d, int& id, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const float od = __shfl_xor(d, j);
      const int oi = __shfl_xor(id, j);
      const bool lower = (lane & j) == 0;            // this lane is the lower of the pair
      const bool asc = (lane & k) == 0 || k == 64;   // direction of this block
      const bool other_less = od < d || (od == d && oi < id);
      const bool take = (lower == asc) ? other_less : !other_less && !(od == d && oi == id);
      d = take ? od : d;
      id = take ? oi : id;
    }
  }
}

// One query: kd position `i` of tree `t` (wave-uniform).  On return lane j holds the j-th nearest (bd ascending, bid = kd position,
// -1 / +inf where the tree has fewer points).  stack: >= 2 * kKdMaxDepth words of LDS owned by this wave.
__device__ __forceinline__ float wave_read_f(float v, int lane_uniform) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane_uniform)); }

// k <= 64 (wave-uniform): the list lives across the lanes of one wave
__device__ __forceinline__ void knn_wave_query(const KdView& t, uint32_t i, int k, int lane, uint32_t* __restrict__ stack, float& bd, int& bid) {
  const float4 q = t.pts[i];
  const int D = t.depth;
  const int Db = D > kKnnWaveBatchLevels ? D - kKnnWaveBatchLevels : 0;
  bd = INFINITY;
  bid = -1;
  float tau = INFINITY;  // the k-th best distance so far (wave-uniform)

  // one batch: the points under node `nd` of depth Db, offered to the list
  bool first_batch = true;
  auto scan = [&](uint32_t nd) {
    const uint32_t kk = nd - (1u << Db);
    const uint32_t first = kd_bound(t.n, Db, kk), end = kd_bound(t.n, Db, kk + 1);
    const uint32_t pos = first + static_cast<uint32_t>(lane);
    float cd = INFINITY;
    int cid = -1;
    if (pos < end) {
      const float4 c = t.pts[pos];
      cd = kd_dist2(c.x, c.y, c.z, q.x, q.y, q.z);
      cid = static_cast<int>(pos);
      if (!(cd < INFINITY)) cd = INFINITY, cid = -1;  // a non-finite point is nobody's neighbour
    }
    if (first_batch) {  // wave-uniform: the list is empty, the batch becomes the list
      first_batch = false;
      wave_sort64(cd, cid, lane);
      bd = cd;
      bid = cid;
      tau = wave_read_f(bd, k - 1);
      return;
    }
    for (;;) {
      const unsigned long long m = __ballot(cd < tau);
      if (m == 0ull) break;
      const int src = __ffsll(static_cast<long long>(m)) - 1;  // candidates in lane (= kd) order
      const float v = wave_read_f(cd, src);
      const int vid = __builtin_amdgcn_readlane(cid, src);
      const int at = __popcll(__ballot(bd <= v));  // behind the entries that are not farther: equal distances keep their arrival order
      const float up_d = __shfl_up(bd, 1);
      const int up_i = __shfl_up(bid, 1);
      bd = lane < at ? bd : (lane == at ? v : up_d);
      bid = lane < at ? bid : (lane == at ? vid : up_i);
      tau = wave_read_f(bd, k - 1);
      if (lane == src) cd = INFINITY;
    }
  };

  // the query's own batch
  const double inv = t.n > 0 ? static_cast<double>(1u << Db) / static_cast<double>(t.n) : 0.0;
  const uint32_t own = __builtin_amdgcn_readfirstlane((1u << Db) + kd_leaf_rank(i, t.n, Db, inv));
  scan(own);
  if (Db == 0) return;
  // the siblings along the root path, their box distances all at once: lane l looks at the sibling of the path node of depth Db - l
  float sib_d = INFINITY;
  if (lane < Db) sib_d = kd_box_dist2(t, (own >> lane) ^ 1u, q.x, q.y, q.z);
  for (int l = 0; l < Db; l++) {  // nearest levels first: the deep siblings are the neighbours in space
    const float dl = wave_read_f(sib_d, l);
    if (!(dl <= tau)) continue;  // (tau only shrinks: skipped for good)
    // walk the sibling's sub-tree: follow the nearer child, stack the farther one with its box distance
    int sp = 0;
    uint32_t nd = (own >> l) ^ 1u;
    int depth = Db - l;
    for (;;) {
      while (depth < Db) {
        const uint32_t c0 = 2u * nd, c1 = c0 + 1u;
        // (every lane computes the same two numbers; read back from lane 0 they are scalars for the compiler too: the walk's control flow stays on the scalar unit)
        const float d0 = wave_read_f(kd_box_dist2(t, c0, q.x, q.y, q.z), 0), d1 = wave_read_f(kd_box_dist2(t, c1, q.x, q.y, q.z), 0);
        const bool left_first = d0 <= d1;
        const uint32_t near_n = left_first ? c0 : c1, far_n = left_first ? c1 : c0;
        const float near_d = left_first ? d0 : d1, far_d = left_first ? d1 : d0;
        depth++;
        if (far_d <= tau) {
          if (lane == 0) {
            stack[2 * sp] = far_n;
            stack[2 * sp + 1] = __float_as_uint(far_d);
          }
          sp++;
        }
        if (!(near_d <= tau)) {
          nd = 0u;  // nothing down here
          break;
        }
        nd = near_n;
      }
      if (nd != 0u) scan(nd);
      bool found = false;
      while (sp > 0) {
        sp--;
        __builtin_amdgcn_wave_barrier();
        const uint32_t pn = __builtin_amdgcn_readfirstlane(stack[2 * sp]);
        const float pd = __uint_as_float(__builtin_amdgcn_readfirstlane(stack[2 * sp + 1]));
        if (pd <= tau) {
          nd = pn;
          depth = 31 - __clz(nd);
          found = true;
          break;
        }
      }
      if (!found) break;
    }
  }
}

}  // namespace sga
