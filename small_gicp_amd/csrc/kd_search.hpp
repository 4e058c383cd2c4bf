// Exact nearest neighbour over an implicit, perfectly balanced kd-tree (the registration hot path's search; the GPU
// counterpart of ann/kdtree.hpp:80-126 build + :193-233 search in the reference tree /root/reference).  Device code, gfx950.
//
// Layout.  The target points are stored in kd order.  Node (depth d, rank k) owns the contiguous range
// [B(d,k), B(d,k+1)) with B(d,k) = floor(k * n / 2^d); it is split at m = B(d+1, 2k+1): left = [B(d,k), m), right = [m, B(d,k+1)),
// exactly the reference's median split (kdtree.hpp:118-126: left = [first, median), right = [median, last), threshold =
// coordinate of the median element).  Nodes are a 1-based heap (node = 2^d + k, children 2*node and 2*node+1) of
// {threshold, axis}: 8 bytes, no child pointers, no leaf records.  Leaves sit at the fixed depth D with <= 8 points.
// The same tree is stored a second time as 16-byte "pair" records for the 1-NN walk: one record per node of EVEN depth holding
// its own split and the splits of its two children, so that one dwordx4 load serves two levels of a descent (the walk is
// bound by the chain of dependent loads, not by bytes).
//
// Search = the reference's recursion (descend to the near side; visit the far side iff worst > cut^2, kdtree.hpp:207-230)
// unrolled onto an explicit per-lane stack kept in LDS as [level][lane] (conflict-free).  The far side is re-tested against
// the CURRENT best when it is popped, like the recursion does after the near side has returned.
// Ties (exactly equal distances): the candidate met first in the fixed traversal order wins, the lowest position inside a leaf —
// deterministic for a given tree and query; the reference's rule is traversal-order dependent as well (knn_result.hpp:81-83).
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace sga {

constexpr int kKdLeafMax = 8;    // points per leaf (<=); 8 x 16 B = one 128-byte line
constexpr int kKdMaxDepth = 24;  // stack slots per lane; tree depth D <= 24 (n <= 2^27)

struct KdView {
  const float4* __restrict__ pts;     // kd order, w = original index bits
  const float2* __restrict__ nodes;   // [2^D] heap: x = threshold, y = bitcast(axis)
  const float4* __restrict__ nodes4;  // pair records: x = own threshold, y / z = thresholds of the left / right child, w = axes (2 bits each)
  const float4* __restrict__ boxes;   // tight bounding boxes: [2 * node] = min corner, [2 * node + 1] = max corner
  const float4* __restrict__ groups;  // group headers: 8 float4 per node of depth gdepth — the boxes of its leaves (kd_visit_group)
  const float4* __restrict__ leafblk; // leaf blocks: 8 float4 per leaf = x[8], y[8], z[8], original index[8] (one 128-byte line; unused slots far away)
  uint32_t n;
  int depth;    // D; leaves are the 2^D ranges at depth D
  int gdepth;   // D - glevels: the 1-NN walk descends to this depth and handles the leaves below it as one group
  int glevels;  // min(2, D)
};

inline KdView make_kd_view(const sga_index* idx) {
  KdView k;
  k.pts = idx->kd_pts.p;
  k.nodes = idx->kd_nodes.p;
  k.nodes4 = idx->kd_nodes4.p;
  k.boxes = idx->kd_boxes.p;
  k.groups = idx->kd_groups.p;
  k.leafblk = idx->kd_leaf.p;
  k.n = static_cast<uint32_t>(idx->n);
  k.depth = idx->kd_depth;
  k.glevels = idx->kd_depth < 2 ? idx->kd_depth : 2;
  k.gdepth = idx->kd_depth - k.glevels;
  return k;
}

struct KdBest {
  float d2;
  int idx;     // position in the kd-ordered target, -1 = none
  int idx2;    // the runner-up (the second candidate of the warm pass's certificate), -1 = none
  float r2;    // exclusion bound: every target point other than `idx` and `idx2` has a computed squared distance >= r2 (see kd_nearest)
  int leaves;  // leaves scanned (a measure of the walk's length)
};

__host__ __device__ __forceinline__ uint32_t kd_bound(uint32_t n, int d, uint32_t k) { return static_cast<uint32_t>((static_cast<unsigned long long>(k) * n) >> d); }

__device__ __forceinline__ uint32_t kd_leaf_of(uint32_t i, uint32_t n, int d) {
  uint32_t k = static_cast<uint32_t>((static_cast<unsigned long long>(i) << d) / n);
  while (kd_bound(n, d, k + 1) <= i) k++;
  while (kd_bound(n, d, k) > i) k--;
  return k;
}

// Pair records are indexed by the heap number of their (even-depth) node: the record of node v sits at nodes4[v].  The slots of the
// odd depths stay unused (2^D records instead of two thirds of that) — the walk computes the address of a record with one shift
// instead of the eight integer instructions a dense level-by-level layout costs in every descent step.
__host__ __device__ __forceinline__ uint32_t kd_pair_index(int /*even_depth*/, uint32_t node) { return node; }
__host__ __device__ __forceinline__ uint32_t kd_pair_count(int D) { return 1u << D; }

// One stack entry per pending far side, 32 bits: [31:5] = cut^2 (float bits >> 4, i.e. rounded toward zero: conservative),
// [4:0] = depth of the far node.  The far node itself is implied: it is the sibling of the depth-`dd` ancestor of the leaf
// the walk currently stands on, so no node index has to be stored.
__device__ __forceinline__ uint32_t kd_pack(float cut, int depth) { return ((__float_as_uint(cut) >> 4) << 5) | static_cast<uint32_t>(depth); }
__device__ __forceinline__ float kd_cut(uint32_t e) { return __uint_as_float((e >> 5) << 4); }

// min of two floats in ONE instruction: fminf() is preceded by a canonicalisation of each operand (v_max_f32 x, x) under IEEE mode; the
// median of (a, b, -inf) is the same value for the numbers the walk deals in (distances >= 0, +inf) and needs no such step.
__device__ __forceinline__ float kd_min(float a, float b) { return __builtin_amdgcn_fmed3f(a, b, -INFINITY); }

// THE squared distance of the search: every kernel that compares distances of target points to a query (the walk, the
// certificate check of the warm pass) evaluates exactly this expression, so they agree bit for bit.
__device__ __forceinline__ float kd_dist2(float cx, float cy, float cz, float qx, float qy, float qz) {
  const float dx = cx - qx, dy = cy - qy, dz = cz - qz;
  return fmaf(dx, dx, fmaf(dy, dy, dz * dz));
}

// Squared distance from the query to the tight bounding box of `node`, evaluated with the same operations (and rounding) as a
// point distance: it never exceeds the distance to any point inside the box.  The split planes alone are a weak bound on
// surface-like data — a cell reaches far beyond the points it holds — so a pending far side that passed the plane test is opened
// only if its box can still hold a closer point.  (The reference prunes with the planes only, kdtree.hpp:224-230; the result is
// the same nearest neighbour, found in about half the rounds.)
__device__ __forceinline__ float kd_box_dist2(const KdView& t, uint32_t node, float qx, float qy, float qz) {
  const float4 lo = t.boxes[2 * node], hi = t.boxes[2 * node + 1];
  const float dx = fmaxf(fmaxf(lo.x - qx, qx - hi.x), 0.f);
  const float dy = fmaxf(fmaxf(lo.y - qy, qy - hi.y), 0.f);
  const float dz = fmaxf(fmaxf(lo.z - qz, qz - hi.z), 0.f);
  return fmaf(dx, dx, fmaf(dy, dy, dz * dz));
}

#ifdef SGA_KD_PREFETCH
// Experiment build (docs/experiments.md, round 6): a load nobody waits for, to pull a line the walk will want a few steps later into the
// L2 / L1 of this CU (bit 1: the box of a far side when it is pushed; bit 2: the four leaf blocks of a group with its header).  The
// destination register is threaded through the walk's state so that nothing else is allocated to it while loads are in flight.
__device__ __forceinline__ void kd_prefetch(const void* addr, uint32_t& pf) { asm volatile("global_load_dword %0, %1, off" : "+v"(pf) : "v"(addr) : "memory"); }
#endif

// ---- exact nearest neighbour ---------------------------------------------------------------------------------------------------
// The target point minimising (kd_dist2, kd position) lexicographically among the points with kd_dist2 < bound2 — a canonical
// rule (equidistant points: the lowest kd position wins), so the result depends on the tree and the query only, never on the
// traversal order, the seed or earlier calls.  Sub-trees at a distance EQUAL to the best are therefore still opened (they could
// hold an equidistant point of lower position); that costs nothing on real data.
//
// The walk keeps the two nearest points it has seen as 64-bit keys (distance bits, position), the distance of the third nearest
// (one v_med3 per scanned point) and the smallest lower bound (plane cut or box distance) of the sub-trees it discarded.  The
// minimum of the third distance and that bound is an EXCLUSION BOUND r2: every target point except the two candidates is at computed
// squared distance >= r2.  The warm linearization pass (linearize.hip) uses it as a certificate: after the query has moved by delta,
// the nearer of the two candidates is still the exact nearest neighbour if its new distance is below sqrt(r2) - delta.  (Two
// candidates rather than one: a query with two nearly equidistant neighbours — a few per cent of all queries at millimetre motions —
// would otherwise fail its certificate at every step.)
struct KdState {
  unsigned long long win;     // (distance bits << 32) | position of the nearest point seen (start: +inf, no point); distances are >= 0, so bit order = value order
  unsigned long long second;  // the runner-up, same encoding
  float third;                // distance of the third nearest
  float dropped;              // smallest lower bound of a discarded sub-tree
  float prune0;               // nothing at or beyond this distance can win (search bound, or the seed's distance + 1 ulp)
  float prune;                // = min(prune0, distance of the winner)
  float open;                 // sub-trees with a lower bound <= open are explored: prune, or (sqrt(prune) + slack)^2 (see kd_nearest)
  float slack;                // exploration margin in metres (0 = the minimal search)
  int leaves;                 // leaves scanned so far
#ifdef SGA_KD_STALE
  int groups;                 // pricing build (docs/experiments.md, deferred leaf scans): group visits completed
#endif
#ifdef SGA_KD_PREFETCH
  uint32_t pf;                // experiment build: destination of the walk's cache prefetches (never read)
#endif
#ifdef SGA_KD_TRIPS
  int own[8], wav[8];         // diagnostics build: loop-body executions of this lane / of the wave (counted by its first active lane)
#endif
};
#ifdef SGA_KD_TRIPS
// bodies: 0 uniform level, 1 pair step, 2 group header, 3 leaf scan, 4 pop iteration, 5 outer iteration
static __device__ unsigned long long g_kd_trips[16];
#define KD_TRIP(s, k)                                                                      \
  do {                                                                                     \
    (s).own[k]++;                                                                          \
    if (static_cast<int>(threadIdx.x & 63) == __ffsll(static_cast<long long>(__ballot(true))) - 1) (s).wav[k]++; \
  } while (0)
#else
#define KD_TRIP(s, k)
#endif

constexpr unsigned long long kKdNoPoint = (0x7f800000ull << 32) | 0xffffffffull;
__device__ __forceinline__ float kd_open_bound(float prune, float slack) {
  return slack > 0.f ? fmaf(slack, fmaf(2.f, sqrtf(prune), slack), prune) * 1.000001f : prune;  // (sqrt(prune) + slack)^2, rounded up
}
__device__ __forceinline__ KdState kd_state(float prune0, float slack = 0.f) {
  KdState s{};
  s.win = s.second = kKdNoPoint;
  s.third = s.dropped = INFINITY;
  s.prune0 = s.prune = prune0;
  s.open = kd_open_bound(prune0, slack);
  s.slack = slack;
  return s;
}
__device__ __forceinline__ float kd_key_dist(unsigned long long key) { return __uint_as_float(static_cast<uint32_t>(key >> 32)); }

__device__ __forceinline__ KdBest kd_result(const KdState& s, float bound2) {
  KdBest best;
  const float wd2 = kd_key_dist(s.win);
  const bool hit = wd2 < bound2;  // else: the nearest point seen lies beyond the search bound — no neighbour, and no candidates: the bound covers everything
  best.idx = hit ? static_cast<int>(static_cast<uint32_t>(s.win)) : -1;
  best.idx2 = (hit && s.second != kKdNoPoint) ? static_cast<int>(static_cast<uint32_t>(s.second)) : -1;
  best.d2 = hit ? wd2 : bound2;
  best.r2 = hit ? fminf(s.third, s.dropped) : fminf(wd2, s.dropped);
  best.leaves = s.leaves;
  return best;
}

// Leaf scan, branch-free: 8 slots are read unconditionally (the array is padded with 8 points at infinity behind the last
// leaf); the slots behind the leaf's own points belong to its right neighbour and are masked out, so that no point is ever
// scanned twice (it would become its own runner-up).  Per point: the distance, one v_med3 for the third distance, two 64-bit
// compares + selects for the two candidates.
__device__ __forceinline__ void kd_scan_leaf(const KdView& t, uint32_t leaf_node, float qx, float qy, float qz, KdState& s) {
  const uint32_t k = leaf_node - (1u << t.depth);
  const uint32_t first = kd_bound(t.n, t.depth, k);
  const uint32_t count = kd_bound(t.n, t.depth, k + 1) - first;
  const float4* __restrict__ lp = t.pts + first;
  float4 p[kKdLeafMax];
#pragma unroll
  for (int i = 0; i < kKdLeafMax; i++) p[i] = lp[i];
#pragma unroll
  for (int i = 0; i < kKdLeafMax; i++) {
    float d2 = kd_dist2(p[i].x, p[i].y, p[i].z, qx, qy, qz);
    d2 = static_cast<uint32_t>(i) < count ? d2 : INFINITY;
    s.third = __builtin_amdgcn_fmed3f(kd_key_dist(s.second), s.third, d2);  // third smallest of {second, third, new}: the runner-up's distance still the old one
    const unsigned long long key = (static_cast<unsigned long long>(__float_as_uint(d2)) << 32) | (first + static_cast<uint32_t>(i));
    const bool before_win = key < s.win;
    const unsigned long long low = key < s.second ? key : s.second;
    s.second = before_win ? s.win : low;
    s.win = before_win ? key : s.win;
  }
  s.prune = fminf(s.prune0, kd_key_dist(s.win));
#ifdef SGA_KD_STALE
  if (s.groups < SGA_KD_STALE)
#endif
  s.open = kd_open_bound(s.prune, s.slack);
  s.leaves++;
  KD_TRIP(s, 3);
#ifdef SGA_KD_STALE
  if (s.groups == 0) KD_TRIP(s, 6);  // leaf scans of the first group visit
#endif
}

// ---- the fast leaf scan ----------------------------------------------------------------------------------------------------------
// A leaf scan with 64-bit (distance, position) keys costs ~190 instructions and is the largest single item of a walk.  The fast
// state keeps the three nearest points as 32-bit keys instead: the bits of the squared distance with the low 3 bits replaced by the
// point's slot inside its leaf, so that inserting a point into the sorted triple is v_med3_u32, v_med3_u32, v_min_u32 and the
// eight distances of a leaf come from a structure-of-arrays leaf block with packed fp32 arithmetic (v_pk_*: two points per
// instruction, the same operations and rounding per point as kd_dist2).  ~70 instructions per leaf.
// Truncating 3 bits makes the ORDER of two points uncertain when their distances agree in all the remaining bits (relative
// difference < 1e-6).  That is detected at the end of the walk (kd_result: the winner and the runner-up share their truncated
// distance, or the winner sits within 8 ulps of the search bound) and such a query — a few per million on real data, every query
// on a lattice — is searched again with the exact 64-bit keys.  Everything else the walk decides with these keys is conservative:
// pruning uses the UPPER end of the winner's distance interval, the exclusion bound the LOWER end of the third's.
constexpr float kKdFar = 1e18f;             // coordinate of an unused leaf-block slot: squared distance ~3e36, finite
constexpr uint32_t kKdNoneKey = 0x7149f2cau;  // bits of 1e30f: keys at or above it are no points (unused slots, initial state)
struct KdFast {
  uint32_t w1, w2, w3;  // keys of the three nearest points seen, ascending
  uint32_t l1, l2;      // leaf ranks of the first two
  float dropped;        // smallest lower bound of a discarded sub-tree
  float prune0;         // nothing at or beyond this distance can win
  float open;           // sub-trees with a lower bound <= open are explored
  float slack;
  int leaves;
#ifdef SGA_KD_STALE
  int groups;
#endif
#ifdef SGA_KD_PREFETCH
  uint32_t pf;
#endif
#ifdef SGA_KD_TRIPS
  int own[8], wav[8];
#endif
};
__device__ __forceinline__ KdFast kd_fast_state(float prune0, float slack = 0.f) {
  KdFast s{};
  s.w1 = s.w2 = s.w3 = 0xffffffffu;
  s.l1 = s.l2 = 0u;
  s.dropped = INFINITY;
  s.prune0 = prune0;
  s.open = kd_open_bound(prune0, slack);
  s.slack = slack;
  return s;
}
__device__ __forceinline__ uint32_t kd_umed3(uint32_t a, uint32_t b, uint32_t c) { return max(min(a, b), min(max(a, b), c)); }  // v_med3_u32
__device__ __forceinline__ float kd_key_hi(uint32_t key) { return key < kKdNoneKey ? __uint_as_float(key | 7u) : INFINITY; }  // >= the point's distance
__device__ __forceinline__ float kd_key_lo(uint32_t key) { return key < kKdNoneKey ? __uint_as_float(key & ~7u) : INFINITY; }  // <= the point's distance

typedef float kd_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void kd_scan_leaf(const KdView& t, uint32_t leaf_node, float qx, float qy, float qz, KdFast& s) {
  const uint32_t k = leaf_node - (1u << t.depth);
  const float4* __restrict__ b = t.leafblk + 8ull * k;
  const float4 x0 = b[0], x1 = b[1], y0 = b[2], y1 = b[3], z0 = b[4], z1 = b[5];
  const kd_f32x2 X[4] = {{x0.x, x0.y}, {x0.z, x0.w}, {x1.x, x1.y}, {x1.z, x1.w}};
  const kd_f32x2 Y[4] = {{y0.x, y0.y}, {y0.z, y0.w}, {y1.x, y1.y}, {y1.z, y1.w}};
  const kd_f32x2 Z[4] = {{z0.x, z0.y}, {z0.z, z0.w}, {z1.x, z1.y}, {z1.z, z1.w}};
  const uint32_t o1 = s.w1, o2 = s.w2;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const kd_f32x2 dx = X[i] - qx, dy = Y[i] - qy, dz = Z[i] - qz;
    const kd_f32x2 d2 = __builtin_elementwise_fma(dx, dx, __builtin_elementwise_fma(dy, dy, dz * dz));  // kd_dist2, two points at a time
    const uint32_t k0 = (__float_as_uint(d2.x) & ~7u) | static_cast<uint32_t>(2 * i), k1 = (__float_as_uint(d2.y) & ~7u) | static_cast<uint32_t>(2 * i + 1);
    s.w3 = kd_umed3(s.w2, s.w3, k0);
    s.w2 = kd_umed3(s.w1, s.w2, k0);
    s.w1 = min(s.w1, k0);
    s.w3 = kd_umed3(s.w2, s.w3, k1);
    s.w2 = kd_umed3(s.w1, s.w2, k1);
    s.w1 = min(s.w1, k1);
  }
  // whose leaf: a key that changed came from this leaf, except a runner-up that is the old winner moved down
  s.l2 = s.w2 != o2 ? (s.w2 == o1 ? s.l1 : k) : s.l2;
  s.l1 = s.w1 != o1 ? k : s.l1;
#ifdef SGA_KD_STALE
  if (s.groups < SGA_KD_STALE)  // pricing build: the bound stops tightening after that many group visits (what a walk with deferred leaf scans would see)
#endif
  s.open = kd_open_bound(fminf(s.prune0, kd_key_hi(s.w1)), s.slack);
  s.leaves++;
  KD_TRIP(s, 3);
#ifdef SGA_KD_STALE
  if (s.groups == 0) KD_TRIP(s, 6);  // leaf scans of the first group visit
#endif
}

// The bottom of the walk.  The leaves under a node of depth D - 2 (a GROUP: up to 4 leaves, 32 points) are not reached through two
// more levels of splits, pushes, pops and box tests — each a dependent memory access — but through the group's header: the tight
// boxes of its leaves in one 128-byte line.  The lane computes the four box distances at once and scans the leaves that can hold a
// closer point, nearest box first (the bound tightens after every scan); the others are excluded with their box distance.
// (Measured on C3: a search scans 4.4 leaves on average — its own and the neighbours the ball reaches into; with the two-level
// bottom every one of them cost a pop, a box fetch, a record fetch and the leaf, one after the other.)
__device__ __forceinline__ float kd_box_dist2_vals(float lox, float loy, float loz, float hix, float hiy, float hiz, float qx, float qy, float qz) {
  const float dx = fmaxf(fmaxf(lox - qx, qx - hix), 0.f);
  const float dy = fmaxf(fmaxf(loy - qy, qy - hiy), 0.f);
  const float dz = fmaxf(fmaxf(loz - qz, qz - hiz), 0.f);
  return fmaf(dx, dx, fmaf(dy, dy, dz * dz));  // the same operations as kd_box_dist2
}

template <class S>
__device__ __forceinline__ void kd_visit_group(const KdView& t, uint32_t gnode, float qx, float qy, float qz, S& s) {
  const float4* __restrict__ h = t.groups + 8ull * (gnode - (1u << t.gdepth));
#ifdef SGA_KD_PREFETCH
  if (SGA_KD_PREFETCH & 2) {
    const float4* __restrict__ lb = t.leafblk + 8ull * ((gnode << t.glevels) - (1u << t.depth));
    for (int l = 0; l < (1 << t.glevels); l++) {
      kd_prefetch(lb + 8 * l, s.pf);
      kd_prefetch(lb + 8 * l + 4, s.pf);
    }
  }
#endif
  const float4 lox = h[0], loy = h[1], loz = h[2], hix = h[3], hiy = h[4], hiz = h[5];
  float lb0 = kd_box_dist2_vals(lox.x, loy.x, loz.x, hix.x, hiy.x, hiz.x, qx, qy, qz);
  float lb1 = kd_box_dist2_vals(lox.y, loy.y, loz.y, hix.y, hiy.y, hiz.y, qx, qy, qz);
  float lb2 = kd_box_dist2_vals(lox.z, loy.z, loz.z, hix.z, hiy.z, hiz.z, qx, qy, qz);
  float lb3 = kd_box_dist2_vals(lox.w, loy.w, loz.w, hix.w, hiy.w, hiz.w, qx, qy, qz);
  const uint32_t leaf0 = gnode << t.glevels;
  KD_TRIP(s, 2);
  for (int trip = 0; trip < 4; trip++) {
    float m = lb0;
    uint32_t l = 0;
    if (lb1 < m) m = lb1, l = 1;
    if (lb2 < m) m = lb2, l = 2;
    if (lb3 < m) m = lb3, l = 3;
    if (!(m <= s.open)) break;  // nothing left in this group that can hold a closer (or equidistant) point
    kd_scan_leaf(t, leaf0 + l, qx, qy, qz, s);
    lb0 = l == 0 ? INFINITY : lb0;
    lb1 = l == 1 ? INFINITY : lb1;
    lb2 = l == 2 ? INFINITY : lb2;
    lb3 = l == 3 ? INFINITY : lb3;
  }
  s.dropped = kd_min(s.dropped, fminf(fminf(lb0, lb1), fminf(lb2, lb3)));  // the leaves not scanned: nothing in there is closer than their box
#ifdef SGA_KD_STALE
  s.groups++;
#endif
}

// The reference's recursion (descend to the near side; visit the far side iff it can hold a closer point, kdtree.hpp:207-230)
// on an explicit stack, continued from `node` at `depth` (<= gdepth) with `sp` entries already pending, until the stack is empty.
// stack: LDS, gdepth * STRIDE words (STRIDE = threads per workgroup); this lane uses stack[level * STRIDE + tid].
template <int STRIDE, class S>
__device__ __forceinline__ void kd_walk(const KdView& t, float qx, float qy, float qz, S& s, uint32_t node, int depth, int sp, uint32_t* __restrict__ stack, int tid) {
  const int D = t.gdepth;
  for (;;) {
    KD_TRIP(s, 5);
    while (depth < D) {
      KD_TRIP(s, 1);
      // one pair record covers the node itself (if it is of even depth) and the child the walk continues into; it is fetched with
      // ONE 16-byte load per lane
      const int odd = depth & 1;
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      u32x4 raw;
      asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(raw) : "v"(t.nodes4 + kd_pair_index(depth - odd, node >> odd)) : "memory");
      const float4 nd = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w));
      const uint32_t axes = raw.w;
      if (!odd) {
        const uint32_t axis = axes & 3u;
        const float qa = axis == 0 ? qx : (axis == 1 ? qy : qz);
        const float diff = qa - nd.x;
        const float cut = diff * diff;
        depth++;
        // a far side beyond the best cannot hold a closer point: the entry is written unconditionally (no branch) and only
        // kept, i.e. the stack pointer advanced, if it can; a discarded one lowers the exclusion bound
        stack[sp * STRIDE + tid] = kd_pack(cut, depth);
        const bool keep = cut <= s.open;
        sp += keep ? 1 : 0;
        s.dropped = kd_min(s.dropped, keep ? INFINITY : cut);
        node = 2 * node + (diff < 0.f ? 0u : 1u);
#ifdef SGA_KD_PREFETCH
        if ((SGA_KD_PREFETCH & 1) && keep) kd_prefetch(t.boxes + 2 * (node ^ 1u), s.pf);
#endif
      }
      if (depth < D) {
        const uint32_t right = node & 1u;
        const uint32_t axis = (axes >> (2u + 2u * right)) & 3u;
        const float thr = right ? nd.z : nd.y;
        const float qa = axis == 0 ? qx : (axis == 1 ? qy : qz);
        const float diff = qa - thr;
        const float cut = diff * diff;
        depth++;
        stack[sp * STRIDE + tid] = kd_pack(cut, depth);
        const bool keep = cut <= s.open;
        sp += keep ? 1 : 0;
        s.dropped = kd_min(s.dropped, keep ? INFINITY : cut);
        node = 2 * node + (diff < 0.f ? 0u : 1u);
#ifdef SGA_KD_PREFETCH
        if ((SGA_KD_PREFETCH & 1) && keep) kd_prefetch(t.boxes + 2 * (node ^ 1u), s.pf);
#endif
      }
    }
    kd_visit_group(t, node, qx, qy, qz, s);
    // next pending far side that can still hold a closer (or equidistant) point: plane test on the stored cut, then the box test
    uint32_t e = 0;
    bool found = false;
    while (sp > 0 && !found) {
      KD_TRIP(s, 4);
      sp--;
      e = stack[sp * STRIDE + tid];
      float lb = kd_cut(e);
      if (lb <= s.open) {
        lb = fmaxf(lb, kd_box_dist2(t, (node >> (D - static_cast<int>(e & 31u))) ^ 1u, qx, qy, qz));
        found = lb <= s.open;
      }
      s.dropped = kd_min(s.dropped, found ? INFINITY : lb);  // discarded: nothing in there is closer than lb
    }
    if (!found) break;
    depth = static_cast<int>(e & 31u);
    node = (node >> (D - depth)) ^ 1u;  // sibling of the current group's ancestor at that depth
  }
#ifdef SGA_KD_PREFETCH
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(s.pf) : : "memory");
#endif
}

// ---- the walk in pieces, for the queue-fed search kernel (linearize.hip: nn_search_kernel) --------------------------------------
// A lane of that kernel holds one query at a time and takes the next one from a queue when it is done, so the walk is cut into the
// steps of one round: [start at a group | descend to a group] -> visit the group -> pop.  Same arithmetic, same canonical result as
// kd_walk.

// Position -> leaf rank at depth D: the largest k with B(D, k) <= i.  inv = 2^D / n in double: the estimate is off by one at most.
__device__ __forceinline__ uint32_t kd_leaf_rank(uint32_t i, uint32_t n, int d, double inv) {
  uint32_t k = min(static_cast<uint32_t>(static_cast<double>(i) * inv), (1u << d) - 1u);
  k += kd_bound(n, d, k + 1) <= i ? 1u : 0u;
  k -= kd_bound(n, d, k) > i ? 1u : 0u;
  return k;
}

// Group (node of depth gdepth) whose cell contains the query: the plain descent (no pending far sides), the top of it wave-uniform
// through the scalar cache while the lanes of the wave agree (tile mode: the 64 queries are neighbours).
__device__ __forceinline__ uint32_t kd_locate(const KdView& t, float qx, float qy, float qz) {
  const int D = t.gdepth;
  int depth = 0;
  uint32_t node = 1;
  const unsigned long long active = __ballot(true);
  while (depth < D) {
    const uint32_t un = __builtin_amdgcn_readfirstlane(node);
    const float2 nd = t.nodes[un];
    const int axis = __builtin_amdgcn_readfirstlane(__float_as_int(nd.y));
    const float thr = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(nd.x)));
    const float qa = axis == 0 ? qx : (axis == 1 ? qy : qz);
    const unsigned long long right = __ballot(!(qa - thr < 0.f));
    if (right != 0ull && right != active) break;
    depth++;
    node = 2 * un + (right != 0ull ? 1u : 0u);
  }
  while (depth < D) {
    const float2 nd = t.nodes[node];
    const int axis = __float_as_int(nd.y);
    const float qa = axis == 0 ? qx : (axis == 1 ? qy : qz);
    node = 2 * node + (qa - nd.x < 0.f ? 0u : 1u);
    depth++;
  }
  return node;
}

// The far sides along the root path of `leaf` (here: a GROUP node, depth gdepth, the lane has just visited), pushed top to bottom
// exactly as a descent to it would have pushed them — but the records of all ancestors are known from the node's index, so they are
// fetched with INDEPENDENT loads, one latency for the whole path (a descent pays one per two levels), and the plane tests already use
// the bound the visit has set.  Where the query lies on the other side of an ancestor's plane than the group (a seed next to the
// query's cell), the sibling is the query's own side: lower bound 0, always opened.
template <int STRIDE, int RECORDS, class S>
__device__ __forceinline__ void kd_push_path(const KdView& t, uint32_t leaf, int d0, float qx, float qy, float qz, S& s, int& sp, uint32_t* __restrict__ stack, int tid) {
  const int D = t.gdepth;
  float4 rec[RECORDS];
#pragma unroll
  for (int r = 0; r < RECORDS; r++) {
    const int d = d0 + 2 * r;
    rec[r] = t.nodes4[d < D ? kd_pair_index(d, leaf >> (D - d)) : 0u];
  }
#pragma unroll
  for (int r = 0; r < RECORDS; r++) {
    const int d = d0 + 2 * r;
    if (d < D) {  // wave-uniform
      const uint32_t axes = __float_as_uint(rec[r].w);
      {
        const uint32_t axis = axes & 3u;
        const float qa = axis == 0 ? qx : (axis == 1 ? qy : qz);
        const float diff = qa - rec[r].x;
        const uint32_t path_right = (leaf >> (D - d - 1)) & 1u;
        const float cut = ((diff < 0.f ? 0u : 1u) == path_right) ? diff * diff : 0.f;
        stack[sp * STRIDE + tid] = kd_pack(cut, d + 1);
        const bool keep = cut <= s.open;
        sp += keep ? 1 : 0;
        s.dropped = kd_min(s.dropped, keep ? INFINITY : cut);
      }
      if (d + 1 < D) {
        const uint32_t right = (leaf >> (D - d - 1)) & 1u;  // the child on the path
        const uint32_t axis = (axes >> (2u + 2u * right)) & 3u;
        const float thr = right ? rec[r].z : rec[r].y;
        const float qa = axis == 0 ? qx : (axis == 1 ? qy : qz);
        const float diff = qa - thr;
        const uint32_t path_right = (leaf >> (D - d - 2)) & 1u;
        const float cut = ((diff < 0.f ? 0u : 1u) == path_right) ? diff * diff : 0.f;
        stack[sp * STRIDE + tid] = kd_pack(cut, d + 2);
        const bool keep = cut <= s.open;
        sp += keep ? 1 : 0;
        s.dropped = kd_min(s.dropped, keep ? INFINITY : cut);
      }
    }
  }
}

// descend from (node, depth) to a group, pushing the far sides (the inner loop of kd_walk)
template <int STRIDE, class S>
__device__ __forceinline__ void kd_descend(const KdView& t, float qx, float qy, float qz, S& s, uint32_t& node, int& depth, int& sp, uint32_t* __restrict__ stack, int tid) {
  const int D = t.gdepth;
  while (depth < D) {
    const int odd = depth & 1;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 raw;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(raw) : "v"(t.nodes4 + kd_pair_index(depth - odd, node >> odd)) : "memory");
    const float4 nd = make_float4(__uint_as_float(raw.x), __uint_as_float(raw.y), __uint_as_float(raw.z), __uint_as_float(raw.w));
    const uint32_t axes = raw.w;
    if (!odd) {
      const uint32_t axis = axes & 3u;
      const float qa = axis == 0 ? qx : (axis == 1 ? qy : qz);
      const float diff = qa - nd.x;
      const float cut = diff * diff;
      depth++;
      stack[sp * STRIDE + tid] = kd_pack(cut, depth);
      const bool keep = cut <= s.open;
      sp += keep ? 1 : 0;
      s.dropped = kd_min(s.dropped, keep ? INFINITY : cut);
      node = 2 * node + (diff < 0.f ? 0u : 1u);
    }
    if (depth < D) {
      const uint32_t right = node & 1u;
      const uint32_t axis = (axes >> (2u + 2u * right)) & 3u;
      const float thr = right ? nd.z : nd.y;
      const float qa = axis == 0 ? qx : (axis == 1 ? qy : qz);
      const float diff = qa - thr;
      const float cut = diff * diff;
      depth++;
      stack[sp * STRIDE + tid] = kd_pack(cut, depth);
      const bool keep = cut <= s.open;
      sp += keep ? 1 : 0;
      s.dropped = kd_min(s.dropped, keep ? INFINITY : cut);
      node = 2 * node + (diff < 0.f ? 0u : 1u);
    }
  }
}

// next pending far side that can still hold a closer (or equidistant) point: false = the walk is over
template <int STRIDE, class S>
__device__ __forceinline__ bool kd_pop(const KdView& t, float qx, float qy, float qz, S& s, uint32_t& node, int& depth, int& sp, const uint32_t* __restrict__ stack, int tid) {
  const int D = t.gdepth;
  uint32_t e = 0;
  bool found = false;
  while (sp > 0 && !found) {
    sp--;
    e = stack[sp * STRIDE + tid];
    float lb = kd_cut(e);
    if (lb <= s.open) {
      lb = fmaxf(lb, kd_box_dist2(t, (node >> (D - static_cast<int>(e & 31u))) ^ 1u, qx, qy, qz));
      found = lb <= s.open;
    }
    s.dropped = kd_min(s.dropped, found ? INFINITY : lb);
  }
  if (found) {
    depth = static_cast<int>(e & 31u);
    node = (node >> (D - depth)) ^ 1u;
  }
  return found;
}

__device__ __forceinline__ float kd_next_up(float d2) { return __uint_as_float(__float_as_uint(d2) + 1u); }  // the next float above a finite d2 >= 0

// Top-down search (cold pass).
// bound2: only points with d2 < bound2 can win (pass max_sq nudged up by one ulp so that d2 == max_sq is still found).
// seed:   kd position of a target point believed to be close to the query (the neighbour found for this source point at the
//         previous pose) or -1.  Its distance (+ 1 ulp, so that the seed itself stays in reach) only tightens the pruning bound from
//         the first descent on; the seed is found again by the walk like any other point.
// slack:  exploration margin in metres.  0 = the minimal search: a sub-tree is opened only if it can hold a closer (or equidistant)
//         point.  > 0: everything within (nearest distance + slack) is explored as well — not needed for the answer, but it pushes
//         the exclusion bound out to that distance, so that the certificate of the warm pass survives a motion of ~slack / 2.
//         (The minimal search leaves, for a few per cent of the queries, an unexplored leaf just beyond the neighbour.)
template <int STRIDE>
__device__ __forceinline__ KdBest kd_nearest(const KdView& t, float qx, float qy, float qz, float bound2, int seed, uint32_t* __restrict__ stack, int tid, float slack = 0.f) {
  if (t.n == 0) return {bound2, -1, -1, INFINITY, 0};
  float prune0 = bound2;
  if (seed >= 0 && static_cast<uint32_t>(seed) < t.n) {
    const float4 c = t.pts[seed];
    const float d2 = kd_dist2(c.x, c.y, c.z, qx, qy, qz);
    prune0 = d2 < prune0 ? kd_next_up(d2) : prune0;
  }
  KdState s = kd_state(prune0, slack);
  const int D = t.gdepth;
  int sp = 0, depth = 0;
  uint32_t node = 1;
  // Wave-uniform top of the first descent.  The 64 queries of a wave are neighbours (the source is sorted by target leaf), so
  // they take the same branches for the first ~12 levels: as long as a ballot says so, the node is read ONCE per wave through the
  // scalar cache and each lane only evaluates its own plane distance (for the push test).  The first disagreement hands over to
  // the per-lane walk at the node reached.
  {
    const unsigned long long active = __ballot(true);
    while (depth < D) {
      KD_TRIP(s, 0);
      const uint32_t un = __builtin_amdgcn_readfirstlane(node);
      const float2 nd = t.nodes[un];  // uniform address: scalar load
      const int axis = __builtin_amdgcn_readfirstlane(__float_as_int(nd.y));
      const float thr = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(nd.x)));
      const float qa = axis == 0 ? qx : (axis == 1 ? qy : qz);
      const float diff = qa - thr;
      const unsigned long long right = __ballot(!(diff < 0.f));
      if (right != 0ull && right != active) break;  // the lanes part ways here
      const float cut = diff * diff;
      depth++;
      stack[sp * STRIDE + tid] = kd_pack(cut, depth);
      const bool keep = cut <= s.open;
      sp += keep ? 1 : 0;
      s.dropped = kd_min(s.dropped, keep ? INFINITY : cut);  // discarded at once: everything beyond this plane is at >= cut
      node = 2 * un + (right != 0ull ? 1u : 0u);
    }
  }
  kd_walk<STRIDE>(t, qx, qy, qz, s, node, depth, sp, stack, tid);
#if defined(SGA_KD_TRIPS) && !defined(SGA_KD_NO_COUNT)
  for (int k = 0; k < 8; k++) {
    atomicAdd(&g_kd_trips[k], static_cast<unsigned long long>(s.own[k]));
    atomicAdd(&g_kd_trips[8 + k], static_cast<unsigned long long>(s.wav[k]));
  }
#endif
  return kd_result(s, bound2);
}

// The same search with the fast leaf scan (KdFast).  `ambiguous` = the truncated keys cannot tell the winner from the runner-up, or
// the winner from the search bound: the caller repeats the search of that query with kd_nearest (exact keys).  When it is false the
// result equals kd_nearest's canonical neighbour; the runner-up and the exclusion bound are valid but may differ.
struct KdBestFast {
  KdBest best;
  bool ambiguous;
};
__device__ __forceinline__ KdBestFast kd_result(const KdView& t, const KdFast& s, float bound2) {
  KdBestFast r;
  const bool none = s.w1 >= kKdNoneKey;
  const float lo = kd_key_lo(s.w1), hi = kd_key_hi(s.w1);
  const bool hit = !none && hi < bound2;
  const bool miss = none ? bound2 <= 1e30f : !(lo < bound2);
  r.ambiguous = !(hit || miss) || (hit && ((s.w1 ^ s.w2) < 8u));
  r.best.idx = hit ? static_cast<int>(kd_bound(t.n, t.depth, s.l1) + (s.w1 & 7u)) : -1;
  r.best.idx2 = (hit && s.w2 < kKdNoneKey) ? static_cast<int>(kd_bound(t.n, t.depth, s.l2) + (s.w2 & 7u)) : -1;
  r.best.d2 = hit ? lo : bound2;
  r.best.r2 = hit ? fminf(kd_key_lo(s.w3), s.dropped) : fminf(lo, s.dropped);
  r.best.leaves = s.leaves;
  return r;
}

template <int STRIDE>
__device__ __forceinline__ KdBestFast kd_nearest_fast(const KdView& t, float qx, float qy, float qz, float bound2, int seed, uint32_t* __restrict__ stack, int tid, float slack = 0.f) {
  if (t.n == 0) return {{bound2, -1, -1, INFINITY, 0}, false};
  float prune0 = bound2;
  if (seed >= 0 && static_cast<uint32_t>(seed) < t.n) {
    const float4 c = t.pts[seed];
    const float d2 = kd_dist2(c.x, c.y, c.z, qx, qy, qz);
    prune0 = d2 < prune0 ? kd_next_up(d2) : prune0;
  }
  KdFast s = kd_fast_state(prune0, slack);
  const int D = t.gdepth;
  int sp = 0, depth = 0;
  uint32_t node = 1;
  {  // wave-uniform top of the first descent (see kd_nearest)
    const unsigned long long active = __ballot(true);
    while (depth < D) {
      KD_TRIP(s, 0);
      const uint32_t un = __builtin_amdgcn_readfirstlane(node);
      const float2 nd = t.nodes[un];
      const int axis = __builtin_amdgcn_readfirstlane(__float_as_int(nd.y));
      const float thr = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(nd.x)));
      const float qa = axis == 0 ? qx : (axis == 1 ? qy : qz);
      const float diff = qa - thr;
      const unsigned long long right = __ballot(!(diff < 0.f));
      if (right != 0ull && right != active) break;
      const float cut = diff * diff;
      depth++;
      stack[sp * STRIDE + tid] = kd_pack(cut, depth);
      const bool keep = cut <= s.open;
      sp += keep ? 1 : 0;
      s.dropped = kd_min(s.dropped, keep ? INFINITY : cut);
      node = 2 * un + (right != 0ull ? 1u : 0u);
    }
  }
  kd_walk<STRIDE>(t, qx, qy, qz, s, node, depth, sp, stack, tid);
#if defined(SGA_KD_TRIPS) && !defined(SGA_KD_NO_COUNT)
  for (int k = 0; k < 8; k++) {
    atomicAdd(&g_kd_trips[k], static_cast<unsigned long long>(s.own[k]));
    atomicAdd(&g_kd_trips[8 + k], static_cast<unsigned long long>(s.wav[k]));
  }
#endif
  return kd_result(t, s, bound2);
}


// ---- k nearest neighbours (traits::knn_search; normal / covariance estimation) -----------------------------------------------------
// Same walk as kd_nearest with the k-th best distance as the pruning bound.  The k-best list lives in LDS as [k][STRIDE]
// (lane-contiguous, conflict-free) and is kept UNSORTED while the walk runs: a better candidate replaces the current worst entry
// and the new worst is found by one sweep over the k slots — k independent LDS reads that pipeline, where a sorted insertion is
// a chain of up to k dependent read-compare-write steps (the old form of this function spent most of its time in that chain).
// The list is sorted once at the end, ascending by (distance, position): KnnResult<-1>::push (ann/knn_result.hpp:80-100) with a
// canonical tie rule.  sd / si must be initialised to +inf / -1 by the caller.  bound2: ignore points with d2 >= bound2.
// pre_first / pre_end / pre_pts (optional): a range of kd positions, staged in LDS by the caller, that is scanned up front, before
// the walk.  Queries that are themselves
// points of the tree (normal / covariance estimation) pass the positions around their own: kd order keeps spatial neighbours
// close, the 64 lanes of a wave read the same candidates (one broadcast load per candidate), and the list is all but final before
// the walk starts — which then runs with a tight bound, skips the leaves inside the range and mostly just proves that nothing
// closer is left.  (Without it every query pays ~k/8 walk rounds of dependent loads with the whole chip waiting on a few waves.)
template <int STRIDE>
__device__ __forceinline__ void kd_knn(const KdView& t, float qx, float qy, float qz, int k, float bound2, float* __restrict__ sd, int* __restrict__ si, uint32_t* __restrict__ stack, int tid, bool sort_result = true,
                                       uint32_t pre_first = 0, uint32_t pre_end = 0, const float4* __restrict__ pre_pts = nullptr /* LDS copy of pts[pre_first, pre_end) */,
                                       uint32_t pre_own_first = 0, uint32_t pre_own_end = 0 /* the part of the range to scan first */) {
  if (t.n == 0) return;
  const int D = t.depth;
  float worst = bound2;  // admission bound: the k-th best (distance, position) once the list is full
  int worst_id = 0x7fffffff;
  int worst_slot = 0, count = 0;
  auto scan_range = [&](uint32_t first, uint32_t end, bool skip_pre) {
    for (uint32_t j0 = first; j0 < end; j0 += 4) {
      float4 p[4];
#pragma unroll
      for (int u = 0; u < 4; u++) p[u] = skip_pre ? t.pts[min(j0 + u, end - 1)] : pre_pts[min(j0 + u, end - 1) - pre_first];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        if (j0 + u >= end) continue;
        if (skip_pre && j0 + u >= pre_first && j0 + u < pre_end) continue;  // scanned up front
        const int id = static_cast<int>(j0 + u);
        const float d2 = kd_dist2(p[u].x, p[u].y, p[u].z, qx, qy, qz);
        if (!(d2 < worst || (d2 == worst && id < worst_id))) continue;
        if (count < k) {  // filling up: any slot will do
          sd[count * STRIDE + tid] = d2;
          si[count * STRIDE + tid] = id;
          count++;
          if (count < k) continue;
        } else {
          sd[worst_slot * STRIDE + tid] = d2;
          si[worst_slot * STRIDE + tid] = id;
        }
        // the list is full: its worst entry is the new admission bound.  Four slots per step: eight LDS reads in flight, one wait
        // (slots k .. k+3 exist — the callers pad the list — and hold (+inf, -1) only until the list is full, never afterwards:
        // they are excluded by the j + u < k test)
        float wd = -1.f;
        int wi = -1, ws = 0;
        for (int j = 0; j < k; j += 4) {
          float pd[4];
          int pi[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            pd[u] = sd[(j + u) * STRIDE + tid];
            pi[u] = si[(j + u) * STRIDE + tid];
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const bool worse = j + u < k && (pd[u] > wd || (pd[u] == wd && pi[u] > wi));
            wd = worse ? pd[u] : wd;
            wi = worse ? pi[u] : wi;
            ws = worse ? j + u : ws;
          }
        }
        worst = wd;
        worst_id = wi;
        worst_slot = ws;
      }
    }
  };
  auto leaf_scan = [&](uint32_t leaf_node) {
    const uint32_t kk = leaf_node - (1u << D);
    const uint32_t first = kd_bound(t.n, D, kk), end = kd_bound(t.n, D, kk + 1);
    if (first >= pre_first && end <= pre_end) return;  // the whole leaf was scanned up front
    scan_range(first, end, true);
  };
  if (pre_end > pre_first && pre_pts != nullptr) {
    // nearest positions first (wave-uniform order): the wave's own stretch, then chunks alternately to its right and to its left —
    // the admission bound is tight after the first few dozen candidates and most of the rest fail the single compare
    const uint32_t own_first = min(max(pre_own_first, pre_first), pre_end), own_end = min(max(pre_own_end, own_first), pre_end);
    scan_range(own_first, own_end, false);
    for (uint32_t step = 0;; step++) {
      const uint32_t r0 = own_end + 16u * step, l1 = own_first >= 16u * step ? own_first - 16u * step : 0u;
      const bool right = r0 < pre_end, left = l1 > pre_first;
      if (!right && !left) break;
      if (right) scan_range(r0, min(r0 + 16u, pre_end), false);
      if (left) scan_range(l1 > pre_first + 16u ? l1 - 16u : pre_first, l1, false);
    }
  }
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
        const int dd = static_cast<int>(e & 31u);
        const uint32_t far_node = (node >> (D - dd)) ^ 1u;
        if (kd_box_dist2(t, far_node, qx, qy, qz) <= worst) {  // plane test, then the tight box of the pending sub-tree
          depth = dd;
          node = far_node;
          found = true;
          break;
        }
      }
    }
    if (!found) break;
  }
  if (!sort_result) return;
  // once, at the end: selection sort of the `count` entries, ascending by (distance, position); unused slots stay (+inf, -1)
  for (int a = 0; a + 1 < count; a++) {
    float bd = sd[a * STRIDE + tid];
    int bi = si[a * STRIDE + tid], bs = a;
    for (int j = a + 1; j < count; j++) {
      const float pd = sd[j * STRIDE + tid];
      const int pi = si[j * STRIDE + tid];
      const bool better = pd < bd || (pd == bd && pi < bi);
      bd = better ? pd : bd;
      bi = better ? pi : bi;
      bs = better ? j : bs;
    }
    if (bs != a) {
      sd[bs * STRIDE + tid] = sd[a * STRIDE + tid];
      si[bs * STRIDE + tid] = si[a * STRIDE + tid];
      sd[a * STRIDE + tid] = bd;
      si[a * STRIDE + tid] = bi;
    }
  }
}

// ---- k nearest neighbours of the tree's OWN points, k known at compile time (normal / covariance estimation, k = 10 / 20) -------
// The list lives in registers, sorted ascending; an insertion is a branch-free shift network (7 VALU operations per slot), so the 64
// lanes of a wave never serialise on each other's insertions the way the LDS list of kd_knn does (any lane inserting makes the whole
// wave run the insertion path: with k = 20 that path ran for nearly every candidate).  Candidates first come from `window`, an LDS
// copy of the kd positions [pre_first, pre_end) around the wave's own — kd order keeps spatial neighbours close and all lanes read
// the same candidate (a broadcast) — nearest stretch first; the walk that follows runs with the k-th distance found so far as its
// bound, skips what the window covered and mostly proves that nothing closer is left.  Distances tie-break by arrival order (a fixed
// order): deterministic; only the choice among equidistant k-th candidates can differ from kd_knn's canonical rule.
template <int K>
struct KnnRegs {
  float d[K];
  int id[K];
};

template <int K>
__device__ __forceinline__ void knn_regs_insert(KnnRegs<K>& L, float v, int vid) {
#pragma unroll
  for (int j = K - 1; j >= 1; j--) {
    const bool shift = v < L.d[j - 1];        // the new element lands before slot j - 1: that one moves up
    const bool here = !shift && v < L.d[j];   // it lands exactly here
    L.d[j] = shift ? L.d[j - 1] : (here ? v : L.d[j]);
    L.id[j] = shift ? L.id[j - 1] : (here ? vid : L.id[j]);
  }
  const bool first = v < L.d[0];
  L.d[0] = first ? v : L.d[0];
  L.id[0] = first ? vid : L.id[0];
}

template <int K, int STRIDE>
__device__ __forceinline__ void kd_knn_own_points_exact(const KdView& t, float qx, float qy, float qz, KnnRegs<K>& L, const float4* __restrict__ window, uint32_t pre_first, uint32_t pre_end, uint32_t own_first, uint32_t own_end,
                                                        uint32_t* __restrict__ stack, int tid) {
#pragma unroll
  for (int j = 0; j < K; j++) {
    L.d[j] = INFINITY;
    L.id[j] = -1;
  }
  if (t.n == 0) return;
  auto offer = [&](float4 c, int id, bool valid) {
    const float d2 = kd_dist2(c.x, c.y, c.z, qx, qy, qz);
    const bool take = valid && d2 < L.d[K - 1];
    if (__ballot(take) == 0ull) return;  // wave-uniform: nobody wants this candidate
    knn_regs_insert<K>(L, take ? d2 : INFINITY, id);
  };
  auto scan_window = [&](uint32_t first, uint32_t end) {
    for (uint32_t j = first; j < end; j++) offer(window[j - pre_first], static_cast<int>(j), true);
  };
  scan_window(own_first, own_end);
  for (uint32_t step = 0;; step++) {
    const uint32_t r0 = own_end + 16u * step, l1 = own_first >= 16u * step ? own_first - 16u * step : 0u;
    const bool right = r0 < pre_end, left = l1 > pre_first;
    if (!right && !left) break;
    if (right) scan_window(r0, min(r0 + 16u, pre_end));
    if (left) scan_window(l1 > pre_first + 16u ? l1 - 16u : pre_first, l1);
  }
  // the walk: kd_knn's, with the register list
  const int D = t.depth;
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
      if (cut <= L.d[K - 1]) {
        stack[sp * STRIDE + tid] = kd_pack(cut, depth);
        sp++;
      }
      node = 2 * node + (diff < 0.f ? 0u : 1u);
    }
    {
      const uint32_t kk = node - (1u << D);
      const uint32_t first = kd_bound(t.n, D, kk), end = kd_bound(t.n, D, kk + 1);
      if (!(first >= pre_first && end <= pre_end)) {  // not covered by the window
        float4 p[kKdLeafMax];
#pragma unroll
        for (int i = 0; i < kKdLeafMax; i++) p[i] = t.pts[first + i];  // 8 slots are always readable (padding behind the last leaf)
#pragma unroll
        for (int i = 0; i < kKdLeafMax; i++) {
          const uint32_t pos = first + i;
          const bool valid = pos < end && !(pos >= pre_first && pos < pre_end);
          const float d2 = kd_dist2(p[i].x, p[i].y, p[i].z, qx, qy, qz);
          if (valid && d2 < L.d[K - 1]) knn_regs_insert<K>(L, d2, static_cast<int>(pos));  // rare once the window has been scanned
        }
      }
    }
    bool found = false;
    while (sp > 0) {
      sp--;
      const uint32_t e = stack[sp * STRIDE + tid];
      if (kd_cut(e) <= L.d[K - 1]) {
        const int dd = static_cast<int>(e & 31u);
        const uint32_t far_node = (node >> (D - dd)) ^ 1u;
        if (kd_box_dist2(t, far_node, qx, qy, qz) <= L.d[K - 1]) {
          depth = dd;
          node = far_node;
          found = true;
          break;
        }
      }
    }
    if (!found) break;
  }
}

// The same search with the window phase on PACKED KEYS (round 6).  What made the function above slow is its insertion
// network: 7 VALU operations per list slot, run by the whole wave for every window candidate ANY lane wants — 128 candidates x 140
// operations, four fifths of the kernel.  A candidate of the window is known by its slot (7 bits); put into the low bits of the bits of
// its squared distance it makes a 32-bit key whose unsigned order is the order of the distances (>= 0) truncated to a multiple of 128
// ulps, and inserting a key into a sorted list is ONE v_med3_u32 per slot (new[j] = med3(old[j-1], old[j], x)) — unconditional, no
// ballot, no branch.  The list holds K + 1 keys: the K nearest and the best loser.
// Truncation can misorder two candidates only if their distances agree in the remaining 17 mantissa bits (relative difference < 1.6e-5);
// for the SET of the K nearest that matters only at the boundary, and it is detected there: the search ends AMBIGUOUS when the K-th
// entry and the best loser share their truncated distance — a few queries per ten thousand — and such a query runs the exact function
// above.  The walk behind the window continues on the truncated distances (new candidates are truncated the same way; the list stays
// sorted by them) and prunes with the UPPER end of the K-th entry's interval, so nothing that could belong to the set is skipped.
constexpr uint32_t kKnnSlotBits = 7, kKnnSlotMask = (1u << kKnnSlotBits) - 1u;  // kFeatWindow = 128 candidates

template <int K, int STRIDE>
__device__ __forceinline__ void kd_knn_own_points(const KdView& t, float qx, float qy, float qz, KnnRegs<K>& L, const float4* __restrict__ window, uint32_t pre_first, uint32_t pre_end, uint32_t own_first, uint32_t own_end,
                                                  uint32_t* __restrict__ stack, int tid) {
  if (t.n == 0) {
#pragma unroll
    for (int j = 0; j < K; j++) {
      L.d[j] = INFINITY;
      L.id[j] = -1;
    }
    return;
  }
  // ---- the window, packed keys
  uint32_t kk[K + 1];
#pragma unroll
  for (int j = 0; j <= K; j++) kk[j] = 0xffffffffu;
  for (uint32_t w = 0; w < pre_end - pre_first; w++) {  // (wave-uniform trip count: the candidate is a broadcast read)
    const float4 c = window[w];
    const float d2 = kd_dist2(c.x, c.y, c.z, qx, qy, qz);
    const uint32_t x = (__float_as_uint(d2) & ~kKnnSlotMask) | w;
#pragma unroll
    for (int j = K; j >= 1; j--) kk[j] = kd_umed3(kk[j - 1], kk[j], x);  // (the compiler folds the pattern into v_med3_u32)
    kk[0] = min(kk[0], x);
  }
  // ---- unpack: truncated distances (lower ends of their intervals) and kd positions; K + 1 entries
  KnnRegs<K + 1> M;
#pragma unroll
  for (int j = 0; j <= K; j++) {
    const bool some = (kk[j] & ~kKnnSlotMask) < 0x7f800000u;  // a finite distance: a real candidate
    M.d[j] = some ? __uint_as_float(kk[j] & ~kKnnSlotMask) : INFINITY;
    M.id[j] = some ? static_cast<int>(pre_first + (kk[j] & kKnnSlotMask)) : -1;
  }
  auto upper = [](float dt) { return dt < INFINITY ? __uint_as_float(__float_as_uint(dt) | kKnnSlotMask) : INFINITY; };  // the largest distance that truncates to dt
  // ---- the walk: kd_knn's, with the register list of truncated distances
  const int D = t.depth;
  int sp = 0, depth = 0;
  uint32_t node = 1;
  for (;;) {
    float bound = upper(M.d[K - 1]);
    while (depth < D) {
      const float2 nd = t.nodes[node];
      const int axis = __float_as_int(nd.y);
      const float qa = axis == 0 ? qx : (axis == 1 ? qy : qz);
      const float diff = qa - nd.x;
      const float cut = diff * diff;
      depth++;
      if (cut <= bound) {
        stack[sp * STRIDE + tid] = kd_pack(cut, depth);
        sp++;
      }
      node = 2 * node + (diff < 0.f ? 0u : 1u);
    }
    {
      const uint32_t kkk = node - (1u << D);
      const uint32_t first = kd_bound(t.n, D, kkk), end = kd_bound(t.n, D, kkk + 1);
      if (!(first >= pre_first && end <= pre_end)) {  // not covered by the window
        float4 p[kKdLeafMax];
#pragma unroll
        for (int i = 0; i < kKdLeafMax; i++) p[i] = t.pts[first + i];  // 8 slots are always readable (padding behind the last leaf)
#pragma unroll
        for (int i = 0; i < kKdLeafMax; i++) {
          const uint32_t pos = first + i;
          const bool valid = pos < end && !(pos >= pre_first && pos < pre_end);
          const float d2 = kd_dist2(p[i].x, p[i].y, p[i].z, qx, qy, qz);
          const float dt = __uint_as_float(__float_as_uint(d2) & ~kKnnSlotMask);
          if (valid && dt < M.d[K]) knn_regs_insert<K + 1>(M, dt, static_cast<int>(pos));  // (it beats the best loser at least) rare once the window has been scanned
        }
      }
    }
    bound = upper(M.d[K - 1]);
    bool found = false;
    while (sp > 0) {
      sp--;
      const uint32_t e = stack[sp * STRIDE + tid];
      if (kd_cut(e) <= bound) {
        const int dd = static_cast<int>(e & 31u);
        const uint32_t far_node = (node >> (D - dd)) ^ 1u;
        if (kd_box_dist2(t, far_node, qx, qy, qz) <= bound) {
          depth = dd;
          node = far_node;
          found = true;
          break;
        }
      }
    }
    if (!found) break;
  }
  // ---- the K-th entry and the best loser cannot be told apart: the exact search decides (a few queries per ten thousand)
  const bool ambiguous = M.id[K] >= 0 && M.d[K] == M.d[K - 1];
  if (ambiguous) {
    kd_knn_own_points_exact<K, STRIDE>(t, qx, qy, qz, L, window, pre_first, pre_end, own_first, own_end, stack, tid);
    return;
  }
#pragma unroll
  for (int j = 0; j < K; j++) {
    L.d[j] = M.d[j];
    L.id[j] = M.id[j];
  }
}

}  // namespace sga
