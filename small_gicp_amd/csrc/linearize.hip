// K1 (linearize) and K2 (error): the per-iteration hot loop of Registration<>::align on gfx950.
//
// Replaces ParallelReductionOMP::linearize / ::error (registration/reduction_omp.hpp:24-70) fused with
// {GICP,PointToPlaneICP,ICP}Factor::linearize / ::error (factors/*.hpp), RobustFactor (robust_kernel.hpp:70-98),
// DistanceRejector (rejector.hpp:19-28) and the nearest-neighbour search (ann/kdtree.hpp:193-233 or
// ann/incremental_voxelmap.hpp:99-119).  The 6x6 solve stays on the host (optimizer.hip).
//
// One linearization pass = ONE search + factor launch and reduce_rows_kernel:
//   search_linearize_kernel   one query per lane, one wave per workgroup; cold passes (full walk) and warm passes after larger
//                             motions (certificate check first); the wave then evaluates the factors of its own 64 points
//   nn_search_queue_kernel    warm passes after small motions: certificate check per tile, the few walks fed from a queue, the
//                             factors of the chunk (4 tiles) at the end
//   nn_search_kernel + linearize_kernel   the same in two launches: host-rejector callback, fp64 per-pair arithmetic;
//                             linearize_kernel alone for voxel-map targets (the lookup happens inside it)
// Factor stage in MOMENT form (accumulate_moments / derived_entry): 74 sums per pass that are both the normal equations and the
// coefficients of the quadratic error model sga_error answers from without a pass over the cloud; error_kernel (K2) only for robust
// factors, the async API and tests.  Sums: DPP inside a wave (fp32), fp64 across waves, one row of 96 doubles per tile / chunk /
// workgroup, added in fixed order by reduce_rows_kernel (no floating-point atomics: bit-reproducible).
#include <algorithm>
#include <chrono>
#include <thread>
#include <utility>

#include "common.hpp"
#include "device_math.hpp"
#include "cell_grid.hpp"
#include "kd_search.hpp"
#include "voxel_hash.hpp"

void sga_profile_collect_pending(sga_context* ctx);

namespace sga {

int comm_allreduce_sum(sga_context* ctx, double* d_buf, size_t count);
// cell_grid.hip: one exact search pass over the target's cell grid (nn / nn2 / rex of every source point), and the rings a reach needs
template <typename Real>
int grid_search_pass(sga_context* ctx, const sga_index* idx, const float4* src_pts, int n, const Rigid<Real>& T, float reach2, int* nn, int* nn2, float* rex, uint32_t* stats);
int grid_rings_for(const sga_index* idx, double reach);

#ifndef SGA_SPACING_REF
#define SGA_SPACING_REF 0.5486  // length scale (geometric mean leaf diagonal) of the C3 target (scripts/spacing_probe.py), the scene the routing thresholds were tuned on
#endif
constexpr int kTile = 256;           // threads per workgroup = source points per tile
constexpr int kRow = 96;             // doubles per partial row: [0, 29) the system (21 H, 6 b, e, inliers), [32, 95) the quadratic error model
constexpr int kCols = 128;           // columns the reduction kernels handle (>= kRow)
constexpr int kModelOff = 32;        // error model: [32, 41) sum p_a g_j, [41, 59) sum p_a M'_c, [59, 95) sum p_a p_b M'_c
constexpr int kModelCols = 95;
constexpr int kStatsCol = 30;        // spare columns 30, 31: search statistics of a grid pass (cell_grid.hip), not sums over points
constexpr int kSearchBlock = 64;      // search kernels: one wave per workgroup
#ifndef SGA_MAX_BLOCKS
#define SGA_MAX_BLOCKS 2048
#endif
constexpr int kMaxBlocks = SGA_MAX_BLOCKS;  // linearize_kernel / error_kernel / certify_linearize_kernel: 8 workgroups per CU

// Small grids (a 15k-point scan is 60 workgroups) fold the final reduction into the producer kernel: every workgroup publishes its
// partial row (agent-scope write-through stores), takes a ticket, and the workgroup that arrives last adds the rows in fixed order
// and hands the result over — one launch and one dependent-launch gap less per pass.  (With the 2048 workgroups of a 1M-point pass
// the ticket contention costs more than the launch: those keep the separate reduce_rows_kernel.)  Where the gain ends, measured late in
// round 6 on VGICP iterations of 3k ... 400k points (the workgroups of a streaming kernel finish together and take the ticket one after
// the other, an agent-scope acquire / release each): 12 workgroups -1.6 us per pass, 24 -0.6, 40 +1.1, 63 +3.4, 120 (30k points) +10,
// 235 (60k) +25, 245 (250k points at four per lane) +30 us — the limit was 256 since round 3.
constexpr int kFuseMaxBlocks = 32;
constexpr int kSeqWord = 128;  // h_accum: [0, 128) a result, word 128 the sequence number of the last published one
struct FusedTail {
  int enabled;
  unsigned* ticket;
  double* out;        // device result (out_n doubles)
  int out_n;
  double* host;       // pinned, device-mapped host result or null
  unsigned long long seq;
};

__host__ __device__ inline double derived_entry(int col, const double* m);
__host__ __device__ inline bool is_derived_col(int c);

// derive: the row holds a linearization in moment form — its derived columns are filled in from the totals (derived_entry)
__device__ __forceinline__ void fused_tail(const FusedTail& f, const double* __restrict__ partials, int nrows, int ncols, int row_stride, bool derive = false) {
  __shared__ unsigned sh_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this workgroup's row has left the CU
  __syncthreads();
  // agent-scope release (this workgroup's row, ordered before by the barrier) / acquire (the rows of the workgroups that arrived earlier)
  if (threadIdx.x == 0) sh_last = __hip_atomic_fetch_add(f.ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == static_cast<unsigned>(nrows - 1) ? 1u : 0u;
  __syncthreads();
  if (!sh_last) return;  // workgroup-uniform
  // 2 slices of 128 columns: slice s adds rows s, s + 2, ... (independent loads), then the slices are added in fixed order
  __shared__ double sh_slice[2][kCols];
  {
    const int c = threadIdx.x & (kCols - 1), sl = threadIdx.x / kCols;
    double t = 0.0;
    if (c < ncols && sl < 2)
      for (int r = sl; r < nrows; r += 2) t += __hip_atomic_load(&partials[static_cast<size_t>(r) * row_stride + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (sl < 2) sh_slice[sl][c] = t;
  }
  __syncthreads();
  if (derive) {  // workgroup-uniform
    if (threadIdx.x < kCols) sh_slice[0][threadIdx.x] += sh_slice[1][threadIdx.x];
    __syncthreads();
    if (threadIdx.x < kCols) sh_slice[1][threadIdx.x] = 0.0;
    __syncthreads();
  }
  if (threadIdx.x < kCols) {
    const int c = threadIdx.x;
    const double t = (derive && is_derived_col(c)) ? derived_entry(c, sh_slice[0]) : sh_slice[0][c] + sh_slice[1][c];
    if (c < f.out_n) {
      const double v = c < ncols ? t : 0.0;
      f.out[c] = v;
      if (f.host != nullptr) f.host[c] = v;
    }
  }
  if (threadIdx.x == 0) __hip_atomic_store(f.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch on this stream
  if (f.host != nullptr) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(f.host + kSeqWord), f.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

template <typename Real>
struct LinParams {
  const float4* __restrict__ src_pts;
  const Cov8* __restrict__ src_cov;
  int n;
  int num_tiles;
  const float4* __restrict__ tgt_pts;
  const float4* __restrict__ tgt_nrm;
  const Cov8* __restrict__ tgt_cov;
  KdView kd;
  VoxelView vox;
  FlatView flat;
  int* __restrict__ corr;
  const int* __restrict__ hint;  // exact nearest neighbour per source point at this pose (kd position) or -1, from nn_search_kernel
  const unsigned char* __restrict__ reject;  // optional: verdict of a host rejector per source point in the CALLER's order (1 = reject)
  Real* __restrict__ maha;  // n*6
  int store_maha;  // cache the mahalanobis matrices for the error kernel (robust factors); otherwise they are recomputed if ever asked for
  Rigid<Real> T;
  float max_sq;  // INFINITY = no rejector
  float bound2;  // a neighbour counts only if kd_dist2 < bound2 (max_sq nudged up by an ulp, or INFINITY); the walks reach a little farther
  int robust_kind;
  Real robust_c;
  double* __restrict__ partials;
  FusedTail tail;
  // warm pass with the certificate check inside the factor kernel (certify_linearize_kernel): the certificate of the previous
  // linearization pose T_prev is checked per point on the way through; a point whose certificate fails contributes nothing to the
  // streaming part, is flagged (rex[i] = -(exploration slack) < 0) and walks at the end of its workgroup's step
  int* __restrict__ cert_nn;
  int* __restrict__ cert_nn2;
  float* __restrict__ cert_rex;
  uint32_t* __restrict__ cert_walked;
  Rigid<Real> T_prev;
  float cert_within2, cert_slack_min, cert_slack_max;
  float cert_pad;  // headroom of the certificate check (see certify)
};

// XCD-aware tile schedule: workgroup b runs on XCD b % 8 (observed placement; used for L2 affinity only).  Each XCD
// gets one contiguous 1/8th of the (spatially sorted) tiles so that neighbouring tiles share an L2.
__device__ __forceinline__ void tile_schedule(int num_tiles, int& first, int& stride, int& end, int nblocks = 0) {
  if (nblocks == 0) nblocks = gridDim.x;
  if (nblocks % 8 == 0 && num_tiles >= nblocks) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd_blocks = nblocks >> 3;
    const int t0 = static_cast<int>((static_cast<long long>(num_tiles) * xcd) >> 3);
    const int t1 = static_cast<int>((static_cast<long long>(num_tiles) * (xcd + 1)) >> 3);
    first = t0 + slot;
    stride = per_xcd_blocks;
    end = t1;
  } else {
    first = blockIdx.x;
    stride = nblocks;
    end = num_tiles;
  }
}

template <typename Real>
__device__ __forceinline__ Sym3<Real> load_sym(const Cov8* __restrict__ c, int i) {
  const float4 a = reinterpret_cast<const float4*>(c)[2 * i];
  const float4 b = reinterpret_cast<const float4*>(c)[2 * i + 1];
  return {Real(a.x), Real(a.y), Real(a.z), Real(a.w), Real(b.x), Real(b.y)};
}

// exclusion radius stored per point: sqrt of the walk's exclusion bound, rounded down
__device__ __forceinline__ float rex_from_r2(float r2) { return sqrtf(r2) * 0.9999995f; }

// The walks search a little farther than the rejector reaches: a source point without a neighbour inside max_dist then carries the
// certificate "nothing within max_dist * (1 + margin)" and stays settled while it moves by less than the margin — without it those
// points (isolated clutter: the longest walks there are) would be searched again in every pass.
constexpr float kSearchMargin = 0.05f;

// The search: exact nearest neighbour of every transformed source point: nn[i] = its kd position (or -1 when nothing lies within the
// search bound), rex[i] = the exclusion radius the walk certifies (kd_search.hpp: every OTHER target point is farther than rex[i]).
//
// COLD pass (check = 0): the full walk for every point, seeded with the previous neighbour.
// WARM pass (check = 1): nn[] / rex[] describe the previous linearization pose T_prev.  The query has moved by
// delta = |T p - T_prev p| since; if the old neighbour is now closer than rex[i] - delta, no other point can be closer (triangle
// inequality): it is still the exact nearest neighbour, the lane shrinks the radius by delta and is done — no tree access at all.
// Only the lanes whose certificate fails walk (seeded with the old neighbour, which usually is the answer).  A small relative margin
// covers the rounding of the fp32 distances; it only ever sends a lane into the walk, never changes a result.  Late LM iterations
// move the points by micrometres: their passes are a stream over 40 bytes per point.
//
// One wave per workgroup (a finished wave frees its slot and its 4 KB of stack at once; the hardware dispatcher balances the uneven
// walks) at 7 - 8 waves per SIMD (72 / 64 VGPRs), which the walk needs.  The factor stage either follows inside the same wave
// (search_linearize_kernel: the moment form needs 42 VGPRs at one point per lane) or runs as linearize_kernel over nn[].
template <typename Real>
struct NNParams {
  const float4* __restrict__ src_pts;
  int n;
  KdView kd;
  Rigid<Real> T;
  float bound2;      // the walks find neighbours with kd_dist2 < bound2 (the rejector's reach + kSearchMargin)
  float within2;     // a neighbour counts for the rejector only if kd_dist2 < within2
  float slack_min, slack_max;  // exploration slack of a re-walk = clamp(motion, slack_min, slack_max) (SGA_SLACK_MIN / SGA_SLACK_MAX)
  float cert_pad;    // headroom of the certificate check as a share of the point's motion (see certify; SGA_CERT_PAD)
  int* __restrict__ nn;
  int* __restrict__ nn2;   // the runner-up of every walk: second candidate of the certificate
  float* __restrict__ rex;
  int check;         // warm pass
  Rigid<Real> T_prev;
  uint32_t* __restrict__ walked;  // statistics, one counter per wave tile: lanes of warm passes that had to walk
  const uint32_t* __restrict__ tile_order;  // launch slot -> tile (longest tile first), or null: slot = tile
  uint32_t* __restrict__ tile_cost;         // out, or null: duration of the tile's wave (100 MHz ticks)
  int* __restrict__ leaves;  // diagnostics (sga_problem_set_search_stats): leaves scanned per source point in this pass, or null
  double inv_leaf;   // 2^depth / n (kd_leaf_rank)
  int chunk_tiles;   // queue-fed kernel: tiles of 64 queries per wave
  int fast;          // one-query-per-lane kernels: walk with the fast leaf scan (exact repeat where it cannot decide)
  GridView grid;     // the target's cell grid (cell_grid.hpp), if grid_walk
  int grid_walk;     // the walkers of certify_linearize_kernel try ring 1 of the grid before they walk the tree
};

// Returns the shrunken radius (relative to the new pose) or a negative value if the certificate fails.
// Headroom (round 5): the DECISION asks for pad * moved metres more than the certificate needs, the radius handed on is the true one.
// A point that is sent into the walk although its certificate holds is found again exactly, so nothing but the number of walkers
// depends on it.  Why: the steps of an LM run shrink geometrically, and a certificate that survives this pass by less than the motion
// still to come fails in one of the LATE passes — where the streaming kernel pays for the tail of a single walk —
// whereas this pass walks thousands of points anyway and the re-walk (exploration slack = this pass's motion) buys a certificate that
// lasts.  Measured on C3 (profiles/r05_cert_pad.txt): walkers of the passes after 1.7 / 0.29 / 0.05 mm 14 021 / 2 625 / 413 -> 2 091 /
// 89 / 6 at pad 0.6, those passes 128 / 96 / 68 -> 92 / 78 / 62 us.  pad = 0: the plain check, bit for bit.
__device__ __forceinline__ float certify(float rex, float moved, bool has_neighbour, float d2_new, float within2, float pad) {
  const float lim = rex - moved * 1.000001f;
  const float lim_d = lim - moved * pad;
  const float lim2 = lim_d > 0.f ? lim_d * lim_d * 0.999995f : -1.f;
  const bool ok = has_neighbour ? d2_new < lim2 : lim2 > within2;  // no neighbour within reach before: still none
  return ok ? lim * 0.9999995f : -1.f;
}

// The walk of ONE source point, top-down and seeded, with the fast leaf scan (kd_search.hpp); the rare query it cannot decide (two
// candidates within 1e-6 of each other, or of the search bound) is searched again with the exact keys.  Stores nn / nn2 / rex.
// `tid` = the lane's column of kd_stack ([level][BLOCK] words): threadIdx.x in the kernels whose workgroup is BLOCK wide; a kernel that gives
// every wave of a wider workgroup its own stack passes the lane number (with threadIdx.x there, wave w's rows were shifted by w
// and a full stack of wave 3 reached past the allocation).
template <typename Real, int BLOCK>
__device__ __forceinline__ int walk_lane(const NNParams<Real>& p, int i, float fx, float fy, float fz, int seed, float slack, uint32_t* __restrict__ kd_stack, int tid) {
  KdBest nb{};
  bool exact = p.fast == 0;
  if (!exact) {
    const KdBestFast f = kd_nearest_fast<BLOCK>(p.kd, fx, fy, fz, p.bound2, seed, kd_stack, tid, slack);
    nb = f.best;
    exact = f.ambiguous;
  }
  if (exact) nb = kd_nearest<BLOCK>(p.kd, fx, fy, fz, p.bound2, seed, kd_stack, tid, slack);
  p.nn[i] = nb.idx;
  p.nn2[i] = nb.idx2;
  p.rex[i] = rex_from_r2(nb.r2);
  if (p.leaves != nullptr) p.leaves[i] = nb.leaves;
  return nb.idx;
}

// The search of ONE source point (a lane of a one-query-per-lane kernel): certificate check (warm) or walk, results stored to nn[] /
// nn2[] / rex[]; returns the neighbour's kd position or -1.  Called by every lane of the wave that holds a point (`i < n`), also
// under divergence: the wave-level steps inside kd_nearest only involve the lanes that walk.
template <typename Real, int BLOCK, bool CHECK>  // CHECK: warm pass (two instantiations: the cold one carries no certificate / margin state in registers)
__device__ __forceinline__ int search_lane(const NNParams<Real>& p, int tile, int i, const float4& ps, float fx, float fy, float fz, uint32_t* __restrict__ kd_stack) {
  int seed = p.nn[i];
  float slack = 0.f;
  if constexpr (CHECK) {
    Real ox, oy, oz;
    transform_point<Real>(p.T_prev, ps.x, ps.y, ps.z, ox, oy, oz);
    const float moved = sqrtf(kd_dist2(static_cast<float>(ox), static_cast<float>(oy), static_cast<float>(oz), fx, fy, fz));
    const int cand2 = p.nn2[i];
    float d1 = INFINITY, d2 = INFINITY;
    if (seed >= 0) {
      const float4 c = p.kd.pts[seed];
      d1 = kd_dist2(c.x, c.y, c.z, fx, fy, fz);
    }
    if (cand2 >= 0) {
      const float4 c = p.kd.pts[cand2];
      d2 = kd_dist2(c.x, c.y, c.z, fx, fy, fz);
    }
    // the nearer of the two candidates (the canonical rule again: equidistant -> lower position)
    const bool swap = cand2 >= 0 && (d2 < d1 || (d2 == d1 && cand2 < seed));
    const int best = swap ? cand2 : seed;
    const float r = certify(p.rex[i], moved, best >= 0, swap ? d2 : d1, p.within2, p.cert_pad);
    if (r >= 0.f) {
      p.rex[i] = r;
      if (swap) {
        p.nn[i] = cand2;
        p.nn2[i] = seed;
      }
      return best;
    }
    seed = best;
    // this point's certificate did not survive: walk again, and explore a margin around the new neighbour proportional to the motion,
    // so that the certificate survives the following (smaller) steps
    slack = fminf(fmaxf(moved, p.slack_min), p.slack_max);
    const unsigned long long walking = __ballot(true);
    if (threadIdx.x == __ffsll(static_cast<long long>(walking)) - 1) p.walked[tile] += static_cast<uint32_t>(__popcll(walking));  // the tile belongs to this wave: no atomic
    // A point whose certificate failed sits next to a surface (isolated points carry wide certificates and rarely fail): ring 1 of the
    // cell grid (cell_grid.hpp) settles it exactly with two dependent loads and gives the new certificate the tightest radius there is
    // (the third-nearest distance); the walk — with its exploration margin — only for what the ring does not settle.
    if (p.grid_walk & 2) {
      int g_nn, g_nn2;
      float g_rex, g_seen;
      bool g_face = false;
      // (a radius that ends at the ring's face with less than the re-walk's slack to spare is left to the walk: see certify_linearize_kernel)
      if (grid_ring1_lane(p.grid, fx, fy, fz, p.bound2, g_nn, g_nn2, g_rex, g_seen, &g_face) && !(g_face && g_nn >= 0 && (p.grid_walk & 32) && g_rex - g_seen < slack)) {
        p.nn[i] = g_nn;
        p.nn2[i] = g_nn2;
        p.rex[i] = g_rex;
        return g_nn;
      }
    }
  }
  return walk_lane<Real, BLOCK>(p, i, fx, fy, fz, seed, CHECK ? slack : 0.f, kd_stack, threadIdx.x);
}

// XCD-aware tile order: workgroup b runs on XCD b % 8 (observed placement; only speed depends on it), and the source is sorted by
// target leaf, so giving each XCD one contiguous eighth of the tiles makes its L2 hold one eighth of the target instead of all of it
__device__ __forceinline__ int search_tile_of_block(int nblk = 0) {
  if (nblk == 0) nblk = gridDim.x;
  const int per_xcd = nblk >> 3, b = blockIdx.x;
  return b < 8 * per_xcd ? (b & 7) * per_xcd + (b >> 3) : b;
}

#ifdef SGA_KD_TRIPS
static __device__ unsigned long long g_kd_wave_times[2 * 32768];  // diagnostics build: start / end (100 MHz wall clock) of every search wave
#endif

template <typename Real, int BLOCK, bool CHECK>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void nn_search_kernel(const NNParams<Real> p) {
  extern __shared__ uint32_t kd_stack[];  // tree depth x BLOCK traversal stack slots
#ifdef SGA_KD_TRIPS
  const unsigned long long wave_t0 = wall_clock64();
#endif
  const int tile = search_tile_of_block();
  const int i = tile * BLOCK + threadIdx.x;
  if (i >= p.n) return;
  const float4 ps = p.src_pts[i];
  Real x, y, z;
  transform_point<Real>(p.T, ps.x, ps.y, ps.z, x, y, z);
  search_lane<Real, BLOCK, CHECK>(p, tile, i, ps, static_cast<float>(x), static_cast<float>(y), static_cast<float>(z), kd_stack);
#ifdef SGA_KD_TRIPS
  if (blockIdx.x < 32768) {
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == __ffsll(static_cast<long long>(__ballot(true))) - 1) {
      g_kd_wave_times[2 * blockIdx.x] = wave_t0;
      g_kd_wave_times[2 * blockIdx.x + 1] = t1;
    }
  }
#endif
}

// The search, queue-fed.  The walks of neighbouring queries differ in length (one leaf for most, twenty for a few): with one query per
// lane for the lifetime of a wave, a wave runs as long as its longest walk and most lanes idle most of the time (measured: about a
// fifth of the VALU lanes active).  Here a wave owns a CHUNK of consecutive tiles and a small queue in LDS:
//   stage:  a tile of 64 queries is prepared by all 64 lanes at once (coalesced loads; in a warm pass the certificate check — lanes
//           whose certificate holds are done right there) and the queries that need a walk are appended to the queue, each with the
//           group (kd_search.hpp: the leaves under a node of depth D - 2) to start from: the one of its previous neighbour (or, for a
//           point without one, the group of its own cell:
//           remembered in nn2[] as -2 - rank, or located by a plain descent).
//   walk:   a lane without a query takes the next one from the queue.  Its first round visits the start group and fetches the records
//           of all ancestors of that leaf at once (kd_push_path: independent loads, one latency); later rounds are the usual
//           descend -> scan -> pop.  A lane whose walk is over writes its result and is refilled in the next round.
// The result is the canonical nearest neighbour (kd_search.hpp) whatever the order; only the exclusion radii may differ from the
// one-query-per-lane kernel's (both are valid bounds).
constexpr int kQueueCap = 128;        // entries; a tile is staged while at most kQueueCap - 64 are waiting
constexpr int kPathRecords = 10;      // pair records fetched at once by kd_push_path: covers depth 20 (8 M points); deeper trees take a second batch

template <typename Real, int FACTOR, int TARGET, int PTS, bool FRESH_NN = false, bool CERT = false, bool STAGED = false>
__device__ __forceinline__ void linearize_group(const LinParams<Real>& p, int first, int stride, int limit, double* __restrict__ acc_row, int lane, unsigned long long* __restrict__ failed_masks = nullptr);
template <typename Real, int FACTOR>
__device__ __forceinline__ bool pair_moments(const LinParams<Real>& p, int i, int j, bool within_bound, Real qx, Real qy, Real qz, Real tx, Real ty, Real tz, Sym3<Real>& Mp, Real* g, Real& e, Sym3<Real>& M_out);
template <typename Real, int PTS>
__device__ __forceinline__ void accumulate_moments(const Real (&P)[PTS][3], const Sym3<Real> (&Mp)[PTS], const Real (&G)[PTS][3], const Real (&E)[PTS], int inliers, double* __restrict__ acc_row, int lane);

// FACTOR >= 0: when the chunk's searches are done the wave also evaluates the factors of its chunk (four tiles at a time, like
// linearize_kernel) and writes ONE partial row per chunk — in the warm passes this kernel runs, the memory system and the VALUs are
// mostly idle, so the factor kernel's work all but disappears inside it (and its launch, start-up and tail with it).
template <typename Real, bool CHECK, int FACTOR = -1>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 6))) void nn_search_queue_kernel(const NNParams<Real> p, const LinParams<Real> lp) {
  extern __shared__ uint32_t kd_stack[];  // tree depth x 64 traversal stack slots
  __shared__ float4 q_pt[kQueueCap];      // query (x, y, z), w = exploration slack
  __shared__ int q_idx[kQueueCap];        // source point
  __shared__ uint32_t q_leaf[kQueueCap];  // start group (heap node of depth gdepth)
  const int lane = threadIdx.x;
  const int D = p.kd.gdepth;  // the walk's unit is the group (kd_search.hpp)
  const int num_tiles = (p.n + 63) >> 6;
  // XCD-aware chunk order (workgroup b runs on XCD b % 8): each XCD gets one contiguous eighth of the chunks
  const int nblk = static_cast<int>(gridDim.x), per_xcd = nblk >> 3, b = blockIdx.x;
  const int chunk = b < 8 * per_xcd ? (b & 7) * per_xcd + (b >> 3) : b;
  int tile = chunk * p.chunk_tiles;
  const int tile_end = min(tile + p.chunk_tiles, num_tiles);
  int q_head = 0, q_count = 0;  // wave-uniform
#ifdef SGA_KD_TRIPS
  const unsigned long long qt0 = wall_clock64();
  unsigned long long qt_stage = 0, qt_last = qt0;
#endif

  bool busy = false, fresh = false;
  float qx = 0.f, qy = 0.f, qz = 0.f;
  int qi = -1;
  uint32_t node = 1;
  int depth = 0, sp = 0;
  KdState s = kd_state(p.bound2, 0.f);

  for (;;) {
    // ---- stage tiles while the queue has room
    while (tile < tile_end && q_count <= kQueueCap - 64) {
      const int i = tile * 64 + lane;
      bool need = i < p.n;
      float fx = 0.f, fy = 0.f, fz = 0.f, slack = 0.f;
      int seed = -1, cand2 = -1;
      if (need) {
        const float4 ps = p.src_pts[i];
        Real x, y, z;
        transform_point<Real>(p.T, ps.x, ps.y, ps.z, x, y, z);
        fx = static_cast<float>(x), fy = static_cast<float>(y), fz = static_cast<float>(z);
        seed = p.nn[i];
        cand2 = p.nn2[i];
        if constexpr (CHECK) {
          Real ox, oy, oz;
          transform_point<Real>(p.T_prev, ps.x, ps.y, ps.z, ox, oy, oz);
          const float moved = sqrtf(kd_dist2(static_cast<float>(ox), static_cast<float>(oy), static_cast<float>(oz), fx, fy, fz));
          float d1 = INFINITY, d2 = INFINITY;
          if (seed >= 0) {
            const float4 c = p.kd.pts[seed];
            d1 = kd_dist2(c.x, c.y, c.z, fx, fy, fz);
          }
          if (cand2 >= 0) {
            const float4 c = p.kd.pts[cand2];
            d2 = kd_dist2(c.x, c.y, c.z, fx, fy, fz);
          }
          const bool swap = cand2 >= 0 && (d2 < d1 || (d2 == d1 && cand2 < seed));  // the canonical rule: equidistant -> lower position
          const int best = swap ? cand2 : seed;
          const float r = certify(p.rex[i], moved, best >= 0, swap ? d2 : d1, p.within2, p.cert_pad);
          if (r >= 0.f) {
            p.rex[i] = r;
            if (swap) {
              p.nn[i] = cand2;
              p.nn2[i] = seed;
            }
            need = false;
          }
          seed = best;
          slack = fminf(fmaxf(moved, p.slack_min), p.slack_max);
        }
      }
      // start group: the seed's; a point without a neighbour remembers the group of its cell; else locate it
      uint32_t leaf = 0;
      const bool seeded = need && seed >= 0, remembered = need && seed < 0 && cand2 <= -2;
      if (seeded) leaf = (1u << D) + (kd_leaf_rank(static_cast<uint32_t>(seed), p.kd.n, p.kd.depth, p.inv_leaf) >> p.kd.glevels);
      if (remembered) leaf = (1u << D) + min(static_cast<uint32_t>(-2 - cand2), (1u << D) - 1u);
      if (__ballot(need && !seeded && !remembered) != 0ull) {  // wave-uniform branch
        if (need && !seeded && !remembered) leaf = kd_locate(p.kd, fx, fy, fz);
      }
      const unsigned long long mask = __ballot(need);
      if (mask != 0ull) {
        if (need) {
          const int rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
          const int slot = (q_head + q_count + rank) & (kQueueCap - 1);
          q_pt[slot] = make_float4(fx, fy, fz, slack);
          q_idx[slot] = i;
          q_leaf[slot] = leaf;
        }
        const int added = __popcll(mask);
        if constexpr (CHECK) {
          if (lane == 0) p.walked[tile] += static_cast<uint32_t>(added);  // the tile belongs to this wave: no atomic
        }
        q_count += added;
      }
      tile++;
    }
#ifdef SGA_KD_TRIPS
    {
      const unsigned long long now = wall_clock64();
      qt_stage += now - qt_last;  // (the first iteration: the staging of the whole chunk; later ones: refills)
      qt_last = now;
    }
#endif
    // ---- lanes without a query take the next ones
    const unsigned long long idle = __ballot(!busy);
    if (q_count > 0 && idle != 0ull) {
      const int rank = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(idle >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(idle), 0u));
      if (!busy && rank < q_count) {
        const int slot = (q_head + rank) & (kQueueCap - 1);
        const float4 e = q_pt[slot];
        qx = e.x, qy = e.y, qz = e.z;
        qi = q_idx[slot];
        node = q_leaf[slot];
        depth = D;
        sp = 0;
        s = kd_state(p.bound2, CHECK ? e.w : 0.f);
        busy = true;
        fresh = true;
      }
      const int taken = min(__popcll(idle), q_count);
      q_head = (q_head + taken) & (kQueueCap - 1);
      q_count -= taken;
      if constexpr (CHECK) {
        // the walkers of a warm pass sit next to a surface: ring 1 of the cell grid settles nearly all of them with two dependent loads
        // (see search_lane); a lane whose query the ring settles is free again for the next one
        if (p.grid_walk & 4) {  // wave-uniform
          const bool trying = busy && fresh;
          if (__ballot(trying) != 0ull) {
            if (trying) {
              int g_nn, g_nn2;
              float g_rex, g_seen;
              bool g_face = false;
              // (a radius that ends at the ring's face with less than the walk's slack to spare is left to the walk: see certify_linearize_kernel)
              if (grid_ring1_lane(p.grid, qx, qy, qz, p.bound2, g_nn, g_nn2, g_rex, g_seen, &g_face) && !(g_face && g_nn >= 0 && (p.grid_walk & 32) && g_rex - g_seen < s.slack)) {
                p.nn[qi] = g_nn;
                p.nn2[qi] = g_nn2;
                p.rex[qi] = g_rex;
                busy = false;
                fresh = false;
              }
            }
          }
        }
      }
    }
    if (__ballot(busy) == 0ull) {
      if (tile >= tile_end && q_count == 0) break;  // (the ring may have settled every query just taken while more are queued)
      continue;
    }
    // ---- one round of the walk
    if (busy) {
      kd_descend<64>(p.kd, qx, qy, qz, s, node, depth, sp, kd_stack, lane);  // nothing to do for a fresh lane: it stands on its start group
      kd_visit_group(p.kd, node, qx, qy, qz, s);
    }
    if (__ballot(fresh) != 0ull) {  // wave-uniform
      if (fresh) {
        kd_push_path<64, kPathRecords>(p.kd, node, 0, qx, qy, qz, s, sp, kd_stack, lane);
        if (D > 2 * kPathRecords) kd_push_path<64, kKdMaxDepth / 2 - kPathRecords>(p.kd, node, 2 * kPathRecords, qx, qy, qz, s, sp, kd_stack, lane);
        fresh = false;
      }
    }
    if (busy) {
      if (!kd_pop<64>(p.kd, qx, qy, qz, s, node, depth, sp, kd_stack, lane)) {
        const KdBest nb = kd_result(s, p.bound2);
        p.nn[qi] = nb.idx;
        // no neighbour within reach: remember the group the walk ended on instead of a runner-up (the next pass starts there)
        p.nn2[qi] = nb.idx >= 0 ? nb.idx2 : -2 - static_cast<int>(node - (1u << D));
        p.rex[qi] = rex_from_r2(nb.r2);
        busy = false;
      }
    }
  }
#ifdef SGA_KD_TRIPS
  const unsigned long long qt1 = wall_clock64();
#endif
  if constexpr (FACTOR >= 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the chunk's neighbours are written
    __syncthreads();                                   // one wave: the stacks are free
    double* row = reinterpret_cast<double*>(kd_stack);
    for (int c = lane; c < kRow; c += 64) row[c] = 0.0;
    __syncthreads();
    const int limit = min(p.n, tile_end * 64);
    if (p.chunk_tiles == 1)  // (small clouds) one point per lane: no arithmetic on three absent tiles
      linearize_group<Real, FACTOR, 0, 1, true>(lp, chunk * 64 + lane, 64, limit, row, lane);
    else
      for (int t0 = chunk * p.chunk_tiles; t0 < tile_end; t0 += 4) linearize_group<Real, FACTOR, 0, 4, true>(lp, t0 * 64 + lane, 64, limit, row, lane);
    __syncthreads();
    for (int c = lane; c < kRow; c += 64) lp.partials[static_cast<size_t>(chunk) * kRow + c] = row[c];
  }
#ifdef SGA_KD_TRIPS
  if (lane == 0) {  // [12] staging, [13] walks, [14] factor stage, [15] waves; [11] latest end - earliest start is derived from the wave times
    const unsigned long long qt2 = wall_clock64();
    atomicAdd(&g_kd_trips[12], qt_stage);
    atomicAdd(&g_kd_trips[13], (qt1 - qt0) - qt_stage);
    atomicAdd(&g_kd_trips[14], qt2 - qt1);
    atomicAdd(&g_kd_trips[15], 1ull);
    if (blockIdx.x < 32768) {
      g_kd_wave_times[2 * blockIdx.x] = qt0;
      g_kd_wave_times[2 * blockIdx.x + 1] = qt2;
    }
  }
#endif
}

// One correspondence (source point i at q = T p, target candidate j at t): rejector, fused mahalanobis, robust weight, the 28
// values of the pair's system.  Returns whether the pair is an inlier; caches the mahalanobis (GICP).
template <typename Real, int FACTOR>
__device__ __forceinline__ bool pair_factor(const LinParams<Real>& p, int i, int j, bool within_bound, Real px, Real py, Real pz, Real qx, Real qy, Real qz, Real tx, Real ty, Real tz, Real* vals, Sym3<Real>* Mp_out = nullptr,
                                            Real* g_out = nullptr) {
  const Real rx = tx - qx, ry = ty - qy, rz = tz - qz;
  const Real d2 = rx * rx + ry * ry + rz * rz;
  const bool inlier = (j >= 0) && within_bound && !(d2 > static_cast<Real>(p.max_sq));
  if (inlier) {
    Sym3<Real> M;
    if constexpr (FACTOR == SGA_GICP) {
      const Sym3<Real> Cs = load_sym<Real>(p.src_cov, i);
      const Sym3<Real> Ct = load_sym<Real>(p.tgt_cov, j);
      const Sym3<Real> RCR = rotate_sym(p.T.r, Cs);
      M = inverse_sym<Real>({Ct.xx + RCR.xx, Ct.xy + RCR.xy, Ct.xz + RCR.xz, Ct.yy + RCR.yy, Ct.yz + RCR.yz, Ct.zz + RCR.zz});
      Real* m = p.maha + static_cast<size_t>(i) * 6;
      m[0] = M.xx;
      m[1] = M.xy;
      m[2] = M.xz;
      m[3] = M.yy;
      m[4] = M.yz;
      m[5] = M.zz;
    } else if constexpr (FACTOR == SGA_PLANE_ICP) {
      const float4 nn = p.tgt_nrm[j];
      M = {Real(nn.x) * Real(nn.x), Real(0), Real(0), Real(nn.y) * Real(nn.y), Real(0), Real(nn.z) * Real(nn.z)};
    } else {
      M = {Real(1), Real(0), Real(0), Real(1), Real(0), Real(1)};
    }
    Real w = Real(1);
    if (p.robust_kind != SGA_ROBUST_NONE) {
      const Real vx = M.xx * rx + M.xy * ry + M.xz * rz, vy = M.xy * rx + M.yy * ry + M.yz * rz, vz = M.xz * rx + M.yz * ry + M.zz * rz;
      w = robust_weight<Real>(p.robust_kind, p.robust_c, Real(0.5) * (rx * vx + ry * vy + rz * vz));
    }
    pair_system<Real>(p.T.r, px, py, pz, rx, ry, rz, M, w, vals, Mp_out, g_out);
  }
  return inlier;
}

// The sums of one linearization, in MOMENT form.  With M' = R^T M R and g = R^T M r of a pair (source frame, device_math.hpp) and
// p = the source point, everything the optimizer needs is linear in
//   sum M'            sum g            sum p_a M'          sum p_a g          sum p_a p_b M'          sum e, #inliers
//   (6 = H_tt)        (3 = -b_t)       (18)                (9)                (36)
// because H_rt = sum skew(p) M', H_rr = sum skew(p) M' skew(p)^T and b_r = -sum skew(p) g only recombine those (derived_entry,
// evaluated once per pass by the reducing workgroup).  The same sums ARE the coefficients of the quadratic ERROR MODEL:
// Reduction::error (reduction_omp.hpp:61-70) evaluates sum_i 1/2 r_i^T M_i r_i at a trial pose with the correspondences and
// mahalanobis matrices CACHED by the last linearization (gicp_factor.hpp:80-89); with those frozen it is a quadratic polynomial in
// Y = [R^T R_n - I | R^T (tau_n - tau)] (trial pose (R_n, tau_n) relative to the linearization pose (R, tau)):
//   e(T_n) = e0 - sum_a Y[:,a] . S1[a] + 1/2 sum_ab Y[:,a]^T S2[a][b] Y[:,b],   S1[a] = sum p~_a g,  S2[a][b] = sum p~_a p~_b M'
// with p~ = (p, 1).  sga_error evaluates it on the host: no pass over the cloud, no device round trip.
// 74 sums instead of the 29 of the direct form, but a lane adds up PTS points before a value goes through the wave reduction (the
// DPP chain is what a sum costs): 74 * (PTS + 6) / PTS instructions per point instead of 29 * 7 + the skew products.
// Row layout: [0, 21) H, [21, 27) b, 27 e, 28 inliers, [32, 41) sum p_a g_j, [41, 59) sum p_a M'_c, [59, 95) sum p_a p_b M'_c
// (c = xx, xy, xz, yy, yz, zz; pairs ab = 00, 01, 02, 11, 12, 22); columns [0, 15) and [21, 24) are derived.
__host__ __device__ inline double derived_entry(int col, const double* m) {
  const int S[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
  auto A = [&](int a, int j, int k) { return m[kModelOff + 9 + 6 * a + S[j][k]]; };                   // sum p_a M'_jk
  auto B = [&](int a, int b, int j, int k) { return m[kModelOff + 27 + 6 * S[a][b] + S[j][k]]; };     // sum p_a p_b M'_jk
  auto G = [&](int a, int j) { return m[kModelOff + 3 * a + j]; };                                    // sum p_a g_j
  // K = skew(p) M': K_ik = p_i1 M'_i2,k - p_i2 M'_i1,k  (i1 = i + 1, i2 = i + 2 mod 3)
  auto PK = [&](int l, int i, int k) { const int i1 = (i + 1) % 3, i2 = (i + 2) % 3; return B(l, i1, i2, k) - B(l, i2, i1, k); };  // sum p_l K_ik
  if (col >= 21) {  // b_r = -sum p x g
    const int i = col - 21, i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    return G(i2, i1) - G(i1, i2);
  }
  // upper triangle of H, row-wise: row i starts at 6 i - i (i - 1) / 2
  const int i = col < 6 ? 0 : (col < 11 ? 1 : 2);
  const int j = col - (6 * i - i * (i - 1) / 2) + i;  // column in the 6x6
  if (j >= 3) {  // H_rt = sum K
    const int k = j - 3, i1 = (i + 1) % 3, i2 = (i + 2) % 3;
    return A(i1, i2, k) - A(i2, i1, k);
  }
  // H_rr = sum K skew(p)^T: [i][j] = p_j1 K_i,j2 - p_j2 K_i,j1
  const int j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return PK(j1, i, j2) - PK(j2, i, j1);
}
__host__ __device__ inline bool is_derived_col(int c) { return c < 15 || (c >= 21 && c < 24); }

// M' and g of one correspondence (weighted by the robust kernel), its error, and whether it is an inlier; caches the mahalanobis
// matrix for the error pass.  (The direct form — the 28 values of pair_system — is pair_factor below; the per-point export uses it.)
template <typename Real, int FACTOR>
__device__ __forceinline__ bool pair_moments(const LinParams<Real>& p, int i, int j, bool within_bound, Real qx, Real qy, Real qz, Real tx, Real ty, Real tz, Sym3<Real>& Mp, Real* g, Real& e, Sym3<Real>& M_out) {
  const Real rx = tx - qx, ry = ty - qy, rz = tz - qz;
  const Real d2 = rx * rx + ry * ry + rz * rz;
  const bool inlier = (j >= 0) && within_bound && !(d2 > static_cast<Real>(p.max_sq));
  Mp = Sym3<Real>{};
  g[0] = g[1] = g[2] = Real(0);
  e = Real(0);
  if (inlier) {
    Sym3<Real> M;
    if constexpr (FACTOR == SGA_GICP) {
      const Sym3<Real> Cs = load_sym<Real>(p.src_cov, i);
      const Sym3<Real> Ct = load_sym<Real>(p.tgt_cov, j);
      const Sym3<Real> RCR = rotate_sym(p.T.r, Cs);
      M = inverse_sym<Real>({Ct.xx + RCR.xx, Ct.xy + RCR.xy, Ct.xz + RCR.xz, Ct.yy + RCR.yy, Ct.yz + RCR.yz, Ct.zz + RCR.zz});
      M_out = M;  // the caller caches it for the error pass (gicp_factor.hpp:80-89)
    } else if constexpr (FACTOR == SGA_PLANE_ICP) {
      const float4 nn = p.tgt_nrm[j];
      M = {Real(nn.x) * Real(nn.x), Real(0), Real(0), Real(nn.y) * Real(nn.y), Real(0), Real(nn.z) * Real(nn.z)};
    } else {
      M = {Real(1), Real(0), Real(0), Real(1), Real(0), Real(1)};
    }
    const Real vx = M.xx * rx + M.xy * ry + M.xz * rz, vy = M.xy * rx + M.yy * ry + M.yz * rz, vz = M.xz * rx + M.yz * ry + M.zz * rz;
    const Real e0 = Real(0.5) * (rx * vx + ry * vy + rz * vz);
    const Real w = p.robust_kind != SGA_ROBUST_NONE ? robust_weight<Real>(p.robust_kind, p.robust_c, e0) : Real(1);
    const Real* R = p.T.r;
    g[0] = w * (R[0] * vx + R[3] * vy + R[6] * vz);
    g[1] = w * (R[1] * vx + R[4] * vy + R[7] * vz);
    g[2] = w * (R[2] * vx + R[5] * vy + R[8] * vz);
    Mp = rotate_sym_t(R, M);
    Mp = {w * Mp.xx, w * Mp.xy, w * Mp.xz, w * Mp.yy, w * Mp.yz, w * Mp.zz};
    e = w * e0;
  }
  return inlier;
}

// Adds the moments of PTS points per lane (zero M' / g / e for the points that are no inliers) to the wave's fp64 row in LDS.  The
// lane adds its points up in registers; the 72 fp32 sums then go through ONE transposing wave reduction (device_math.hpp:
// wave_transpose_sum, ~3 instructions per sum instead of a 6-step DPP chain each) that leaves the totals spread over the lanes —
// lane l holds the sums number `slot` and 64 + slot — and every lane adds its own two to the row.  e is summed in fp64.
// Must be called by all 64 lanes of the wave (cross-lane operations).
// Sum number s -> row column: s < 6: H_tt (15 + s); s < 9: b_t (24 + s - 6); else the error-model block (kModelOff + s - 9).
constexpr int kMomentSums = 72;
__host__ __device__ constexpr int moment_column(int s) { return s < 6 ? 15 + s : (s < 9 ? 18 + s : kModelOff - 9 + s); }

// the S-th sum of one lane's PTS points (S is a template parameter: every index below is a compile-time constant, so the 72 values
// live in registers — a run-time-indexed array of them would be placed in scratch memory)
template <typename Real, int PTS, int S>
__device__ __forceinline__ Real moment_sum(const Real (&P)[PTS][3], const Sym3<Real> (&Mp)[PTS], const Real (&G)[PTS][3]) {
  auto m6 = [&](int u, int c) -> Real { return c == 0 ? Mp[u].xx : (c == 1 ? Mp[u].xy : (c == 2 ? Mp[u].xz : (c == 3 ? Mp[u].yy : (c == 4 ? Mp[u].yz : Mp[u].zz)))); };
  Real v = Real(0);
  if constexpr (S < 6) {
#pragma unroll
    for (int u = 0; u < PTS; u++) v += m6(u, S);
  } else if constexpr (S < 9) {
#pragma unroll
    for (int u = 0; u < PTS; u++) v -= G[u][S - 6];
  } else if constexpr (S < 18) {
    constexpr int a = (S - 9) / 3, j = (S - 9) % 3;
#pragma unroll
    for (int u = 0; u < PTS; u++) v += P[u][a] * G[u][j];
  } else if constexpr (S < 36) {
    constexpr int a = (S - 18) / 6, c = (S - 18) % 6;
#pragma unroll
    for (int u = 0; u < PTS; u++) v += P[u][a] * m6(u, c);
  } else {
    constexpr int pair = (S - 36) / 6, c = (S - 36) % 6;
    constexpr int a = pair < 3 ? 0 : (pair < 5 ? 1 : 2), b = pair < 3 ? pair : (pair < 5 ? pair - 2 : 2);
#pragma unroll
    for (int u = 0; u < PTS; u++) v += (P[u][a] * P[u][b]) * m6(u, c);
  }
  return v;
}
template <typename Real, int PTS, int BASE, int... S>  // v[S] = sum number BASE + S
__device__ __forceinline__ void moment_sums(const Real (&P)[PTS][3], const Sym3<Real> (&Mp)[PTS], const Real (&G)[PTS][3], Real (&v)[sizeof...(S)], std::integer_sequence<int, S...>) {
  ((v[S] = moment_sum<Real, PTS, BASE + S>(P, Mp, G)), ...);
}

template <typename Real, int PTS>
__device__ __forceinline__ void accumulate_moments(const Real (&P)[PTS][3], const Sym3<Real> (&Mp)[PTS], const Real (&G)[PTS][3], const Real (&E)[PTS], int inliers, double* __restrict__ acc_row, int lane) {
  double es = 0.0;
#pragma unroll
  for (int u = 0; u < PTS; u++) es += static_cast<double>(E[u]);
  const double et = wave_sum_f64(es);
  if (lane == 0) {
    acc_row[27] += et;
    acc_row[28] += static_cast<double>(inliers);
  }
  if constexpr (sizeof(Real) == 4 && PTS == 1) {
    // inside the one-query-per-lane search kernels (64 VGPRs): two reductions of 36 sums each — all 72 at once do not fit the register
    // budget there, and the spills (60 bytes per lane through scratch memory) showed up as 60 MB of HBM traffic per pass
    constexpr int kHalf = kMomentSums / 2;
#pragma unroll
    for (int part = 0; part < 2; part++) {
      float v[kHalf];
      if (part == 0)
        moment_sums<float, PTS, 0>(P, Mp, G, v, std::make_integer_sequence<int, kHalf>{});
      else
        moment_sums<float, PTS, kHalf>(P, Mp, G, v, std::make_integer_sequence<int, kHalf>{});
      float lo, hi;
      int slot;
      wave_transpose_sum<kHalf>(v, lane, lo, hi, slot);
      if (slot < kHalf) acc_row[moment_column(part * kHalf + slot)] += static_cast<double>(lo);
    }
    return;
  }
  Real v[kMomentSums];
  moment_sums<Real, PTS, 0>(P, Mp, G, v, std::make_integer_sequence<int, kMomentSums>{});
  if constexpr (sizeof(Real) == 4) {
    float lo, hi;
    int slot;
    wave_transpose_sum<kMomentSums>(v, lane, lo, hi, slot);
    acc_row[moment_column(slot)] += static_cast<double>(lo);
    if (slot + 64 < kMomentSums) acc_row[moment_column(slot + 64)] += static_cast<double>(hi);
  } else {
#pragma unroll
    for (int s = 0; s < kMomentSums; s++) {
      const double t = wave_sum_f64(v[s]);
      if (lane == 0) acc_row[moment_column(s)] += t;
    }
  }
}

// The kernel's FIRST argument (a LinParams) read again from the kernel-argument segment through a pointer the compiler cannot see through:
// the scalar loads of the fields a stage uses are issued in that stage and their registers die with it.  Without this the compiler loads
// every field at the kernel's start and keeps all of them for its whole length — more than the 100 scalar registers a wave has, so it
// parks them in the lanes of vector registers and fetches them back one v_readlane at a time (certify_linearize_kernel: 448 of its ~3 200
// vector instructions per wave).  Only valid inside a kernel whose first parameter is the LinParams<Real> passed by value.
template <typename Real>
__device__ __forceinline__ const LinParams<Real>& kernarg_lin_params() {
  using Args = const __attribute__((address_space(4))) LinParams<Real>;
  Args* a = (Args*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(a));
  return *(const LinParams<Real>*)a;
}

// The factor stage as a kernel of its own.  TARGET: 0 kd-tree (the neighbours come from the search kernel), 1 Gaussian voxel map, 2 flat voxel map (the lookup of a
// voxel target happens right here).  Streaming + two gathers; a lane handles PTS points (PTS x kTile consecutive points per
// workgroup step), their products are added up in registers, reduced with DPP inside the wave, in fp64 across waves.
// The factors of PTS points per lane — points first, first + stride, ... below `limit` — added to the wave's row.
// FRESH_NN: hint[] was written earlier in this very kernel (by any lane of this wave): read it past the vector L1.
// STAGED (the caller is a kernel whose first argument is `p0` itself): every stage reads the parameters it uses afresh (kernarg_lin_params).
template <typename Real, int FACTOR, int TARGET, int PTS, bool FRESH_NN, bool CERT, bool STAGED>
__device__ __forceinline__ void linearize_group(const LinParams<Real>& p0, int first, int stride, int limit, double* __restrict__ acc_row, int lane, unsigned long long* __restrict__ failed_masks) {
#define SGA_STAGE_PARAMS(name) const LinParams<Real>& name = STAGED ? kernarg_lin_params<Real>() : p0
  SGA_STAGE_PARAMS(p);
  Real P[PTS][3], G[PTS][3], E[PTS];
  Sym3<Real> Mp[PTS];
  int inliers = 0;
  // The PTS points of a lane go through the stages TOGETHER — source point + neighbour index, neighbour point, covariances — so
  // that the loads of a stage are in flight at once (one latency per stage, not per point); the stores (mahalanobis cache,
  // correspondence) come after the last load, or they would pin the loads of the next point behind them.
  float4 ps4[PTS];
  int jn[PTS];
  bool act[PTS];
#pragma unroll
  for (int u = 0; u < PTS; u++) {
    const int i = first + u * stride;
    act[u] = i < limit;
    ps4[u] = act[u] ? p.src_pts[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    jn[u] = -1;
    if constexpr (TARGET == 0 && CERT)
      jn[u] = act[u] ? p.cert_nn[i] : -1;  // (the same array as hint[]: read through the pointer it is written through)
    else if constexpr (TARGET == 0)
      jn[u] = act[u] ? (FRESH_NN ? __hip_atomic_load(&p.hint[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p.hint[i]) : -1;
  }
  Real Q[PTS][3], Tg[PTS][3];
  bool within[PTS];
  {
  SGA_STAGE_PARAMS(p);
#pragma unroll
  for (int u = 0; u < PTS; u++) {
    P[u][0] = ps4[u].x, P[u][1] = ps4[u].y, P[u][2] = ps4[u].z;  // multiplied by zero M' / g when the point is no inlier
    Q[u][0] = Q[u][1] = Q[u][2] = Real(0);
    if (act[u]) transform_point(p.T, P[u][0], P[u][1], P[u][2], Q[u][0], Q[u][1], Q[u][2]);
    Tg[u][0] = Tg[u][1] = Tg[u][2] = Real(0);
    within[u] = true;
    if constexpr (TARGET == 2) {
      if (act[u]) {
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
        jn[u] = flat_nearest<Real>(p.flat, p.tgt_pts, Q[u][0], Q[u][1], Q[u][2], m);
        Tg[u][0] = m.x, Tg[u][1] = m.y, Tg[u][2] = m.z;
      }
    } else if constexpr (TARGET == 1) {
      if (act[u]) {
        if (p.vox.offsets == 1)  // (wave-uniform) the default: the query's own voxel, no distance to compare
          jn[u] = voxel_lookup(p.vox, static_cast<float>(Q[u][0]), static_cast<float>(Q[u][1]), static_cast<float>(Q[u][2]));
        else
          jn[u] = voxel_nearest<Real>(p.vox, p.tgt_pts, Q[u][0], Q[u][1], Q[u][2]);
      }
    }
  }
  }
  if constexpr (TARGET != 2) {
    float4 m4[PTS];
    if constexpr (CERT && TARGET == 0) {
      // The certificate check of the warm pass (search_lane / nn_search_queue_kernel: the same arithmetic, bit for bit) on the way through:
      // both candidates of the previous pass are fetched, the nearer one (canonical rule) is the neighbour if its new distance is
      // below the exclusion radius minus the point's motion; otherwise the point is flagged for the walkers' kernel and skipped here.
      int c2[PTS];
      float rx[PTS];
      float4 m4b[PTS];
      {
        SGA_STAGE_PARAMS(p);
#pragma unroll
        for (int u = 0; u < PTS; u++) {
          const int i = first + u * stride;
          c2[u] = act[u] ? p.cert_nn2[i] : -1;
          rx[u] = act[u] ? p.cert_rex[i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < PTS; u++) {
          m4[u] = jn[u] >= 0 ? p.tgt_pts[jn[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
          m4b[u] = c2[u] >= 0 ? p.tgt_pts[c2[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      SGA_STAGE_PARAMS(p);
#pragma unroll
      for (int u = 0; u < PTS; u++) {
        const int i = first + u * stride;
        bool failed = false;
        if (act[u]) {
          const float fx = static_cast<float>(Q[u][0]), fy = static_cast<float>(Q[u][1]), fz = static_cast<float>(Q[u][2]);
          Real ox, oy, oz;
          transform_point<Real>(p.T_prev, P[u][0], P[u][1], P[u][2], ox, oy, oz);
          const float moved = sqrtf(kd_dist2(static_cast<float>(ox), static_cast<float>(oy), static_cast<float>(oz), fx, fy, fz));
          const float d1 = jn[u] >= 0 ? kd_dist2(m4[u].x, m4[u].y, m4[u].z, fx, fy, fz) : INFINITY;
          const float d2 = c2[u] >= 0 ? kd_dist2(m4b[u].x, m4b[u].y, m4b[u].z, fx, fy, fz) : INFINITY;
          const bool swap = c2[u] >= 0 && (d2 < d1 || (d2 == d1 && c2[u] < jn[u]));  // the canonical rule: equidistant -> lower position
          const int best = swap ? c2[u] : jn[u];
          const float r = certify(rx[u], moved, best >= 0, swap ? d2 : d1, p.cert_within2, p.cert_pad);
          failed = !(r >= 0.f);
          // settled: the shrunken radius; failed: the flag of the walkers' kernel, which is also the exploration slack of the re-walk
          p.cert_rex[i] = failed ? -fminf(fmaxf(moved, p.cert_slack_min), p.cert_slack_max) : r;
          if (swap) {  // (for a walker: its seed is the nearer candidate)
            p.cert_nn[i] = c2[u];
            p.cert_nn2[i] = jn[u];
            m4[u] = m4b[u];
          }
          jn[u] = failed ? -1 : best;
        }
        const unsigned long long fm = __ballot(failed);
        if (lane == 0) {
          failed_masks[u] = fm;  // (LDS) which of this wave's 64 points of sub-step u walk
          if (fm != 0ull) p.cert_walked[i >> 6] += static_cast<uint32_t>(__popcll(fm));  // the 64 points belong to this wave: no atomic
        }
        if (failed) act[u] = false;  // the walk phase writes its correspondence
      }
    } else {
#pragma unroll
      for (int u = 0; u < PTS; u++) m4[u] = jn[u] >= 0 ? p.tgt_pts[jn[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    bool decided64[PTS];
#pragma unroll
    for (int u = 0; u < PTS; u++) decided64[u] = false;
    if constexpr (TARGET == 0 && !CERT && sizeof(Real) == 8) {
      // fp64 per-pair arithmetic: the reference compares DOUBLE distances (ann/knn_result.hpp:80-100, double queries against the stored
      // points).  The walk compared fp32 distances of the fp32-rounded query; it also kept its runner-up — the only other point that can
      // be the nearest in double when the two agree to fp32 rounding.  Both candidates are measured again here, in double, against the
      // double query: the nearer one is the correspondence (equidistant: the lower kd position, the canonical rule), and the rejector's
      // test (rejector.hpp:19-28: reject iff sq_dist > max_dist_sq) runs on that double distance.  hint[] / hint2[] / rex[] — the state of
      // the SEARCH — stay what the walk wrote.
      if (p.cert_nn2 != nullptr) {
#pragma unroll
        for (int u = 0; u < PTS; u++) {
          const int i = first + u * stride;
          if (!act[u] || jn[u] < 0) continue;
          const int j2 = p.cert_nn2[i];
          const double ax = static_cast<double>(m4[u].x) - Q[u][0], ay = static_cast<double>(m4[u].y) - Q[u][1], az = static_cast<double>(m4[u].z) - Q[u][2];
          double d1 = ax * ax + ay * ay + az * az;
          if (j2 >= 0) {
            const float4 c = p.tgt_pts[j2];
            const double bx = static_cast<double>(c.x) - Q[u][0], by = static_cast<double>(c.y) - Q[u][1], bz = static_cast<double>(c.z) - Q[u][2];
            const double d2 = bx * bx + by * by + bz * bz;
            if (d2 < d1 || (d2 == d1 && j2 < jn[u])) {
              d1 = d2;
              jn[u] = j2;
              m4[u] = c;
            }
          }
          within[u] = d1 <= static_cast<double>(p.max_sq);
          if (p.reject != nullptr) within[u] = within[u] && p.reject[__float_as_uint(ps4[u].w)] == 0;
          decided64[u] = true;
        }
      }
    }
    SGA_STAGE_PARAMS(p);
#pragma unroll
    for (int u = 0; u < PTS; u++) {
      Tg[u][0] = m4[u].x, Tg[u][1] = m4[u].y, Tg[u][2] = m4[u].z;
      if constexpr (TARGET == 0) {
        if (jn[u] >= 0 && !decided64[u]) {
          // the search reaches a little beyond the rejector (kSearchMargin) and a certified neighbour may have drifted out of
          // reach: a neighbour counts only inside the reach of a plain search, whichever way it was found
          within[u] = kd_dist2(m4[u].x, m4[u].y, m4[u].z, static_cast<float>(Q[u][0]), static_cast<float>(Q[u][1]), static_cast<float>(Q[u][2])) < p.bound2;
          if (p.reject != nullptr) within[u] = within[u] && p.reject[__float_as_uint(ps4[u].w)] == 0;
        }
      }
    }
  }
  Sym3<Real> Mh[PTS];  // the mahalanobis matrices, stored after the last load
  bool inl[PTS];
  {
  SGA_STAGE_PARAMS(p);
#pragma unroll
  for (int u = 0; u < PTS; u++) {
    const int i = first + u * stride;
    inl[u] = false;
    Mp[u] = Sym3<Real>{};
    Mh[u] = Sym3<Real>{};
    G[u][0] = G[u][1] = G[u][2] = E[u] = Real(0);
    if (act[u]) inl[u] = pair_moments<Real, FACTOR>(p, i, jn[u], within[u], Q[u][0], Q[u][1], Q[u][2], Tg[u][0], Tg[u][1], Tg[u][2], Mp[u], G[u], E[u], Mh[u]);
    inliers += __popcll(__ballot(inl[u]));
  }
  }
  SGA_STAGE_PARAMS(pw);
#pragma unroll
  for (int u = 0; u < PTS; u++) {
    const int i = first + u * stride;
    if (act[u]) {
      pw.corr[i] = inl[u] ? jn[u] : -1;
      if constexpr (FACTOR == SGA_GICP) {
        if (inl[u] && pw.store_maha) {  // only robust factors read it back (error kernel); otherwise it is recomputed on demand
          Real* m = pw.maha + static_cast<size_t>(i) * 6;
          m[0] = Mh[u].xx, m[1] = Mh[u].xy, m[2] = Mh[u].xz, m[3] = Mh[u].yy, m[4] = Mh[u].yz, m[5] = Mh[u].zz;
        }
      }
    }
  }
  if (inliers == 0) return;  // wave-uniform
  accumulate_moments<Real, PTS>(P, Mp, G, E, inliers, acc_row, lane);
#undef SGA_STAGE_PARAMS
}

template <typename Real, int FACTOR, int TARGET, int PTS>
__global__ __launch_bounds__(kTile) __attribute__((amdgpu_waves_per_eu(4))) void linearize_kernel(const LinParams<Real> p) {
  __shared__ double sh_acc[kTile / 64][kRow];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = lane; c < kRow; c += 64) sh_acc[wave][c] = 0.0;  // each wave owns its row: no workgroup barrier needed until the end
  double* acc_row = sh_acc[wave];

  int tile, stride, tile_end;
  tile_schedule(p.num_tiles, tile, stride, tile_end);
  for (; tile < tile_end; tile += stride) {
    linearize_group<Real, FACTOR, TARGET, PTS>(p, tile * PTS * kTile + static_cast<int>(threadIdx.x), kTile, p.n, acc_row, lane);
  }
  __syncthreads();
  if (threadIdx.x < kRow) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kTile / 64; w++) t += sh_acc[w][threadIdx.x];
    if (p.tail.enabled)
      __hip_atomic_store(&p.partials[static_cast<size_t>(blockIdx.x) * kRow + threadIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      p.partials[static_cast<size_t>(blockIdx.x) * kRow + threadIdx.x] = t;
  }
  if (p.tail.enabled) fused_tail(p.tail, p.partials, gridDim.x, kModelCols, kRow, true);
}

// K1 of a warm pass after a small motion (large clouds): linearize_kernel with the certificate check on the way through.
// The streaming part is the factor kernel as it is (4 points per lane, everything of a stage in flight together: the kernel that reaches
// half of the HBM peak) plus, per point, the second candidate and the 12 bytes of certificate; the few points whose certificate fails are
// left out, flagged, and collected per step of 1024 points (one ballot per wave and sub-step into LDS: a fixed order).  Wave 0 of the
// workgroup then walks them, 64 at a time (seeded top-down walk with exploration slack, walk_lane), and adds their factors to its row —
// while the other workgroups stream on.  One launch, one partial row per workgroup; the old form (nn_search_queue_kernel: certificate
// check, queue-fed walks and factors per chunk of 4 tiles in one wave) streams at half this rate and stays for small clouds.
#ifndef SGA_CERT_WAVES
#define SGA_CERT_WAVES 4
#endif
template <typename Real, int FACTOR, int PTS>
__global__ __launch_bounds__(kTile) __attribute__((amdgpu_waves_per_eu(SGA_CERT_WAVES))) void certify_linearize_kernel(const LinParams<Real> p, const NNParams<Real> q_arg) {
  extern __shared__ uint32_t kd_stack[];  // 4 x tree depth x 64 words: the traversal stacks of the waves' walks
  __shared__ double sh_acc[kTile / 64][kRow];
  __shared__ unsigned long long sh_failed[kTile / 64][PTS];
  __shared__ int sh_list[PTS * kTile];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = lane; c < kRow; c += 64) sh_acc[wave][c] = 0.0;
  double* acc_row = sh_acc[wave];

  int tile, stride, tile_end;
  tile_schedule(p.num_tiles, tile, stride, tile_end);
  for (; tile < tile_end; tile += stride) {
    const int base = tile * PTS * kTile;
    linearize_group<Real, FACTOR, 0, PTS, false, true, true>(p, base + static_cast<int>(threadIdx.x), kTile, p.n, acc_row, lane, sh_failed[wave]);
    __syncthreads();
    {  // the walkers of this step: point of (sub-step u, wave w, lane l) = base + u * kTile + w * 64 + l; every wave counts them, wave 0 lists them
      int total = 0;
#pragma unroll
      for (int u = 0; u < PTS; u++)
#pragma unroll
        for (int w = 0; w < kTile / 64; w++) {
          const unsigned long long m = sh_failed[w][u];
          if (wave == 0 && ((m >> lane) & 1ull)) sh_list[total + __popcll(m & ((1ull << lane) - 1ull))] = base + u * kTile + w * 64 + lane;
          total += __popcll(m);
        }
      __syncthreads();
      // What the walk phase needs of the search parameters (tree, grid, certificate arrays: ~60 scalar registers) is read from the kernel
      // argument segment HERE, through a pointer the compiler cannot see through — otherwise it loads all of it at the kernel's start,
      // keeps it live across the streaming part, runs out of scalar registers and re-reads the pose of the streaming part from spilled
      // lanes (v_readlane: 720 of its 3187 vector instructions).
      using SearchArgs = const __attribute__((address_space(4))) NNParams<Real>;
      static_assert(alignof(NNParams<Real>) <= 8 && sizeof(LinParams<Real>) % 8 == 0, "kernel argument layout");
      SearchArgs* qa = (SearchArgs*)((const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr() + sizeof(LinParams<Real>));
      asm volatile("" : "+s"(qa));
      const NNParams<Real>& q = *(const NNParams<Real>*)qa;
      // spread over the four waves (entry 4 l + w of every 256 goes to lane l of wave w: a fixed assignment): four times the walks in flight
      uint32_t* my_stack = kd_stack + static_cast<size_t>(wave) * static_cast<size_t>(max(q.kd.depth, 1)) * 64;
      // A handful of walkers (the late passes of a registration): what they cost is the LATENCY of one walk, paid by the whole pass, while
      // nearly all lanes idle.  Then every walker gets a group of Gw lanes (the largest power of two that serves all of them in one
      // round: 64 lanes each for <= 4 walkers ... 2 lanes each for <= 128) which scan ring 1 together (grid_ring1_group).
      int Gw = 1;
      if ((q.grid_walk & 8) && total > 0 && total <= kTile / 2) {
        Gw = 64;
        while (Gw * total > kTile) Gw >>= 1;
      }
      const int per_round = kTile / Gw;  // walkers per round of the workgroup
      for (int k0 = 0; k0 < total; k0 += per_round) {
        const int sub = lane / Gw, gl = lane & (Gw - 1);
        const int k = k0 + 4 * sub + wave;  // (Gw = 1: entry 4 l + w of every 256 goes to lane l of wave w)
        const bool has = k < total, mine = has && gl == 0;
        const int i = has ? sh_list[k] : 0;
        Real P[1][3] = {{Real(0), Real(0), Real(0)}}, G[1][3] = {{Real(0), Real(0), Real(0)}}, E[1] = {Real(0)};
        Sym3<Real> Mp[1] = {Sym3<Real>{}};
        Sym3<Real> Mh{};
        bool inl = false;
        bool g_settled = false;
        int g_nn = -1, g_nn2 = -1;
        float g_rex = 0.f, g_seen = INFINITY;
        bool g_face = false;
        if (Gw > 1) {  // workgroup-uniform
          float gx = 0.f, gy = 0.f, gz = 0.f;
          if (has) {
            const float4 ps = p.src_pts[i];
            Real t[3];
            transform_point<Real>(p.T, Real(ps.x), Real(ps.y), Real(ps.z), t[0], t[1], t[2]);
            gx = static_cast<float>(t[0]), gy = static_cast<float>(t[1]), gz = static_cast<float>(t[2]);
          }
          g_settled = grid_ring1_group(q.grid, Gw, gl, has, gx, gy, gz, q.bound2, g_nn, g_nn2, g_rex, g_seen, &g_face);
        }
        if (mine) {
          const float4 ps = p.src_pts[i];
          P[0][0] = ps.x, P[0][1] = ps.y, P[0][2] = ps.z;
          Real t[3];
          transform_point<Real>(p.T, P[0][0], P[0][1], P[0][2], t[0], t[1], t[2]);
          const float fx = static_cast<float>(t[0]), fy = static_cast<float>(t[1]), fz = static_cast<float>(t[2]);
          // A walker of a warm pass sits next to a surface (its certificate failed by millimetres): ring 1 of the cell grid settles it
          // with a chain of two dependent loads where the seeded kd walk has a dozen, and its 27 cells give the new certificate a radius
          // of at least a cell.  The rare walker ring 1 does not settle walks the tree.
          int j = -1;
          bool settled = g_settled;
          if (Gw == 1 && (q.grid_walk & 1)) settled = grid_ring1_lane(q.grid, fx, fy, fz, q.bound2, g_nn, g_nn2, g_rex, g_seen, &g_face);  // wave-uniform condition
          // The ring's certificate ends at the face of its 27 cells at the latest, with no exploration slack: a walker whose neighbour lies
          // just inside that face would fail again in every later pass (the lone walker the last passes of a registration keep: ~5 us of
          // tail each).  Such a walker — radius set by the face, less than the re-walk's slack beyond the neighbour — walks the tree
          // once instead: the walk's radius ends at the third target point or the slack, whichever comes first.  (A radius that ends at
          // a third point is all a walk would find too: those stay with the ring.)
          if (settled && g_face && g_nn >= 0 && (q.grid_walk & 32)) {
            const float need = -__hip_atomic_load(&q.rex[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // what the check left there: the re-walk's slack
            if (g_rex - g_seen < need) settled = false;
          }
          if (settled) {
            q.nn[i] = g_nn;
            q.nn2[i] = g_nn2;
            q.rex[i] = g_rex;
            j = g_nn;
          }
          if (!settled) {
            // seed: the nearer candidate (the check put it first); slack: what the check left in rex[] — both written by this workgroup
            const int seed = __hip_atomic_load(&q.nn[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float slack = -__hip_atomic_load(&q.rex[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            j = walk_lane<Real, 64>(q, i, fx, fy, fz, seed, slack, my_stack, lane);
          }
          if (j >= 0) {
            const float4 m = p.tgt_pts[j];
            const bool within = kd_dist2(m.x, m.y, m.z, fx, fy, fz) < p.bound2;
            inl = pair_moments<Real, FACTOR>(p, i, j, within, t[0], t[1], t[2], Real(m.x), Real(m.y), Real(m.z), Mp[0], G[0], E[0], Mh);
          }
          p.corr[i] = inl ? j : -1;
          if constexpr (FACTOR == SGA_GICP) {
            if (inl && p.store_maha) {
              Real* mm = p.maha + static_cast<size_t>(i) * 6;
              mm[0] = Mh.xx, mm[1] = Mh.xy, mm[2] = Mh.xz, mm[3] = Mh.yy, mm[4] = Mh.yz, mm[5] = Mh.zz;
            }
          }
        }
        const int inliers = __popcll(__ballot(inl));
        if (inliers > 0) accumulate_moments<Real, 1>(P, Mp, G, E, inliers, acc_row, lane);
      }
    }
    __syncthreads();  // (sh_failed / sh_list are reused by the next step)
  }
  __syncthreads();
  if (threadIdx.x < kRow) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < kTile / 64; w++) t += sh_acc[w][threadIdx.x];
    if (p.tail.enabled)
      __hip_atomic_store(&p.partials[static_cast<size_t>(blockIdx.x) * kRow + threadIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      p.partials[static_cast<size_t>(blockIdx.x) * kRow + threadIdx.x] = t;
  }
  if (p.tail.enabled) fused_tail(p.tail, p.partials, gridDim.x, kModelCols, kRow, true);  // small grids: the last workgroup adds the rows and hands the result over
}

// K1, fused: the search wave also evaluates the factors of its own 64 source points — their neighbours are in registers, no nn[]
// round trip, no second kernel whose start waits for the slowest search wave.  The factor algebra in moment form is lean enough
// (42 VGPRs at one point per lane) to live inside the search kernel's register budget (72 VGPRs for 7 waves per SIMD, which the walks need).
// One partial row per wave (= per tile of 64 points, whatever workgroup took it: the row order, and with it the fp64 sum, does not
// depend on the placement).  The row is built in the LDS the traversal stacks occupied.
// 7 waves per SIMD (72 VGPRs): as fast as 8 (64 VGPRs; measured 179 / 190 / 200 against 182 / 185 / 198 us for the cold passes of a C3
// registration) and without the two registers the 64-VGPR build spills around the walk (20 bytes per lane through scratch memory =
// 2 x 20 MB of HBM traffic per cold pass).
// The warm form (CHECK: certificate check, ring 1 of the cell grid for the walkers, then the walk) needs 16 registers more: at 7 waves it
// spills them around the walk (68 bytes per lane through scratch = 2 x 68 MB of HBM traffic per pass of 1M points, rocprofv3 FETCH_SIZE /
// WRITE_SIZE), at 6 waves per SIMD (80 VGPRs) it does not.
#ifndef SGA_SL_WAVES
#define SGA_SL_WAVES 7
#endif
#ifndef SGA_SL_WAVES_WARM
#define SGA_SL_WAVES_WARM 6
#endif
template <typename Real, int FACTOR, bool CHECK>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(CHECK ? SGA_SL_WAVES_WARM : SGA_SL_WAVES, CHECK ? SGA_SL_WAVES_WARM : SGA_SL_WAVES))) void search_linearize_kernel(const NNParams<Real> p, const LinParams<Real> lp) {
  extern __shared__ uint32_t kd_stack[];  // max(tree depth, 3) x 64 words: the traversal stacks, then the wave's row of kRow doubles
#ifdef SGA_KD_TRIPS
  const unsigned long long wave_t0 = wall_clock64();
#endif
  const int lane = threadIdx.x;
  const int slot = search_tile_of_block();
  const int tile = p.tile_order != nullptr ? static_cast<int>(p.tile_order[slot]) : slot;  // wave-uniform
  const unsigned long long cost_t0 = p.tile_cost != nullptr ? wall_clock64() : 0ull;
  const int i = tile * 64 + lane;
  const bool active = i < p.n;
  const float4 ps = active ? p.src_pts[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  Real q[3] = {Real(0), Real(0), Real(0)};
  int j = -1;
  if (active) {
    transform_point<Real>(p.T, ps.x, ps.y, ps.z, q[0], q[1], q[2]);
    j = search_lane<Real, 64, CHECK>(p, tile, i, ps, static_cast<float>(q[0]), static_cast<float>(q[1]), static_cast<float>(q[2]), kd_stack);
  }
  // ---- the factors of the wave's 64 points (linearize_kernel with one point per lane)
  Real P[1][3] = {{ps.x, ps.y, ps.z}}, G[1][3] = {{Real(0), Real(0), Real(0)}}, E[1] = {Real(0)};
  Sym3<Real> Mp[1] = {Sym3<Real>{}};
  Sym3<Real> Mh{};
  bool inl = false;
  if (j >= 0) {
    const float4 m = lp.tgt_pts[j];
    // the search reaches a little beyond the rejector (kSearchMargin) and a certified neighbour may have drifted out of reach: a
    // neighbour counts only inside the reach of a plain search, whichever way it was found
    const bool within = kd_dist2(m.x, m.y, m.z, static_cast<float>(q[0]), static_cast<float>(q[1]), static_cast<float>(q[2])) < lp.bound2;
    inl = pair_moments<Real, FACTOR>(lp, i, j, within, q[0], q[1], q[2], Real(m.x), Real(m.y), Real(m.z), Mp[0], G[0], E[0], Mh);
  }
  if (active) {
    lp.corr[i] = inl ? j : -1;
    if constexpr (FACTOR == SGA_GICP) {
      if (inl && lp.store_maha) {
        Real* mm = lp.maha + static_cast<size_t>(i) * 6;
        mm[0] = Mh.xx, mm[1] = Mh.xy, mm[2] = Mh.xz, mm[3] = Mh.yy, mm[4] = Mh.yz, mm[5] = Mh.zz;
      }
    }
  }
  const int inliers = __popcll(__ballot(inl));
  __syncthreads();  // one wave: every lane is done with its stack
  double* row = reinterpret_cast<double*>(kd_stack);
  for (int c = lane; c < kRow; c += 64) row[c] = 0.0;
  __syncthreads();
  if (inliers > 0) accumulate_moments<Real, 1>(P, Mp, G, E, inliers, row, lane);
  __syncthreads();
  for (int c = lane; c < kRow; c += 64) lp.partials[static_cast<size_t>(tile) * kRow + c] = row[c];
  if (p.tile_cost != nullptr && lane == 0) p.tile_cost[tile] = static_cast<uint32_t>(wall_clock64() - cost_t0);
#ifdef SGA_KD_TRIPS
  if (blockIdx.x < 32768 && lane == 0) {
    g_kd_wave_times[2 * blockIdx.x] = wave_t0;
    g_kd_wave_times[2 * blockIdx.x + 1] = wall_clock64();
  }
#endif
}

// Per-point export of the same factors (the reference's Python binding exposes Factor::linearize per source point,
// src/python/factors.cpp:52-101): the 28 values of every pair instead of their sum.  Runs after a linearize pass at the same pose
// (the neighbours come from hint[]); not on the hot path.
template <typename Real, int FACTOR>
__global__ __launch_bounds__(256) void per_point_kernel(const LinParams<Real> p, double* __restrict__ out28, unsigned char* __restrict__ ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n) return;
  const float4 ps4 = p.src_pts[i];
  const Real px = ps4.x, py = ps4.y, pz = ps4.z;
  Real qx, qy, qz;
  transform_point(p.T, px, py, pz, qx, qy, qz);
  Real vals[28];
#pragma unroll
  for (int k = 0; k < 28; k++) vals[k] = Real(0);
  const int j = p.hint[i];
  Real tx = 0, ty = 0, tz = 0;
  bool within = false;
  if (j >= 0) {
    const float4 m = p.tgt_pts[j];
    tx = m.x;
    ty = m.y;
    tz = m.z;
    within = kd_dist2(m.x, m.y, m.z, static_cast<float>(qx), static_cast<float>(qy), static_cast<float>(qz)) < p.bound2;
  }
  const bool inlier = pair_factor<Real, FACTOR>(p, i, j, within, px, py, pz, qx, qy, qz, tx, ty, tz, vals);
  const uint32_t orig = __float_as_uint(ps4.w);  // the caller's source order
  ok[orig] = inlier ? 1 : 0;
#pragma unroll
  for (int k = 0; k < 28; k++) out28[static_cast<size_t>(orig) * 28 + k] = static_cast<double>(vals[k]);
}

template <typename Real>
struct ErrParams {
  const float4* __restrict__ src_pts;
  int n;
  int num_tiles;
  const float4* __restrict__ tgt_pts;
  const float4* __restrict__ tgt_nrm;
  const int* __restrict__ corr;
  const Real* __restrict__ maha;
  Rigid<Real> T;
  int robust_kind;
  Real robust_c;
  double* __restrict__ partials;
  FusedTail tail;
};

template <typename Real, int FACTOR>
__global__ __launch_bounds__(kTile) void error_kernel(const ErrParams<Real> p) {
  __shared__ double sh_e[kTile / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc = 0.0;
  int tile, stride, tile_end;
  tile_schedule(p.num_tiles, tile, stride, tile_end);
  for (; tile < tile_end; tile += stride) {
    const int i = tile * kTile + threadIdx.x;
    Real e = Real(0);
    if (i < p.n) {
      const int j = p.corr[i];
      if (j >= 0) {
        const float4 ps4 = p.src_pts[i];
        Real qx, qy, qz;
        transform_point<Real>(p.T, ps4.x, ps4.y, ps4.z, qx, qy, qz);
        const float4 t4 = p.tgt_pts[j];
        const Real rx = Real(t4.x) - qx, ry = Real(t4.y) - qy, rz = Real(t4.z) - qz;
        if constexpr (FACTOR == SGA_GICP) {
          const Real* m = p.maha + static_cast<size_t>(i) * 6;
          const Real vx = m[0] * rx + m[1] * ry + m[2] * rz, vy = m[1] * rx + m[3] * ry + m[4] * rz, vz = m[2] * rx + m[4] * ry + m[5] * rz;
          e = Real(0.5) * (rx * vx + ry * vy + rz * vz);
        } else if constexpr (FACTOR == SGA_PLANE_ICP) {
          const float4 nn = p.tgt_nrm[j];
          // same association as pair_system's 1/2 r^T diag(n^2) r
          e = Real(0.5) * (rx * (Real(nn.x) * Real(nn.x) * rx) + ry * (Real(nn.y) * Real(nn.y) * ry) + rz * (Real(nn.z) * Real(nn.z) * rz));
        } else {
          e = Real(0.5) * (rx * rx + ry * ry + rz * rz);
        }
        if (p.robust_kind != SGA_ROBUST_NONE) e *= robust_weight<Real>(p.robust_kind, p.robust_c, e);
      }
    }
    acc += wave_sum_f64(static_cast<double>(e));
  }
  if (lane == 0) sh_e[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.0;
    for (int w = 0; w < kTile / 64; w++) s += sh_e[w];
    if (p.tail.enabled)
      __hip_atomic_store(&p.partials[blockIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      p.partials[blockIdx.x] = s;
  }
  if (p.tail.enabled) fused_tail(p.tail, p.partials, gridDim.x, 1, 1);
}

// Deterministic fp64 sum of `nrows` partial rows of `ncols` (<= 32) doubles in ONE launch of G = 32 workgroups: workgroup g sums
// rows g, g+G, g+2G, ... into row g of `stage`; the workgroup that finishes LAST (a ticket counter) adds the G stage rows in fixed
// order — which workgroup that is changes nothing in the arithmetic — and writes out[ncols] (+ zero padding up to out_n).
// Hand-off between workgroups (per-CU L1s and per-XCD L2s are not coherent): a workgroup writes its stage row, a barrier orders the
// row before lane 0's ticket increment, which is an agent-scope release / acquire (the row accesses themselves are agent-scope
// relaxed atomics, i.e. write-through stores and cache-bypassing loads); only these <= 64 workgroups touch the ticket.  (Putting the ticket into the 2048 workgroups of the producer kernel was measured: +20 us.)
// When `host` is given the result is handed to the host right here: copied into pinned, device-mapped host memory, then a
// sequence number is published (system-scope release) on which the host spins.  This replaces hipMemcpyAsync +
// hipStreamSynchronize, whose fixed cost is paid twice per optimizer iteration.
constexpr int kReduceGroups = 64;

constexpr int kReduceSlices = 8;  // 1024 threads = 8 slices of 128 columns
__global__ __launch_bounds__(kReduceSlices * kCols) void reduce_rows_kernel(
  const double* __restrict__ partials, int nrows, int ncols, int row_stride, double* __restrict__ stage, unsigned* __restrict__ ticket, double* __restrict__ out, int out_n, double* __restrict__ host,
  unsigned long long seq, int derive, const uint32_t* __restrict__ stats) {
  __shared__ double sh[kReduceSlices][kCols];
  __shared__ unsigned sh_ticket;
  const int c = threadIdx.x & (kCols - 1), s = threadIdx.x / kCols;
  const int G = gridDim.x;
  // stage 1: (workgroup g, slice s) adds rows g + G * s, g + G * (s + 8), ...: four independent chains, the loads of a chain in flight together
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  if (c < ncols) {
    const int step = G * kReduceSlices;
    int r = blockIdx.x + G * s;
    for (; r + 3 * step < nrows; r += 4 * step) {
      const double v0 = partials[static_cast<size_t>(r) * row_stride + c], v1 = partials[static_cast<size_t>(r + step) * row_stride + c];
      const double v2 = partials[static_cast<size_t>(r + 2 * step) * row_stride + c], v3 = partials[static_cast<size_t>(r + 3 * step) * row_stride + c];
      a0 += v0, a1 += v1, a2 += v2, a3 += v3;
    }
    for (; r < nrows; r += step) a0 += partials[static_cast<size_t>(r) * row_stride + c];
  }
  sh[s][c] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  auto fold = [&]() {  // sh[0][c] = sum over the slices, fixed order
    if (threadIdx.x < kCols) {
      double t = 0.0;
#pragma unroll
      for (int k = 0; k < kReduceSlices; k++) t += sh[k][threadIdx.x];
      sh[0][threadIdx.x] = t;
    }
    __syncthreads();
  };
  fold();
  if (G > 1) {
    if (threadIdx.x < kCols) __hip_atomic_store(&stage[blockIdx.x * kCols + threadIdx.x], sh[0][threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) sh_ticket = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);  // release this workgroup's stage row, acquire the earlier ones
    __syncthreads();
    if (sh_ticket != static_cast<unsigned>(G - 1)) return;  // workgroup-uniform
    // the last workgroup adds the G <= 64 stage rows: slice s takes rows s, s + 8, ...: at most 8 loads per thread, all in flight
    double v[kReduceGroups / kReduceSlices];
#pragma unroll
    for (int k = 0; k < kReduceGroups / kReduceSlices; k++) {
      const int g = s + k * kReduceSlices;
      v[k] = g < G ? __hip_atomic_load(&stage[g * kCols + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    }
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < kReduceGroups / kReduceSlices; k++) t += v[k];
    __syncthreads();
    sh[s][c] = t;
    __syncthreads();
    fold();
  }
  if (threadIdx.x < kCols) {
    const int cc = threadIdx.x;
    const double t = (derive && is_derived_col(cc)) ? derived_entry(cc, sh[0]) : sh[0][cc];  // moment form: H_rr, H_rt, b_r from the totals
    if (cc < out_n) {
      double r = cc < ncols ? t : 0.0;
      if (stats != nullptr && (cc == kStatsCol || cc == kStatsCol + 1)) r = static_cast<double>(stats[cc - kStatsCol]);  // a grid pass's search statistics ride along in two spare columns
      out[cc] = r;
      if (host != nullptr) host[cc] = r;
    }
  }
  if (G > 1 && threadIdx.x == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next launch on this stream
  if (host != nullptr) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(host + kSeqWord), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// Longest tile first.  A pass of the one-query-per-lane kernel is 1.9 rounds of waves, and 35 - 40 % of it is the drain of the waves
// that happened to start last and run long.  Which tiles run long is only known afterwards — but after a small
// motion the next pass's costs resemble this one's (correlation 0.9 at <= 0.1 m, scripts/diag_lpt.py).  This kernel sorts the tiles of
// every XCD's share (search_tile_of_block: slot s of XCD x is its (s - x * per_xcd)-th workgroup to start) by the duration their
// wave just recorded, longest first: a counting sort over 64 duration classes of 2.56 us, one workgroup per XCD.  The order within a class is
// whatever the LDS atomics make it: nothing but speed depends on the launch order (rows are indexed by tile).
__global__ __launch_bounds__(1024) void tile_order_kernel(const uint32_t* __restrict__ cost, uint32_t* __restrict__ order, int num_tiles) {
  __shared__ uint32_t hist[64], start[64];
  const int per_xcd = num_tiles >> 3;
  const int x = blockIdx.x, first = x * per_xcd;
  if (threadIdx.x < 64) hist[threadIdx.x] = 0u;
  __syncthreads();
  for (int k = threadIdx.x; k < per_xcd; k += blockDim.x) atomicAdd(&hist[63u - min(63u, cost[first + k] >> 8)], 1u);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t acc = 0u;
    for (int b = 0; b < 64; b++) {
      start[b] = acc;
      acc += hist[b];
    }
  }
  __syncthreads();
  for (int k = threadIdx.x; k < per_xcd; k += blockDim.x) {
    const uint32_t pos = atomicAdd(&start[63u - min(63u, cost[first + k] >> 8)], 1u);
    order[first + pos] = static_cast<uint32_t>(first + k);
  }
  if (x == 0)
    for (int t = 8 * per_xcd + threadIdx.x; t < num_tiles; t += blockDim.x) order[t] = static_cast<uint32_t>(t);  // the (< 8) tiles beyond the XCD shares
}

// partial rows (one per workgroup of linearize_kernel / error_kernel, or one per 64 source points when the search kernel does the factor algebra itself);
// the stage-1 rows of the reduction follow them
static size_t partial_rows(size_t n) { return std::max<size_t>(kMaxBlocks, (n + 63) / 64); }

static void launch_reduce(sga_context* ctx, const double* partials, int nrows, int ncols, int row_stride, double* stage, double* out, int out_n, double* host, unsigned long long seq, bool derive = false, const uint32_t* stats = nullptr) {
  const int groups = nrows > 256 ? std::min(kReduceGroups, std::max(8, nrows / 128)) : 1;  // <= 256 rows: one workgroup, no hand-off between workgroups
  hipLaunchKernelGGL(reduce_rows_kernel, dim3(groups), dim3(kReduceSlices * kCols), 0, ctx->stream, partials, nrows, ncols, row_stride, stage, ctx->d_ticket.p, out, out_n, host, seq, derive ? 1 : 0, stats);
}

static bool g_lazy_maha = getenv("SGA_LAZY_MAHA") ? atoi(getenv("SGA_LAZY_MAHA")) != 0 : true;
static int g_fuse_max = getenv("SGA_FUSE_MAX") ? atoi(getenv("SGA_FUSE_MAX")) : kFuseMaxBlocks;
// four points per lane in the standalone factor kernel from this many points on (VGICP iterations of 140k / 200k / 300k / 500k / 1M points:
// 32.5 / 32.7 / 36.4 / 39.0 / 45.1 us at four per lane against 25.5 / 27.0 / 30.4 / 37.0 / 45.4 us at one: the size sweep late in round 6;
// 131072 before)
static int g_lin_pts_min = getenv("SGA_LIN_PTS_MIN") ? atoi(getenv("SGA_LIN_PTS_MIN")) : 750000;
static int grid_blocks(int num_tiles) { return num_tiles < kMaxBlocks ? (num_tiles < 1 ? 1 : num_tiles) : kMaxBlocks; }

// nearest neighbour (caller's target order) and squared distance of every source point in the caller's source order: the input of a
// host rejector callback
template <typename Real>
__global__ void export_neighbours_kernel(const float4* __restrict__ src_pts, const int* __restrict__ nn, int n, const float4* __restrict__ tgt_pts, Rigid<Real> T, long long* __restrict__ out_idx, float* __restrict__ out_d2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 ps = src_pts[i];
  const uint32_t orig = __float_as_uint(ps.w);
  const int j = nn[i];
  long long t = -1;
  float d2 = INFINITY;
  if (j >= 0) {
    Real x, y, z;
    transform_point<Real>(T, ps.x, ps.y, ps.z, x, y, z);
    const float4 m = tgt_pts[j];
    t = static_cast<long long>(__float_as_uint(m.w));
    d2 = kd_dist2(m.x, m.y, m.z, static_cast<float>(x), static_cast<float>(y), static_cast<float>(z));
  }
  out_idx[orig] = t;
  out_d2[orig] = d2;
}

// Points per lane and reduction (linearize_kernel's PTS): large clouds amortise the wave reductions over 4 points; small ones (a
// 15k-point scan) keep one point per lane — they are bound by latency and want every workgroup they can get.
constexpr int kLinPts = 4;
// (fp64 per-pair arithmetic keeps four per lane from 131 072 points on: its factor kernel is arithmetic-bound and loses a quarter of its rate at one —
// 200k / 400k / 700k / 1M points 11 047 / 9 236 / 6 768 / 5 389 against 9 632 / 7 069 / 4 921 / 3 673 iterations/s)
static int g_lin_pts_min_f64 = getenv("SGA_LIN_PTS_MIN_F64") ? atoi(getenv("SGA_LIN_PTS_MIN_F64")) : 131072;
static int linearize_pts(int n, bool fp64) { return n >= (fp64 ? g_lin_pts_min_f64 : g_lin_pts_min) ? kLinPts : 1; }

template <typename Real, int FACTOR, int TARGET>
static void launch_linearize(hipStream_t st, const LinParams<Real>& p, int blocks, int pts) {
  if (pts == kLinPts)
    hipLaunchKernelGGL((linearize_kernel<Real, FACTOR, TARGET, kLinPts>), dim3(blocks), dim3(kTile), 0, st, p);
  else
    hipLaunchKernelGGL((linearize_kernel<Real, FACTOR, TARGET, 1>), dim3(blocks), dim3(kTile), 0, st, p);
}

// Largest displacement |Ta p - Tb p| over the box [lo, hi] (column-major 4x4 poses): the norm of an affine map is convex, so the
// maximum sits at a corner.
static double max_displacement(const double Ta[16], const double Tb[16], const float lo[3], const float hi[3]) {
  double best = 0.0;
  for (int c = 0; c < 8; c++) {
    const double x = (c & 1) ? hi[0] : lo[0], y = (c & 2) ? hi[1] : lo[1], z = (c & 4) ? hi[2] : lo[2];
    double s = 0.0;
    for (int r = 0; r < 3; r++) {
      const double d = (Ta[r] - Tb[r]) * x + (Ta[4 + r] - Tb[4 + r]) * y + (Ta[8 + r] - Tb[8 + r]) * z + (Ta[12 + r] - Tb[12 + r]);
      s += d * d;
    }
    best = std::max(best, s);
  }
  return std::sqrt(best);
}

// The lengths the pass routing compares motions with are properties of the TARGET, not constants of nature: a certificate's exclusion
// radius is a fraction of the distance between neighbouring target points, so "how far may the source have moved before the
// certificates are gone" scales with the target's spacing.  The thresholds below were tuned in metres on the C3 scene; they are applied
// multiplied by  unit = (the target's length scale) / kSpacingRef  — index_spacing(): the geometric mean of the tree's leaf diagonals,
// computed by the build (kd_tail_kernel); kSpacingRef: that number for the C3 target — so a cloud in millimetres, or one ten times sparser,
// routes its passes exactly as its metre-scale twin does (tests/test_scale_free.py).  unit = 1 while the length scale of a
// target is not known (only ever before the first result of a problem has come back: the first pass is cold anyway).
constexpr double kSpacingRef = SGA_SPACING_REF;
double index_spacing(const sga_index* idx);  // index_build.hip
static double routing_unit(const sga_index* idx) {
  static const bool scale_free = !(getenv("SGA_SCALE_FREE") && atoi(getenv("SGA_SCALE_FREE")) == 0);
  const double s = scale_free ? index_spacing(idx) : 0.0;
  return s > 0.0 ? s / kSpacingRef : 1.0;
}

// A pass runs warm (certified neighbours skip the walk) while no source point can have moved farther than this since the previous
// linearization; beyond it hardly any certificate holds and checking them is wasted work.  Environment override SGA_WARM_DELTA
// (metres at the reference spacing, see above), run-time override sga_set_warm_limit; negative = never.  Results do not depend on it.
static double g_warm_delta = getenv("SGA_WARM_DELTA") ? atof(getenv("SGA_WARM_DELTA")) : 0.1;

// Search kernel selection (results do not depend on it).  SGA_SEARCH_QUEUE: 0 = one query per lane always (nn_search_kernel),
// 1 = queue-fed always, 2 (default) = queue-fed for warm passes after a motion of at most SGA_QUEUE_DELTA metres.  Measured on C3
// (1 M points): when most lanes walk (cold passes, the first warm pass) a wave per 64 queries and 8 waves per SIMD win — the walks
// are bound by the number of memory accesses in flight and the queue-fed kernel starts each of them with 10 record fetches; when few
// lanes walk (the later passes of a registration) the queue packs them into full waves: 124 -> 80, 87 -> 73, 68 -> 53 us.
static int g_search_queue = getenv("SGA_SEARCH_QUEUE") ? atoi(getenv("SGA_SEARCH_QUEUE")) : 2;
// 1 (default): the search waves evaluate the factors of their own tiles (search_linearize_kernel); 0: always a separate factor kernel (linearize_kernel)
static bool g_fuse_search = getenv("SGA_FUSE_SEARCH") ? atoi(getenv("SGA_FUSE_SEARCH")) != 0 : true;
// longest tile first (tile_order_kernel): 0 off, 1 = warm passes of the one-query-per-lane kernel (motion <= SGA_WARM_DELTA) when the
// previous pass was such a pass too, 2 = every pass of that kernel that follows another (experiments)
static const int g_lpt = getenv("SGA_LPT") ? atoi(getenv("SGA_LPT")) : 1;
static const float g_slack_min = getenv("SGA_SLACK_MIN") ? static_cast<float>(atof(getenv("SGA_SLACK_MIN"))) : 3e-4f;
static const float g_slack_max = getenv("SGA_SLACK_MAX") ? static_cast<float>(atof(getenv("SGA_SLACK_MAX"))) : 0.02f;
// headroom of the certificate check (certify): a point walks unless its certificate survives PAD x its own motion beyond what this pass needs
static const float g_cert_pad = getenv("SGA_CERT_PAD") ? static_cast<float>(atof(getenv("SGA_CERT_PAD"))) : 0.6f;
static double g_queue_delta = getenv("SGA_QUEUE_DELTA") ? atof(getenv("SGA_QUEUE_DELTA")) : 0.02;
// 1 (default): the one-query-per-lane search kernels walk with the fast leaf scan (32-bit keys, packed fp32; exact repeat of the
// queries it cannot decide); 0: the exact 64-bit keys throughout.  Results do not depend on it.
static int g_fast_scan = getenv("SGA_FAST_SCAN") ? atoi(getenv("SGA_FAST_SCAN")) : 1;
static int g_chunk_tiles_cold = getenv("SGA_CHUNK_COLD") ? atoi(getenv("SGA_CHUNK_COLD")) : 4;
static const bool g_chunk_adapt = getenv("SGA_CHUNK_ADAPT") ? atoi(getenv("SGA_CHUNK_ADAPT")) != 0 : true;
static int g_chunk_tiles_warm = getenv("SGA_CHUNK_WARM") ? atoi(getenv("SGA_CHUNK_WARM")) : 4;

int g_grid_mode = getenv("SGA_GRID") ? atoi(getenv("SGA_GRID")) : 1;  // what the cell grid is used for (see linearize_dispatch); sga_set_grid_mode
long long g_grid_min_points = getenv("SGA_GRID_MIN_POINTS") ? atoll(getenv("SGA_GRID_MIN_POINTS")) : 65536;  // targets below this get no grid (cell_grid.hip)

template <typename Real>
static int linearize_dispatch(sga_context* ctx, sga_problem* pb, const sga_factor_params* fp, const double T[16], double* d_out30, double* host, unsigned long long seq, bool with_model = false) {
  const sga_index* idx = pb->target;
  const bool voxel = idx->kind != SGA_INDEX_KDTREE;  // Gaussian or flat voxel map: the lookup happens inside the factor kernel
  const bool flat = idx->kind == SGA_INDEX_FLATMAP;
  if (fp->factor_kind == SGA_GICP && ((pb->n > 0 && !pb->has_covs) || (idx->n > 0 && !idx->has_covs))) return fail(SGA_ERR_INVALID, "GICP needs covariances on both source and target");
  if (fp->factor_kind == SGA_PLANE_ICP && (voxel || (idx->n > 0 && !idx->has_normals))) return fail(SGA_ERR_UNSUPPORTED, "PLANE_ICP needs a kd-tree index over a target with normals");
  if (fp->factor_kind < 0 || fp->factor_kind > 2) return fail(SGA_ERR_INVALID, "invalid factor_kind %d", fp->factor_kind);

  LinParams<Real> p{};
  p.src_pts = pb->src_pts();
  p.src_cov = pb->src_cov();
  p.n = static_cast<int>(pb->n);
  const int pts = linearize_pts(p.n, sizeof(Real) == 8);
  p.num_tiles = (p.n + kTile * pts - 1) / (kTile * pts);  // steps of kTile * pts points
  p.tgt_pts = voxel ? idx->pts.p : idx->kd_pts.p;
  p.tgt_nrm = idx->nrm.p;
  p.tgt_cov = idx->cov.p;
  if (flat) {
    p.flat.hkeys = idx->hkeys.p;
    p.flat.hvals = idx->hvals.p;
    p.flat.hmask = idx->hmask;
    p.flat.inv_leaf = 1.0 / idx->leaf;
    p.flat.vnum = idx->vcounts.p;
    p.flat.offsets = idx->search_offsets;
    for (int k = 0; k < 3; k++) p.flat.org[k] = idx->origin[k];
  } else if (voxel) {
    p.vox.hkeys = idx->hkeys.p;
    p.vox.hvals = idx->hvals.p;
    p.vox.hmask = idx->hmask;
    p.vox.inv_leaf = 1.0 / idx->leaf;
    for (int k = 0; k < 3; k++) p.vox.org[k] = idx->origin[k];
    p.vox.offsets = idx->search_offsets;
  } else {
    p.kd = make_kd_view(idx);
  }
  p.corr = pb->corr.p;
  p.hint = pb->hint.p;
  p.cert_nn2 = (!voxel && sizeof(Real) == 8) ? pb->hint2.p : nullptr;  // fp64 arithmetic: the factor kernel measures the walk's two candidates again, in double
  if constexpr (sizeof(Real) == 4) {
    p.maha = pb->maha.p;
  } else {
    if (pb->maha64.n < pb->n * 6) SGA_TRY(pb->maha64.alloc(pb->n * 6));
    p.maha = pb->maha64.p;
  }
  p.T = rigid_from_colmajor<Real>(T);
  p.store_maha = (fp->robust_kind != SGA_ROBUST_NONE || !g_lazy_maha) ? 1 : 0;  // the error passes of a robust factor run the error kernel
  const bool host_rejector = pb->rejector_fn != nullptr && !voxel;
  p.max_sq = (fp->max_dist_sq < 0 || host_rejector) ? INFINITY : static_cast<float>(fp->max_dist_sq);  // a user rejector sees every nearest neighbour
  p.bound2 = p.max_sq < 3.0e38f ? p.max_sq * 1.0000002f : INFINITY;  // d2 == max_sq must still be found (strict '>' rejector)
  p.robust_kind = fp->robust_kind;
  p.robust_c = static_cast<Real>(fp->robust_c);
  p.partials = pb->partials.p;
  const int blocks = grid_blocks(p.num_tiles);
  const bool fuse = p.n > 0 && blocks <= g_fuse_max;
  const int ncols = kModelCols, out_n = with_model ? kRow : SGA_ACCUM_DOUBLES;  // the caller's buffer: the system, or the system + its moments
  p.tail = FusedTail{fuse ? 1 : 0, ctx->d_ticket.p, d_out30, out_n, host, seq};

  // warm pass?  Only with certificates from a previous pass in the same arithmetic (the queries must be bit-identical), and only
  // while no source point can have moved farther than the certificates can possibly cover.
  const int math = sizeof(Real) == 4 ? SGA_MATH_FP32 : SGA_MATH_FP64;
  const double displacement = (!voxel && p.n > 0 && pb->prev_valid && pb->prev_math == math) ? max_displacement(T, pb->T_prev, pb->bbox_lo, pb->bbox_hi) : INFINITY;
  const double unit = voxel ? 1.0 : routing_unit(idx);  // the target's own length scale relative to the scene the thresholds were tuned on
  bool warm = displacement <= g_warm_delta * unit;
  // The cell grid (cell_grid.hpp).  SGA_GRID: 0 no grid at all; 1 (default) the WALKERS of warm passes try ring 1 of the grid before they
  // walk the tree (certify_linearize_kernel) and every full search stays with the kd walk — measured on C3 whole grid passes do not beat
  // it; 2 = also the cold passes of a registration but its first (whose queries lie as far from the target as the
  // initial guess is off: there the kd walk's pruning pays) and the warm passes after a larger motion; 3 the first pass too; 4 every pass.
  // If ring 1 of the last grid pass left more than SGA_GRID_MAX_OPEN of the queries open, the next cold pass walks the kd-tree instead.
  const int grid_mode = g_grid_mode;
  static const int grid_max_rings = getenv("SGA_GRID_MAX_RINGS") ? atoi(getenv("SGA_GRID_MAX_RINGS")) : 16;
  static const double grid_max_open = getenv("SGA_GRID_MAX_OPEN") ? atof(getenv("SGA_GRID_MAX_OPEN")) : 0.5;
  const bool first_pass = !pb->prev_valid;
  bool use_grid = false;

  const bool timed = ctx->profiling && (ctx->lin_seq++ % ctx->profile_period) == 0;  // sampled: event records cost ~7 us each
  if (timed) {
    sga_profile_collect_pending(ctx);
    (void)hipEventRecord(ctx->ev0, ctx->stream);
    ctx->pending_warm = warm;
  }
  bool split_fused_tail = false;  // certify_linearize_kernel of a small grid: the row reduction and the hand-off happen inside it
  bool fused_search = false;  // the search kernel evaluates the factors itself ...
  bool record_tiles = false;  // ... and records the duration of every tile's wave (longest tile first, tile_order_kernel)
  int fused_rows = 0;         // ... and leaves this many partial rows
  // The search kernels rewrite hint / hint2 / rex for THIS pose: until the pass has been launched completely the certificates belong
  // to no pose the host knows, so an early return below (allocation, rejector callback, launch error) must not leave them marked
  // valid for T_prev.
  if (!voxel) pb->prev_valid = false;
  pb->state_fresh = false;
  if (p.n > 0 && !voxel) {
    NNParams<Real> q{};
    q.src_pts = pb->src_pts();
    q.n = p.n;
    q.kd = p.kd;
    q.T = p.T;
    q.within2 = p.bound2;
    q.slack_min = static_cast<float>(g_slack_min * unit), q.slack_max = static_cast<float>(g_slack_max * unit);
    q.cert_pad = std::max(g_cert_pad, 0.f);
    q.bound2 = p.bound2 * (1.f + kSearchMargin) * (1.f + kSearchMargin);
    q.nn = pb->hint.p;
    q.nn2 = pb->hint2.p;
    q.rex = pb->rex.p;
    q.check = warm ? 1 : 0;
    q.fast = g_fast_scan;
    if (warm) q.T_prev = rigid_from_colmajor<Real>(pb->T_prev);
    q.walked = pb->walked.p;
    q.leaves = pb->dbg_leaves.n >= pb->n ? pb->dbg_leaves.p : nullptr;
    if (q.leaves != nullptr) SGA_HIP(hipMemsetAsync(q.leaves, 0, pb->n * sizeof(int), ctx->stream));
    const size_t words = static_cast<size_t>(p.kd.depth);  // traversal stack rows
    const dim3 sgrid((p.n + kSearchBlock - 1) / kSearchBlock), sblock(kSearchBlock);
    if (grid_mode != 0 && idx->grid_h > 0.f && q.bound2 < 3.0e38f && !host_rejector && q.leaves == nullptr) {
      const int rings = grid_rings_for(idx, std::sqrt(static_cast<double>(q.bound2)));
      const bool small_warm = warm && displacement <= g_queue_delta * unit;
      use_grid = grid_mode >= 2 && rings > 0 && rings <= grid_max_rings &&
                 (grid_mode >= 4 || (!small_warm && (grid_mode == 3 || (!first_pass && pb->grid_open_frac <= grid_max_open))));
      // which walkers try ring 1 of the grid first (SGA_GRID_WALK, bits): 1 those of certify_linearize_kernel, 2 those of the
      // one-query-per-lane warm pass (search_linearize_kernel<CHECK>), 4 those of the queue-fed kernel, 8 certify_linearize_kernel scans ring 1
      // with a group of lanes per walker when a workgroup has few of them (grid_ring1_group); 32: a ring certificate that ends at the ring's
      // face with less than the re-walk's slack beyond the neighbour is left to the tree walk (the margin rule, certify_linearize_kernel)
      static const int grid_walk_bits = getenv("SGA_GRID_WALK") ? atoi(getenv("SGA_GRID_WALK")) : 47;
      q.grid_walk = grid_walk_bits;
      q.grid = make_grid_view(idx);
      if (first_pass || (!use_grid && !warm)) pb->grid_open_frac = 0.0;  // a kd pass in between: the grid gets another chance afterwards
    }
    if (use_grid) {
      warm = false;  // a full search of every point: the pass counts as cold
      if (timed) ctx->pending_warm = false;
      if (pb->grid_stats.n < 4) {
        SGA_TRY(pb->grid_stats.alloc(4));
        SGA_HIP(hipMemsetAsync(pb->grid_stats.p, 0, 4 * sizeof(uint32_t), ctx->stream));
      }
      SGA_TRY(grid_search_pass<Real>(ctx, idx, pb->src_pts(), p.n, p.T, q.bound2, pb->hint.p, pb->hint2.p, pb->rex.p, pb->grid_stats.p));
      pb->order_tiles = 0;
      pb->grid_passes++;
    }
    const bool queue = !use_grid && (g_search_queue == 1 || (g_search_queue == 2 && warm && displacement <= g_queue_delta * unit));
    // Warm pass of a large cloud after a small motion: the certificates are checked inside the streaming factor kernel
    // (certify_linearize_kernel), whose workgroups walk their few failed points themselves.  SGA_WARM_SPLIT=0: the older form
    // (certificate check, queue-fed walks and factors per chunk in nn_search_queue_kernel).
    static const bool warm_split = getenv("SGA_WARM_SPLIT") ? atoi(getenv("SGA_WARM_SPLIT")) != 0 : true;
    // Measured on C3 (profiles/r04_warm_split.txt): it wins where few points walk (after motions of a millimetre or two: 38 - 54 us
    // against 44 - 66) and loses where many do (the walkers of a pass sit where the points moved most, i.e. in a few workgroups, which
    // then walk alone: 121 against 99 us after a 1 cm motion), hence the second, smaller limit SGA_SPLIT_DELTA.
    static const double split_delta = getenv("SGA_SPLIT_DELTA") ? atof(getenv("SGA_SPLIT_DELTA")) : 0.002;
    // Only from SGA_SPLIT_MIN_POINTS source points on (default 262144; 131072 until the size sweep late in round 6: late passes of 135k / 200k /
    // 300k / 400k / 600k-point pairs 27.3 / 29.1 / 33.8 / 35.0 / 41.0 us with this kernel against 24.6 / 26.6 / 34.2 / 38.5 / 43.2 us with the
    // queue-fed one: the crossover is near 300k): measured on C2 (100k points) and C5 (12k-point
    // scans) the kernel — one point per lane there, the row reduction folded into it (fused_tail) when the grid is small — is no faster
    // than the queue-fed one (C2 warm pass 39.9 against 40.3 us, C5 registration 0.51 against 0.48 ms/scan; with 4 points per lane 53.9 us):
    // at those sizes a pass is a chain of launch, a few dependent loads and the hand-off, whichever kernel runs it.
    static const size_t split_min_points = getenv("SGA_SPLIT_MIN_POINTS") ? static_cast<size_t>(atoll(getenv("SGA_SPLIT_MIN_POINTS"))) : 262144;
    static const int split_pts_env = getenv("SGA_SPLIT_PTS") ? atoi(getenv("SGA_SPLIT_PTS")) : 0;
    const bool split = warm_split && queue && warm && displacement <= split_delta * unit && g_search_queue == 2 && g_fuse_search && !host_rejector && sizeof(Real) == 4 && pb->n >= split_min_points && q.leaves == nullptr;
    fused_search = !use_grid && g_fuse_search && !host_rejector && !queue && sizeof(Real) == 4;  // fp64 math: the fused kernel would spill
    const unsigned order_tiles_before = pb->order_tiles;
    pb->order_tiles = 0;  // (set again below when this pass records its tiles' durations)
    if (use_grid) {
      // searched above; the factors follow as linearize_kernel over nn[]
    } else if (split) {
      // one point per lane unless told otherwise (SGA_SPLIT_PTS=4): the size sweep late in round 6 — late passes of 300k / 600k / 700k-point pairs
      // 34.0 / 40.4 / 42.3 us at four per lane against 31.2 / 38.8 / 40.6 us at one, C3's warm passes 83.8 -> 81.2 us on average (+1.2 % on the headline)
      const int spts = split_pts_env == 1 || split_pts_env == 4 ? split_pts_env : 1;
      LinParams<Real>& pc = p;
      pc.num_tiles = (p.n + kTile * spts - 1) / (kTile * spts);
      const int cblocks = grid_blocks(pc.num_tiles);
      // (never at its sizes since kFuseMaxBlocks is 32: 128 - 256 workgroups taking the tail's ticket one after the other cost the late passes
      // of a 250k-point source 28 us, 59 against 31: scripts/diag_shards.py, N = 4)
      const bool cfuse = cblocks <= g_fuse_max;
      pc.tail = FusedTail{cfuse ? 1 : 0, ctx->d_ticket.p, d_out30, out_n, host, seq};
      split_fused_tail = cfuse;
      p.cert_nn = pb->hint.p;
      p.cert_nn2 = pb->hint2.p;
      p.cert_rex = pb->rex.p;
      p.cert_walked = pb->walked.p;
      p.T_prev = q.T_prev;
      p.cert_within2 = q.within2;
      p.cert_slack_min = q.slack_min, p.cert_slack_max = q.slack_max;
      p.cert_pad = q.cert_pad;
      const size_t lds = std::max<size_t>(words, 1) * 64 * sizeof(uint32_t) * (kTile / 64);
#define SGA_CERTIFY(F)                                                                                                                      \
  do {                                                                                                                                      \
    if (spts == kLinPts)                                                                                                                    \
      hipLaunchKernelGGL((certify_linearize_kernel<Real, F, kLinPts>), dim3(cblocks), dim3(kTile), lds, ctx->stream, p, q);  \
    else                                                                                                                                    \
      hipLaunchKernelGGL((certify_linearize_kernel<Real, F, 1>), dim3(cblocks), dim3(kTile), lds, ctx->stream, p, q);        \
  } while (0)
      switch (fp->factor_kind) {
        case SGA_GICP: SGA_CERTIFY(SGA_GICP); break;
        case SGA_PLANE_ICP: SGA_CERTIFY(SGA_PLANE_ICP); break;
        default: SGA_CERTIFY(SGA_ICP); break;
      }
#undef SGA_CERTIFY
      fused_search = true;
      fused_rows = cblocks;
    } else if (fused_search) {
      // every search wave evaluates the factors of its own tile: one partial row per tile of 64 points, summed by reduce_rows_kernel
      p.tail.enabled = 0;
      fused_rows = static_cast<int>(sgrid.x);
      const dim3 cgrid(sgrid.x);
      if (g_lpt != 0 && sgrid.x >= 8192) {  // fewer tiles than wave slots: all waves start at once, the order means nothing
        SGA_TRY(pb->tile_cost.reserve(sgrid.x));
        SGA_TRY(pb->tile_order.reserve(sgrid.x));
        q.tile_cost = pb->tile_cost.p;
        // (the order was written behind the previous pass's hand-off on that pass's stream: a pass issued on another stream could read it half-written)
        if (order_tiles_before == sgrid.x && pb->order_stream == ctx->stream && (g_lpt == 2 || warm)) q.tile_order = pb->tile_order.p;
        record_tiles = true;
      }
      static const size_t lds_pad = getenv("SGA_LDS_PAD") ? static_cast<size_t>(atoi(getenv("SGA_LDS_PAD"))) : 0;  // experiments: bytes of unused LDS per wave (lowers the occupancy)
      const size_t lds = std::max<size_t>(words, 3) * 64 * sizeof(uint32_t) + lds_pad;
      switch (fp->factor_kind) {
        case SGA_GICP:
          if (warm) hipLaunchKernelGGL((search_linearize_kernel<Real, SGA_GICP, true>), cgrid, sblock, lds, ctx->stream, q, p);
          else hipLaunchKernelGGL((search_linearize_kernel<Real, SGA_GICP, false>), cgrid, sblock, lds, ctx->stream, q, p);
          break;
        case SGA_PLANE_ICP:
          if (warm) hipLaunchKernelGGL((search_linearize_kernel<Real, SGA_PLANE_ICP, true>), cgrid, sblock, lds, ctx->stream, q, p);
          else hipLaunchKernelGGL((search_linearize_kernel<Real, SGA_PLANE_ICP, false>), cgrid, sblock, lds, ctx->stream, q, p);
          break;
        default:
          if (warm) hipLaunchKernelGGL((search_linearize_kernel<Real, SGA_ICP, true>), cgrid, sblock, lds, ctx->stream, q, p);
          else hipLaunchKernelGGL((search_linearize_kernel<Real, SGA_ICP, false>), cgrid, sblock, lds, ctx->stream, q, p);
          break;
      }
    } else if (queue) {
      q.inv_leaf = p.kd.n > 0 ? std::ldexp(1.0, p.kd.depth) / static_cast<double>(p.kd.n) : 0.0;
      q.chunk_tiles = std::max(1, warm ? g_chunk_tiles_warm : g_chunk_tiles_cold);
      // a wave works through its chunk serially (two dependent round trips per tile): small clouds get shorter chunks, down to one tile,
      // as long as that leaves no more than ~4 waves per SIMD (kWaveSlots4 = 256 CUs x 4 SIMDs x 4); C3 keeps its 4 tiles
      if (warm && g_chunk_adapt) q.chunk_tiles = std::min(q.chunk_tiles, std::max(1, (static_cast<int>(sgrid.x) + 4095) / 4096));
      const dim3 qgrid((sgrid.x + q.chunk_tiles - 1) / q.chunk_tiles);
      const size_t lds = std::max<size_t>(words, 3) * 64 * sizeof(uint32_t);
      fused_search = g_fuse_search && !host_rejector && warm && sizeof(Real) == 4;
      if (fused_search) {  // the chunk's wave evaluates the factors as well: one partial row per chunk
        p.tail.enabled = 0;
        fused_rows = static_cast<int>(qgrid.x);
        const dim3 cgrid(qgrid.x);
        switch (fp->factor_kind) {
          case SGA_GICP: hipLaunchKernelGGL((nn_search_queue_kernel<Real, true, SGA_GICP>), cgrid, sblock, lds, ctx->stream, q, p); break;
          case SGA_PLANE_ICP: hipLaunchKernelGGL((nn_search_queue_kernel<Real, true, SGA_PLANE_ICP>), cgrid, sblock, lds, ctx->stream, q, p); break;
          default: hipLaunchKernelGGL((nn_search_queue_kernel<Real, true, SGA_ICP>), cgrid, sblock, lds, ctx->stream, q, p); break;
        }
      } else if (warm)
        hipLaunchKernelGGL((nn_search_queue_kernel<Real, true>), qgrid, sblock, lds, ctx->stream, q, p);
      else
        hipLaunchKernelGGL((nn_search_queue_kernel<Real, false>), qgrid, sblock, lds, ctx->stream, q, p);
    } else if (warm)
      hipLaunchKernelGGL((nn_search_kernel<Real, kSearchBlock, true>), sgrid, sblock, words * kSearchBlock * sizeof(uint32_t), ctx->stream, q);
    else
      hipLaunchKernelGGL((nn_search_kernel<Real, kSearchBlock, false>), sgrid, sblock, words * kSearchBlock * sizeof(uint32_t), ctx->stream, q);
    if (timed) {
      (void)hipEventRecord(ctx->ev_mid, ctx->stream);
      ctx->mid_recorded = true;
    }
    if (host_rejector) {  // neighbours down, verdicts up (rejector.hpp:11-28 through sga_rejector_fn)
      const size_t n = pb->n;
      DevBuf<long long> d_idx;
      DevBuf<float> d_d2;
      SGA_TRY(d_idx.alloc(n));
      SGA_TRY(d_d2.alloc(n));
      SGA_TRY(pb->reject.reserve(n));
      hipLaunchKernelGGL((export_neighbours_kernel<Real>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, pb->src_pts(), pb->hint.p, p.n, p.tgt_pts, p.T, d_idx.p, d_d2.p);
      SGA_HIP(hipGetLastError());
      std::vector<int64_t> h_idx(n);
      std::vector<float> h_d2(n);
      std::vector<unsigned char> h_rej(n, 0);
      SGA_HIP(hipMemcpyAsync(h_idx.data(), d_idx.p, n * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
      SGA_HIP(hipMemcpyAsync(h_d2.data(), d_d2.p, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
      SGA_HIP(hipStreamSynchronize(ctx->stream));
      int cb_rc;
      {
        StreamScope outside(nullptr);  // user code: objects it destroys (a Python GC run) are not stream-ordered frees of this entry point
        cb_rc = pb->rejector_fn(pb->rejector_user, pb->caller_T ? pb->caller_T : T, n, h_idx.data(), h_d2.data(), h_rej.data());  // the rejector sees the CALLER's pose
      }
      if (cb_rc != 0) return fail(SGA_ERR_CALLBACK, "rejector callback failed");
      SGA_HIP(hipMemcpyAsync(pb->reject.p, h_rej.data(), n, hipMemcpyHostToDevice, ctx->stream));
      SGA_HIP(hipStreamSynchronize(ctx->stream));  // h_rej goes out of scope
      p.reject = pb->reject.p;
    }
  }
  if (p.n > 0 && !fused_search) {
    if (flat) {
      if (fp->factor_kind == SGA_GICP)
        launch_linearize<Real, SGA_GICP, 2>(ctx->stream, p, blocks, pts);
      else
        launch_linearize<Real, SGA_ICP, 2>(ctx->stream, p, blocks, pts);
    } else if (voxel) {
      if (fp->factor_kind == SGA_GICP)
        launch_linearize<Real, SGA_GICP, 1>(ctx->stream, p, blocks, pts);
      else
        launch_linearize<Real, SGA_ICP, 1>(ctx->stream, p, blocks, pts);
    } else {
      switch (fp->factor_kind) {
        case SGA_GICP: launch_linearize<Real, SGA_GICP, 0>(ctx->stream, p, blocks, pts); break;
        case SGA_PLANE_ICP: launch_linearize<Real, SGA_PLANE_ICP, 0>(ctx->stream, p, blocks, pts); break;
        default: launch_linearize<Real, SGA_ICP, 0>(ctx->stream, p, blocks, pts); break;
      }
    }
  }
  if (fused_search && split_fused_tail) {
    // (done inside the kernel: fused tail of a small grid)
  } else if (fused_search) {
    launch_reduce(ctx, pb->partials.p, fused_rows, ncols, kRow, pb->partials.p + partial_rows(pb->n) * kRow, d_out30, out_n, host, seq, true);
  }
  else if (!fuse)
    launch_reduce(ctx, pb->partials.p, p.n > 0 ? blocks : 0, ncols, kRow, pb->partials.p + partial_rows(pb->n) * kRow, d_out30, out_n, host, seq, true, use_grid ? pb->grid_stats.p : nullptr);
  pb->grid_stats_pending = use_grid && !fuse && out_n > kStatsCol + 1;
  if (timed) {  // the whole GPU side of the pass: search + factors + the sum of the rows
    (void)hipEventRecord(ctx->ev1, ctx->stream);
    ctx->pending |= 1;
  }
  if (record_tiles) {  // after the reduction has handed the result to the host: the tiles are sorted while the host solves
    hipLaunchKernelGGL(tile_order_kernel, dim3(8), dim3(1024), 0, ctx->stream, pb->tile_cost.p, pb->tile_order.p, fused_rows);
    pb->order_tiles = static_cast<unsigned>(fused_rows);
    pb->order_stream = ctx->stream;
  }
  SGA_HIP(hipGetLastError());
  pb->last_math = math;
  pb->lin_factor = fp->factor_kind;
  memcpy(pb->lin_T, T, sizeof(pb->lin_T));
  pb->maha_valid = p.store_maha != 0;
  if (!voxel) {
    memcpy(pb->T_prev, T, sizeof(pb->T_prev));
    pb->prev_valid = true;
    pb->prev_math = math;
    if (warm)
      pb->warm_passes++;
    else
      pb->cold_passes++;
  }
  return SGA_OK;
}

// The mahalanobis matrices of the last linearization, (C_t + R C_s R^T)^-1 per accepted pair (gicp_factor.hpp:57-60), written on
// demand: the hot loop of a non-robust registration never reads them (the error model answers the error passes), so the factor
// kernels skip the 24-byte store per point; the error kernel and sga_problem_get_factors call this first.
template <typename Real>
__global__ void recompute_maha_kernel(const Cov8* __restrict__ src_cov, const Cov8* __restrict__ tgt_cov, const int* __restrict__ corr, int n, Rigid<Real> T, Real* __restrict__ maha) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int j = corr[i];
  if (j < 0) return;
  const Sym3<Real> Cs = load_sym<Real>(src_cov, i);
  const Sym3<Real> Ct = load_sym<Real>(tgt_cov, j);
  const Sym3<Real> RCR = rotate_sym(T.r, Cs);
  const Sym3<Real> M = inverse_sym<Real>({Ct.xx + RCR.xx, Ct.xy + RCR.xy, Ct.xz + RCR.xz, Ct.yy + RCR.yy, Ct.yz + RCR.yz, Ct.zz + RCR.zz});
  Real* m = maha + static_cast<size_t>(i) * 6;
  m[0] = M.xx, m[1] = M.xy, m[2] = M.xz, m[3] = M.yy, m[4] = M.yz, m[5] = M.zz;
}

int problem_ensure_maha(sga_context* ctx, sga_problem* pb) {
  if (pb->maha_valid || pb->n == 0 || pb->lin_factor != SGA_GICP) return SGA_OK;
  const sga_index* idx = pb->target;
  const int n = static_cast<int>(pb->n);
  if (pb->last_math == SGA_MATH_FP64) {
    if (pb->maha64.n < pb->n * 6) SGA_TRY(pb->maha64.alloc(pb->n * 6));
    hipLaunchKernelGGL((recompute_maha_kernel<double>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, pb->src_cov(), idx->cov.p, pb->corr.p, n, rigid_from_colmajor<double>(pb->lin_T), pb->maha64.p);
  } else {
    hipLaunchKernelGGL((recompute_maha_kernel<float>), dim3((n + 255) / 256), dim3(256), 0, ctx->stream, pb->src_cov(), idx->cov.p, pb->corr.p, n, rigid_from_colmajor<float>(pb->lin_T), pb->maha.p);
  }
  SGA_HIP(hipGetLastError());
  pb->maha_valid = true;
  return SGA_OK;
}

template <typename Real>
static int error_dispatch(sga_context* ctx, sga_problem* pb, const sga_factor_params* fp, const double T[16], double* d_out1, double* host, unsigned long long seq) {
  const sga_index* idx = pb->target;
  ErrParams<Real> p{};
  p.src_pts = pb->src_pts();
  p.n = static_cast<int>(pb->n);
  p.num_tiles = (p.n + kTile - 1) / kTile;
  p.tgt_pts = idx->kind != SGA_INDEX_KDTREE ? idx->pts.p : idx->kd_pts.p;
  p.tgt_nrm = idx->nrm.p;
  p.corr = pb->corr.p;
  if (fp->factor_kind == SGA_GICP) {
    if ((sizeof(Real) == 4) != (pb->last_math == SGA_MATH_FP32)) return fail(SGA_ERR_INVALID, "sga_error in another arithmetic than the last sga_linearize");
    SGA_TRY(problem_ensure_maha(ctx, pb));
  }
  if constexpr (sizeof(Real) == 4) {
    p.maha = pb->maha.p;
  } else {
    if (fp->factor_kind == SGA_GICP && pb->maha64.n < pb->n * 6) return fail(SGA_ERR_INVALID, "sga_error(fp64) before sga_linearize(fp64)");
    p.maha = pb->maha64.p;
  }
  p.T = rigid_from_colmajor<Real>(T);
  p.robust_kind = fp->robust_kind;
  p.robust_c = static_cast<Real>(fp->robust_c);
  p.partials = pb->partials.p;
  const int blocks = grid_blocks(p.num_tiles);
  const bool fuse = p.n > 0 && blocks <= g_fuse_max;
  p.tail = FusedTail{fuse ? 1 : 0, ctx->d_ticket.p, d_out1, 1, host, seq};
  if (fp->factor_kind == SGA_PLANE_ICP && !idx->has_normals) return fail(SGA_ERR_UNSUPPORTED, "PLANE_ICP needs target normals");
  const bool timed = ctx->profiling && (ctx->err_seq++ % ctx->profile_period) == 0;
  if (timed) {
    sga_profile_collect_pending(ctx);
    (void)hipEventRecord(ctx->ev2, ctx->stream);
  }
  if (p.n > 0) {
    switch (fp->factor_kind) {
      case SGA_GICP: hipLaunchKernelGGL((error_kernel<Real, SGA_GICP>), dim3(blocks), dim3(kTile), 0, ctx->stream, p); break;
      case SGA_PLANE_ICP: hipLaunchKernelGGL((error_kernel<Real, SGA_PLANE_ICP>), dim3(blocks), dim3(kTile), 0, ctx->stream, p); break;
      case SGA_ICP: hipLaunchKernelGGL((error_kernel<Real, SGA_ICP>), dim3(blocks), dim3(kTile), 0, ctx->stream, p); break;
      default: return fail(SGA_ERR_INVALID, "invalid factor_kind %d", fp->factor_kind);
    }
  }
  if (timed) {
    (void)hipEventRecord(ctx->ev3, ctx->stream);
    ctx->pending |= 2;
  }
  if (!fuse) launch_reduce(ctx, pb->partials.p, p.n > 0 ? blocks : 0, 1, 1, pb->partials.p + partial_rows(pb->n) * kRow, d_out1, 1, host, seq);
  SGA_HIP(hipGetLastError());
  return SGA_OK;
}

size_t problem_partials_doubles(size_t n) { return partial_rows(n) * kRow + static_cast<size_t>(kReduceGroups) * kCols; }

// Hand `count` doubles to the host after an all-reduce (see reduce_rows_kernel for the protocol).
__global__ void publish_kernel(const double* __restrict__ src, int count, double* __restrict__ host, unsigned long long seq) {
  if (static_cast<int>(threadIdx.x) < count) host[threadIdx.x] = src[threadIdx.x];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(host + kSeqWord), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

static int wait_result(sga_context* ctx, unsigned long long seq) {
  const volatile unsigned long long* flag = reinterpret_cast<const volatile unsigned long long*>(ctx->h_accum + kSeqWord);
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; spins++) {
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return SGA_OK;
    __builtin_ia32_pause();
    if (spins > 200000u) std::this_thread::yield();  // a result normally arrives within tens of microseconds; past ~1 ms stop hogging the core
    if ((spins & 0xfffu) == 0xfffu) {
      // the stream has drained without publishing (a fault), or this is taking implausibly long: let the runtime report it
      const hipError_t q = hipStreamQuery(ctx->stream);
      if (q != hipErrorNotReady || std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
        SGA_HIP(hipStreamSynchronize(ctx->stream));
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return SGA_OK;
        // a faulted or aborted launch may have left the arrival counter of reduce_rows_kernel non-zero: the next result on this
        // context would never be published
        (void)hipMemsetAsync(ctx->d_ticket.p, 0, ctx->d_ticket.n * sizeof(unsigned), ctx->stream);
        return fail(SGA_ERR_HIP, "result was not published by the device");
      }
    }
  }
}

// single GPU: the final reduction kernel publishes; with a communicator the all-reduce comes first
static int fetch_result(sga_context* ctx, const double* d_src, int count, unsigned long long seq, bool published_by_kernel) {
  if (!published_by_kernel) {
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(kCols), 0, ctx->stream, d_src, count, ctx->h_accum_dev, seq);
    SGA_HIP(hipGetLastError());
  }
  return wait_result(ctx, seq);
}

static int check_args(sga_context* ctx, sga_problem* pb, const sga_factor_params* fp, const double* T) {
  if (!ctx || !pb || !fp || !T) return fail(SGA_ERR_INVALID, "null argument");
  if (pb->device != ctx->device) return fail(SGA_ERR_INVALID, "problem lives on device %d, context on %d", pb->device, ctx->device);
  return SGA_OK;
}

}  // namespace sga
// Fold the HIP-event pair of the previous launch into the running averages (no-op while that launch is still in flight).
void sga_profile_collect_pending(sga_context* ctx) {
  // Called before the next launch of either kind: the previous launches have long finished (their results were consumed), the
  // event synchronisation only covers the short gap between the result flag and the event's own completion signal.
  if (!ctx->profiling || ctx->pending == 0) return;
  float ms = 0.f;
  if (ctx->pending & 1) {
    if (hipEventSynchronize(ctx->ev1) == hipSuccess && hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) {
      ctx->lin_ms += ms;
      ctx->lin_calls++;
      if (ctx->pending_warm) {
        ctx->warm_ms += ms;
        ctx->warm_calls++;
      } else {
        ctx->cold_ms += ms;
        ctx->cold_calls++;
      }
      if (ctx->mid_recorded && hipEventElapsedTime(&ms, ctx->ev0, ctx->ev_mid) == hipSuccess) {
        ctx->search_ms += ms;
        ctx->search_calls++;
        if (ctx->pending_warm) ctx->warm_first_ms += ms;
      }
    }
    ctx->mid_recorded = false;
    if (ctx->comm_recorded && hipEventSynchronize(ctx->ev_comm) == hipSuccess && hipEventElapsedTime(&ms, ctx->ev1, ctx->ev_comm) == hipSuccess) {
      ctx->comm_ms += ms;
      ctx->comm_calls++;
    }
    ctx->comm_recorded = false;
  }
  if (ctx->pending & 2) {
    if (hipEventSynchronize(ctx->ev3) == hipSuccess && hipEventElapsedTime(&ms, ctx->ev2, ctx->ev3) == hipSuccess) {
      ctx->err_ms += ms;
      ctx->err_calls++;
    }
  }
  ctx->pending = 0;
}
namespace sga {

}  // namespace sga

namespace sga {
// ---- device frames (common.hpp): what crosses the boundary is converted HERE, everything below works between the two device frames ----
bool problem_framed(const sga_problem* pb) { return !origin_is_zero(pb->src_origin) || !origin_is_zero(pb->target->origin); }
// the caller's pose -> the same rigid motion between the source's and the target's device frames; returns Td (or T itself when both origins are 0)
const double* problem_pose(const sga_problem* pb, const double T[16], double Td[16]) {
  if (!problem_framed(pb)) return T;
  pose_to_device(T, pb->src_origin, pb->target->origin, Td);
  return Td;
}
void problem_system_to_caller(const sga_problem* pb, double H[36], double b[6]) {
  if (!origin_is_zero(pb->src_origin)) system_to_caller(pb->src_origin, H, b);
}
// the 30-double accumulator of the asynchronous entry points, in place on the device (one thread: a 6x6 congruence)
__global__ void frame_accumulator_kernel(double* __restrict__ acc, double ox, double oy, double oz) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double H[36], b[6];
  int k = 0;
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++) {
      H[6 * i + j] = H[6 * j + i] = acc[k];
      k++;
    }
  for (int i = 0; i < 6; i++) b[i] = acc[21 + i];
  const double X[3][3] = {{0, oz, -oy}, {-oz, 0, ox}, {oy, -ox, 0}};  // -skew(o), see system_to_caller (context.hip)
  double HA[6][6];
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      double v = H[6 * i + j];
      if (j < 3)
        for (int c = 0; c < 3; c++) v += H[6 * i + 3 + c] * X[c][j];
      HA[i][j] = v;
    }
  k = 0;
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++) {
      double v = HA[i][j];
      if (i < 3)
        for (int c = 0; c < 3; c++) v += X[c][i] * HA[3 + c][j];
      acc[k++] = v;
    }
  for (int i = 0; i < 3; i++) {
    double v = b[i];
    for (int c = 0; c < 3; c++) v += X[c][i] * b[3 + c];
    acc[21 + i] = v;
  }
}
// Sharded contexts add the ranks' accumulators: their moments are sums over source points in the SOURCE's device frame, so every rank's
// shard must live in the same one (slices of one uploaded cloud do: sga_cloud_slice; separately uploaded shards name a common origin:
// sga_cloud_create_*_origin).  Checked once per problem with one small sum over the ranks: all equal <=> n sum(o^2) == (sum o)^2.
// The ranks of a sharded registration must share ONE source frame (their accumulators are added).  Compared with one all-reduce of sums
// that are EXACT for any number of ranks and any origin (n * sum(o^2) == (sum o)^2 on the doubles themselves rounds differently
// on both sides for 3+ ranks and non-round origins): every origin coordinate is cut into four 16-bit pieces of its bit pattern, each an
// integer < 2^16, and the ranks add the pieces p and their squares p^2.  All of those sums are integers below 2^53 for up to 2^10 ranks, so
// they do not depend on the order of the additions, every rank reads the same numbers and takes the same decision, and by the equality case
// of Cauchy-Schwarz  n * sum(p^2) == (sum p)^2  holds iff all ranks hold the same piece.
constexpr int kFrameCheckDoubles = SGA_FRAME_CHECK_DOUBLES;
void shard_frame_pack(const double origin[3], double out[kFrameCheckDoubles]) {
  for (int k = 0; k < 3; k++) {
    const double o = origin[k] + 0.0;  // -0.0 and +0.0 are one origin
    unsigned long long bits;
    memcpy(&bits, &o, sizeof(bits));
    for (int j = 0; j < 4; j++) {
      const double piece = static_cast<double>((bits >> (16 * j)) & 0xffffull);
      out[4 * k + j] = piece;
      out[12 + 4 * k + j] = piece * piece;
    }
  }
  out[24] = 1.0;  // the number of ranks
  for (int i = 25; i < kFrameCheckDoubles; i++) out[i] = 0.0;
}
bool shard_frame_agree(const double sum[kFrameCheckDoubles]) {
  for (int i = 0; i < 12; i++)
    if (sum[24] * sum[12 + i] != sum[i] * sum[i]) return false;
  return true;
}
int problem_check_shard_frames(sga_context* ctx, sga_problem* pb) {
  if (pb->frame_checked || !ctx->sharded()) return SGA_OK;
  double h[kFrameCheckDoubles];
  shard_frame_pack(pb->src_origin, h);
  DevBuf<double> d;
  SGA_TRY(d.alloc(kFrameCheckDoubles));
  SGA_HIP(hipMemcpyAsync(d.p, h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
  SGA_HIP(hipStreamSynchronize(ctx->stream));
  SGA_TRY(comm_allreduce_sum(ctx, d.p, kFrameCheckDoubles));
  SGA_HIP(hipMemcpyAsync(h, d.p, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  SGA_HIP(hipStreamSynchronize(ctx->stream));
  if (!shard_frame_agree(h))
    return fail(SGA_ERR_INVALID, "the source shards of the ranks live in different device frames (origins differ): slice ONE uploaded cloud (sga_cloud_slice) or upload the shards with a common origin (sga_cloud_create_f64_origin)");
  pb->frame_checked = true;
  return SGA_OK;
}
}  // namespace sga

using namespace sga;

static bool g_error_model = getenv("SGA_ERROR_MODEL") ? atoi(getenv("SGA_ERROR_MODEL")) != 0 : true;

extern "C" {

void sga_debug_shard_frame_pack(const double origin[3], double out[SGA_FRAME_CHECK_DOUBLES]) { shard_frame_pack(origin, out); }
int sga_debug_shard_frame_agree(const double sum[SGA_FRAME_CHECK_DOUBLES]) { return shard_frame_agree(sum) ? 1 : 0; }

void sga_unpack_accumulator(const double acc[SGA_ACCUM_DOUBLES], double H[36], double b[6], double* e, uint64_t* num_inliers) {
  int k = 0;
  for (int i = 0; i < 6; i++)
    for (int j = i; j < 6; j++) {
      H[6 * i + j] = acc[k];
      H[6 * j + i] = acc[k];
      k++;
    }
  for (int i = 0; i < 6; i++) b[i] = acc[21 + i];
  if (e) *e = acc[27];
  if (num_inliers) *num_inliers = static_cast<uint64_t>(acc[28] + 0.5);
}

int sga_linearize_per_point(sga_context* ctx, sga_problem* pb, const sga_factor_params* fp, const double T[16], double* values28, unsigned char* inlier) {
  SGA_TRY(check_args(ctx, pb, fp, T));
  if (!values28 || !inlier) return fail(SGA_ERR_INVALID, "null output");
  if (pb->target->kind != SGA_INDEX_KDTREE) return fail(SGA_ERR_UNSUPPORTED, "per-point factors need a kd-tree target");
  SGA_ENTER(ctx);
  double H[36], b[6], e = 0;
  uint64_t ninl = 0;
  sga_factor_params f32 = *fp;
  f32.math_mode = SGA_MATH_FP64;  // the export is a diagnostic: full precision per pair
  SGA_TRY(sga_linearize(ctx, pb, &f32, T, H, b, &e, &ninl));  // neighbours + factor state at T
  const size_t n = pb->n;
  if (n == 0) return SGA_OK;
  const sga_index* idx = pb->target;
  LinParams<double> p{};
  p.src_pts = pb->src_pts();
  p.src_cov = pb->src_cov();
  p.n = static_cast<int>(n);
  p.tgt_pts = idx->kd_pts.p;
  p.tgt_nrm = idx->nrm.p;
  p.tgt_cov = idx->cov.p;
  p.corr = pb->corr.p;
  p.hint = pb->hint.p;
  p.maha = pb->maha64.p;
  double Td[16];
  p.T = rigid_from_colmajor<double>(problem_pose(pb, T, Td));
  p.max_sq = fp->max_dist_sq < 0 ? INFINITY : static_cast<float>(fp->max_dist_sq);
  p.bound2 = p.max_sq < 3.0e38f ? p.max_sq * 1.0000002f : INFINITY;
  p.robust_kind = fp->robust_kind;
  p.robust_c = fp->robust_c;
  DevBuf<double> d_vals;
  DevBuf<unsigned char> d_ok;
  SGA_TRY(d_vals.alloc(n * 28));
  SGA_TRY(d_ok.alloc(n));
  const dim3 grid((n + 255) / 256), block(256);
  switch (fp->factor_kind) {
    case SGA_GICP: hipLaunchKernelGGL((per_point_kernel<double, SGA_GICP>), grid, block, 0, ctx->stream, p, d_vals.p, d_ok.p); break;
    case SGA_PLANE_ICP: hipLaunchKernelGGL((per_point_kernel<double, SGA_PLANE_ICP>), grid, block, 0, ctx->stream, p, d_vals.p, d_ok.p); break;
    default: hipLaunchKernelGGL((per_point_kernel<double, SGA_ICP>), grid, block, 0, ctx->stream, p, d_vals.p, d_ok.p); break;
  }
  SGA_HIP(hipGetLastError());
  SGA_HIP(hipMemcpyAsync(values28, d_vals.p, n * 28 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  SGA_HIP(hipMemcpyAsync(inlier, d_ok.p, n, hipMemcpyDeviceToHost, ctx->stream));
  SGA_HIP(hipStreamSynchronize(ctx->stream));
  if (!origin_is_zero(pb->src_origin)) {  // per-point systems of the device frame -> the caller's twist convention (common.hpp)
    for (size_t i = 0; i < n; i++) {
      double* v = values28 + 28 * i;
      double Hi[36], bi[6];
      sga_unpack_accumulator(v, Hi, bi, nullptr, nullptr);  // reads [0, 27) only
      system_to_caller(pb->src_origin, Hi, bi);
      int k = 0;
      for (int r = 0; r < 6; r++)
        for (int c = r; c < 6; c++) v[k++] = Hi[6 * r + c];
      for (int r = 0; r < 6; r++) v[21 + r] = bi[r];
    }
  }
  return SGA_OK;
}

int sga_problem_set_rejector(sga_problem* pb, sga_rejector_fn fn, void* user) {
  if (!pb) return fail(SGA_ERR_INVALID, "null argument");
  if (fn && pb->target->kind != SGA_INDEX_KDTREE) return fail(SGA_ERR_UNSUPPORTED, "host rejectors need a kd-tree target");
  pb->rejector_fn = fn;
  pb->rejector_user = user;
  pb->prev_valid = false;  // the search bound changes with the rejector
  return SGA_OK;
}

void sga_set_warm_limit(double warm_delta_m) { g_warm_delta = warm_delta_m; }
double sga_get_warm_limit(void) { return g_warm_delta; }

int sga_linearize_async(sga_context* ctx, sga_problem* pb, const sga_factor_params* fp, const double T[16], double* d_out30) {
  SGA_TRY(check_args(ctx, pb, fp, T));
  if (!d_out30) return fail(SGA_ERR_INVALID, "null output");
  pb->model_valid = false;  // the cached factor state changes: the synchronous error path must not answer from an older model
  SGA_ENTER(ctx);
  double Tdev[16];
  const double* Td = problem_pose(pb, T, Tdev);
  pb->caller_T = T;
  const int rc_dispatch = fp->math_mode == SGA_MATH_FP64 ? linearize_dispatch<double>(ctx, pb, fp, Td, d_out30, nullptr, 0) : linearize_dispatch<float>(ctx, pb, fp, Td, d_out30, nullptr, 0);
  pb->caller_T = nullptr;  // (valid for this dispatch only)
  SGA_TRY(rc_dispatch);
  if (!origin_is_zero(pb->src_origin)) {  // the caller reads H / b in its own twist convention (common.hpp: device frames)
    hipLaunchKernelGGL(frame_accumulator_kernel, dim3(1), dim3(64), 0, ctx->stream, d_out30, pb->src_origin[0], pb->src_origin[1], pb->src_origin[2]);
    SGA_HIP(hipGetLastError());
  }
  return SGA_OK;
}

int sga_error_async(sga_context* ctx, sga_problem* pb, const sga_factor_params* fp, const double T[16], double* d_out1) {
  SGA_TRY(check_args(ctx, pb, fp, T));
  if (!d_out1) return fail(SGA_ERR_INVALID, "null output");
  SGA_ENTER(ctx);
  double Tdev[16];
  const double* Td = problem_pose(pb, T, Tdev);
  return fp->math_mode == SGA_MATH_FP64 ? error_dispatch<double>(ctx, pb, fp, Td, d_out1, nullptr, 0) : error_dispatch<float>(ctx, pb, fp, Td, d_out1, nullptr, 0);
}

}  // extern "C"

namespace sga {
// pb->caller_T points at the caller's (often stack) pose for the duration of ONE dispatch (a host rejector is shown the caller's pose):
// cleared when the entry point returns, whichever way (it used to dangle)
struct CallerPoseScope {
  sga_problem* pb;
  CallerPoseScope(sga_problem* p, const double* T) : pb(p) { pb->caller_T = T; }
  ~CallerPoseScope() { pb->caller_T = nullptr; }
  CallerPoseScope(const CallerPoseScope&) = delete;
  CallerPoseScope& operator=(const CallerPoseScope&) = delete;
};
// sga_linearize in two halves, so that one host thread can keep several devices busy (multi.hip): enqueue = the kernels of the pass, the
// sum over ranks if the context has a communicator, and the hand-off of the result to the host; collect = wait for it (ctx->h_accum
// then holds `count` doubles: the system, or the system and the error model).
int linearize_enqueue(sga_context* ctx, sga_problem* pb, const sga_factor_params* fp, const double T_caller[16], unsigned long long* seq_out, int* count_out) {
  double Tdev[16];
  const double* T = problem_pose(pb, T_caller, Tdev);  // the kernels work between the two device frames (common.hpp)
  CallerPoseScope caller_pose(pb, T_caller);
  SGA_TRY(problem_check_shard_frames(ctx, pb));
  const unsigned long long seq = ++ctx->publish_seq;
  const bool direct = !ctx->sharded();
  double* host = direct ? ctx->h_accum_dev : nullptr;
  const bool model = fp->robust_kind == SGA_ROBUST_NONE && g_error_model;  // a robust kernel's error is not quadratic in the pose
  pb->model_valid = false;
  const int count = model ? kRow : SGA_ACCUM_DOUBLES;
  SGA_TRY(fp->math_mode == SGA_MATH_FP64 ? linearize_dispatch<double>(ctx, pb, fp, T, ctx->d_accum.p, host, seq, model) : linearize_dispatch<float>(ctx, pb, fp, T, ctx->d_accum.p, host, seq, model));
  int rc = comm_allreduce_sum(ctx, ctx->d_accum.p, count);  // source sharded over ranks: sum the shards' systems (and error models)
  if (rc == SGA_OK && !direct && (ctx->pending & 1)) {  // a timed pass: ev1 (behind the row reduction) -> ev_comm = the time inside the collective
    (void)hipEventRecord(ctx->ev_comm, ctx->stream);
    ctx->comm_recorded = true;
  }
  if (rc == SGA_OK && !direct) {
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(kCols), 0, ctx->stream, ctx->d_accum.p, count, ctx->h_accum_dev, seq);
    if (hipGetLastError() != hipSuccess) rc = fail(SGA_ERR_HIP, "publish_kernel launch failed");
  }
  if (rc != SGA_OK) {
    pb->prev_valid = false;  // the pass did not complete: its certificates are not to be trusted
    return rc;
  }
  *seq_out = seq;
  *count_out = count;
  return SGA_OK;
}
int linearize_collect(sga_context* ctx, sga_problem* pb, const double T[16], unsigned long long seq, int count) {
  const int rc = wait_result(ctx, seq);
  if (rc != SGA_OK) {
    pb->prev_valid = false;
    return rc;
  }
  if (pb->grid_stats_pending) {  // search statistics of a grid pass: the share of the queries its first ring left open steers the next cold pass
    pb->grid_stats_pending = false;
    if (!ctx->sharded() && pb->n > 0) {
      pb->grid_open_frac = ctx->h_accum[kStatsCol] / static_cast<double>(pb->n);
      pb->grid_open_total += static_cast<uint64_t>(ctx->h_accum[kStatsCol]);
      pb->grid_ring_total += static_cast<uint64_t>(ctx->h_accum[kStatsCol + 1]);
    }
  }
  if (count == kRow) {  // the error model: moments of the source's device frame, evaluated between device-frame poses (sga_error)
    double Tdev[16];
    memcpy(pb->model, ctx->h_accum, sizeof(pb->model));
    memcpy(pb->model_T, problem_pose(pb, T, Tdev), sizeof(pb->model_T));
    pb->model_valid = true;
  }
  return SGA_OK;
}
}  // namespace sga

extern "C" {

int sga_linearize(sga_context* ctx, sga_problem* pb, const sga_factor_params* fp, const double T[16], double H[36], double b[6], double* e, uint64_t* num_inliers) {
  SGA_TRY(check_args(ctx, pb, fp, T));
  if (!H || !b || !e) return fail(SGA_ERR_INVALID, "null output");
  SGA_ENTER(ctx);
  unsigned long long seq = 0;
  int count = 0;
  SGA_TRY(linearize_enqueue(ctx, pb, fp, T, &seq, &count));
  SGA_TRY(linearize_collect(ctx, pb, T, seq, count));
  sga_unpack_accumulator(ctx->h_accum, H, b, e, num_inliers);
  problem_system_to_caller(pb, H, b);
  return SGA_OK;
}

// e(T_n) from the quadratic error model of the last linearization (accumulate_model): a few hundred flops on the host
static double evaluate_error_model(const double* acc, const double T[16], const double Tn[16]) {
  // Y = [R^T R_n - I | R^T (tau_n - tau)], column-major 4x4 inputs
  double Y[3][4];
  for (int r = 0; r < 3; r++) {
    for (int a = 0; a < 3; a++) {
      double v = 0.0;
      for (int k = 0; k < 3; k++) v += T[4 * r + k] * Tn[4 * a + k];  // (R^T R_n)[r][a] = sum_k R[k][r] R_n[k][a]
      Y[r][a] = v - (r == a ? 1.0 : 0.0);
    }
    double v = 0.0;
    for (int k = 0; k < 3; k++) v += T[4 * r + k] * (Tn[12 + k] - T[12 + k]);
    Y[r][3] = v;
  }
  // S1[a][j] = sum p_h,a g_j; a = 3: sum g = -b_t
  double S1[4][3];
  for (int a = 0; a < 3; a++)
    for (int j = 0; j < 3; j++) S1[a][j] = acc[kModelOff + 3 * a + j];
  for (int j = 0; j < 3; j++) S1[3][j] = -acc[24 + j];
  // S2[a][b] = sum p_h,a p_h,b M' as symmetric 3x3 (xx, xy, xz, yy, yz, zz)
  auto S2 = [&](int a, int b) -> const double* {
    if (a > b) std::swap(a, b);
    if (b == 3) return a == 3 ? acc + 15 : acc + kModelOff + 9 + 6 * a;
    static const int pair_of[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    return acc + kModelOff + 27 + 6 * pair_of[a][b];
  };
  double lin = 0.0, quad = 0.0;
  for (int a = 0; a < 4; a++) {
    for (int j = 0; j < 3; j++) lin += Y[j][a] * S1[a][j];
    for (int b2 = 0; b2 < 4; b2++) {
      const double* m = S2(a, b2);
      const double ya[3] = {Y[0][a], Y[1][a], Y[2][a]}, yb[3] = {Y[0][b2], Y[1][b2], Y[2][b2]};
      const double mv[3] = {m[0] * yb[0] + m[1] * yb[1] + m[2] * yb[2], m[1] * yb[0] + m[3] * yb[1] + m[4] * yb[2], m[2] * yb[0] + m[4] * yb[1] + m[5] * yb[2]};
      quad += ya[0] * mv[0] + ya[1] * mv[1] + ya[2] * mv[2];
    }
  }
  return acc[27] - lin + 0.5 * quad;
}

}  // extern "C"

namespace sga {
double error_model_value(const double* acc96, const double T_lin[16], const double T[16]) { return evaluate_error_model(acc96, T_lin, T); }
bool error_model_enabled() { return g_error_model; }
// sga_error's device pass in two halves (see linearize_enqueue)
int error_enqueue(sga_context* ctx, sga_problem* pb, const sga_factor_params* fp, const double T_caller[16], unsigned long long* seq_out) {
  double Tdev[16];
  const double* T = problem_pose(pb, T_caller, Tdev);
  const unsigned long long seq = ++ctx->publish_seq;
  const bool direct = !ctx->sharded();
  double* host = direct ? ctx->h_accum_dev : nullptr;
  SGA_TRY(fp->math_mode == SGA_MATH_FP64 ? error_dispatch<double>(ctx, pb, fp, T, ctx->d_accum.p, host, seq) : error_dispatch<float>(ctx, pb, fp, T, ctx->d_accum.p, host, seq));
  SGA_TRY(comm_allreduce_sum(ctx, ctx->d_accum.p, 1));
  if (!direct) {
    hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(kCols), 0, ctx->stream, ctx->d_accum.p, 1, ctx->h_accum_dev, seq);
    SGA_HIP(hipGetLastError());
  }
  *seq_out = seq;
  return SGA_OK;
}
int error_collect(sga_context* ctx, unsigned long long seq, double* e) {
  SGA_TRY(wait_result(ctx, seq));
  *e = ctx->h_accum[0];
  return SGA_OK;
}
}  // namespace sga

extern "C" {

int sga_error(sga_context* ctx, sga_problem* pb, const sga_factor_params* fp, const double T[16], double* e) {
  SGA_TRY(check_args(ctx, pb, fp, T));
  if (!e) return fail(SGA_ERR_INVALID, "null output");
  if (pb->model_valid && fp->robust_kind == SGA_ROBUST_NONE && g_error_model) {  // no pass over the cloud: the model of the last linearization
    double Tdev[16];
    *e = evaluate_error_model(pb->model, pb->model_T, problem_pose(pb, T, Tdev));
    return SGA_OK;
  }
  SGA_ENTER(ctx);
  unsigned long long seq = 0;
  SGA_TRY(error_enqueue(ctx, pb, fp, T, &seq));
  return error_collect(ctx, seq, e);
}

int sga_error_model_eval(const double acc96[SGA_MODEL_DOUBLES], const double T_lin[16], const double T[16], double* e) {
  if (!acc96 || !T_lin || !T || !e) return fail(SGA_ERR_INVALID, "null argument");
  *e = evaluate_error_model(acc96, T_lin, T);
  return SGA_OK;
}

// experiments / tests: 0 = every error pass runs the error kernel (the reference's literal procedure)
void sga_set_error_model(int enabled) { g_error_model = enabled != 0; }

// diagnostics: record the leaves scanned per source point by the next passes (one-query-per-lane kernel); out = the last pass's
int sga_problem_set_search_stats(sga_context* ctx, sga_problem* pb, int enabled) {
  if (!ctx || !pb) return fail(SGA_ERR_INVALID, "null argument");
  SGA_ENTER(ctx);
  if (enabled) return pb->dbg_leaves.alloc(pb->n);
  pb->dbg_leaves.release();
  return SGA_OK;
}
// diagnostics: the source points in the engine's order (sorted by target leaf at init_T, then Morton): n x 4 floats (x, y, z, original index bits)
int sga_problem_get_sorted_points(sga_context* ctx, const sga_problem* pb, float* xyzw) {
  if (!ctx || !pb || !xyzw) return fail(SGA_ERR_INVALID, "null argument");
  if (pb->n == 0) return SGA_OK;
  SGA_ENTER(ctx);
  SGA_HIP(hipMemcpyAsync(xyzw, pb->src_pts(), pb->n * sizeof(float4), hipMemcpyDeviceToHost, ctx->stream));
  SGA_HIP(hipStreamSynchronize(ctx->stream));
  return SGA_OK;
}
int sga_problem_get_search_stats(sga_context* ctx, const sga_problem* pb, int* leaves_per_point) {
  if (!ctx || !pb || !leaves_per_point) return fail(SGA_ERR_INVALID, "null argument");
  if (pb->dbg_leaves.n < pb->n) return fail(SGA_ERR_INVALID, "search statistics are not enabled");
  SGA_ENTER(ctx);
  SGA_HIP(hipMemcpyAsync(leaves_per_point, pb->dbg_leaves.p, pb->n * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  SGA_HIP(hipStreamSynchronize(ctx->stream));
  return SGA_OK;
}

#ifdef SGA_KD_TRIPS
// diagnostics build (make trips): loop-body executions of the walk since the last call, [0, 6) per lane, [8, 14) per wave
int sga_debug_kd_trips(unsigned long long* out16) {
  unsigned long long zero[16] = {0};
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_kd_trips), sizeof(zero)) != hipSuccess) return SGA_ERR_HIP;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_kd_trips), zero, sizeof(zero)) != hipSuccess) return SGA_ERR_HIP;
  return SGA_OK;
}
// diagnostics build: start / end (100 MHz wall clock) of the first `waves` search waves of the last fused search launch
int sga_debug_kd_wave_times(unsigned long long* out, int waves) {
  if (waves > 32768) waves = 32768;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_kd_wave_times), sizeof(unsigned long long) * 2 * static_cast<size_t>(waves)) != hipSuccess) return SGA_ERR_HIP;
  return SGA_OK;
}
#else
int sga_debug_kd_trips(unsigned long long*) { return fail(SGA_ERR_UNSUPPORTED, "loop-trip counters exist in the diagnostics build only (make trips)"); }
int sga_debug_kd_wave_times(unsigned long long*, int) { return fail(SGA_ERR_UNSUPPORTED, "wave timers exist in the diagnostics build only (make trips)"); }
#endif

}  // extern "C"

namespace sga {
// The HIP runtime resolves a kernel (code object lookup, kernel descriptor, launch metadata) the first time it is used: 100 - 200 us each,
// paid in the middle of somebody's first registration — the kernels of the late LM iterations are first launched several passes in.
// Called once per process when the first context is created: asking for a kernel's attributes resolves it without launching anything.
void preload_hot_kernels() {
  hipFuncAttributes a;
#define SGA_PRELOAD(...) (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&__VA_ARGS__))
#define SGA_PRELOAD_FACTOR(F)                                        \
  SGA_PRELOAD(search_linearize_kernel<float, F, false>);            \
  SGA_PRELOAD(search_linearize_kernel<float, F, true>);             \
  SGA_PRELOAD(nn_search_queue_kernel<float, true, F>);              \
  SGA_PRELOAD(certify_linearize_kernel<float, F, kLinPts>);         \
  SGA_PRELOAD(certify_linearize_kernel<float, F, 1>);               \
  SGA_PRELOAD(linearize_kernel<float, F, 0, kLinPts>);              \
  SGA_PRELOAD(linearize_kernel<float, F, 0, 1>)
  SGA_PRELOAD_FACTOR(SGA_GICP);
  SGA_PRELOAD_FACTOR(SGA_PLANE_ICP);
  SGA_PRELOAD_FACTOR(SGA_ICP);
  SGA_PRELOAD(linearize_kernel<float, SGA_GICP, 1, kLinPts>);
  SGA_PRELOAD(linearize_kernel<float, SGA_GICP, 1, 1>);
  SGA_PRELOAD(reduce_rows_kernel);
  SGA_PRELOAD(tile_order_kernel);
  SGA_PRELOAD(publish_kernel);
  SGA_PRELOAD(error_kernel<float, SGA_GICP>);
#undef SGA_PRELOAD_FACTOR
#undef SGA_PRELOAD
  (void)hipGetLastError();
}
}  // namespace sga

extern "C" {

// experiments / tests: when the cell grid searches (SGA_GRID) and from how many target points on an index gets one (SGA_GRID_MIN_POINTS;
// affects indices built afterwards); negative values keep the current setting
void sga_set_grid_mode(int mode, long long min_points) {
  if (mode >= 0) sga::g_grid_mode = mode;
  if (min_points >= 0) sga::g_grid_min_points = min_points;
}

// experiments / tests: which search kernel runs (queue != 0: nn_search_queue_kernel with the given tiles per wave; <= 0 keeps a value)
void sga_set_search_mode(int queue, int chunk_tiles_cold, int chunk_tiles_warm) {
  g_search_queue = queue;
  if (chunk_tiles_cold > 0) g_chunk_tiles_cold = chunk_tiles_cold;
  if (chunk_tiles_warm > 0) g_chunk_tiles_warm = chunk_tiles_warm;
}

}  // extern "C"
