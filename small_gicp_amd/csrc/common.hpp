// Internal definitions shared by the HIP translation units of libsmall_gicp_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/small_gicp_amd.h"
#include "../../include/small_gicp_amd_debug.h"

namespace sga {

// ---- error plumbing --------------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int fail(int code, const char* fmt, ...);

#define SGA_HIP(expr)                                                                                             \
  do {                                                                                                            \
    hipError_t _e = (expr);                                                                                       \
    if (_e != hipSuccess) return ::sga::fail(SGA_ERR_HIP, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
  } while (0)
#define SGA_TRY(expr)           \
  do {                          \
    int _rc = (expr);           \
    if (_rc != SGA_OK) return _rc; \
  } while (0)

// ---- device memory --------------------------------------------------------------------------------------------------------
// hipMalloc / hipFree cost tens of microseconds each and hipFree synchronises the device; a registration of a 30k-point scan
// allocates ~30 buffers.  Freed blocks are therefore kept in size-bucketed free lists (context.hip) and handed out again — in
// STREAM ORDER, because work is asynchronous (sga_linearize_async, borrowed streams, several contexts on one device):
//   * a block freed inside an entry point (the thread's current stream, SGA_ENTER) goes to THAT stream's list and is only handed to
//     later allocations on the same stream, which run after everything that touched it;
//   * a block freed outside any entry point (sga_*_destroy) may still be in use by kernels in flight on any stream of its device:
//     an event is recorded on every busy stream and the block becomes reusable once those events have completed;
//   * the lists of a context move to the shared pool when the context is destroyed (after synchronising its stream).
int dev_alloc(void** p, size_t bytes);
void dev_free(void* p);
struct StreamScope {  // the calling thread's current stream for dev_alloc / dev_free
  explicit StreamScope(hipStream_t s);
  ~StreamScope();
  StreamScope(const StreamScope&) = delete;
  StreamScope& operator=(const StreamScope&) = delete;
  hipStream_t prev;
  unsigned long long prev_epoch;
};
#define SGA_ENTER(ctx)                        \
  SGA_HIP(hipSetDevice((ctx)->device));       \
  ::sga::StreamScope _sga_stream_scope((ctx)->stream)

// ---- device buffer -----------------------------------------------------------------------------------------------------
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() {
    if (p) dev_free(p);
    p = nullptr;
    n = 0;
  }
  int alloc(size_t count) {
    release();
    if (count == 0) return SGA_OK;
    const int rc = dev_alloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    if (rc != SGA_OK) {
      p = nullptr;
      return rc;
    }
    n = count;
    return SGA_OK;
  }
  // grow-only
  int reserve(size_t count) { return (count <= n) ? SGA_OK : alloc(count); }
  void swap(DevBuf& o) {
    T* tp = p;
    p = o.p;
    o.p = tp;
    const size_t tn = n;
    n = o.n;
    o.n = tn;
  }
};

// ---- packed records (HBM layout) ---------------------------------------------------------------------------------------
// pts4 : float4 {x, y, z, bitcast(u32 original index)}           16 B, one dwordx4 gather
// nrm4 : float4 {nx, ny, nz, 0}                                   16 B
// cov8 : 2 x float4 {xx, xy, xz, yy | yz, zz, 0, 0}               32 B, two dwordx4 gathers
struct alignas(16) Cov8 {
  float xx, xy, xz, yy, yz, zz, pad0, pad1;
};

}  // namespace sga

// ---- opaque handles ------------------------------------------------------------------------------------------------------
struct sga_context {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  bool registered = false;  // known to the allocator (context.hip)
  // scratch
  sga::DevBuf<double> d_accum;    // 128 doubles: a linearization result (30, or 96 with the error model)
  sga::DevBuf<unsigned> d_ticket; // arrival counter of the reduction kernel (linearize.hip), zero between launches
  double* h_accum = nullptr;      // pinned + device-mapped: [0, 128) a result, word 128 = sequence number of the last published result
  double* h_accum_dev = nullptr;  // device address of h_accum
  // pinned, device-mapped staging ring for uploads from pageable memory (context.hip): the pack kernel reads a slot over PCIe while
  // the host fills the next one; a slot is reused once the event recorded behind its reader has completed
  struct StageSlot {
    void* host = nullptr;
    void* dev = nullptr;
    size_t bytes = 0;
    hipEvent_t done = nullptr;
    bool busy = false;
  };
  static constexpr int kStageSlots = 3;
  StageSlot stage[kStageSlots];
  unsigned stage_next = 0;
  // notes (notes.hpp): small results handed to the host through mapped memory
  unsigned long long* h_notes = nullptr;      // kNoteSlots x kNoteWords words behind h_accum
  unsigned long long* h_notes_dev = nullptr;
  unsigned long long note_seq = 0;
  // voxel grid (preprocess.hip: ds_segments_kernel): look-back status words, {arrival counter, runs, valid points}, launch epoch
  sga::DevBuf<unsigned long long> vg_status;
  sga::DevBuf<uint32_t> vg_scratch;
  unsigned vg_epoch = 0;
  sga::DevBuf<unsigned long long> d_spacing;  // kd_tail_kernel: {sum of log2(leaf diagonal) in 2^-20 units, leaves counted, arrival counter, 0}; zero between launches
  sga::DevBuf<int> d_box;         // bounding-box accumulator of box_reduce_publish: identity values + arrival counter between launches
  int* h_scratch = nullptr;       // 16 pinned ints behind h_accum: small asynchronous read-backs (bounding boxes)
  unsigned long long publish_seq = 0;
  sga::DevBuf<uint8_t> d_temp;    // rocPRIM temp storage (grow-only)
  // profiling
  bool profiling = false;
  unsigned profile_period = 1;  // every profile_period-th linearize / error pass is bracketed with events
  unsigned lin_seq = 0, err_seq = 0;
  int pending = 0;  // bit 0 = the linearize event pair (ev0, ev1) awaits collection, bit 1 = the error pair (ev2, ev3)
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr, ev_mid = nullptr;  // ev_mid: between the search and the factor kernel
  hipEvent_t ev_comm = nullptr;  // behind the all-reduce of a timed pass (sharded contexts): ev1 -> ev_comm = the time inside the collective
  bool comm_recorded = false;
  double comm_ms = 0.0;
  uint64_t comm_calls = 0;
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;  // sga_debug_timer_*: GPU time between two points of the stream
  hipEvent_t ev_aux = nullptr;   // small read-backs that must not wait for the work enqueued behind them (stream-ordered mode)
  bool stream_ordered = false;   // sga_context_set_stream_ordered: preprocessing entry points return once their work is enqueued
  bool mid_recorded = false;
  double search_ms = 0.0;
  unsigned long long search_calls = 0;
  double lin_ms = 0.0, err_ms = 0.0;
  uint64_t lin_calls = 0, err_calls = 0;
  bool pending_warm = false;  // the pending linearize pair brackets a warm pass
  double warm_ms = 0.0, cold_ms = 0.0;  // the same samples split by kind of pass
  double warm_first_ms = 0.0;           // warm passes: the certificate + factor kernel alone (the rest: fallback search)
  uint64_t warm_calls = 0, cold_calls = 0;
  int num_cus = 256;
  // multi-GPU (comm.hip): RCCL communicator over the ranks that share one registration, or null
  void* comm = nullptr;
  int comm_ranks = 1;
  sga_allreduce_fn comm_fn = nullptr;  // caller-supplied sum over ranks (sga_comm_init_callback) instead of RCCL
  void* comm_user = nullptr;
  bool sharded() const { return comm != nullptr || comm_fn != nullptr; }
};

// Producer / consumer ordering across streams.  In stream-ordered mode (sga_context_set_stream_ordered) an index build or an attribute
// estimation returns while its kernels are still in flight on the producing context's stream.  The object remembers that stream and an
// event recorded behind the work; an entry point that consumes the object on ANOTHER stream first makes its stream wait for the event
// (sga::wait_ready), so outputs never race with a half-built input whichever context consumes them.  (Same stream: stream order suffices.)
// Producers that return early in stream-ordered mode: sga_index_build_kdtree, sga_estimate_normals_covariances, sga_index_clone; every
// entry point that takes a cloud or an index waits (problem creation, index build, estimation, refresh, voxel grid, voxel-map insert, kNN,
// downloads, clone).  All other producers (voxel grid, voxel-map insert / build, uploads) synchronise their stream before they return.
namespace sga {
struct Ready {
  hipEvent_t event = nullptr;
  hipStream_t stream = nullptr;
  bool pending = false;
  ~Ready() {
    if (event) (void)hipEventDestroy(event);
  }
  Ready() = default;
  Ready(const Ready&) = delete;
  Ready& operator=(const Ready&) = delete;
};
int mark_ready(sga_context* ctx, Ready& r);            // behind the producing work; no-op unless the context is stream-ordered
int wait_ready(sga_context* ctx, const Ready& r);      // before consuming on ctx's stream

// ---- device frames (round 5) ---------------------------------------------------------------------------------------------------
// The reference stores and computes in double (points/point_cloud.hpp:69-71, factors/gicp_factor.hpp:35-73), so clouds kilometres from
// the origin (UTM / ENU maps) register to full precision.  The device keeps fp32: every cloud / index therefore carries a host-side
// `origin` (double[3]) and the device holds p' = fl32(p - origin), the subtraction done in double.  origin = kOriginQuantum * round(bbox
// centre / kOriginQuantum): clouds whose box is centred within 64 m of the origin (every config of BASELINE.json) keep origin 0 and are
// stored bit for bit as before.  Poses cross the boundary through pose_to_device (t' = R o_s + t - o_t: the same rigid motion between
// the two device frames, evaluated in double); H and b come back through system_to_caller (the 6x6 adjoint of the source-side shift), so
// the caller's twist convention (util/lie.hpp:73-96, J = [R skew(p), -R], gicp_factor.hpp:62-66) — and with it the LM damping of
// optimizer.hpp:100-144, which is not invariant under that adjoint — is exactly the reference's.  Residuals, covariances, normals, the
// error, the quadratic error model (evaluated between two device-frame poses) do not depend on the frames.
constexpr double kOriginQuantum = 128.0;
void choose_origin(const double lo[3], const double hi[3], double origin[3]);  // lo > hi (empty / non-finite box): origin 0
inline bool origin_is_zero(const double o[3]) { return o[0] == 0.0 && o[1] == 0.0 && o[2] == 0.0; }
// T, T_dev column-major 4x4; o_s / o_t: origins of the source and target device frames
void pose_to_device(const double T[16], const double o_s[3], const double o_t[3], double T_dev[16]);
// H (row-major 6x6, twist order [rx ry rz tx ty tz]) and b of the device frame -> the caller's frame: H = A^T H' A, b = A^T b', A = [[I, 0], [-skew(o_s), I]]
void system_to_caller(const double o_s[3], double H[36], double b[6]);
}  // namespace sga

struct sga_cloud {
  int device = 0;
  mutable sga::Ready ready;
  size_t n = 0;
  double origin[3] = {0, 0, 0};  // the device holds p - origin (see "device frames" above)
  // bounding box of the finite records (device frame), when the producer knows it for free (uploads): lets the voxel grid sort short keys
  bool has_box = false;
  float box_lo[3] = {0, 0, 0}, box_hi[3] = {0, 0, 0};
  bool has_normals = false, has_covs = false;
  sga::DevBuf<float4> pts;   // w = bitcast(original index)
  sga::DevBuf<float4> nrm;
  sga::DevBuf<sga::Cov8> cov;
};

enum { SGA_INDEX_KDTREE = 0, SGA_INDEX_VOXELMAP = 1, SGA_INDEX_FLATMAP = 2 };
constexpr int kFlatCap = 16;  // point slots per voxel of a flat map (max_num_points_in_cell <= 16)

struct sga_index {
  int kind = SGA_INDEX_KDTREE;
  int device = 0;
  mutable sga::Ready ready;
  size_t n = 0;  // points (kd-tree) or voxels (voxel map)
  double origin[3] = {0, 0, 0};  // device frame of the fp32 records (kd-tree: the cloud's; voxel maps: of the exported means / points — voxel
                                 // coordinates, hash keys and the fp64 state of incremental maps stay in the caller's frame)
  bool has_normals = false, has_covs = false;
  // implicit balanced kd-tree over the target (kd_search.hpp): points in kd order + {threshold, axis} heap
  sga::DevBuf<float4> kd_pts;       // kd order; w = original index bits
  sga::DevBuf<float4> nrm;          // kd order
  sga::DevBuf<sga::Cov8> cov;       // kd order.  Voxel maps: mean covariances in voxel-id order
  sga::DevBuf<float2> kd_nodes;     // 2^kd_depth entries (index 0 unused)
  sga::DevBuf<float4> kd_nodes4;    // pair records of the even depths (kd_search.hpp)
  sga::DevBuf<float4> kd_boxes;     // tight bounding box of every node: [2 * node] = min corner, [2 * node + 1] = max corner
  sga::DevBuf<float4> kd_groups;    // group headers of the 1-NN walk: the boxes of the (up to) 4 leaves under every node of depth kd_depth - 2
  sga::DevBuf<float4> kd_leaf;      // leaf blocks of the 1-NN walk: per leaf x[8], y[8], z[8], original index[8] (kd_search.hpp: the fast leaf scan)
  int kd_depth = 0;
  float bbox_lo[3] = {0, 0, 0}, bbox_hi[3] = {0, 0, 0};
  // the target's own length scale (notes.hpp: late notes): geometric mean of the diagonals of the tree's leaf boxes, i.e. the size of a
  // neighbourhood of 8 points; the pass routing of linearize.hip measures motions in units of it.  0 = not known (yet)
  mutable double spacing = 0.0;
  mutable unsigned long long spacing_seq = 0;  // the late note that carries it; 0 = none
  // uniform cell grid over the same points (cell_grid.hpp / cell_grid.hip): the exact search of cold passes near the optimum; grid_h == 0: none
  sga::DevBuf<float4> grid_pts;       // cell order, w = kd position bits
  sga::DevBuf<uint32_t> grid_start;   // cells + 1
  float grid_h = 0.f, grid_eps = 0.f;
  float grid_org[3] = {0, 0, 0};
  int grid_dim[3] = {0, 0, 0};
  // voxel map
  sga::DevBuf<float4> pts;                // voxel means in voxel-id order, w = voxel id bits
  double leaf = 0.0;
  sga::DevBuf<unsigned long long> hkeys;  // open addressing, EMPTY = ~0ull
  sga::DevBuf<uint32_t> hvals;            // voxel id
  uint32_t hmask = 0;
  sga::DevBuf<int> vcoords;               // n*3
  sga::DevBuf<uint32_t> vcounts;          // n
  // incremental voxel maps (voxelmap.hip: repeated insert() with a pose, running means in fp64, LRU removal)
  bool incremental = false;
  size_t vcap = 0;                        // capacity of the per-voxel arrays (voxels)
  sga::DevBuf<double> vmean64;            // 3 per voxel: finalized mean
  sga::DevBuf<double> vcov64;             // 6 per voxel: finalized mean covariance (xx, xy, xz, yy, yz, zz)
  sga::DevBuf<uint32_t> vlru;             // insert counter at the voxel's last update
  uint32_t lru_counter = 0, lru_horizon = 100, lru_clear_cycle = 10;
  // flat maps (IncrementalVoxelMap<FlatContainerCov>): voxels keep up to flat_max of the inserted points; pts / cov then hold
  // kFlatCap slots per voxel (slot = voxel * kFlatCap + i), vcounts the number of points of a voxel
  sga::DevBuf<double> fpts64;             // 3 per slot
  sga::DevBuf<double> fcov64;             // 6 per slot
  uint32_t flat_max = 10;                 // flat_container.hpp:20
  double flat_min_sq = 0.1 * 0.1;         // flat_container.hpp:19
  int search_offsets = 1;                 // 1, 7 or 27 (incremental_voxelmap.hpp:157-186)
};

struct sga_problem {
  int device = 0;
  const sga_index* target = nullptr;
  size_t n = 0;  // source points
  double src_origin[3] = {0, 0, 0};  // device frame of the source records (the target's is target->origin, which an insert into a voxel map may move)
  bool frame_checked = false;        // sharded contexts: the ranks' source origins were compared (they must agree: the accumulators are summed)
  bool has_normals = false, has_covs = false;
  sga::DevBuf<float4> pts;       // spatially sorted copy of the source; w = original index bits
  sga::DevBuf<sga::Cov8> cov;
  // a problem created from the source's own index (sga_problem_create_from_index) borrows that index's kd-ordered arrays instead
  // of copying them: the index must outlive the problem (like the target index)
  const float4* pts_view = nullptr;
  const sga::Cov8* cov_view = nullptr;
  const float4* src_pts() const { return pts_view ? pts_view : pts.p; }
  const sga::Cov8* src_cov() const { return cov_view ? cov_view : cov.p; }
  // factor state
  sga::DevBuf<int> corr;         // kd position of the matched target point / voxel id (voxelmap); -1 = outlier
  sga::DevBuf<int> hint;         // exact nearest neighbour per source point at the last linearization pose, rejected or not (kd targets)
  sga::DevBuf<int> hint2;        // the runner-up of that search: second candidate of the warm pass's certificate
  sga::DevBuf<float> rex;        // exclusion radius around the query (kd_search.hpp): every target point but the two candidates lies beyond it
  sga::DevBuf<int> dbg_leaves;   // diagnostics (sga_problem_set_search_stats): leaves scanned per source point in the last pass
  sga::DevBuf<uint32_t> walked;  // statistics, one counter per 64 source points: lanes of warm passes that had to walk
  bool state_fresh = false;      // hint / hint2 / corr still hold the "none" problem_state_init_kernel wrote: the first registration skips its own reset
  double T_prev[16] = {0};       // pose of the last linearization (column-major), valid iff prev_valid
  bool prev_valid = false;
  int prev_math = 0;
  int lin_factor = -1;           // factor kind, pose of the last linearize of any kind, and whether it stored its mahalanobis matrices
  double lin_T[16] = {0};
  bool maha_valid = false;
  int last_math = 0;             // arithmetic of the last linearize of any kind (which mahalanobis cache is current)
  float bbox_lo[3] = {0, 0, 0}, bbox_hi[3] = {0, 0, 0};  // bounding box of the source (source frame): bounds the motion between two poses
  uint64_t cold_passes = 0, warm_passes = 0;  // passes against a kd-tree since the problem was created
  uint64_t grid_passes = 0;                   // cold passes searched through the cell grid (cell_grid.hip)
  sga::DevBuf<uint32_t> grid_stats;           // [0] queries ring 1 left open in the current pass, [1] sum of the rings they then scanned
  bool grid_stats_pending = false;            // the result being fetched carries the statistics of a grid pass in its spare columns
  uint64_t grid_open_total = 0, grid_ring_total = 0;  // totals of those statistics since the problem was created
  double grid_open_frac = 0.0;                // share of the queries the last grid pass's ring 1 left open (policy: too many -> kd walk)
  // launch order of the one-query-per-lane search kernel (linearize.hip, "longest tile first"): the duration of every tile's wave
  // in the last such pass, and the tiles of each XCD's share sorted by it; order_tiles != 0 iff tile_order belongs to that pass
  sga::DevBuf<uint32_t> tile_cost, tile_order;
  unsigned order_tiles = 0;
  hipStream_t order_stream = nullptr;  // the stream tile_order was (or is being) written on: a pass on another stream ignores it
  sga::DevBuf<float> maha;       // n*6 (fp32 mode) — fused mahalanobis of the last linearize
  sga::DevBuf<double> maha64;    // n*6 (fp64 mode, allocated on first use)
  // custom CorrespondenceRejector on the host (sga_problem_set_rejector): reject flag per source point (caller's order) for the current pass
  sga_rejector_fn rejector_fn = nullptr;
  void* rejector_user = nullptr;
  const double* caller_T = nullptr;  // the pose as the caller passed it to the linearization being dispatched (what a host rejector is shown)
  sga::DevBuf<unsigned char> reject;
  // the quadratic error model of the last linearization (linearize.hip): valid for trial poses until the next linearization
  bool model_valid = false;
  double model[96] = {0};        // the reduced row: system + model sums
  double model_T[16] = {0};      // its pose
  // reduction scratch
  sga::DevBuf<double> partials;  // partial rows + the stage rows of reduce_rows_kernel
};
