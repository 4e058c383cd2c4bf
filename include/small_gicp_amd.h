/*
 * small_gicp_amd — C-ABI of the MI355X-native registration hot path.
 *
 * Drop-in boundary for koide3/small_gicp's per-iteration linearization (reference tree /root/reference, v1.0.1).
 * Every entry point names the reference interface it replaces (file:line relative to /root/reference).
 * Plain C: opaque handles, caller-owned host buffers, int status codes, no exceptions, no torch types.
 *
 * Conventions
 *   - 4x4 transforms are column-major double[16] (Eigen::Isometry3d::matrix().data()).
 *   - 6x6 H is row-major double[36] (symmetric), b is double[6]; twist order is [rx ry rz tx ty tz] (util/lie.hpp:73-77).
 *   - Symmetric 3x3 matrices ("cov6", "mahalanobis6") are packed xx,xy,xz,yy,yz,zz.
 *   - All device work of a context is serialised on ONE HIP stream; blocking calls synchronise that stream only.
 *   - A context is bound to one GPU; multi-GPU = one process (or context) per GPU with the source cloud sharded: after
 *     sga_comm_init() every sga_linearize all-reduces its 96-double accumulator (system + error-model moments) over the ranks on the
 *     context's stream before the host reads it (the 30-double sga_linearize_async() form is the building block for other transports).
 *   - Diagnostics (search statistics, debug counters) are declared in small_gicp_amd_debug.h; environment switches (SGA_*) read when the
 *     library loads select between equivalent kernels for experiments and are NOT part of this ABI.
 */
#ifndef SMALL_GICP_AMD_H
#define SMALL_GICP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sga_context sga_context; /* one GPU + one stream */
typedef struct sga_cloud sga_cloud;     /* device-resident point cloud: points [+ normals] [+ covariances]          (points/point_cloud.hpp:15-71) */
typedef struct sga_index sga_index;     /* nearest-neighbour search target: kd-tree (replaces ann/kdtree.hpp KdTree) or GaussianVoxelMap */
typedef struct sga_problem sga_problem; /* a (target index, source cloud) pairing + per-source-point factor state (registration.hpp:41 std::vector<PointFactor>) */

enum sga_status {
  SGA_OK = 0,
  SGA_ERR_INVALID = 1,     /* bad argument */
  SGA_ERR_HIP = 2,         /* a HIP runtime call failed; see sga_last_error() */
  SGA_ERR_NO_DEVICE = 3,   /* no usable gfx950 device: the product path NEVER falls back to the CPU */
  SGA_ERR_UNSUPPORTED = 4, /* combination not available (e.g. PLANE_ICP against a voxel map) */
  SGA_ERR_CALLBACK = 5     /* a user callback returned non-zero */
};

/* registration_helper.hpp:38 RegistrationSetting::RegistrationType (VGICP = GICP factor against a GaussianVoxelMap index) */
enum sga_factor_kind { SGA_ICP = 0, SGA_PLANE_ICP = 1, SGA_GICP = 2 };
/* factors/robust_kernel.hpp:11-59 */
enum sga_robust_kind { SGA_ROBUST_NONE = 0, SGA_ROBUST_HUBER = 1, SGA_ROBUST_CAUCHY = 2 };
enum sga_optimizer_kind { SGA_LEVENBERG_MARQUARDT = 0, SGA_GAUSS_NEWTON = 1 };
enum sga_math_mode { SGA_MATH_FP32 = 0, SGA_MATH_FP64 = 1 }; /* per-pair arithmetic; data in HBM is fp32 either way, sums are fp64 */

/* Thread-local description of the last failure on the calling thread. Never NULL. */
const char* sga_last_error(void);
/* Library version string. */
const char* sga_version(void);
/* Diagnostics of the library's caching device allocator: hipMalloc calls, allocations served by the calling stream's own free list, by
 * the shared pool, by blocks whose deferred release had completed, and frees that had to be deferred behind busy streams. */
void sga_allocator_stats(uint64_t out[5]);
/* Number of visible HIP devices (0 when there is no GPU / no driver). */
int sga_device_count(void);

/* ---- context -------------------------------------------------------------------------------------------------- */
int sga_context_create(int device, sga_context** out);
/* Borrow an existing hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) instead of creating one. */
int sga_context_create_on_stream(int device, void* hip_stream, sga_context** out);
int sga_context_destroy(sga_context* ctx);
int sga_context_synchronize(sga_context* ctx);
void* sga_context_stream(sga_context* ctx); /* the hipStream_t */

/* ---- clouds (points/traits.hpp:15-78 accessor protocol: size / point / normal / cov) --------------------------------- */
/* Pinned host memory for the caller's scans (round 6).  The reference's drivers hold every scan in host memory before they time a frame
 * (benchmark/benchmark_odom.hpp:36-47, odometry_benchmark.cpp: the points are read into std::vector<Eigen::Vector4f> first); a caller of
 * this library that reads its scans into memory from sga_host_alloc (or any hipHostMalloc / hipHostRegister'd buffer) spares the upload
 * its only CPU pass: sga_cloud_create_f32 recognises such arrays and lets the device read them in place.  Ordinary (pageable) arrays
 * work as before: they are copied once into the context's pinned staging ring. */
int sga_host_alloc(size_t bytes, void** out);
int sga_host_free(void* p);
/* fp32 input: xyz n*3, normals n*3 or NULL, cov6 n*6 or NULL.  The arrays are free for reuse when the call returns (in stream-ordered
 * mode too: a pageable array has been copied, a pinned one has been read). */
int sga_cloud_create_f32(sga_context* ctx, const float* xyz, const float* normals, const float* cov6, size_t n, sga_cloud** out);
/* Reference PointCloud layout (points/point_cloud.hpp:69-71): xyzw n*4 doubles, normals n*4 doubles or NULL, covs n*16 doubles (4x4) or NULL */
int sga_cloud_create_f64(sga_context* ctx, const double* xyzw, const double* normals4, const double* cov4x4, size_t n, sga_cloud** out);
/* Device frames (round 5).  The reference stores and computes in double (points/point_cloud.hpp:69-71), so clouds kilometres from the origin
 * (UTM / ENU maps) register to full precision; the device keeps fp32.  Every cloud therefore has an ORIGIN (double[3]) and the device
 * holds fl32(p - origin), the subtraction done in double: sga_cloud_create_f32 / _f64 choose it themselves (the centre of the bounding
 * box rounded to a multiple of 128 m — so a cloud centred within 64 m of the origin keeps origin 0 and is stored as before); poses,
 * H, b, downloaded points, voxel coordinates and kNN queries are always the CALLER's frame, the library converts at its entry points
 * (pose: t' = R o_source + t - o_target in double; H, b: the 6x6 adjoint of the source shift, so that the twist convention of
 * util/lie.hpp:73-96 and the LM damping of optimizer.hpp:100-144 are the reference's).  The _origin forms let the caller name the origin:
 *   sga_cloud_create_f32_origin: xyz_rel are ALREADY relative to origin (true position = xyz_rel + origin; origin NULL = 0) — how an
 *     fp32 caller hands over a geo-referenced cloud without losing its millimetres;
 *   sga_cloud_create_f64_origin: absolute doubles, recentred about `origin` (NULL: chosen as above) — ranks that upload their own shard of
 *     a sharded source name a common origin (the shards' accumulators are added; slices made with sga_cloud_slice share one anyway). */
int sga_cloud_create_f32_origin(sga_context* ctx, const float* xyz_rel, const float* normals, const float* cov6, size_t n, const double origin[3], sga_cloud** out);
int sga_cloud_create_f64_origin(sga_context* ctx, const double* xyzw, const double* normals4, const double* cov4x4, size_t n, const double origin[3], sga_cloud** out);
int sga_cloud_origin(const sga_cloud* cloud, double origin[3]);
int sga_index_origin(const sga_index* index, double origin[3]);
/* the library's rule: origin of the device frame for a bounding box (empty / non-finite box: 0) */
void sga_choose_origin(const double lo[3], const double hi[3], double origin[3]);
/* A new cloud holding points [first, first + count) of `cloud` with their normals / covariances (device copy): the source shard of one
 * rank when a registration is spread over GPUs (reduction_omp.hpp:32-58 is the loop being partitioned). */
int sga_cloud_slice(sga_context* ctx, const sga_cloud* cloud, size_t first, size_t count, sga_cloud** out);
int sga_cloud_destroy(sga_cloud* cloud);
int sga_cloud_size(const sga_cloud* cloud, size_t* n);
int sga_cloud_has(const sga_cloud* cloud, int* has_normals, int* has_covs);
/* Any of xyz / normals / cov6 may be NULL. */
int sga_cloud_download(sga_context* ctx, const sga_cloud* cloud, float* xyz, float* normals, float* cov6);
/* the same with the points in double (device record + origin, added in double): what a caller far from the origin wants back */
int sga_cloud_download_f64(sga_context* ctx, const sga_cloud* cloud, double* xyz, float* normals, float* cov6);

/* ---- preprocessing (registration_helper.cpp:22-34 preprocess_points) ----------------------------------------------- */
/* util/downsampling.hpp:23-78 voxelgrid_sampling: centroid per occupied voxel, output in ascending packed-key order. */
int sga_voxelgrid_sampling(sga_context* ctx, const sga_cloud* in, double leaf_size, sga_cloud** out);
/* util/normal_estimation.hpp:65-92 estimate_local_features: kNN(k, incl. self) -> mean/cov -> eigvecs -> normal / covariance.
 * index: a kd-tree built over `cloud`, or NULL to build a temporary one.  flags: bit0 = normals, bit1 = covariances. */
int sga_estimate_normals_covariances(sga_context* ctx, sga_cloud* cloud, const sga_index* index, int num_neighbors, int flags);

/* ---- search indices --------------------------------------------------------------------------------------------- */
/* Replaces KdTree<PointCloud>(points) (ann/kdtree.hpp:80-126, :250-252): exact nearest neighbour / kNN over `target`.
 * An implicit, perfectly balanced kd-tree (median splits like the reference, no pointers) built on the GPU; the index keeps its
 * own kd-ordered copy of the target's points / normals / covariances. */
int sga_index_build_kdtree(sga_context* ctx, const sga_cloud* target, sga_index** out);
/* Replaces create_gaussian_voxelmap (registration_helper.cpp:50-54; ann/incremental_voxelmap.hpp:55-92, gaussian_voxelmap.hpp:32-53):
 * one-shot insert of a cloud WITH covariances; voxel ids follow first-insertion order like the reference. */
int sga_index_build_gaussian_voxelmap(sga_context* ctx, const sga_cloud* points_with_covs, double leaf_size, sga_index** out);
/* The same kind of index from voxels that already exist on the host — the reference's GaussianVoxelMap object as it is (flat order:
 * coords n*3 int32, means n*3 doubles, cov6 n*6 doubles xx,xy,xz,yy,yz,zz; ann/incremental_voxelmap.hpp:39-92): the voxel ids are the
 * caller's.  What lets Registration<GICPFactor, ParallelReductionHIP>::align(voxelmap, source, voxelmap) (registration_helper.cpp:125-137) work. */
int sga_index_create_voxelmap_from_voxels(sga_context* ctx, double leaf_size, const int32_t* coords, const double* means3, const double* cov6, size_t n, sga_index** out);
/* A flat voxel map (IncrementalVoxelMap<FlatContainer*>, ann/flat_container.hpp:21-58) from voxels that exist on the host, flat order: coords
 * n*3, counts n (<= 16 each), points n*16*3 doubles and cov6 n*16*6 doubles (16 slots per voxel, the first counts[v] valid; cov6 may be NULL
 * for ICP), searched over 1, 7 or 27 voxels.  Target indices are (voxel_id << 32) | point_id with the caller's ids. */
int sga_index_create_flatmap_from_voxels(sga_context* ctx, double leaf_size, const int32_t* coords, const uint32_t* counts, const double* points3, const double* cov6, int search_offsets, size_t n, sga_index** out);
/* Re-copy normals / covariances from `cloud` (the cloud the tree was built over) into the index's kd-ordered arrays, e.g. after
 * attributes were estimated or set once the index already existed (reference flow: KdTree first, estimate_covariances second). */
int sga_index_refresh_attributes(sga_context* ctx, sga_index* index, const sga_cloud* cloud);
/* A deep copy of an index on ctx's device, which may differ from the source's (peer copy): a replicated target reaches the other GPUs of a
 * sharded registration without being built once per GPU. */
int sga_index_clone(sga_context* ctx, const sga_index* src, sga_index** out);
int sga_index_destroy(sga_index* index);
/* Number of target points (kd-tree) or voxels (voxel map): traits::size(target). */
int sga_index_size(const sga_index* index, size_t* n);
/* Voxel map contents in voxel-id order: coords n*3 int32, means n*3, cov6 n*6, counts n (any may be NULL). */
int sga_index_voxelmap_download(sga_context* ctx, const sga_index* index, int32_t* coords, float* means, float* cov6, uint32_t* counts);
/* Incremental GaussianVoxelMap (ann/incremental_voxelmap.hpp:55-92, the scan-to-model target): an empty map, then any number of
 * insert(points_with_covs, T): the points are moved by T (column-major 4x4, NULL = identity), voxels keep running means over
 * everything ever inserted while they live, voxel ids follow the creation order, and every `clear_cycle` inserts the voxels not
 * touched for more than `horizon` inserts are removed (defaults 100 / 10, incremental_voxelmap.hpp:46).  Usable as the target of
 * sga_problem_create / sga_align like a one-shot map; problems created before an insert must be re-created. */
int sga_voxelmap_create(sga_context* ctx, double leaf_size, sga_index** out);
int sga_voxelmap_insert(sga_context* ctx, sga_index* voxelmap, const sga_cloud* points_with_covs, const double T[16]);
int sga_voxelmap_set_lru(sga_index* voxelmap, uint32_t horizon, uint32_t clear_cycle);
/* IncrementalVoxelMap<FlatContainerCov> (ann/flat_container.hpp:15-100): voxels that keep up to max_num_points_in_cell (default 10,
 * at most 16) of the inserted points, at least sqrt(min_sq_dist_in_cell) (default 0.1 m) apart, with their covariances — the
 * scan-to-model GICP target (odometry_benchmark_small_gicp_model_omp.cpp).  Inserted with sga_voxelmap_insert / _set_lru like a
 * Gaussian map; searched over 1, 7 or 27 voxels (incremental_voxelmap.hpp:157-186); target indices are (voxel_id << 32) | point_id. */
int sga_flatmap_create(sga_context* ctx, double leaf_size, sga_index** out);
int sga_flatmap_set_setting(sga_index* flatmap, double min_sq_dist_in_cell, uint32_t max_num_points_in_cell);
/* Voxels visited around the query's own (incremental_voxelmap.hpp:157-186): 1 (default), 7 or 27 — for flat AND Gaussian maps, incremental,
 * one-shot or created from host voxels; every visited Gaussian voxel offers its mean, the nearest wins (the first of equal distances). */
int sga_voxelmap_set_search_offsets(sga_index* voxelmap, int num_offsets);
/* coords n*3, counts n, points n*16*3 and cov6 n*16*6 (16 slots per voxel, the first counts[v] of them valid); any pointer may be NULL */
int sga_flatmap_download(sga_context* ctx, const sga_index* flatmap, int32_t* coords, uint32_t* counts, float* points, float* cov6);
/* traits::knn_search / nearest_neighbor_search (ann/traits.hpp:22-57) for m host queries (m*3 floats):
 * idx m*k int64 (original target indices, -1 = none), sq_dist m*k floats ascending (inf = none).
 * max_sq_dist < 0 means unbounded.  k <= 116 for kd-trees.  Voxel maps (Gaussian and flat, incremental_voxelmap.hpp:99-149): any
 * k <= 128, the voxels at the map's search offsets in the reference's order, KnnResult::push semantics; idx is then the reference's global
 * index (voxel_id << 32) | point_id (point_id = 0 for a Gaussian voxel, the slot within its voxel for a flat map). */
int sga_index_knn(sga_context* ctx, const sga_index* index, const float* queries, size_t m, int k, double max_sq_dist, int64_t* idx, float* sq_dist);
/* The same with the reference's types (double queries m*3, double squared distances): the search runs on the fp32 roundings of the
 * queries, the squared distances of the neighbours found are then evaluated in double against the double queries (the reference's
 * Python tests compare them with scipy to 1e-6 at ranges where fp32 resolves 1e-4, src/test/python_test.py:194-257). */
int sga_index_knn_f64(sga_context* ctx, const sga_index* index, const double* queries, size_t m, int k, double max_sq_dist, int64_t* idx, double* sq_dist);

/* ---- the hot path: Reduction::linearize / Reduction::error (registration/reduction_omp.hpp:24-70) ------------------------- */
typedef struct sga_factor_params {
  int factor_kind;    /* sga_factor_kind */
  int robust_kind;    /* sga_robust_kind */
  double robust_c;    /* robust kernel width (robust_kernel.hpp:16,42) */
  double max_dist_sq; /* DistanceRejector::max_dist_sq (rejector.hpp:19-28); < 0 = NullRejector */
  int math_mode;      /* sga_math_mode */
} sga_factor_params;
void sga_factor_params_default(sga_factor_params* p);

/* Pair a target index with a source cloud.  The problem keeps a spatially sorted copy of the source (sorted by the target kd-leaf
 * init_T * p falls into) and the per-point factor state (target index + cached mahalanobis), i.e. registration.hpp:41. */
int sga_problem_create(sga_context* ctx, const sga_index* target, const sga_cloud* source, const double init_T[16], sga_problem** out);
/* The same with the source given by ITS OWN kd-tree index (the odometry loop, src/benchmark/odometry_benchmark_small_gicp_omp.cpp:22-38:
 * every scan is indexed once, for its covariances and as the next target): the problem takes the index's kd-ordered points and
 * covariances as they are — the order is spatially coherent already, so no sort.  Factor state is reported in the ORIGINAL order of the
 * cloud the index was built over, exactly as with sga_problem_create.  Attributes must be present in the index (build it after
 * estimating them, or call sga_index_refresh_attributes). */
int sga_problem_create_from_index(sga_context* ctx, const sga_index* target, const sga_index* source_index, const double init_T[16], sga_problem** out);
int sga_problem_destroy(sga_problem* problem);
/* Sum_i (H_i, b_i, e_i) at T over all source points with a correspondence; refreshes the factor state. */
int sga_linearize(sga_context* ctx, sga_problem* problem, const sga_factor_params* params, const double T[16], double H[36], double b[6], double* e, uint64_t* num_inliers);
/* Sum_i e_i at T with the correspondences and mahalanobis cached by the last sga_linearize (gicp_factor.hpp:80-89).  With those
 * frozen the sum is a quadratic polynomial in T: sga_linearize accumulates its coefficients next to H / b (63 more sums) and this
 * call evaluates it on the host — no pass over the cloud, no device round trip.  Robust kernels (not quadratic) and calls after
 * sga_linearize_async run the error kernel. */
int sga_error(sga_context* ctx, sga_problem* problem, const sga_factor_params* params, const double T[16], double* e);
/* Enqueue-only forms for multi-GPU: results stay in device memory so they can be all-reduced (RCCL) on the same stream before
 * the host reads them.  d_out30: [0..20] upper triangle of H row-wise, [21..26] b, [27] e, [28] num_inliers (as double), [29] 0.
 * d_out1: e.  Both must be device pointers valid on the context's device; no host synchronisation is performed. */
#define SGA_ACCUM_DOUBLES 30
int sga_linearize_async(sga_context* ctx, sga_problem* problem, const sga_factor_params* params, const double T[16], double* d_out30);
int sga_error_async(sga_context* ctx, sga_problem* problem, const sga_factor_params* params, const double T[16], double* d_out1);
/* Native multi-GPU form: give the context an RCCL communicator over the ranks that share ONE registration (source sharded, target
 * replicated).  From then on sga_linearize / sga_error / sga_align all-reduce their accumulators over the ranks on the context's
 * stream before reading them back, so every rank returns the system of the whole source cloud.  Rank 0 creates the 128-byte id
 * and hands it to the others out of band (MPI, torch.distributed, a file).  librccl is bound with dlopen on first use. */
int sga_comm_unique_id(unsigned char id[128]);
int sga_comm_init(sga_context* ctx, int nranks, int rank, const unsigned char id[128]);
int sga_comm_destroy(sga_context* ctx);
/* The same protocol with the sum over ranks supplied by the caller: wherever the RCCL form runs ncclAllReduce (same place in the
 * stream order: behind the row reduction, before the result is handed to the host), the library copies the accumulator to the host,
 * calls fn(user, values, count) — which must replace values[0..count) by their sum over all ranks (MPI_Allreduce, gloo, a pipe) and
 * return 0 — and copies the sums back.  For transports other than RCCL, and for running the N-rank code path on a single device
 * (tests/test_distributed_gpu.py).  count is 96 (system + error-model moments) or 30 / 1 (robust factors). */
typedef int (*sga_allreduce_fn)(void* user, double* values, size_t count);
int sga_comm_init_callback(sga_context* ctx, int nranks, int rank, sga_allreduce_fn fn, void* user);
/* Host only (no device): the error at trial pose T from the 96-double accumulator of a linearization at T_lin — what sga_error answers
 * from after sga_linearize.  acc96: [0, 30) the system (SGA_ACCUM_DOUBLES layout), [32, 95) the moments sum p_a g (9), sum p_a M' (18),
 * sum p_a p_b M' (36) with M' = R^T M R, g = R^T M r in the source frame; sums over ranks of such accumulators are
 * accumulators. */
#define SGA_MODEL_DOUBLES 96
int sga_error_model_eval(const double acc96[SGA_MODEL_DOUBLES], const double T_lin[16], const double T[16], double* e);
/* Expand a 30-double accumulator (host memory) into H[36], b[6], e, num_inliers. */
void sga_unpack_accumulator(const double acc30[SGA_ACCUM_DOUBLES], double H[36], double b[6], double* e, uint64_t* num_inliers);
/* A custom CorrespondenceRejector on the host (registration/rejector.hpp:11-28 is a duck-typed functor `bool operator()(target, source, T,
 * target_index, source_index, sq_dist)`, true = reject).  Batch form: once per linearization the callback receives, for every source
 * point in the caller's order, the index of its nearest target point (caller's target order, -1 = the target is empty) and the
 * squared distance, and fills reject[i] (1 = reject).  While a callback is set the search is unbounded and params->max_dist_sq is
 * ignored, exactly as with a user-supplied rejector type in the reference.  Costs a device->host->device round trip per
 * linearization: a correctness path, not a fast one.  fn = NULL restores the built-in DistanceRejector / NullRejector.  kd-tree targets. */
typedef int (*sga_rejector_fn)(void* user, const double T[16], size_t n, const int64_t* target_index, const float* sq_dist, unsigned char* reject);
int sga_problem_set_rejector(sga_problem* problem, sga_rejector_fn fn, void* user);
/* Factor::linearize per source point, as the reference's Python binding exposes it (src/python/factors.cpp:52-101; gicp_factor.hpp:35-73,
 * icp_factor.hpp:20-54, plane_icp_factor.hpp:19-57): runs one linearization at T, then returns for every source point (caller's order)
 * values28 n*28 doubles = [0..20] upper triangle of H_i row-wise, [21..26] b_i, [27] e_i, and inlier n bytes (0: no correspondence, all
 * values 0).  kd-tree targets; per-pair arithmetic in fp64.  A diagnostic / binding entry point, not the hot path. */
int sga_linearize_per_point(sga_context* ctx, sga_problem* problem, const sga_factor_params* params, const double T[16], double* values28, unsigned char* inlier);
/* Factor state in the caller's source order: target_index n int64 (-1 = outlier; voxel id for voxel maps), mahalanobis6 n*6 floats (GICP only). */
int sga_problem_get_factors(sga_context* ctx, const sga_problem* problem, int64_t* target_index, float* mahalanobis6);
/* Stream-ordered mode (default off): sga_index_build_kdtree and sga_estimate_normals_covariances return as soon as their kernels are
 * enqueued on the context's stream instead of waiting for them (a 15k-point odometry scan spends a fifth of its time in such waits).
 * Later calls on the SAME context see their results in stream order; before their outputs are used from another context / stream,
 * call sga_context_synchronize.  Errors of the enqueued kernels are reported by the next synchronising call. */
int sga_context_set_stream_ordered(sga_context* ctx, int enabled);
/* Average device time (ms) of the linearize / error kernel chains measured with HIP events on the context's stream (0 if profiling
 * off).  enabled = 0: off; 1: every pass is bracketed with events; N > 1: every N-th pass (an event record costs microseconds). */
int sga_context_set_profiling(sga_context* ctx, int enabled);
int sga_context_get_kernel_ms(sga_context* ctx, double* linearize_ms, uint64_t* linearize_calls, double* error_ms, uint64_t* error_calls);
/* The part of a COLD pass's time spent in the nearest-neighbour search kernel (the rest: factor evaluation + block reduction). */
int sga_context_get_search_ms(sga_context* ctx, double* search_ms, uint64_t* search_calls);
/* Sharded contexts (sga_comm_init*): average time (ms) of the timed passes between the end of the row reduction and the end of the all-reduce
 * of the accumulator on the context's stream — the time inside the collective, incl. waiting for the slowest rank. */
int sga_context_get_comm_ms(sga_context* ctx, double* comm_ms, uint64_t* comm_calls);
/* linearize_ms split by kind of pass against a kd-tree.  cold = the exact nearest-neighbour walk (kdtree.hpp:193-233) for every source
 * point; warm = the neighbour of the previous linearization is kept wherever it is certified still exact (its exclusion radius minus
 * the point's motion since, triangle inequality) and only the other points walk.  Same results either way, bit for bit in the
 * correspondences.  warm_search_ms: the part of warm_ms spent in the search kernel. */
int sga_context_get_pass_ms(sga_context* ctx, double* cold_ms, uint64_t* cold_calls, double* warm_ms, uint64_t* warm_calls, double* warm_search_ms);
/* A pass runs warm while no source point can have moved farther than warm_delta_m metres since the previous linearization (default
 * 0.1; negative: never, i.e. every pass walks in full).  Process-wide; results do not depend on it, only speed does. */
void sga_set_warm_limit(double warm_delta_m);
/* 0: sga_error always runs the error kernel (the reference's literal procedure; tests compare the two). Default 1. */
void sga_set_error_model(int enabled);
/* Which nearest-neighbour kernel the linearization runs (results do not depend on it; tests compare the two): 1 = the queue-fed
 * kernel (a wave owns chunk_tiles x 64 source points and refills its lanes from a queue), 0 = one query per lane, 2 (default) =
 * queue-fed for warm passes after a small motion, one query per lane otherwise.  chunk_tiles_* <= 0 keep the current value (4 / 4). */
void sga_set_search_mode(int queue, int chunk_tiles_cold, int chunk_tiles_warm);
double sga_get_warm_limit(void);
/* Passes of each kind since the problem was created, and the number of source points that had to walk in the warm passes. */
int sga_problem_get_pass_stats(sga_context* ctx, const sga_problem* problem, uint64_t* cold_passes, uint64_t* warm_passes, uint64_t* walked_points);

/* ---- the driver: Registration<>::align + optimizers (registration/registration.hpp:33-54, optimizer.hpp:24-149) -------- */
typedef struct sga_registration_setting {
  sga_factor_params factor;  /* PointFactor + CorrespondenceRejector */
  int optimizer;             /* sga_optimizer_kind */
  int max_iterations;        /* optimizer.hpp: 20 */
  int max_inner_iterations;  /* LM: 10 */
  double init_lambda;        /* LM: 1e-3 */
  double lambda_factor;      /* LM: 10 */
  double gn_lambda;          /* GN: 1e-6 */
  double translation_eps;    /* termination_criteria.hpp: 1e-3 */
  double rotation_eps;       /* 0.1 deg in rad */
  int verbose;
  /* general_factor.hpp:41-75 RestrictDoFFactor: if restrict_dof_lambda > 0, H += lambda * diag(|mask - 1|) */
  double restrict_dof_lambda;
  double restrict_dof_mask[6];
} sga_registration_setting;
void sga_registration_setting_default(sga_registration_setting* s);

/* registration_result.hpp:11-30, field for field */
typedef struct sga_result {
  double T_target_source[16];
  int converged;
  uint64_t iterations;
  uint64_t num_inliers;
  double H[36];
  double b[6];
  double error;
} sga_result;

/* Registration<Factor, ParallelReductionHIP>::align(target, source, target_tree, init_T): everything device-resident, LM/GN on the host. */
int sga_align(sga_context* ctx, const sga_index* target, const sga_cloud* source, const double init_T[16], const sga_registration_setting* setting, sga_result* out);
/* Same, on an existing problem (re-uses the sorted source and the factor buffers). */
int sga_align_problem(sga_context* ctx, sga_problem* problem, const double init_T[16], const sga_registration_setting* setting, sga_result* out);

/* ---- one registration over several GPUs of THIS process ------------------------------------------------------------------------
 * The single-process form of the sharded path and the analogue of ParallelReductionOMP::num_threads (registration/reduction_omp.hpp:22,72:
 * the loop over the source points, :32-58, is what gets partitioned).  Shard g owns the source points [g n / G, (g + 1) n / G) of the
 * caller's order and their factor state on device devices[g]; the target and its search index are replicated on every device.  A
 * linearization enqueues the pass on every device from the calling thread, collects the G accumulators and adds them in shard order on
 * the host (bit-reproducible; no collective, no communicator, no launcher).  The same device may be listed more than once (logical
 * shards on one GPU: how the path is tested on single-GPU machines).  Inputs use the reference's PointCloud layout like
 * sga_cloud_create_f64.  The process-per-GPU form with an RCCL all-reduce on the stream is sga_comm_init. */
typedef struct sga_multi sga_multi;
int sga_multi_create(const int* devices, int num_devices, sga_multi** out);
int sga_multi_destroy(sga_multi* m);
int sga_multi_num_devices(const sga_multi* m);
int sga_multi_set_target_f64(sga_multi* m, const double* xyzw, const double* normals4, const double* cov4x4, size_t n);
/* the same with fp32 arrays in the layout of sga_cloud_create_f32 (xyz n*3, normals n*3, cov6 n*6: xx xy xz yy yz zz) */
int sga_multi_set_target_f32(sga_multi* m, const float* xyz, const float* normals3, const float* cov6, size_t n);
int sga_multi_set_source_f32(sga_multi* m, const float* xyz, const float* normals3, const float* cov6, size_t n, const double init_T[16]);
/* the same with xyz_rel relative to `origin` (see sga_cloud_create_f32_origin): a caller that repacks double clouds subtracts in double while it repacks */
int sga_multi_set_target_f32_origin(sga_multi* m, const float* xyz_rel, const float* normals3, const float* cov6, size_t n, const double origin[3]);
int sga_multi_set_source_f32_origin(sga_multi* m, const float* xyz_rel, const float* normals3, const float* cov6, size_t n, const double origin[3], const double init_T[16]);
/* a Gaussian voxel map as the target (see sga_index_create_voxelmap_from_voxels): replicated on every device */
int sga_multi_set_target_voxels(sga_multi* m, double leaf_size, const int32_t* coords, const double* means3, const double* cov6, size_t n);
/* A custom CorrespondenceRejector for the whole registration (see sga_problem_set_rejector): the batch callback is invoked once per shard and
 * linearization with the shard's range [first, first + n) of the caller's source order; fn = NULL restores the built-in rejectors. */
typedef int (*sga_multi_rejector_fn)(void* user, const double T[16], size_t first, size_t n, const int64_t* target_index, const float* sq_dist, unsigned char* reject);
int sga_multi_set_rejector(sga_multi* m, sga_multi_rejector_fn fn, void* user);
/* search offsets of the voxel-map target on every device (sga_voxelmap_set_search_offsets) */
int sga_multi_set_search_offsets(sga_multi* m, int num_offsets);
/* a flat voxel map as the target (see sga_index_create_flatmap_from_voxels): replicated on every device */
int sga_multi_set_target_flat_voxels(sga_multi* m, double leaf_size, const int32_t* coords, const uint32_t* counts, const double* points3, const double* cov6, int search_offsets, size_t n);
int sga_multi_set_source_f64(sga_multi* m, const double* xyzw, const double* normals4, const double* cov4x4, size_t n, const double init_T[16]);
/* Reduction::linearize / ::error over all shards (reduction_omp.hpp:24-70), Registration<>::align (registration.hpp:33-43) on the host */
int sga_multi_linearize(sga_multi* m, const sga_factor_params* params, const double T[16], double H[36], double b[6], double* e, uint64_t* num_inliers);
int sga_multi_error(sga_multi* m, const sga_factor_params* params, const double T[16], double* e);
int sga_multi_align(sga_multi* m, const double init_T[16], const sga_registration_setting* setting, sga_result* out);
/* every registration starts without search hints (sga_multi_align calls it; callers that drive linearize / error themselves call it per align) */
int sga_multi_reset_search_state(sga_multi* m);
/* factor state of the whole source in the caller's order (see sga_problem_get_factors) */
int sga_multi_get_factors(sga_multi* m, int64_t* target_index, float* mahalanobis6);

/* The optimizer alone, over user reductions (the reference's Optimizer::optimize with a pluggable Reduction, optimizer.hpp:27-36):
 * used for sharded multi-GPU runs where linearize = local kernels + all-reduce.  Callbacks return 0 on success. */
typedef int (*sga_linearize_fn)(void* user, const double T[16], double H[36], double b[6], double* e, uint64_t* num_inliers);
typedef int (*sga_error_fn)(void* user, const double T[16], double* e);
int sga_optimize(const sga_registration_setting* setting, const double init_T[16], sga_linearize_fn linearize, sga_error_fn error, void* user, sga_result* out);

/* util/lie.hpp:77-96 se3_exp (host). */
void sga_se3_exp(const double twist[6], double T[16]);

#ifdef __cplusplus
}
#endif
#endif /* SMALL_GICP_AMD_H */
