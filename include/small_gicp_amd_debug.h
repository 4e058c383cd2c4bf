/*
 * small_gicp_amd — diagnostics of the MI355X registration hot path.  NOT part of the drop-in boundary (small_gicp_amd.h): these entry
 * points expose how the searches went (for the scripts under scripts/ and for DESIGN.md's measurements), never what they returned.
 */
#ifndef SMALL_GICP_AMD_DEBUG_H
#define SMALL_GICP_AMD_DEBUG_H

#include "small_gicp_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Record, for the following linearization passes of this problem, the number of kd-tree leaves each source point's search scanned
 * (source order of the engine, i.e. sorted); get returns the last pass's counts (n ints).  While enabled every pass walks the kd-tree. */
int sga_problem_set_search_stats(sga_context* ctx, sga_problem* pb, int enabled);
int sga_problem_get_search_stats(sga_context* ctx, const sga_problem* pb, int* leaves_per_point);
/* The source points in the engine's order (n x 4 floats: x, y, z, original index as bits). */
int sga_problem_get_sorted_points(sga_context* ctx, const sga_problem* pb, float* xyzw);
/* Cell-grid passes (cell_grid.hip) since the problem was created: out[0] = passes searched through the grid, out[1] = queries their
 * first ring left open (finished by the second kernel), out[2] = sum of the rings those queries then scanned, out[3] = cell edge in
 * micrometres (0: the target has no grid); out[4], out[5]: always 0 (counters of experiments removed in round 6). */
int sga_problem_get_grid_stats(const sga_problem* pb, uint64_t out[6]);
/* What the cell grid (a second, flat search structure of large kd-tree indices) is used for; results do not depend on it (tests compare
 * the searches): mode 0 nothing (no grid is built); 1 (default) the walkers of warm passes try its first ring before they walk the tree;
 * 2 also the cold passes of a registration but its first; 3 the first pass too; 4 every pass.  min_points: targets with fewer points get
 * no grid (default 65536; applies to indices built afterwards).  Negative arguments keep the current value.
 * Environment: SGA_GRID, SGA_GRID_MIN_POINTS. */
void sga_set_grid_mode(int mode, long long min_points);
/* Normal / covariance estimation: clouds of at most max_points points search their neighbours with one wave per query (csrc/knn_wave.hpp:
 * the form for clouds that do not fill the chip), larger ones with one query per lane.  Both are exact; the tests compare them.
 * Default 32768 (environment: SGA_KNN_WAVE_MAX); 0 = never. */
void sga_set_knn_wave_max(long long max_points);
/* The length scale of a kd-tree index: the geometric mean of the diagonals of its leaf boxes (a leaf = a neighbourhood of <= 8 points),
 * computed by the build.  The pass routing of the linearization measures the source's motion in units of it (csrc/linearize.hip:
 * routing_unit), so that the choice between the exact search kernels does not depend on the unit of length or the density of the cloud.
 * 0 while the build's kernels have not run yet (the value arrives as a late note, csrc/notes.hpp) or for other kinds of index. */
int sga_index_spacing(const sga_index* index, double* spacing);
/* GPU time (HIP events) between two points of the context's stream: start() records an event, stop() records another, waits for it and
 * returns the milliseconds in between — the kernel times of bench.py's per-stage roofline lines (voxel grid, index build, covariances). */
int sga_debug_timer_start(sga_context* ctx);
int sga_debug_timer_stop(sga_context* ctx, double* ms);
/* The host arithmetic of the frame check of sharded registrations (linearize.hip: problem_check_shard_frames), exposed so that it can be
 * tested without a device: pack() turns a source origin into the SGA_FRAME_CHECK_DOUBLES values a rank contributes to the all-reduce,
 * agree() says whether the ranks whose contributions were summed all named the same origin (exact for any origin, up to 1024 ranks). */
#define SGA_FRAME_CHECK_DOUBLES 32
void sga_debug_shard_frame_pack(const double origin[3], double out[SGA_FRAME_CHECK_DOUBLES]);
int sga_debug_shard_frame_agree(const double sum[SGA_FRAME_CHECK_DOUBLES]);
/* Diagnostics build only (make trips): loop-trip counters of the kd walk and the start / end clock of every search wave. */
int sga_debug_kd_trips(unsigned long long* out16);
int sga_debug_kd_wave_times(unsigned long long* out, int waves);

#ifdef __cplusplus
}
#endif
#endif /* SMALL_GICP_AMD_DEBUG_H */
