// sga_multi: ONE registration spread over several GPUs of this process — the single-process form of the sharded path and the analogue of
// ParallelReductionOMP::num_threads (registration/reduction_omp.hpp:22,72: the loop over the source points, :32-58, is what gets
// partitioned).  Shard g owns the source points [g n / G, (g + 1) n / G) of the caller's order with their factor state; the target and
// its search index are replicated on every device (52 MB per million points: nothing next to 288 GB).  A linearization enqueues the pass
// on every device from one host thread, then collects the G rows of 96 doubles (the system AND the error-model moments are sums over
// source points) and adds them in shard order on the host: bit-reproducible, no collective, no second thread.  The process-per-GPU form
// with an RCCL all-reduce on the stream is sga_comm_init (comm.hip); this one needs no launcher and no communicator.
#include <memory>
#include <vector>

#include "common.hpp"

namespace sga {
int linearize_enqueue(sga_context* ctx, sga_problem* pb, const sga_factor_params* fp, const double T[16], unsigned long long* seq_out, int* count_out);
int linearize_collect(sga_context* ctx, sga_problem* pb, const double T[16], unsigned long long seq, int count);
int error_enqueue(sga_context* ctx, sga_problem* pb, const sga_factor_params* fp, const double T[16], unsigned long long* seq_out);
int error_collect(sga_context* ctx, unsigned long long seq, double* e);
double error_model_value(const double* acc96, const double T_lin[16], const double T[16]);
bool error_model_enabled();
const double* problem_pose(const sga_problem* pb, const double T[16], double Td[16]);
void problem_system_to_caller(const sga_problem* pb, double H[36], double b[6]);
int cloud_create_f32_about(sga_context* ctx, const float* xyz, const float* normals, const float* cov6, size_t n, const double origin[3], sga_cloud** out);
void host_bbox_f32(const float* xyz, size_t n, double lo[3], double hi[3]);
void host_bbox_f64(const double* xyzw, size_t n, double lo[3], double hi[3]);
}  // namespace sga

using namespace sga;

struct sga_multi {
  struct Shard {
    int device = 0;
    sga_context* ctx = nullptr;
    sga_cloud* target = nullptr;
    sga_index* index = nullptr;
    sga_cloud* source = nullptr;
    sga_problem* problem = nullptr;
    size_t first = 0, count = 0;  // its range of the caller's source order
    sga_multi* owner = nullptr;   // for the rejector trampoline
  };
  sga_multi_rejector_fn rejector_fn = nullptr;
  void* rejector_user = nullptr;
  std::vector<Shard> shards;
  size_t n_target = 0, n_source = 0;
  bool has_target = false, has_source = false;
  double model[SGA_MODEL_DOUBLES] = {0};  // sum over the shards of the last linearization's rows
  double model_T[16] = {0};
  bool model_valid = false;
  ~sga_multi() {
    for (auto& s : shards) {
      if (s.problem) sga_problem_destroy(s.problem);
      if (s.source) sga_cloud_destroy(s.source);
      if (s.index) sga_index_destroy(s.index);
      if (s.target) sga_cloud_destroy(s.target);
      if (s.ctx) sga_context_destroy(s.ctx);
    }
  }
};

namespace {
struct Entered {  // SGA_ENTER for a loop over contexts: device + the thread's current stream, restored when the scope ends
  explicit Entered(sga_context* c) : scope(c->stream) {}
  StreamScope scope;
};

// before a new target: the problems of the source refer to the old index
void drop_target(sga_multi* m) {
  m->model_valid = false;
  for (auto& s : m->shards) {
    if (s.problem) sga_problem_destroy(s.problem);
    s.problem = nullptr;
    if (s.index) sga_index_destroy(s.index);
    s.index = nullptr;
    if (s.target) sga_cloud_destroy(s.target);
    s.target = nullptr;
  }
  m->has_target = false;
}

int multi_lin_cb(void* user, const double T[16], double H[36], double b[6], double* e, uint64_t* inl);
int multi_err_cb(void* user, const double T[16], double* e);
// a shard's problem asks about ITS source points: the caller is told where they start in its own order
int shard_rejector(void* user, const double T[16], size_t n, const int64_t* target_index, const float* sq_dist, unsigned char* reject) {
  auto* s = static_cast<sga_multi::Shard*>(user);
  return s->owner->rejector_fn(s->owner->rejector_user, T, s->first, n, target_index, sq_dist, reject);
}
void install_rejector(sga_multi* m) {
  for (auto& s : m->shards) {
    s.owner = m;
    if (s.problem) (void)sga_problem_set_rejector(s.problem, m->rejector_fn ? shard_rejector : nullptr, &s);
  }
}
struct MultiReduction {
  sga_multi* m;
  const sga_factor_params* fp;
};
}  // namespace

extern "C" {

int sga_multi_create(const int* devices, int num_devices, sga_multi** out) {
  if (!devices || num_devices < 1 || num_devices > 64 || !out) return fail(SGA_ERR_INVALID, "sga_multi_create: need 1 .. 64 devices");
  *out = nullptr;
  std::unique_ptr<sga_multi> m(new sga_multi);
  m->shards.resize(static_cast<size_t>(num_devices));
  for (int g = 0; g < num_devices; g++) {
    m->shards[g].device = devices[g];
    SGA_TRY(sga_context_create(devices[g], &m->shards[g].ctx));  // the same device twice = two logical shards on it (tests on 1-GPU boxes)
  }
  *out = m.release();
  return SGA_OK;
}

int sga_multi_destroy(sga_multi* m) {
  delete m;
  return SGA_OK;
}

int sga_multi_num_devices(const sga_multi* m) { return m ? static_cast<int>(m->shards.size()) : 0; }

int sga_multi_set_target_f64(sga_multi* m, const double* xyzw, const double* normals4, const double* cov4x4, size_t n) {
  if (!m || (n > 0 && !xyzw)) return fail(SGA_ERR_INVALID, "null argument");
  drop_target(m);
  {  // built ONCE, on the first device; the other shards get copies of the finished index (sga_index_clone: peer copies)
    auto& s0 = m->shards[0];
    SGA_TRY(sga_cloud_create_f64(s0.ctx, xyzw, normals4, cov4x4, n, &s0.target));
    SGA_TRY(sga_index_build_kdtree(s0.ctx, s0.target, &s0.index));
    for (size_t g = 1; g < m->shards.size(); g++) SGA_TRY(sga_index_clone(m->shards[g].ctx, s0.index, &m->shards[g].index));
  }
  m->n_target = n;
  m->has_target = true;
  return SGA_OK;
}

// The same two setters with fp32 arrays (xyz n*3, normals n*3, cov6 n*6: the layout of sga_cloud_create_f32): a caller that repacks its
// clouds anyway (ParallelReductionHIP: from the reference's AoS doubles) converts while it repacks, on its own threads, and spares the
// serial double -> float pass of the f64 entry points (150 ms of a 250 ms first bind at 2 x 1M points).
// `relative`: xyz are given relative to `origin` (sga_cloud_create_f32_origin); otherwise absolute, the device frame chosen here — ONE for
// the whole cloud, so that the shards' accumulators add up (common.hpp: device frames)
static int multi_set_target_f32(sga_multi* m, const float* xyz, const float* normals3, const float* cov6, size_t n, const double* origin, bool relative) {
  if (!m || (n > 0 && !xyz)) return fail(SGA_ERR_INVALID, "null argument");
  drop_target(m);
  {  // built ONCE, on the first device; the other shards get copies of the finished index (sga_index_clone: peer copies) — a first bind
     // of G devices costs one build + G - 1 copies instead of G builds one after the other
    auto& s0 = m->shards[0];
    if (relative)
      SGA_TRY(sga_cloud_create_f32_origin(s0.ctx, xyz, normals3, cov6, n, origin, &s0.target));
    else
      SGA_TRY(sga_cloud_create_f32(s0.ctx, xyz, normals3, cov6, n, &s0.target));
    SGA_TRY(sga_index_build_kdtree(s0.ctx, s0.target, &s0.index));
    for (size_t g = 1; g < m->shards.size(); g++) SGA_TRY(sga_index_clone(m->shards[g].ctx, s0.index, &m->shards[g].index));
  }
  m->n_target = n;
  m->has_target = true;
  return SGA_OK;
}

static int multi_set_source_f32(sga_multi* m, const float* xyz, const float* normals3, const float* cov6, size_t n, const double init_T[16], const double* origin_in, bool relative) {
  if (!m || (n > 0 && !xyz)) return fail(SGA_ERR_INVALID, "null argument");
  if (!m->has_target) return fail(SGA_ERR_INVALID, "sga_multi_set_source_f32 before a target was set");
  m->model_valid = false;
  m->has_source = false;
  double origin[3] = {0, 0, 0};
  if (relative) {
    if (origin_in)
      for (int k = 0; k < 3; k++) origin[k] = origin_in[k];
  } else {
    double lo[3], hi[3];
    host_bbox_f32(xyz, n, lo, hi);
    sga_choose_origin(lo, hi, origin);
  }
  const size_t G = m->shards.size();
  for (size_t g = 0; g < G; g++) {
    auto& s = m->shards[g];
    if (s.problem) sga_problem_destroy(s.problem);
    s.problem = nullptr;
    if (s.source) sga_cloud_destroy(s.source);
    s.source = nullptr;
    s.first = n * g / G;
    s.count = n * (g + 1) / G - s.first;
    const float *px = xyz + 3 * s.first, *pn = normals3 ? normals3 + 3 * s.first : nullptr, *pc = cov6 ? cov6 + 6 * s.first : nullptr;
    if (relative)
      SGA_TRY(sga_cloud_create_f32_origin(s.ctx, px, pn, pc, s.count, origin, &s.source));
    else
      SGA_TRY(cloud_create_f32_about(s.ctx, px, pn, pc, s.count, origin, &s.source));
    SGA_TRY(sga_problem_create(s.ctx, s.index, s.source, init_T, &s.problem));
  }
  m->n_source = n;
  m->has_source = true;
  if (m->rejector_fn) install_rejector(m);  // (a rejector set earlier stays in force for the new problems)
  return SGA_OK;
}

int sga_multi_set_target_f32(sga_multi* m, const float* xyz, const float* normals3, const float* cov6, size_t n) { return multi_set_target_f32(m, xyz, normals3, cov6, n, nullptr, false); }
int sga_multi_set_target_f32_origin(sga_multi* m, const float* xyz_rel, const float* normals3, const float* cov6, size_t n, const double origin[3]) {
  return multi_set_target_f32(m, xyz_rel, normals3, cov6, n, origin, true);
}
int sga_multi_set_source_f32(sga_multi* m, const float* xyz, const float* normals3, const float* cov6, size_t n, const double init_T[16]) {
  return multi_set_source_f32(m, xyz, normals3, cov6, n, init_T, nullptr, false);
}
int sga_multi_set_source_f32_origin(sga_multi* m, const float* xyz_rel, const float* normals3, const float* cov6, size_t n, const double origin[3], const double init_T[16]) {
  return multi_set_source_f32(m, xyz_rel, normals3, cov6, n, init_T, origin, true);
}

int sga_multi_set_target_voxels(sga_multi* m, double leaf, const int32_t* coords, const double* means3, const double* cov6, size_t n) {
  if (!m) return fail(SGA_ERR_INVALID, "null argument");
  drop_target(m);
  SGA_TRY(sga_index_create_voxelmap_from_voxels(m->shards[0].ctx, leaf, coords, means3, cov6, n, &m->shards[0].index));
  for (size_t g = 1; g < m->shards.size(); g++) SGA_TRY(sga_index_clone(m->shards[g].ctx, m->shards[0].index, &m->shards[g].index));
  m->n_target = n;
  m->has_target = true;
  return SGA_OK;
}

int sga_multi_set_rejector(sga_multi* m, sga_multi_rejector_fn fn, void* user) {
  if (!m) return fail(SGA_ERR_INVALID, "null argument");
  if (fn && m->has_target && m->shards[0].index && m->shards[0].index->kind != SGA_INDEX_KDTREE) return fail(SGA_ERR_UNSUPPORTED, "host rejectors need a kd-tree target");
  m->rejector_fn = fn;
  m->rejector_user = user;
  m->model_valid = false;
  install_rejector(m);
  return SGA_OK;
}

int sga_multi_set_search_offsets(sga_multi* m, int num_offsets) {
  if (!m || !m->has_target) return fail(SGA_ERR_INVALID, "sga_multi_set_search_offsets before a voxel map target was set");
  for (auto& s : m->shards) SGA_TRY(sga_voxelmap_set_search_offsets(s.index, num_offsets));
  m->model_valid = false;
  return SGA_OK;
}

int sga_multi_set_target_flat_voxels(sga_multi* m, double leaf, const int32_t* coords, const uint32_t* counts, const double* points3, const double* cov6, int search_offsets, size_t n) {
  if (!m) return fail(SGA_ERR_INVALID, "null argument");
  drop_target(m);
  SGA_TRY(sga_index_create_flatmap_from_voxels(m->shards[0].ctx, leaf, coords, counts, points3, cov6, search_offsets, n, &m->shards[0].index));
  for (size_t g = 1; g < m->shards.size(); g++) SGA_TRY(sga_index_clone(m->shards[g].ctx, m->shards[0].index, &m->shards[g].index));
  m->n_target = n;
  m->has_target = true;
  return SGA_OK;
}

int sga_multi_set_source_f64(sga_multi* m, const double* xyzw, const double* normals4, const double* cov4x4, size_t n, const double init_T[16]) {
  if (!m || (n > 0 && !xyzw)) return fail(SGA_ERR_INVALID, "null argument");
  if (!m->has_target) return fail(SGA_ERR_INVALID, "sga_multi_set_source_f64 before sga_multi_set_target_f64");
  m->model_valid = false;
  m->has_source = false;
  double lo[3], hi[3], origin[3];
  host_bbox_f64(xyzw, n, lo, hi);
  sga_choose_origin(lo, hi, origin);  // ONE device frame for all shards: their accumulators are added (common.hpp)
  const size_t G = m->shards.size();
  for (size_t g = 0; g < G; g++) {
    auto& s = m->shards[g];
    if (s.problem) sga_problem_destroy(s.problem);
    s.problem = nullptr;
    if (s.source) sga_cloud_destroy(s.source);
    s.source = nullptr;
    s.first = n * g / G;
    s.count = n * (g + 1) / G - s.first;
    SGA_TRY(sga_cloud_create_f64_origin(s.ctx, xyzw + 4 * s.first, normals4 ? normals4 + 4 * s.first : nullptr, cov4x4 ? cov4x4 + 16 * s.first : nullptr, s.count, origin, &s.source));
    SGA_TRY(sga_problem_create(s.ctx, s.index, s.source, init_T, &s.problem));
  }
  m->n_source = n;
  m->has_source = true;
  if (m->rejector_fn) install_rejector(m);  // (a rejector set earlier stays in force for the new problems)
  return SGA_OK;
}

int sga_multi_linearize(sga_multi* m, const sga_factor_params* fp, const double T[16], double H[36], double b[6], double* e, uint64_t* num_inliers) {
  if (!m || !fp || !T || !H || !b || !e) return fail(SGA_ERR_INVALID, "null argument");
  if (!m->has_source) return fail(SGA_ERR_INVALID, "sga_multi_linearize without clouds");
  const size_t G = m->shards.size();
  std::vector<unsigned long long> seq(G, 0);
  std::vector<int> count(G, 0);
  m->model_valid = false;
  for (size_t g = 0; g < G; g++) {  // every device gets its pass before the host waits for any
    auto& s = m->shards[g];
    SGA_HIP(hipSetDevice(s.device));
    Entered in(s.ctx);
    const int rc = linearize_enqueue(s.ctx, s.problem, fp, T, &seq[g], &count[g]);
    if (rc != SGA_OK) {  // the passes already enqueued on the earlier shards are collected before the error is reported
      const std::string why = sga_last_error();
      for (size_t k = 0; k < g; k++) {
        auto& e = m->shards[k];
        (void)hipSetDevice(e.device);
        Entered in2(e.ctx);
        (void)linearize_collect(e.ctx, e.problem, T, seq[k], count[k]);
      }
      return fail(rc, "%s", why.c_str());
    }
  }
  double acc[SGA_MODEL_DOUBLES] = {0};
  for (size_t g = 0; g < G; g++) {
    auto& s = m->shards[g];
    SGA_HIP(hipSetDevice(s.device));
    Entered in(s.ctx);
    SGA_TRY(linearize_collect(s.ctx, s.problem, T, seq[g], count[g]));
    for (int c = 0; c < count[g]; c++) acc[c] += s.ctx->h_accum[c];  // shard order: a fixed summation order
  }
  sga_unpack_accumulator(acc, H, b, e, num_inliers);
  problem_system_to_caller(m->shards[0].problem, H, b);  // all shards share one source frame (the setters see to it)
  if (count[0] == SGA_MODEL_DOUBLES) {
    double Tdev[16];
    memcpy(m->model, acc, sizeof(acc));
    memcpy(m->model_T, problem_pose(m->shards[0].problem, T, Tdev), sizeof(m->model_T));  // the model lives between the device frames
    m->model_valid = true;
  }
  return SGA_OK;
}

int sga_multi_error(sga_multi* m, const sga_factor_params* fp, const double T[16], double* e) {
  if (!m || !fp || !T || !e) return fail(SGA_ERR_INVALID, "null argument");
  if (!m->has_source) return fail(SGA_ERR_INVALID, "sga_multi_error without clouds");
  if (m->model_valid && fp->robust_kind == SGA_ROBUST_NONE && error_model_enabled()) {
    double Tdev[16];
    *e = error_model_value(m->model, m->model_T, problem_pose(m->shards[0].problem, T, Tdev));
    return SGA_OK;
  }
  const size_t G = m->shards.size();
  std::vector<unsigned long long> seq(G, 0);
  for (size_t g = 0; g < G; g++) {
    auto& s = m->shards[g];
    SGA_HIP(hipSetDevice(s.device));
    Entered in(s.ctx);
    SGA_TRY(error_enqueue(s.ctx, s.problem, fp, T, &seq[g]));
  }
  double sum = 0.0;
  for (size_t g = 0; g < G; g++) {
    auto& s = m->shards[g];
    SGA_HIP(hipSetDevice(s.device));
    Entered in(s.ctx);
    double part = 0.0;
    SGA_TRY(error_collect(s.ctx, seq[g], &part));
    sum += part;
  }
  *e = sum;
  return SGA_OK;
}

int sga_multi_get_factors(sga_multi* m, int64_t* target_index, float* mahalanobis6) {
  if (!m) return fail(SGA_ERR_INVALID, "null argument");
  if (!m->has_source) return fail(SGA_ERR_INVALID, "sga_multi_get_factors without clouds");
  for (auto& s : m->shards) {
    if (s.count == 0) continue;
    SGA_TRY(sga_problem_get_factors(s.ctx, s.problem, target_index ? target_index + s.first : nullptr, mahalanobis6 ? mahalanobis6 + 6 * s.first : nullptr));
  }
  return SGA_OK;
}

// every registration starts without search hints (sga_align_problem does the same): its result must not depend on earlier calls
int sga_multi_reset_search_state(sga_multi* m) {
  if (!m) return fail(SGA_ERR_INVALID, "null argument");
  for (auto& s : m->shards) {
    if (!s.problem) continue;
    SGA_HIP(hipSetDevice(s.device));
    Entered in(s.ctx);
    sga_problem* pb = s.problem;
    if (pb->n > 0 && pb->hint.n >= pb->n) SGA_HIP(hipMemsetAsync(pb->hint.p, 0xff, pb->n * sizeof(int), s.ctx->stream));
    if (pb->n > 0 && pb->hint2.n >= pb->n) SGA_HIP(hipMemsetAsync(pb->hint2.p, 0xff, pb->n * sizeof(int), s.ctx->stream));
    pb->prev_valid = false;
  }
  m->model_valid = false;
  return SGA_OK;
}

int sga_multi_align(sga_multi* m, const double init_T[16], const sga_registration_setting* setting, sga_result* out) {
  if (!m || !setting || !out) return fail(SGA_ERR_INVALID, "null argument");
  if (!m->has_source) return fail(SGA_ERR_INVALID, "sga_multi_align without clouds");
  static const double I16[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  if (m->n_target <= 10) std::fprintf(stderr, "warning: target point cloud is too small. |target|=%zu\n", m->n_target);  // registration.hpp:34-39
  if (m->n_source <= 10) std::fprintf(stderr, "warning: source point cloud is too small. |source|=%zu\n", m->n_source);
  SGA_TRY(sga_multi_reset_search_state(m));
  MultiReduction r{m, &setting->factor};
  return sga_optimize(setting, init_T ? init_T : I16, multi_lin_cb, multi_err_cb, &r, out);
}

}  // extern "C"

namespace {
int multi_lin_cb(void* user, const double T[16], double H[36], double b[6], double* e, uint64_t* inl) {
  auto* r = static_cast<MultiReduction*>(user);
  return sga_multi_linearize(r->m, r->fp, T, H, b, e, inl);
}
int multi_err_cb(void* user, const double T[16], double* e) {
  auto* r = static_cast<MultiReduction*>(user);
  return sga_multi_error(r->m, r->fp, T, e);
}
}  // namespace
