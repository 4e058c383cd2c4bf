// Context (one GPU + one stream), error plumbing and device-resident clouds.
#include "common.hpp"

#include <cmath>

#include <map>
#include <mutex>

namespace sga {
void preload_hot_kernels();  // linearize.hip

// The first kernel of a queue that needs scratch memory (a few spilled registers are enough: certify_linearize_kernel has 20 bytes per
// lane) makes the runtime allocate the queue's scratch arena: ~0.2 ms, paid in the middle of somebody's first registration.  A context
// pays it when it is created instead: one tiny launch with 256 bytes of private memory per lane on its stream.
__global__ void scratch_prime_kernel(int* out, int n) {
  volatile int buf[64];
  for (int i = 0; i < 64; i++) buf[i] = i * n;
  int t = 0;
  for (int i = 0; i < 64; i++) t += buf[(i * 7 + n) & 63];
  if (n < 0) *out = t;  // never taken: the array must not be optimised away
}
}  // namespace sga

#include <unordered_map>

namespace sga {

static thread_local char g_err[1024] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// ---- caching device allocator, stream ordered (see common.hpp) -------------------------------------------------------------------
namespace {
thread_local hipStream_t g_cur_stream = nullptr;  // set by SGA_ENTER for the duration of an entry point

struct FreeKey {
  int device;
  hipStream_t stream;  // nullptr = the shared pool: blocks nobody is using any more
  size_t bucket;
  bool operator<(const FreeKey& o) const {
    if (device != o.device) return device < o.device;
    if (stream != o.stream) return stream < o.stream;
    return bucket < o.bucket;
  }
};
struct PendingBlock {
  void* p;
  int device;
  size_t bucket;
  std::vector<hipEvent_t> events;  // one per stream that was busy when the block was freed
};
struct DevCache {
  std::mutex mu;
  std::unordered_map<void*, std::pair<int, size_t>> live;  // every block handed out: device, bucket size
  std::map<FreeKey, std::vector<void*>> free_blocks;
  std::vector<PendingBlock> pending;                       // freed outside an entry point: reusable once their events have completed
  std::vector<std::pair<int, hipStream_t>> streams;        // the streams of the live contexts
  std::vector<hipEvent_t> event_pool;
  size_t cached_bytes = 0;
  int contexts = 0;
  uint64_t n_malloc = 0, n_stream_hit = 0, n_pool_hit = 0, n_pending_hit = 0, n_deferred = 0;  // statistics (sga_allocator_stats)
};
DevCache& dev_cache() {
  static DevCache* c = new DevCache;  // never destroyed: no HIP calls during static destruction
  return *c;
}
constexpr size_t kCacheLimitBytes = 8ull << 30;

// < 1 MiB: next power of two (>= 256 B); above: 8 buckets per octave (<= 12.5 % slack)
size_t bucket_bytes(size_t bytes) {
  size_t p2 = 256;
  while (p2 < bytes) p2 <<= 1;
  if (p2 <= (1ull << 20)) return p2;
  const size_t step = p2 >> 4;  // p2/2 < bytes <= p2: steps of (p2/2)/8
  return ((bytes + step - 1) / step) * step;
}

void recycle_events(DevCache& c, std::vector<hipEvent_t>& evs) {
  for (hipEvent_t e : evs) c.event_pool.push_back(e);
  evs.clear();
}

// move the pending blocks whose events have all completed into the shared pool
void collect_pending_locked(DevCache& c) {
  size_t w = 0;
  for (size_t i = 0; i < c.pending.size(); i++) {
    PendingBlock& b = c.pending[i];
    bool done = true;
    for (hipEvent_t e : b.events)
      if (hipEventQuery(e) == hipErrorNotReady) {
        done = false;
        break;
      }
    if (done) {
      recycle_events(c, b.events);
      c.free_blocks[{b.device, nullptr, b.bucket}].push_back(b.p);
    } else {
      if (w != i) c.pending[w] = std::move(b);
      w++;
    }
  }
  c.pending.resize(w);
  (void)hipGetLastError();  // hipEventQuery's hipErrorNotReady is sticky in hipGetLastError
}

void release_cached_locked(DevCache& c) {
  for (PendingBlock& b : c.pending) {
    for (hipEvent_t e : b.events) (void)hipEventSynchronize(e);
    recycle_events(c, b.events);
    (void)hipFree(b.p);
  }
  c.pending.clear();
  for (auto& kv : c.free_blocks)
    for (void* q : kv.second) (void)hipFree(q);
  c.free_blocks.clear();
  c.cached_bytes = 0;
}

void* take_locked(DevCache& c, const FreeKey& key) {
  auto it = c.free_blocks.find(key);
  if (it == c.free_blocks.end() || it->second.empty()) return nullptr;
  void* p = it->second.back();
  it->second.pop_back();
  return p;
}
}  // namespace

StreamScope::StreamScope(hipStream_t s) : prev(g_cur_stream) { g_cur_stream = s; }
StreamScope::~StreamScope() { g_cur_stream = prev; }

int dev_alloc(void** p, size_t bytes) {
  *p = nullptr;
  if (bytes == 0) return SGA_OK;
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) return fail(SGA_ERR_HIP, "hipGetDevice failed");
  const size_t bucket = bucket_bytes(bytes);
  DevCache& c = dev_cache();
  std::lock_guard<std::mutex> lock(c.mu);
  // same stream first (stream order makes the reuse safe), then blocks nobody uses, then blocks whose last users have finished
  if (g_cur_stream != nullptr && (*p = take_locked(c, {device, g_cur_stream, bucket})) != nullptr) c.n_stream_hit++;
  if (!*p && (*p = take_locked(c, {device, nullptr, bucket})) != nullptr) c.n_pool_hit++;
  if (!*p && !c.pending.empty()) {
    collect_pending_locked(c);
    if ((*p = take_locked(c, {device, nullptr, bucket})) != nullptr) c.n_pending_hit++;
  }
  if (*p) {
    c.cached_bytes -= bucket;
  } else {
    c.n_malloc++;
    hipError_t e = hipMalloc(p, bucket);
    if (e != hipSuccess) {  // out of memory with blocks parked in the cache: give them back and retry once
      (void)hipGetLastError();
      release_cached_locked(c);
      e = hipMalloc(p, bucket);
    }
    if (e != hipSuccess) {
      *p = nullptr;
      return fail(SGA_ERR_HIP, "hipMalloc(%zu bytes) -> %s", bucket, hipGetErrorString(e));
    }
  }
  c.live[*p] = {device, bucket};
  return SGA_OK;
}

void dev_free(void* p) {
  if (!p) return;
  DevCache& c = dev_cache();
  std::lock_guard<std::mutex> lock(c.mu);
  auto it = c.live.find(p);
  if (it == c.live.end()) {
    (void)hipFree(p);
    return;
  }
  const int device = it->second.first;
  const size_t bucket = it->second.second;
  c.live.erase(it);
  if (c.contexts == 0 || c.cached_bytes + bucket > kCacheLimitBytes) {
    (void)hipFree(p);  // synchronises the device: safe whatever is in flight
    return;
  }
  c.cached_bytes += bucket;
  if (g_cur_stream != nullptr) {
    // Inside an entry point: later work on the same stream may get this block at once — provided no OTHER stream of the device has
    // work in flight.  Buffers of long-lived shared objects (an index's attributes, a problem's mahalanobis cache, the rejector
    // flags) may be read by kernels another context has enqueued on its own stream (sga_linearize_async on context B while
    // context A refreshes the index): with such a stream busy the block takes the event-deferred path below.
    bool others_busy = false;
    for (const auto& ds : c.streams)
      if (ds.first == device && ds.second != g_cur_stream && hipStreamQuery(ds.second) == hipErrorNotReady) others_busy = true;
    (void)hipGetLastError();
    if (!others_busy) {
      c.free_blocks[{device, g_cur_stream, bucket}].push_back(p);
      return;
    }
  }
  // outside an entry point (destroy functions): kernels on any stream of the device may still use the block
  PendingBlock b{p, device, bucket, {}};
  int cur = -1;
  (void)hipGetDevice(&cur);
  for (const auto& ds : c.streams) {
    if (ds.first != device) continue;
    if (hipStreamQuery(ds.second) != hipErrorNotReady) continue;  // idle: nothing of it can touch the block
    if (cur != device) {
      (void)hipSetDevice(device);
      cur = device;
    }
    hipEvent_t e = nullptr;
    if (!c.event_pool.empty()) {
      e = c.event_pool.back();
      c.event_pool.pop_back();
    } else if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
      e = nullptr;
    }
    if (e == nullptr || hipEventRecord(e, ds.second) != hipSuccess) {
      (void)hipStreamSynchronize(ds.second);  // cannot track it: wait for it instead
      if (e) c.event_pool.push_back(e);
      continue;
    }
    b.events.push_back(e);
  }
  (void)hipGetLastError();
  if (b.events.empty()) {
    c.free_blocks[{device, nullptr, bucket}].push_back(p);
  } else {
    c.n_deferred++;
    c.pending.push_back(std::move(b));
  }
}

static void dev_cache_context_created(int device, hipStream_t stream) {
  DevCache& c = dev_cache();
  std::lock_guard<std::mutex> lock(c.mu);
  c.contexts++;
  c.streams.push_back({device, stream});
}
// the context's stream has been synchronised: its blocks join the shared pool
static void dev_cache_context_destroyed(int device, hipStream_t stream) {
  DevCache& c = dev_cache();
  std::lock_guard<std::mutex> lock(c.mu);
  for (size_t i = 0; i < c.streams.size(); i++)
    if (c.streams[i].first == device && c.streams[i].second == stream) {
      c.streams.erase(c.streams.begin() + i);
      break;
    }
  for (auto it = c.free_blocks.begin(); it != c.free_blocks.end();) {
    if (it->first.device == device && it->first.stream == stream && stream != nullptr) {
      auto& pool = c.free_blocks[{device, nullptr, it->first.bucket}];
      pool.insert(pool.end(), it->second.begin(), it->second.end());
      it = c.free_blocks.erase(it);
    } else {
      ++it;
    }
  }
  if (--c.contexts <= 0) {
    c.contexts = 0;
    release_cached_locked(c);
    for (hipEvent_t e : c.event_pool) (void)hipEventDestroy(e);
    c.event_pool.clear();
  }
}

__global__ void pack_cloud_f32_kernel(const float* __restrict__ xyz, const float* __restrict__ nrm, const float* __restrict__ cov6, size_t n, float4* __restrict__ pts, float4* __restrict__ onrm, Cov8* __restrict__ ocov) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  pts[i] = make_float4(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], __uint_as_float(static_cast<uint32_t>(i)));
  if (nrm) onrm[i] = make_float4(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2], 0.f);
  if (cov6) {
    Cov8 c;
    c.xx = cov6[6 * i];
    c.xy = cov6[6 * i + 1];
    c.xz = cov6[6 * i + 2];
    c.yy = cov6[6 * i + 3];
    c.yz = cov6[6 * i + 4];
    c.zz = cov6[6 * i + 5];
    c.pad0 = c.pad1 = 0.f;
    ocov[i] = c;
  }
}

__global__ void unpack_cloud_kernel(const float4* __restrict__ pts, const float4* __restrict__ nrm, const Cov8* __restrict__ cov, size_t n, float* __restrict__ xyz, float* __restrict__ onrm, float* __restrict__ cov6) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= n) return;
  if (xyz) {
    const float4 p = pts[i];
    xyz[3 * i] = p.x;
    xyz[3 * i + 1] = p.y;
    xyz[3 * i + 2] = p.z;
  }
  if (onrm) {
    const float4 q = nrm[i];
    onrm[3 * i] = q.x;
    onrm[3 * i + 1] = q.y;
    onrm[3 * i + 2] = q.z;
  }
  if (cov6) {
    const Cov8 c = cov[i];
    cov6[6 * i] = c.xx;
    cov6[6 * i + 1] = c.xy;
    cov6[6 * i + 2] = c.xz;
    cov6[6 * i + 3] = c.yy;
    cov6[6 * i + 4] = c.yz;
    cov6[6 * i + 5] = c.zz;
  }
}

__global__ void slice_cloud_kernel(const float4* __restrict__ pts, const float4* __restrict__ nrm, const Cov8* __restrict__ cov, size_t first, size_t count, float4* __restrict__ opts, float4* __restrict__ onrm, Cov8* __restrict__ ocov) {
  const size_t i = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
  if (i >= count) return;
  float4 p = pts[first + i];
  p.w = __uint_as_float(static_cast<uint32_t>(i));  // indices of the slice start at 0
  opts[i] = p;
  if (nrm) onrm[i] = nrm[first + i];
  if (cov) ocov[i] = cov[first + i];
}

// ---- device frames (common.hpp) ----------------------------------------------------------------------------------------------
void choose_origin(const double lo[3], const double hi[3], double origin[3]) {
  for (int k = 0; k < 3; k++) {
    origin[k] = 0.0;
    if (!(lo[k] <= hi[k])) continue;  // empty or non-finite
    const double c = 0.5 * (lo[k] + hi[k]);
    if (c - c != 0.0) continue;
    origin[k] = kOriginQuantum * std::nearbyint(c / kOriginQuantum);
  }
}

void pose_to_device(const double T[16], const double o_s[3], const double o_t[3], double Td[16]) {
  for (int i = 0; i < 16; i++) Td[i] = T[i];
  for (int r = 0; r < 3; r++) Td[12 + r] = (T[r] * o_s[0] + T[4 + r] * o_s[1] + T[8 + r] * o_s[2]) + (T[12 + r] - o_t[r]);  // R o_s + (t - o_t)
}

void system_to_caller(const double o[3], double H[36], double b[6]) {
  // A = [[I, 0], [X, I]], X = -skew(o):  H = A^T H' A, b = A^T b'  (J = J' A with J' = [R skew(p'), -R], p = p' + o)
  const double X[3][3] = {{0, o[2], -o[1]}, {-o[2], 0, o[0]}, {o[1], -o[0], 0}};
  double HA[6][6];  // H' A: columns 0..2 get H'[:, 3..5] X added
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      double v = H[6 * i + j];
      if (j < 3)
        for (int k = 0; k < 3; k++) v += H[6 * i + 3 + k] * X[k][j];
      HA[i][j] = v;
    }
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      double v = HA[i][j];
      if (i < 3)
        for (int k = 0; k < 3; k++) v += X[k][i] * HA[3 + k][j];  // (A^T)[i][3 + k] = X[k][i]
      H[6 * i + j] = v;
    }
  for (int i = 0; i < 3; i++)
    for (int k = 0; k < 3; k++) b[i] += X[k][i] * b[3 + k];
}

int ensure_temp(sga_context* ctx, size_t bytes) { return ctx->d_temp.reserve(bytes); }

}  // namespace sga

void sga_profile_collect_pending(sga_context* ctx);

using namespace sga;

extern "C" {

const char* sga_last_error(void) { return g_err; }

void sga_allocator_stats(uint64_t out[5]) {
  DevCache& c = dev_cache();
  std::lock_guard<std::mutex> lock(c.mu);
  out[0] = c.n_malloc;
  out[1] = c.n_stream_hit;
  out[2] = c.n_pool_hit;
  out[3] = c.n_pending_hit;
  out[4] = c.n_deferred;
}
const char* sga_version(void) { return "small_gicp_amd 0.1.0 (gfx950)"; }

int sga_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static int context_create_impl(int device, void* stream, bool borrow, sga_context** out) {
  if (!out) return fail(SGA_ERR_INVALID, "null out");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail(SGA_ERR_NO_DEVICE, "no HIP device available (%s): small_gicp_amd has no CPU fallback", e == hipSuccess ? "count=0" : hipGetErrorString(e));
  if (device < 0 || device >= n) return fail(SGA_ERR_INVALID, "device %d out of range [0,%d)", device, n);
  SGA_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  SGA_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(SGA_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
  auto* ctx = new sga_context;
  ctx->device = device;
  ctx->num_cus = prop.multiProcessorCount;
  if (borrow) {
    ctx->stream = static_cast<hipStream_t>(stream);
    ctx->owns_stream = false;
  } else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
      delete ctx;
      return fail(SGA_ERR_HIP, "hipStreamCreate failed");
    }
    ctx->owns_stream = true;
  }
  dev_cache_context_created(device, ctx->stream);
  ctx->registered = true;
  StreamScope scope(ctx->stream);
  int rc = ctx->d_accum.alloc(128);
  if (rc == SGA_OK) rc = ctx->d_ticket.alloc(16);
  if (rc == SGA_OK && hipMemsetAsync(ctx->d_ticket.p, 0, 16 * sizeof(unsigned), ctx->stream) != hipSuccess) rc = fail(SGA_ERR_HIP, "hipMemsetAsync failed");
  if (rc == SGA_OK && hipHostMalloc(reinterpret_cast<void**>(&ctx->h_accum), 160 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) rc = fail(SGA_ERR_HIP, "hipHostMalloc failed");
  if (rc == SGA_OK) {
    std::memset(ctx->h_accum, 0, 160 * sizeof(double));
    ctx->h_scratch = reinterpret_cast<int*>(ctx->h_accum + 136);  // doubles [136, 144) of the pinned block
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->h_accum_dev), ctx->h_accum, 0) != hipSuccess) rc = fail(SGA_ERR_HIP, "hipHostGetDevicePointer failed");
  }
  if (rc == SGA_OK && (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess || hipEventCreate(&ctx->ev2) != hipSuccess || hipEventCreate(&ctx->ev3) != hipSuccess || hipEventCreate(&ctx->ev_mid) != hipSuccess || hipEventCreate(&ctx->ev_comm) != hipSuccess || hipEventCreateWithFlags(&ctx->ev_aux, hipEventDisableTiming) != hipSuccess)) rc = fail(SGA_ERR_HIP, "hipEventCreate failed");
  if (rc != SGA_OK) {
    sga_context_destroy(ctx);
    return rc;
  }
  hipLaunchKernelGGL(scratch_prime_kernel, dim3(1), dim3(64), 0, ctx->stream, static_cast<int*>(nullptr), 1);
  (void)hipGetLastError();
  {  // once per process: resolve the hot kernels now instead of in the middle of the first registration (linearize.hip)
    static std::once_flag preload_once;
    std::call_once(preload_once, [] { preload_hot_kernels(); });
  }
  *out = ctx;
  return SGA_OK;
}

int sga_context_create(int device, sga_context** out) { return context_create_impl(device, nullptr, false, out); }
int sga_context_create_on_stream(int device, void* hip_stream, sga_context** out) { return context_create_impl(device, hip_stream, true, out); }

int sga_context_destroy(sga_context* ctx) {
  if (!ctx) return SGA_OK;
  (void)hipSetDevice(ctx->device);
  (void)sga_comm_destroy(ctx);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  if (ctx->ev2) (void)hipEventDestroy(ctx->ev2);
  if (ctx->ev3) (void)hipEventDestroy(ctx->ev3);
  if (ctx->ev_mid) (void)hipEventDestroy(ctx->ev_mid);
  if (ctx->ev_comm) (void)hipEventDestroy(ctx->ev_comm);
  if (ctx->ev_aux) (void)hipEventDestroy(ctx->ev_aux);
  if (ctx->h_accum) (void)hipHostFree(ctx->h_accum);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  const int device = ctx->device;
  const hipStream_t stream = ctx->stream;
  const bool registered = ctx->registered, owns = ctx->owns_stream;
  {
    StreamScope scope(stream);  // the context's own buffers: the stream is idle (synchronised above)
    delete ctx;
  }
  if (registered) dev_cache_context_destroyed(device, stream);  // the last context gives the cached device memory back
  if (owns && stream) (void)hipStreamDestroy(stream);
  return SGA_OK;
}

int sga_context_synchronize(sga_context* ctx) {
  if (!ctx) return fail(SGA_ERR_INVALID, "null context");
  SGA_HIP(hipStreamSynchronize(ctx->stream));
  return SGA_OK;
}

void* sga_context_stream(sga_context* ctx) { return ctx ? static_cast<void*>(ctx->stream) : nullptr; }

}  // extern "C"
namespace sga {
int mark_ready(sga_context* ctx, Ready& r) {
  if (!ctx->stream_ordered) {  // the entry point synchronised its stream before returning: nothing in flight
    r.pending = false;
    return SGA_OK;
  }
  if (!r.event) SGA_HIP(hipEventCreateWithFlags(&r.event, hipEventDisableTiming));
  SGA_HIP(hipEventRecord(r.event, ctx->stream));
  r.stream = ctx->stream;
  r.pending = true;
  return SGA_OK;
}
int wait_ready(sga_context* ctx, const Ready& r) {
  if (r.pending && r.event && r.stream != ctx->stream) SGA_HIP(hipStreamWaitEvent(ctx->stream, r.event, 0));
  return SGA_OK;
}
}  // namespace sga
extern "C" {

int sga_context_set_stream_ordered(sga_context* ctx, int enabled) {
  if (!ctx) return fail(SGA_ERR_INVALID, "null context");
  if (!enabled && ctx->stream_ordered) (void)hipStreamSynchronize(ctx->stream);
  ctx->stream_ordered = enabled != 0;
  return SGA_OK;
}

int sga_context_set_profiling(sga_context* ctx, int enabled) {
  if (!ctx) return fail(SGA_ERR_INVALID, "null context");
  ctx->profiling = enabled != 0;
  ctx->profile_period = enabled > 1 ? static_cast<unsigned>(enabled) : 1u;
  ctx->lin_seq = ctx->err_seq = 0;
  ctx->lin_ms = ctx->err_ms = 0.0;
  ctx->lin_calls = ctx->err_calls = 0;
  ctx->search_ms = 0.0;
  ctx->search_calls = 0;
  ctx->warm_ms = ctx->cold_ms = ctx->warm_first_ms = 0.0;
  ctx->warm_calls = ctx->cold_calls = 0;
  ctx->comm_ms = 0.0;
  ctx->comm_calls = 0;
  ctx->comm_recorded = false;
  ctx->pending = 0;
  return SGA_OK;
}

int sga_context_get_kernel_ms(sga_context* ctx, double* lin_ms, uint64_t* lin_calls, double* err_ms, uint64_t* err_calls) {
  if (!ctx) return fail(SGA_ERR_INVALID, "null context");
  sga_profile_collect_pending(ctx);
  if (lin_ms) *lin_ms = ctx->lin_calls ? ctx->lin_ms / ctx->lin_calls : 0.0;
  if (lin_calls) *lin_calls = ctx->lin_calls;
  if (err_ms) *err_ms = ctx->err_calls ? ctx->err_ms / ctx->err_calls : 0.0;
  if (err_calls) *err_calls = ctx->err_calls;
  return SGA_OK;
}

int sga_context_get_pass_ms(sga_context* ctx, double* cold_ms, uint64_t* cold_calls, double* warm_ms, uint64_t* warm_calls, double* warm_search_ms) {
  if (!ctx) return fail(SGA_ERR_INVALID, "null context");
  sga_profile_collect_pending(ctx);
  if (cold_ms) *cold_ms = ctx->cold_calls ? ctx->cold_ms / ctx->cold_calls : 0.0;
  if (cold_calls) *cold_calls = ctx->cold_calls;
  if (warm_ms) *warm_ms = ctx->warm_calls ? ctx->warm_ms / ctx->warm_calls : 0.0;
  if (warm_calls) *warm_calls = ctx->warm_calls;
  if (warm_search_ms) *warm_search_ms = ctx->warm_calls ? ctx->warm_first_ms / ctx->warm_calls : 0.0;
  return SGA_OK;
}

int sga_context_get_comm_ms(sga_context* ctx, double* comm_ms, uint64_t* comm_calls) {
  if (!ctx) return fail(SGA_ERR_INVALID, "null context");
  sga_profile_collect_pending(ctx);
  if (comm_ms) *comm_ms = ctx->comm_calls ? ctx->comm_ms / ctx->comm_calls : 0.0;
  if (comm_calls) *comm_calls = ctx->comm_calls;
  return SGA_OK;
}

int sga_context_get_search_ms(sga_context* ctx, double* search_ms, uint64_t* search_calls) {
  if (!ctx) return fail(SGA_ERR_INVALID, "null context");
  sga_profile_collect_pending(ctx);
  if (search_ms) *search_ms = ctx->search_calls ? ctx->search_ms / ctx->search_calls : 0.0;
  if (search_calls) *search_calls = ctx->search_calls;
  return SGA_OK;
}

// the context's pinned staging buffer for host -> device uploads (grow-only; the stream must be idle: uploads synchronise)
static int ctx_pinned_stage(sga_context* ctx, size_t bytes, void** out) {
  if (ctx->h_stage_bytes < bytes) {
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    ctx->h_stage = nullptr;
    ctx->h_stage_bytes = 0;
    size_t want = 1u << 20;
    while (want < bytes) want <<= 1;
    if (hipHostMalloc(&ctx->h_stage, want, hipHostMallocDefault) != hipSuccess) return fail(SGA_ERR_HIP, "hipHostMalloc(%zu bytes) failed", want);
    ctx->h_stage_bytes = want;
  }
  *out = ctx->h_stage;
  return SGA_OK;
}

}  // extern "C"

// The cloud whose fp32 coordinates are given RELATIVE to `origin` (true position = xyz_rel + origin): the records go to the device as they are.
static int cloud_create_rel(sga_context* ctx, const float* xyz, const float* normals, const float* cov6, size_t n, const double origin[3], const double* recentre_by, sga_cloud** out) {
  if (!ctx || !out || (n > 0 && !xyz)) return fail(SGA_ERR_INVALID, "null argument");
  if (n >= (1ull << 31)) return fail(SGA_ERR_INVALID, "cloud too large (%zu points; limit 2^31-1)", n);
  *out = nullptr;
  SGA_ENTER(ctx);
  auto* c = new sga_cloud;
  c->device = ctx->device;
  c->n = n;
  for (int k = 0; k < 3; k++) c->origin[k] = origin ? origin[k] : 0.0;
  c->has_normals = normals != nullptr;
  c->has_covs = cov6 != nullptr;
  DevBuf<float> sx, sn, sc;
  int rc = c->pts.alloc(n);
  if (rc == SGA_OK && normals) rc = c->nrm.alloc(n);
  if (rc == SGA_OK && cov6) rc = c->cov.alloc(n);
  if (rc == SGA_OK) rc = sx.alloc(n * 3);
  if (rc == SGA_OK && normals) rc = sn.alloc(n * 3);
  if (rc == SGA_OK && cov6) rc = sc.alloc(n * 6);
  if (rc != SGA_OK) {
    delete c;
    return rc;
  }
  if (n > 0) {
    // The caller's buffers are pageable: copied from there the runtime has to lock their pages first, which costs up to tens of
    // milliseconds per fresh megabyte-sized buffer on some hosts (measured: 15-28 ms for a 1.4 MB scan).  They are staged through the
    // context's own pinned buffer instead (a CPU memcpy at memory speed).
    const size_t fx = n * 3, fn = normals ? n * 3 : 0, fc = cov6 ? n * 6 : 0;
    float* stage = nullptr;
    rc = ctx_pinned_stage(ctx, (fx + fn + fc) * sizeof(float), reinterpret_cast<void**>(&stage));
    if (rc != SGA_OK) {
      delete c;
      return rc;
    }
    if (recentre_by == nullptr) {
      std::memcpy(stage, xyz, fx * sizeof(float));
    } else {  // absolute fp32 coordinates far from the origin: the subtraction in double, while the points are staged anyway
      for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) stage[3 * i + k] = static_cast<float>(static_cast<double>(xyz[3 * i + k]) - recentre_by[k]);
    }
    if (normals) std::memcpy(stage + fx, normals, fn * sizeof(float));
    if (cov6) std::memcpy(stage + fx + fn, cov6, fc * sizeof(float));
    hipError_t e = hipMemcpyAsync(sx.p, stage, fx * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && normals) e = hipMemcpyAsync(sn.p, stage + fx, fn * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && cov6) e = hipMemcpyAsync(sc.p, stage + fx + fn, fc * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
      hipLaunchKernelGGL(pack_cloud_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, sx.p, sn.p, sc.p, n, c->pts.p, c->nrm.p, c->cov.p);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
      delete c;
      return fail(SGA_ERR_HIP, "cloud upload failed: %s", hipGetErrorString(e));
    }
  }
  *out = c;
  return SGA_OK;
}

// bounding box over the finite coordinates of n points with `stride` values per point
template <typename S>
static void host_bbox(const S* xyz, size_t n, size_t stride, double lo[3], double hi[3]) {
  for (int k = 0; k < 3; k++) lo[k] = INFINITY, hi[k] = -INFINITY;
  for (size_t i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) {
      const double v = static_cast<double>(xyz[stride * i + k]);
      if (v - v == 0.0) {  // finite
        lo[k] = v < lo[k] ? v : lo[k];
        hi[k] = v > hi[k] ? v : hi[k];
      }
    }
}

namespace sga {
// absolute fp32 coordinates, recentred about a GIVEN origin (multi.hip: the shards of one source share a device frame)
int cloud_create_f32_about(sga_context* ctx, const float* xyz, const float* normals, const float* cov6, size_t n, const double origin[3], sga_cloud** out) {
  return cloud_create_rel(ctx, xyz, normals, cov6, n, origin, origin_is_zero(origin) ? nullptr : origin, out);
}
void host_bbox_f32(const float* xyz, size_t n, double lo[3], double hi[3]) { host_bbox(xyz, n, 3, lo, hi); }
void host_bbox_f64(const double* xyzw, size_t n, double lo[3], double hi[3]) { host_bbox(xyzw, n, 4, lo, hi); }
}  // namespace sga
extern "C" {

int sga_cloud_create_f32_origin(sga_context* ctx, const float* xyz_rel, const float* normals, const float* cov6, size_t n, const double origin[3], sga_cloud** out) {
  return cloud_create_rel(ctx, xyz_rel, normals, cov6, n, origin, nullptr, out);
}

int sga_cloud_create_f32(sga_context* ctx, const float* xyz, const float* normals, const float* cov6, size_t n, sga_cloud** out) {
  if (!ctx || !out || (n > 0 && !xyz)) return fail(SGA_ERR_INVALID, "null argument");
  double lo[3], hi[3], origin[3];
  host_bbox(xyz, n, 3, lo, hi);
  choose_origin(lo, hi, origin);
  return cloud_create_rel(ctx, xyz, normals, cov6, n, origin, origin_is_zero(origin) ? nullptr : origin, out);
}

int sga_cloud_create_f64_origin(sga_context* ctx, const double* xyzw, const double* normals4, const double* cov4x4, size_t n, const double origin_in[3], sga_cloud** out) {
  if (!ctx || !out || (n > 0 && !xyzw)) return fail(SGA_ERR_INVALID, "null argument");
  double origin[3] = {0, 0, 0};
  if (origin_in) {
    for (int k = 0; k < 3; k++) origin[k] = origin_in[k];
  } else {
    double lo[3], hi[3];
    host_bbox(xyzw, n, 4, lo, hi);
    choose_origin(lo, hi, origin);
  }
  std::vector<float> xyz(n * 3), nrm, cov;
  for (size_t i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) xyz[3 * i + k] = static_cast<float>(xyzw[4 * i + k] - origin[k]);  // in double, then rounded: what fp32 can hold of the cloud is its shape, not its place
  if (normals4) {
    nrm.resize(n * 3);
    for (size_t i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) nrm[3 * i + k] = static_cast<float>(normals4[4 * i + k]);
  }
  if (cov4x4) {
    cov.resize(n * 6);
    for (size_t i = 0; i < n; i++) {
      const double* m = cov4x4 + 16 * i;  // symmetric: storage order irrelevant
      cov[6 * i + 0] = static_cast<float>(m[0]);
      cov[6 * i + 1] = static_cast<float>(m[1]);
      cov[6 * i + 2] = static_cast<float>(m[2]);
      cov[6 * i + 3] = static_cast<float>(m[5]);
      cov[6 * i + 4] = static_cast<float>(m[6]);
      cov[6 * i + 5] = static_cast<float>(m[10]);
    }
  }
  return cloud_create_rel(ctx, xyz.data(), normals4 ? nrm.data() : nullptr, cov4x4 ? cov.data() : nullptr, n, origin, nullptr, out);
}

int sga_cloud_create_f64(sga_context* ctx, const double* xyzw, const double* normals4, const double* cov4x4, size_t n, sga_cloud** out) {
  return sga_cloud_create_f64_origin(ctx, xyzw, normals4, cov4x4, n, nullptr, out);
}

int sga_cloud_origin(const sga_cloud* cloud, double origin[3]) {
  if (!cloud || !origin) return fail(SGA_ERR_INVALID, "null argument");
  for (int k = 0; k < 3; k++) origin[k] = cloud->origin[k];
  return SGA_OK;
}

int sga_index_origin(const sga_index* index, double origin[3]) {
  if (!index || !origin) return fail(SGA_ERR_INVALID, "null argument");
  for (int k = 0; k < 3; k++) origin[k] = index->origin[k];
  return SGA_OK;
}

void sga_choose_origin(const double lo[3], const double hi[3], double origin[3]) { choose_origin(lo, hi, origin); }

int sga_cloud_slice(sga_context* ctx, const sga_cloud* cloud, size_t first, size_t count, sga_cloud** out) {
  if (!ctx || !cloud || !out) return fail(SGA_ERR_INVALID, "null argument");
  if (first > cloud->n || count > cloud->n - first) return fail(SGA_ERR_INVALID, "slice [%zu, %zu) outside a cloud of %zu points", first, first + count, cloud->n);
  if (cloud->device != ctx->device) return fail(SGA_ERR_INVALID, "cloud lives on another device");
  *out = nullptr;
  SGA_ENTER(ctx);
  auto* c = new sga_cloud;
  c->device = ctx->device;
  c->n = count;
  for (int k = 0; k < 3; k++) c->origin[k] = cloud->origin[k];  // the slice stays in its cloud's device frame: shards of one registration share it
  c->has_normals = cloud->has_normals;
  c->has_covs = cloud->has_covs;
  int rc = c->pts.alloc(count);
  if (rc == SGA_OK && cloud->has_normals) rc = c->nrm.alloc(count);
  if (rc == SGA_OK && cloud->has_covs) rc = c->cov.alloc(count);
  if (rc != SGA_OK) {
    delete c;
    return rc;
  }
  if (count > 0) {
    hipLaunchKernelGGL(slice_cloud_kernel, dim3((count + 255) / 256), dim3(256), 0, ctx->stream, cloud->pts.p, cloud->has_normals ? cloud->nrm.p : nullptr, cloud->has_covs ? cloud->cov.p : nullptr, first, count, c->pts.p, c->nrm.p, c->cov.p);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
      delete c;
      return fail(SGA_ERR_HIP, "slice kernel: %s", hipGetErrorString(e));
    }
  }
  *out = c;
  return SGA_OK;
}

int sga_cloud_destroy(sga_cloud* cloud) {
  if (cloud) {
    (void)hipSetDevice(cloud->device);
    delete cloud;
  }
  return SGA_OK;
}

int sga_cloud_size(const sga_cloud* cloud, size_t* n) {
  if (!cloud || !n) return fail(SGA_ERR_INVALID, "null argument");
  *n = cloud->n;
  return SGA_OK;
}

int sga_cloud_has(const sga_cloud* cloud, int* has_normals, int* has_covs) {
  if (!cloud) return fail(SGA_ERR_INVALID, "null argument");
  if (has_normals) *has_normals = cloud->has_normals;
  if (has_covs) *has_covs = cloud->has_covs;
  return SGA_OK;
}

static int cloud_download_impl(sga_context* ctx, const sga_cloud* cloud, float* xyz, double* xyz64, float* normals, float* cov6) {
  if (!ctx || !cloud) return fail(SGA_ERR_INVALID, "null argument");
  if (normals && !cloud->has_normals) return fail(SGA_ERR_INVALID, "cloud has no normals");
  if (cov6 && !cloud->has_covs) return fail(SGA_ERR_INVALID, "cloud has no covariances");
  const size_t n = cloud->n;
  if (n == 0) return SGA_OK;
  SGA_ENTER(ctx);
  SGA_TRY(wait_ready(ctx, cloud->ready));
  DevBuf<float> sx, sn, sc;
  const bool framed = !origin_is_zero(cloud->origin);
  std::vector<float> rel;
  float* xyz_dst = xyz;
  if (xyz64 || (xyz && framed)) {  // the device frame -> the caller's: the origin is added in double
    rel.resize(n * 3);
    xyz_dst = rel.data();
  }
  if (xyz_dst) SGA_TRY(sx.alloc(n * 3));
  if (normals) SGA_TRY(sn.alloc(n * 3));
  if (cov6) SGA_TRY(sc.alloc(n * 6));
  hipLaunchKernelGGL(unpack_cloud_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, cloud->pts.p, cloud->nrm.p, cloud->cov.p, n, sx.p, sn.p, sc.p);
  SGA_HIP(hipGetLastError());
  if (xyz_dst) SGA_HIP(hipMemcpyAsync(xyz_dst, sx.p, n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (normals) SGA_HIP(hipMemcpyAsync(normals, sn.p, n * 3 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  if (cov6) SGA_HIP(hipMemcpyAsync(cov6, sc.p, n * 6 * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  SGA_HIP(hipStreamSynchronize(ctx->stream));
  if (!rel.empty())
    for (size_t i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) {
        const double v = static_cast<double>(rel[3 * i + k]) + cloud->origin[k];
        if (xyz64) xyz64[3 * i + k] = v;
        if (xyz) xyz[3 * i + k] = static_cast<float>(v);
      }
  return SGA_OK;
}

int sga_cloud_download(sga_context* ctx, const sga_cloud* cloud, float* xyz, float* normals, float* cov6) { return cloud_download_impl(ctx, cloud, xyz, nullptr, normals, cov6); }

int sga_cloud_download_f64(sga_context* ctx, const sga_cloud* cloud, double* xyz, float* normals, float* cov6) { return cloud_download_impl(ctx, cloud, nullptr, xyz, normals, cov6); }

}  // extern "C"
