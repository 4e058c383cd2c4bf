// Context (one GPU + one stream), error plumbing and device-resident clouds.
#include "common.hpp"
#include "notes.hpp"

#include <atomic>
#include <cstdlib>
#include <chrono>
#include <cmath>
#include <thread>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <map>
#include <mutex>

// HIP streams share a small pool of hardware queues (4 per device by default), and two streams that land on one queue run their
// kernels one after the other.  The library's unit of concurrency is a context = a stream (pipelined preprocessing, several registrations
// side by side, one context per policy thread): measured in round 6, two side-by-side C3 registrations gave 10 800 iterations/s on
// distinct queues and 8 600 (nothing) when the runtime had put their streams on one — which depended on how many streams the process had
// created before.  Unless the user has set it, ask for 8 queues before the HIP runtime initialises (it reads the variable once).
namespace {
struct HwQueuesDefault {
  HwQueuesDefault() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }
} g_hw_queues_default;
}  // namespace

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
// Every entry-point invocation has a process-wide unique number.  A block that is freed by the invocation that allocated it (a temporary:
// sort keys, scan flags, staging arrays) was never visible to anybody else — no other stream can hold work that touches it — so it goes
// straight back to its stream's list.  (Round 6: the general path asks every other stream of the device whether it is busy and records an
// event on each that is: ~25 us per free with four streams in flight, under the allocator's lock — the pipelined odometry driver spent
// more time there than its kernels took.)
thread_local unsigned long long g_cur_epoch = 0;
std::atomic<unsigned long long> g_epoch_counter{0};

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
};
// Blocks freed where another stream may still use them wait for events — recorded LAZILY and shared: a free only parks the block
// (`unbatched`); the next allocation that misses the free lists records ONE event on every busy stream of the device for all the blocks
// parked since (an event recorded after the free covers everything that was in flight at the free, and more: conservative), and a
// batch's blocks join the shared pool when its events have completed.  (Round 6: an event per block and stream, recorded at the free,
// cost ~25 us per free with four streams in flight.)
struct PendingBatch {
  int device;
  std::vector<hipEvent_t> events;
  std::vector<PendingBlock> blocks;
};
struct DevCache {
  std::mutex mu;
  struct Live {
    int device;
    size_t bucket;
    unsigned long long epoch;  // the entry-point invocation that allocated it (0: outside any)
    hipStream_t stream;        // ... and its stream
  };
  std::unordered_map<void*, Live> live;  // every block handed out
  std::map<FreeKey, std::vector<void*>> free_blocks;
  std::vector<PendingBlock> unbatched;                     // freed where other streams may be using them; no event recorded yet
  std::vector<PendingBatch> pending;                       // ... with their events: reusable once those have completed
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

// batches whose events have all completed -> the shared pool; the blocks parked since the last call get their batch (one event per busy
// stream of their device; no busy stream: straight to the pool)
void collect_pending_locked(DevCache& c) {
  size_t w = 0;
  for (size_t i = 0; i < c.pending.size(); i++) {
    PendingBatch& b = c.pending[i];
    bool done = true;
    for (hipEvent_t e : b.events)
      if (hipEventQuery(e) == hipErrorNotReady) {
        done = false;
        break;
      }
    if (done) {
      recycle_events(c, b.events);
      for (const PendingBlock& k : b.blocks) c.free_blocks[{k.device, nullptr, k.bucket}].push_back(k.p);
    } else {
      if (w != i) c.pending[w] = std::move(b);
      w++;
    }
  }
  c.pending.resize(w);
  if (!c.unbatched.empty()) {
    int cur = -1;
    (void)hipGetDevice(&cur);
    const int restore = cur;
    std::vector<PendingBatch> fresh;
    for (PendingBlock& k : c.unbatched) {
      PendingBatch* batch = nullptr;
      for (PendingBatch& f : fresh)
        if (f.device == k.device) batch = &f;
      if (batch == nullptr) {
        fresh.push_back(PendingBatch{k.device, {}, {}});
        batch = &fresh.back();
        for (const auto& ds : c.streams) {
          if (ds.first != k.device) continue;
          if (hipStreamQuery(ds.second) != hipErrorNotReady) continue;  // idle: nothing of it can touch the blocks
          if (cur != k.device) {
            (void)hipSetDevice(k.device);
            cur = k.device;
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
          batch->events.push_back(e);
        }
      }
      batch->blocks.push_back(k);
    }
    c.unbatched.clear();
    if (cur != restore && restore >= 0) (void)hipSetDevice(restore);
    for (PendingBatch& f : fresh) {
      if (f.events.empty()) {
        for (const PendingBlock& k : f.blocks) c.free_blocks[{k.device, nullptr, k.bucket}].push_back(k.p);
      } else {
        c.n_deferred += f.blocks.size();
        c.pending.push_back(std::move(f));
      }
    }
  }
  (void)hipGetLastError();  // hipEventQuery's / hipStreamQuery's hipErrorNotReady is sticky in hipGetLastError
}

void release_cached_locked(DevCache& c) {
  for (PendingBatch& b : c.pending) {
    for (hipEvent_t e : b.events) (void)hipEventSynchronize(e);
    recycle_events(c, b.events);
    for (const PendingBlock& k : b.blocks) (void)hipFree(k.p);
  }
  c.pending.clear();
  for (const PendingBlock& k : c.unbatched) (void)hipFree(k.p);  // (hipFree synchronises the device: safe whatever is in flight)
  c.unbatched.clear();
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

StreamScope::StreamScope(hipStream_t s) : prev(g_cur_stream), prev_epoch(g_cur_epoch) {
  g_cur_stream = s;
  g_cur_epoch = s != nullptr ? ++g_epoch_counter : 0ull;
}
StreamScope::~StreamScope() {
  g_cur_stream = prev;
  g_cur_epoch = prev_epoch;
}

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
  if (!*p && (!c.pending.empty() || !c.unbatched.empty())) {
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
  c.live[*p] = DevCache::Live{device, bucket, g_cur_epoch, g_cur_stream};
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
  const int device = it->second.device;
  const size_t bucket = it->second.bucket;
  const bool temporary = g_cur_stream != nullptr && it->second.epoch == g_cur_epoch && it->second.epoch != 0ull && it->second.stream == g_cur_stream;
  c.live.erase(it);
  if (c.contexts == 0 || c.cached_bytes + bucket > kCacheLimitBytes) {
    (void)hipFree(p);  // synchronises the device: safe whatever is in flight
    return;
  }
  c.cached_bytes += bucket;
  if (temporary) {  // allocated by this very invocation: nobody else has seen it
    c.free_blocks[{device, g_cur_stream, bucket}].push_back(p);
    return;
  }
  // Anything else — a buffer of a long-lived object (an index's attributes, a problem's mahalanobis cache, the rejector flags: kernels
  // another context has enqueued on its own stream may be reading it) or a block freed outside any entry point (destroy
  // functions) — may still be in use on any stream of the device: parked until the streams have passed this point (collect_pending_locked)
  c.unbatched.push_back(PendingBlock{p, device, bucket});
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

// Upload: xyz (3 floats per point, AoS), optional normals (3) and covariances (6, symmetric) -> the 16 / 16 / 32-byte device records.
// The inputs may live in pinned HOST memory (the staging ring, or the caller's own pinned buffer): every dword is then read over PCIe
// exactly once, with unit-stride loads through LDS (a per-thread stride of 12 bytes would touch every line three times, and mapped host
// memory is not cached).  recentre: records = fl32(double(x) - origin) (common.hpp, device frames).  d_box != nullptr: the bounding
// box of the finite INPUT coordinates is handed to the host as a note (notes.hpp) — how a pinned upload learns its origin.
__global__ __launch_bounds__(256) void pack_cloud_kernel(const float* __restrict__ xyz, const float* __restrict__ nrm, const float* __restrict__ cov6, size_t n, double ox, double oy, double oz, int recentre, float4* __restrict__ pts,
                                                         float4* __restrict__ onrm, Cov8* __restrict__ ocov, int* __restrict__ d_box, unsigned long long* __restrict__ note_slot, unsigned long long seq) {
  __shared__ float sh[256 * 6];
  const size_t base = blockIdx.x * static_cast<size_t>(256);
  const size_t i = base + threadIdx.x;
  const int t = threadIdx.x;
  {
    const size_t f0 = base * 3, fend = n * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const size_t f = f0 + k * 256 + t;
      sh[k * 256 + t] = f < fend ? xyz[f] : 0.f;
    }
  }
  __syncthreads();
  float x = sh[3 * t], y = sh[3 * t + 1], z = sh[3 * t + 2];
  float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  if (i < n) {
    if (d_box != nullptr) {  // finite coordinates only (what the origin is chosen from)
      if (fabsf(x) <= 3.4028234e38f) lo[0] = hi[0] = x;
      if (fabsf(y) <= 3.4028234e38f) lo[1] = hi[1] = y;
      if (fabsf(z) <= 3.4028234e38f) lo[2] = hi[2] = z;
    }
    if (recentre) {
      x = static_cast<float>(static_cast<double>(x) - ox);
      y = static_cast<float>(static_cast<double>(y) - oy);
      z = static_cast<float>(static_cast<double>(z) - oz);
    }
    pts[i] = make_float4(x, y, z, __uint_as_float(static_cast<uint32_t>(i)));
  }
  if (nrm != nullptr) {
    __syncthreads();
    const size_t f0 = base * 3, fend = n * 3;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const size_t f = f0 + k * 256 + t;
      sh[k * 256 + t] = f < fend ? nrm[f] : 0.f;
    }
    __syncthreads();
    if (i < n) onrm[i] = make_float4(sh[3 * t], sh[3 * t + 1], sh[3 * t + 2], 0.f);
  }
  if (cov6 != nullptr) {
    __syncthreads();
    const size_t f0 = base * 6, fend = n * 6;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const size_t f = f0 + k * 256 + t;
      sh[k * 256 + t] = f < fend ? cov6[f] : 0.f;
    }
    __syncthreads();
    if (i < n) {
      Cov8 c;
      c.xx = sh[6 * t];
      c.xy = sh[6 * t + 1];
      c.xz = sh[6 * t + 2];
      c.yy = sh[6 * t + 3];
      c.yz = sh[6 * t + 4];
      c.zz = sh[6 * t + 5];
      c.pad0 = c.pad1 = 0.f;
      ocov[i] = c;
    }
  }
  if (d_box != nullptr) box_reduce_publish(lo, hi, d_box, note_slot, seq);
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

constexpr int kNotesAt = 160;  // h_accum: doubles [0, 129) results + sequence word, [136, 144) scratch ints, [160, 192) notes
constexpr int kPinnedDoubles = kNotesAt + sga::kNoteSlots * sga::kNoteWords;
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
  if (rc == SGA_OK && hipHostMalloc(reinterpret_cast<void**>(&ctx->h_accum), kPinnedDoubles * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) rc = fail(SGA_ERR_HIP, "hipHostMalloc failed");
  if (rc == SGA_OK) {
    std::memset(ctx->h_accum, 0, kPinnedDoubles * sizeof(double));
    ctx->h_scratch = reinterpret_cast<int*>(ctx->h_accum + 136);  // doubles [136, 144) of the pinned block
    if (hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->h_accum_dev), ctx->h_accum, 0) != hipSuccess) rc = fail(SGA_ERR_HIP, "hipHostGetDevicePointer failed");
    ctx->h_notes = reinterpret_cast<unsigned long long*>(ctx->h_accum + kNotesAt);  // words [160, 192): the notes (notes.hpp)
    ctx->h_notes_dev = reinterpret_cast<unsigned long long*>(ctx->h_accum_dev + kNotesAt);
  }
  if (rc == SGA_OK) rc = ctx->d_spacing.alloc(4);
  if (rc == SGA_OK && hipMemsetAsync(ctx->d_spacing.p, 0, 4 * sizeof(unsigned long long), ctx->stream) != hipSuccess) rc = fail(SGA_ERR_HIP, "hipMemsetAsync failed");
  if (rc == SGA_OK) rc = ctx->d_box.alloc(8);
  if (rc == SGA_OK) {
    const int init[8] = {kBoxEncPosInf, kBoxEncPosInf, kBoxEncPosInf, kBoxEncNegInf, kBoxEncNegInf, kBoxEncNegInf, 0, 0};
    if (hipMemcpyAsync(ctx->d_box.p, init, sizeof(init), hipMemcpyHostToDevice, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) rc = fail(SGA_ERR_HIP, "box accumulator init failed");
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
  if (ctx->ev_t0) (void)hipEventDestroy(ctx->ev_t0);
  if (ctx->ev_t1) (void)hipEventDestroy(ctx->ev_t1);
  if (ctx->h_accum) (void)hipHostFree(ctx->h_accum);
  for (auto& slot : ctx->stage) {
    if (slot.host) (void)hipHostFree(slot.host);
    if (slot.done) (void)hipEventDestroy(slot.done);
  }
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

// GPU time between two points of the context's stream (bench.py: the per-stage roofline lines of the preprocessing kernels)
int sga_debug_timer_start(sga_context* ctx) {
  if (!ctx) return fail(SGA_ERR_INVALID, "null context");
  SGA_HIP(hipSetDevice(ctx->device));
  if (!ctx->ev_t0) SGA_HIP(hipEventCreate(&ctx->ev_t0));
  if (!ctx->ev_t1) SGA_HIP(hipEventCreate(&ctx->ev_t1));
  SGA_HIP(hipEventRecord(ctx->ev_t0, ctx->stream));
  return SGA_OK;
}
int sga_debug_timer_stop(sga_context* ctx, double* ms) {
  if (!ctx || !ms || !ctx->ev_t0) return fail(SGA_ERR_INVALID, "no timer running");
  SGA_HIP(hipEventRecord(ctx->ev_t1, ctx->stream));
  SGA_HIP(hipEventSynchronize(ctx->ev_t1));
  float f = 0.f;
  SGA_HIP(hipEventElapsedTime(&f, ctx->ev_t0, ctx->ev_t1));
  *ms = f;
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

// ---- notes (notes.hpp) -------------------------------------------------------------------------------------------------------
}  // extern "C"
namespace sga {
unsigned long long note_begin(sga_context* ctx, unsigned long long** dev_slot) {
  const unsigned long long seq = ++ctx->note_seq;
  *dev_slot = ctx->h_notes_dev + (seq % kNoteSlots) * kNoteWords;
  return seq;
}
int note_wait(sga_context* ctx, unsigned long long seq, unsigned long long payload[kNoteWords - 1]) {
  const unsigned long long* slot = ctx->h_notes + (seq % kNoteSlots) * kNoteWords;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spins = 0;; spins++) {
    if (__atomic_load_n(slot, __ATOMIC_ACQUIRE) == seq) break;
    __builtin_ia32_pause();
    if (spins > 200000u) std::this_thread::yield();
    if ((spins & 0xfffu) == 0xfffu) {
      // the stream has drained without publishing (a fault), or this is taking implausibly long: let the runtime report it
      const hipError_t q = hipStreamQuery(ctx->stream);
      if (q != hipErrorNotReady || std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
        (void)hipGetLastError();
        SGA_HIP(hipStreamSynchronize(ctx->stream));
        if (__atomic_load_n(slot, __ATOMIC_ACQUIRE) == seq) break;
        return fail(SGA_ERR_HIP, "a note was not published by the device");
      }
      (void)hipGetLastError();  // hipErrorNotReady is sticky in hipGetLastError
    }
  }
  for (int k = 0; k < kNoteWords - 1; k++) payload[k] = slot[1 + k];
  return SGA_OK;
}
}  // namespace sga

namespace sga {
namespace {
struct LateRing {
  std::mutex mu;
  unsigned long long* host = nullptr;
  bool failed = false;
  std::atomic<unsigned long long> seq{0};
};
LateRing& late_ring() {
  static LateRing* r = new LateRing;  // never destroyed: no HIP calls during static destruction
  return *r;
}
}  // namespace
unsigned long long late_note_begin(int device, unsigned long long** dev_slot) {
  LateRing& r = late_ring();
  *dev_slot = nullptr;
  {
    std::lock_guard<std::mutex> lock(r.mu);
    if (r.host == nullptr && !r.failed) {
      void* p = nullptr;
      if (hipHostMalloc(&p, sizeof(unsigned long long) * kLateSlots * kLateWords, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess) {
        (void)hipGetLastError();
        r.failed = true;
      } else {
        std::memset(p, 0, sizeof(unsigned long long) * kLateSlots * kLateWords);
        r.host = static_cast<unsigned long long*>(p);
      }
    }
    if (r.host == nullptr) return 0;
  }
  void* dev = nullptr;
  (void)device;
  if (hipHostGetDevicePointer(&dev, r.host, 0) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  const unsigned long long seq = ++r.seq;
  *dev_slot = static_cast<unsigned long long*>(dev) + (seq % kLateSlots) * kLateWords;
  return seq;
}
int late_note_peek(unsigned long long seq, unsigned long long payload[kLateWords - 1]) {
  LateRing& r = late_ring();
  if (seq == 0 || r.host == nullptr) return -1;
  const unsigned long long* slot = r.host + (seq % kLateSlots) * kLateWords;
  const unsigned long long have = __atomic_load_n(slot + kLateWords - 1, __ATOMIC_ACQUIRE);
  if (have == seq) {
    for (int k = 0; k < kLateWords - 1; k++) payload[k] = slot[k];
    if (__atomic_load_n(slot + kLateWords - 1, __ATOMIC_ACQUIRE) != seq) return -1;  // overwritten while it was being read
    return 1;
  }
  return have < seq ? 0 : -1;
}
}  // namespace sga

// ---- uploads ------------------------------------------------------------------------------------------------------------------
// A slot of the context's pinned staging ring with room for `bytes` (grow-only).  A slot handed out before is reused only after the
// event recorded behind its reader (stage_release) has completed.
static int stage_acquire(sga_context* ctx, size_t bytes, sga_context::StageSlot** out) {
  sga_context::StageSlot& slot = ctx->stage[ctx->stage_next++ % sga_context::kStageSlots];
  if (slot.busy) {
    SGA_HIP(hipEventSynchronize(slot.done));
    slot.busy = false;
  }
  if (slot.bytes < bytes) {
    if (slot.host) (void)hipHostFree(slot.host);
    slot.host = slot.dev = nullptr;
    slot.bytes = 0;
    size_t want = 1u << 20;
    while (want < bytes) want <<= 1;
    if (hipHostMalloc(&slot.host, want, hipHostMallocMapped) != hipSuccess) return fail(SGA_ERR_HIP, "hipHostMalloc(%zu bytes) failed", want);
    if (hipHostGetDevicePointer(&slot.dev, slot.host, 0) != hipSuccess) return fail(SGA_ERR_HIP, "hipHostGetDevicePointer failed");
    slot.bytes = want;
  }
  *out = &slot;
  return SGA_OK;
}
// behind the launch that reads the slot
static int stage_release(sga_context* ctx, sga_context::StageSlot* slot) {
  if (!slot->done) SGA_HIP(hipEventCreateWithFlags(&slot->done, hipEventDisableTiming));
  SGA_HIP(hipEventRecord(slot->done, ctx->stream));
  slot->busy = true;
  return SGA_OK;
}

// One pass over a pageable xyz array: copy it into the staging slot AND take the bounding box of its finite coordinates (the origin of
// the device frame is chosen from it).  Twelve running minima / maxima (four points) so that the compiler keeps them in vector registers;
// a 115k-point scan (1.4 MB) went through a scalar box pass and a memcpy before: two passes, ~0.25 ms.
#if defined(__x86_64__)
// 24 floats (8 points) per step: three 8-wide vectors whose lanes keep their coordinate (24 is a multiple of 3)
__attribute__((target("avx2"))) static void copy_with_box_wide(const float* __restrict__ src, float* __restrict__ dst, size_t count /* floats, a multiple of 24 */, float lo24[24], float hi24[24]) {
  const __m256 absmask = _mm256_castsi256_ps(_mm256_set1_epi32(0x7fffffff)), fmax = _mm256_set1_ps(3.4028234e38f);
  __m256 lo0 = _mm256_loadu_ps(lo24), lo1 = _mm256_loadu_ps(lo24 + 8), lo2 = _mm256_loadu_ps(lo24 + 16);
  __m256 hi0 = _mm256_loadu_ps(hi24), hi1 = _mm256_loadu_ps(hi24 + 8), hi2 = _mm256_loadu_ps(hi24 + 16);
  for (size_t i = 0; i < count; i += 24) {
    const __m256 a = _mm256_loadu_ps(src + i), b = _mm256_loadu_ps(src + i + 8), c = _mm256_loadu_ps(src + i + 16);
    _mm256_storeu_ps(dst + i, a);
    _mm256_storeu_ps(dst + i + 8, b);
    _mm256_storeu_ps(dst + i + 16, c);
    // non-finite values (NaN compares false, inf fails <= FLT_MAX) are replaced by the running bound: they change nothing
    const __m256 ma = _mm256_cmp_ps(_mm256_and_ps(a, absmask), fmax, _CMP_LE_OQ), mb = _mm256_cmp_ps(_mm256_and_ps(b, absmask), fmax, _CMP_LE_OQ), mc = _mm256_cmp_ps(_mm256_and_ps(c, absmask), fmax, _CMP_LE_OQ);
    lo0 = _mm256_min_ps(_mm256_blendv_ps(lo0, a, ma), lo0);
    lo1 = _mm256_min_ps(_mm256_blendv_ps(lo1, b, mb), lo1);
    lo2 = _mm256_min_ps(_mm256_blendv_ps(lo2, c, mc), lo2);
    hi0 = _mm256_max_ps(_mm256_blendv_ps(hi0, a, ma), hi0);
    hi1 = _mm256_max_ps(_mm256_blendv_ps(hi1, b, mb), hi1);
    hi2 = _mm256_max_ps(_mm256_blendv_ps(hi2, c, mc), hi2);
  }
  _mm256_storeu_ps(lo24, lo0), _mm256_storeu_ps(lo24 + 8, lo1), _mm256_storeu_ps(lo24 + 16, lo2);
  _mm256_storeu_ps(hi24, hi0), _mm256_storeu_ps(hi24 + 8, hi1), _mm256_storeu_ps(hi24 + 16, hi2);
}
#endif
static void copy_with_box_plain(const float* __restrict__ src, float* __restrict__ dst, size_t first, size_t count, float lo24[24], float hi24[24]) {
  for (size_t f = first; f < count; f++) {
    const float v = src[f];
    dst[f] = v;
    const int j = static_cast<int>(f % 24);  // 24 is a multiple of 3: lane j keeps coordinate j % 3
    if (__builtin_fabsf(v) <= 3.4028234e38f) {
      lo24[j] = v < lo24[j] ? v : lo24[j];
      hi24[j] = v > hi24[j] ? v : hi24[j];
    }
  }
}
static void copy_with_box(const float* src, float* dst, size_t n, double lo[3], double hi[3]) {
  float lo24[24], hi24[24];
  for (int j = 0; j < 24; j++) lo24[j] = INFINITY, hi24[j] = -INFINITY;
  const size_t count = n * 3;
  size_t body = 0;
#if defined(__x86_64__)
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2) {
    body = count / 24 * 24;
    copy_with_box_wide(src, dst, body, lo24, hi24);
  }
#endif
  copy_with_box_plain(src, dst, body, count, lo24, hi24);
  for (int k = 0; k < 3; k++) {
    lo[k] = INFINITY, hi[k] = -INFINITY;
    for (int j = k; j < 24; j += 3) {
      lo[k] = lo24[j] < lo[k] ? lo24[j] : lo[k];
      hi[k] = hi24[j] > hi[k] ? hi24[j] : hi[k];
    }
  }
}

// Is [p, p + bytes) pinned host memory a kernel of this device can read (hipHostMalloc / hipHostRegister / sga_host_alloc)?  -> its device address
static const void* pinned_device_view(const void* p, size_t bytes) {
  if (p == nullptr) return nullptr;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();  // an ordinary (pageable) pointer: not an error of ours
    return nullptr;
  }
  if (a.type != hipMemoryTypeHost || a.devicePointer == nullptr) return nullptr;
  (void)bytes;
  return a.devicePointer;
}

// How an upload learns its frame: `origin` given (the records are relative to it, or become so: recentre) or chosen from the bounding box.
enum class UploadFrame { Given, GivenRecentre, FromBox };

// The cloud of n points from host arrays: xyz (3 floats per point), optional normals (3) and covariances (6).
//   * pageable arrays are copied once into a slot of the context's pinned staging ring (the box is taken in the same pass) and the pack
//     kernel reads the slot over PCIe; in stream-ordered mode the call returns with the kernel in flight (the slot is reused only after
//     the event recorded behind it) — the caller's arrays are free as soon as the call returns either way;
//   * arrays that already live in pinned host memory (sga_host_alloc, hipHostMalloc) are read by the pack kernel where they are: no CPU
//     pass at all; the box comes back from the kernel as a note (notes.hpp), so the kernel has finished reading when the call returns.
static int cloud_upload(sga_context* ctx, const float* xyz, const float* normals, const float* cov6, size_t n, UploadFrame frame, const double origin_in[3], sga_cloud** out) {
  if (!ctx || !out || (n > 0 && !xyz)) return fail(SGA_ERR_INVALID, "null argument");
  if (n >= (1ull << 31)) return fail(SGA_ERR_INVALID, "cloud too large (%zu points; limit 2^31-1)", n);
  *out = nullptr;
  SGA_ENTER(ctx);
  std::unique_ptr<sga_cloud> c(new sga_cloud);
  c->device = ctx->device;
  c->n = n;
  for (int k = 0; k < 3; k++) c->origin[k] = (frame != UploadFrame::FromBox && origin_in) ? origin_in[k] : 0.0;
  c->has_normals = normals != nullptr;
  c->has_covs = cov6 != nullptr;
  SGA_TRY(c->pts.alloc(n));
  if (normals) SGA_TRY(c->nrm.alloc(n));
  if (cov6) SGA_TRY(c->cov.alloc(n));
  if (n == 0) {
    *out = c.release();
    return SGA_OK;
  }
  const size_t fx = n * 3, fn = normals ? n * 3 : 0, fc = cov6 ? n * 6 : 0;
  const dim3 grid((n + 255) / 256), block(256);
  static const bool zero_copy = !(getenv("SGA_UPLOAD_PINNED") && atoi(getenv("SGA_UPLOAD_PINNED")) == 0);
  const float* dx = zero_copy ? static_cast<const float*>(pinned_device_view(xyz, fx * sizeof(float))) : nullptr;
  const float* dn = (dx && normals) ? static_cast<const float*>(pinned_device_view(normals, fn * sizeof(float))) : nullptr;
  const float* dc = (dx && cov6) ? static_cast<const float*>(pinned_device_view(cov6, fc * sizeof(float))) : nullptr;
  double lo[3], hi[3];
  if (dx && (!normals || dn) && (!cov6 || dc)) {
    // ---- the caller's arrays are pinned: the kernel reads them in place
    if (frame == UploadFrame::FromBox) {
      unsigned long long* slot = nullptr;
      const unsigned long long seq = note_begin(ctx, &slot);
      hipLaunchKernelGGL(pack_cloud_kernel, grid, block, 0, ctx->stream, dx, dn, dc, n, 0.0, 0.0, 0.0, 0, c->pts.p, c->nrm.p, c->cov.p, ctx->d_box.p, slot, seq);
      SGA_HIP(hipGetLastError());
      unsigned long long payload[kNoteWords - 1];
      SGA_TRY(note_wait(ctx, seq, payload));
      float flo[3], fhi[3];
      box_note_decode(payload, flo, fhi);
      for (int k = 0; k < 3; k++) lo[k] = flo[k], hi[k] = fhi[k];
      choose_origin(lo, hi, c->origin);
      if (!origin_is_zero(c->origin)) {  // far from the origin (rare): once more, the subtraction in double
        hipLaunchKernelGGL(pack_cloud_kernel, grid, block, 0, ctx->stream, dx, nullptr, nullptr, n, c->origin[0], c->origin[1], c->origin[2], 1, c->pts.p, static_cast<float4*>(nullptr), static_cast<Cov8*>(nullptr), static_cast<int*>(nullptr),
                           static_cast<unsigned long long*>(nullptr), 0ull);
        SGA_HIP(hipGetLastError());
        SGA_HIP(hipStreamSynchronize(ctx->stream));  // the caller's buffer is being read
      }
    } else {
      hipLaunchKernelGGL(pack_cloud_kernel, grid, block, 0, ctx->stream, dx, dn, dc, n, c->origin[0], c->origin[1], c->origin[2], frame == UploadFrame::GivenRecentre ? 1 : 0, c->pts.p, c->nrm.p, c->cov.p, static_cast<int*>(nullptr),
                         static_cast<unsigned long long*>(nullptr), 0ull);
      SGA_HIP(hipGetLastError());
      SGA_HIP(hipStreamSynchronize(ctx->stream));  // the caller's buffer is being read
      lo[0] = INFINITY;  // (no box)
    }
  } else {
    // ---- pageable arrays: one CPU pass into the staging ring
    sga_context::StageSlot* slot = nullptr;
    SGA_TRY(stage_acquire(ctx, (fx + fn + fc) * sizeof(float), &slot));
    float* stage = static_cast<float*>(slot->host);
    const float* dstage = static_cast<const float*>(slot->dev);
    copy_with_box(xyz, stage, n, lo, hi);
    if (normals) std::memcpy(stage + fx, normals, fn * sizeof(float));
    if (cov6) std::memcpy(stage + fx + fn, cov6, fc * sizeof(float));
    if (frame == UploadFrame::FromBox) choose_origin(lo, hi, c->origin);
    const bool recentre = frame == UploadFrame::GivenRecentre || (frame == UploadFrame::FromBox && !origin_is_zero(c->origin));
    hipLaunchKernelGGL(pack_cloud_kernel, grid, block, 0, ctx->stream, dstage, normals ? dstage + fx : nullptr, cov6 ? dstage + fx + fn : nullptr, n, c->origin[0], c->origin[1], c->origin[2], recentre ? 1 : 0, c->pts.p, c->nrm.p, c->cov.p,
                       static_cast<int*>(nullptr), static_cast<unsigned long long*>(nullptr), 0ull);
    SGA_HIP(hipGetLastError());
    if (ctx->stream_ordered) {
      SGA_TRY(stage_release(ctx, slot));
    } else {
      SGA_HIP(hipStreamSynchronize(ctx->stream));
    }
  }
  if (lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2] && frame != UploadFrame::Given) {
    // the box of the records (device frame): outward-rounded fp32 of (box - origin)
    c->has_box = true;
    for (int k = 0; k < 3; k++) {
      c->box_lo[k] = std::nextafterf(static_cast<float>(lo[k] - c->origin[k]), -INFINITY);
      c->box_hi[k] = std::nextafterf(static_cast<float>(hi[k] - c->origin[k]), INFINITY);
    }
  } else if (lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]) {
    c->has_box = true;  // records relative to a given origin: the box of the inputs IS the box of the records
    for (int k = 0; k < 3; k++) c->box_lo[k] = static_cast<float>(lo[k]), c->box_hi[k] = static_cast<float>(hi[k]);
  }
  SGA_TRY(mark_ready(ctx, c->ready));
  *out = c.release();
  return SGA_OK;
}

// The cloud whose fp32 coordinates are given RELATIVE to `origin` (true position = xyz_rel + origin): the records go to the device as they are.
// recentre_by != nullptr: absolute fp32 coordinates, records = fl32(double(x) - recentre_by).
static int cloud_create_rel(sga_context* ctx, const float* xyz, const float* normals, const float* cov6, size_t n, const double origin[3], const double* recentre_by, sga_cloud** out) {
  static const double zero[3] = {0, 0, 0};
  return cloud_upload(ctx, xyz, normals, cov6, n, recentre_by ? UploadFrame::GivenRecentre : UploadFrame::Given, origin ? origin : zero, out);
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
  return cloud_upload(ctx, xyz, normals, cov6, n, UploadFrame::FromBox, nullptr, out);  // the origin: chosen from the box the upload takes in passing
}

// Pinned host memory for the caller's scans: sga_cloud_create_f32 reads arrays that live in it in place (no staging copy on the CPU).
int sga_host_alloc(size_t bytes, void** out) {
  if (!out) return fail(SGA_ERR_INVALID, "null argument");
  *out = nullptr;
  if (bytes == 0) return SGA_OK;
  if (hipHostMalloc(out, bytes, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
    (void)hipGetLastError();
    *out = nullptr;
    return fail(SGA_ERR_HIP, "hipHostMalloc(%zu bytes) failed", bytes);
  }
  return SGA_OK;
}
int sga_host_free(void* p) {
  if (p && hipHostFree(p) != hipSuccess) {
    (void)hipGetLastError();
    return fail(SGA_ERR_HIP, "hipHostFree failed");
  }
  return SGA_OK;
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
