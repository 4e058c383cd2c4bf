// Notes: a few words handed from a kernel to the host WITHOUT a copy command and without a stream synchronisation.
//
// A preprocessing chain (upload -> voxel grid -> kd-tree -> covariances, registration_helper.cpp:22-34) needs three small facts on the
// host before it can size its next launch: the bounding box of an uploaded cloud (-> the origin of its device frame), the number of
// occupied voxels, the bounding box of the downsampled cloud.  Fetched with hipMemcpyAsync + hipStreamSynchronize each of them costs two
// commands and a blocking wait (measured on a C5 scan: the voxel-grid stage took 240 us of wall time for 120 us of kernels).  Here the
// kernel that produces the fact stores it into the context's pinned, device-mapped note block and then publishes a sequence number with
// a system-scope release — the hand-off the linearization results already use (linearize.hip: reduce_rows_kernel / wait_result) — and
// the host spins on that word: its wait ends a microsecond or two after the kernel's last store.
#pragma once
#include "common.hpp"

namespace sga {

constexpr int kNoteSlots = 4;  // notes in flight per context (slot = seq % kNoteSlots); every producer waits for its own note before it returns
constexpr int kNoteWords = 8;  // per slot: word 0 = the sequence number, words 1..7 = payload

// host side (context.hip)
unsigned long long note_begin(sga_context* ctx, unsigned long long** dev_slot);             // next sequence number + the device address of its slot
int note_wait(sga_context* ctx, unsigned long long seq, unsigned long long payload[kNoteWords - 1]);  // spin (bounded; then the runtime reports what happened)

// LATE notes: results nobody waits for.  A process-wide ring of kLateSlots slots in pinned, device-mapped, portable host memory; the
// producer remembers the sequence number it was given, the kernel writes {payload..., seq} into slot seq % kLateSlots when it gets there,
// and whoever wants the value later looks: the slot shows the number -> the payload is valid; an older number -> not there yet; a newer
// one -> overwritten (kLateSlots later producers have passed), the value is lost — acceptable for what travels this way: heuristics
// (the target's length scale, which only steers the choice between exact kernels).
constexpr int kLateSlots = 1024;
constexpr int kLateWords = 4;  // words 0..2 payload, word 3 = the sequence number
unsigned long long late_note_begin(int device, unsigned long long** dev_slot);  // null slot: no ring (allocation failed): the value stays unknown
// 1 = payload read, 0 = not there yet, -1 = lost
int late_note_peek(unsigned long long seq, unsigned long long payload[kLateWords - 1]);

// order-preserving int encoding of floats (atomicMin / atomicMax on ints)
__host__ __device__ inline int box_enc(float f) {
  int i;
  memcpy(&i, &f, 4);
  return i >= 0 ? i : i ^ 0x7fffffff;
}
__host__ __device__ inline float box_dec(int i) {
  const int j = i >= 0 ? i : i ^ 0x7fffffff;
  float f;
  memcpy(&f, &j, 4);
  return f;
}
constexpr int kBoxEncPosInf = 0x7f800000;
constexpr int kBoxEncNegInf = static_cast<int>(0xff800000u) ^ 0x7fffffff;

#if defined(__HIPCC__)
// by ONE thread, after the payload stores of its workgroup are visible to it (a __syncthreads() before)
__device__ __forceinline__ void note_publish(unsigned long long* slot, unsigned long long seq) {
  __threadfence_system();
  __hip_atomic_store(slot, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Bounding box over a launch of 256-thread workgroups -> note.  Every thread brings the box of its own points (lo = +inf / hi = -inf
// for none); d_box = the context's eight ints {min x y z, max x y z (encoded), arrival counter, 0}, which every launch leaves as it
// found them (identity values, counter 0).  The last workgroup to arrive writes payload words 1..3 = (min, max) pairs per axis,
// words 4.. = extra[], and publishes.
__device__ __forceinline__ void box_reduce_publish(float lo[3], float hi[3], int* __restrict__ d_box, unsigned long long* __restrict__ slot, unsigned long long seq, const unsigned long long* extra = nullptr, int num_extra = 0) {
  __shared__ float sh_box[4][6];
  __shared__ bool sh_last;
#pragma unroll
  for (int k = 0; k < 3; k++) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[k] = fminf(lo[k], __shfl_xor(lo[k], off));
      hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off));
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
      sh_box[wave][k] = lo[k];
      sh_box[wave][3 + k] = hi[k];
    }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int k = threadIdx.x;
    float v = sh_box[0][k];
    for (int w = 1; w < static_cast<int>(blockDim.x >> 6); w++) v = k < 3 ? fminf(v, sh_box[w][k]) : fmaxf(v, sh_box[w][k]);
    if (k < 3)
      atomicMin(&d_box[k], box_enc(v));
    else
      atomicMax(&d_box[k], box_enc(v));
  }
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) sh_last = __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(&d_box[6]), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
  __syncthreads();
  if (!sh_last) return;  // workgroup-uniform
  if (threadIdx.x < 3) {
    const int k = threadIdx.x;
    const unsigned a = static_cast<unsigned>(__hip_atomic_load(&d_box[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    const unsigned b = static_cast<unsigned>(__hip_atomic_load(&d_box[3 + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    slot[1 + k] = static_cast<unsigned long long>(a) | (static_cast<unsigned long long>(b) << 32);
    __hip_atomic_store(&d_box[k], kBoxEncPosInf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&d_box[3 + k], kBoxEncNegInf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x >= 8 && static_cast<int>(threadIdx.x) < 8 + num_extra && static_cast<int>(threadIdx.x) - 8 < kNoteWords - 4) slot[4 + threadIdx.x - 8] = extra[threadIdx.x - 8];
  if (threadIdx.x == 0) __hip_atomic_store(reinterpret_cast<unsigned*>(&d_box[6]), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) note_publish(slot, seq);
}
#endif

// payload words 1..3 of a box note -> floats
inline void box_note_decode(const unsigned long long payload[kNoteWords - 1], float lo[3], float hi[3]) {
  for (int k = 0; k < 3; k++) {
    lo[k] = box_dec(static_cast<int>(static_cast<unsigned>(payload[k] & 0xffffffffull)));
    hi[k] = box_dec(static_cast<int>(static_cast<unsigned>(payload[k] >> 32)));
  }
}

}  // namespace sga
