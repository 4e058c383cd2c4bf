// Multi-GPU: all-reduce of the linearization accumulator (96 doubles: the system + the error-model moments; 30 / 1 for robust
// factors) enqueued on the context's stream between the reduction kernel and the device->host read-back, so that sga_linearize /
// sga_error / sga_align work unchanged on a source cloud sharded over ranks (the loop being partitioned is
// registration/reduction_omp.hpp:32-58).  Transport: RCCL (ncclAllReduce on the stream), or a caller-supplied host function
// (sga_comm_init_callback).  librccl is bound at run time with dlopen — the library has no link-time
// dependency on it and single-GPU users never load it.  One communicator per context; ranks exchange the 128-byte unique id out
// of band (bench.py broadcasts it with torch.distributed).
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "common.hpp"

namespace sga {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
  ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
  const char* (*get_error_string)(ncclResult_t) = nullptr;
};

static RcclApi* rccl() {
  static RcclApi api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    // a librccl the process has already mapped (PyTorch ships its own copy) first: two RCCL instances in one process would each
    // bring their own device state
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      api.handle = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
      if (api.handle) break;
    }
    if (!api.handle) {
      for (const char* name : {"librccl.so.1", "librccl.so"}) {
        api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (api.handle) break;
      }
    }
    if (api.handle) {
      api.get_unique_id = reinterpret_cast<decltype(api.get_unique_id)>(dlsym(api.handle, "ncclGetUniqueId"));
      api.comm_init_rank = reinterpret_cast<decltype(api.comm_init_rank)>(dlsym(api.handle, "ncclCommInitRank"));
      api.all_reduce = reinterpret_cast<decltype(api.all_reduce)>(dlsym(api.handle, "ncclAllReduce"));
      api.comm_destroy = reinterpret_cast<decltype(api.comm_destroy)>(dlsym(api.handle, "ncclCommDestroy"));
      api.get_error_string = reinterpret_cast<decltype(api.get_error_string)>(dlsym(api.handle, "ncclGetErrorString"));
      if (!api.get_unique_id || !api.comm_init_rank || !api.all_reduce || !api.comm_destroy) api.handle = nullptr;
    }
  }
  return api.handle ? &api : nullptr;
}

// called by linearize.hip between the reduction and the read-back; no-op without a communicator
int comm_allreduce_sum(sga_context* ctx, double* d_buf, size_t count) {
  if (ctx->comm_fn) {  // the caller's transport: down, sum over ranks on the host, up — at the point of the stream order where RCCL would run
    double h[128];
    if (count > 128) return fail(SGA_ERR_INVALID, "accumulator of %zu doubles", count);
    SGA_HIP(hipMemcpyAsync(h, d_buf, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SGA_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->comm_fn(ctx->comm_user, h, count) != 0) return fail(SGA_ERR_CALLBACK, "all-reduce callback failed");
    SGA_HIP(hipMemcpyAsync(d_buf, h, count * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SGA_HIP(hipStreamSynchronize(ctx->stream));  // h goes out of scope
    return SGA_OK;
  }
  if (!ctx->comm) return SGA_OK;
  RcclApi* api = rccl();
  if (!api) return fail(SGA_ERR_HIP, "librccl is not available");
  const ncclResult_t r = api->all_reduce(d_buf, d_buf, count, ncclDouble, ncclSum, static_cast<ncclComm_t>(ctx->comm), ctx->stream);
  if (r != ncclSuccess) return fail(SGA_ERR_HIP, "ncclAllReduce -> %s", api->get_error_string ? api->get_error_string(r) : "error");
  return SGA_OK;
}

}  // namespace sga

using namespace sga;

extern "C" {

int sga_comm_unique_id(unsigned char id[128]) {
  if (!id) return fail(SGA_ERR_INVALID, "null argument");
  RcclApi* api = rccl();
  if (!api) {
    const char* why = dlerror();  // one call: the second one returns NULL
    return fail(SGA_ERR_HIP, "librccl is not available: %s", why ? why : "dlopen failed");
  }
  ncclUniqueId uid;
  const ncclResult_t r = api->get_unique_id(&uid);
  if (r != ncclSuccess) return fail(SGA_ERR_HIP, "ncclGetUniqueId -> %s", api->get_error_string ? api->get_error_string(r) : "error");
  static_assert(sizeof(uid) == 128, "ncclUniqueId is 128 bytes");
  memcpy(id, &uid, 128);
  return SGA_OK;
}

int sga_comm_init(sga_context* ctx, int nranks, int rank, const unsigned char id[128]) {
  if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(SGA_ERR_INVALID, "bad argument");
  if (ctx->sharded()) return fail(SGA_ERR_INVALID, "context already has a communicator");
  RcclApi* api = rccl();
  if (!api) return fail(SGA_ERR_HIP, "librccl is not available");
  SGA_ENTER(ctx);
  ncclUniqueId uid;
  memcpy(&uid, id, 128);
  ncclComm_t comm = nullptr;
  const ncclResult_t r = api->comm_init_rank(&comm, nranks, uid, rank);
  if (r != ncclSuccess) return fail(SGA_ERR_HIP, "ncclCommInitRank -> %s", api->get_error_string ? api->get_error_string(r) : "error");
  ctx->comm = comm;
  ctx->comm_ranks = nranks;
  return SGA_OK;
}

int sga_comm_init_callback(sga_context* ctx, int nranks, int rank, sga_allreduce_fn fn, void* user) {
  if (!ctx || !fn || nranks < 1 || rank < 0 || rank >= nranks) return fail(SGA_ERR_INVALID, "bad argument");
  if (ctx->sharded()) return fail(SGA_ERR_INVALID, "context already has a communicator");
  ctx->comm_fn = fn;
  ctx->comm_user = user;
  ctx->comm_ranks = nranks;
  return SGA_OK;
}

int sga_comm_destroy(sga_context* ctx) {
  if (!ctx) return fail(SGA_ERR_INVALID, "null argument");
  if (ctx->comm_fn) {
    (void)hipStreamSynchronize(ctx->stream);
    ctx->comm_fn = nullptr;
    ctx->comm_user = nullptr;
    ctx->comm_ranks = 1;
  }
  if (ctx->comm) {
    RcclApi* api = rccl();
    (void)hipStreamSynchronize(ctx->stream);
    if (api) api->comm_destroy(static_cast<ncclComm_t>(ctx->comm));
    ctx->comm = nullptr;
    ctx->comm_ranks = 1;
  }
  return SGA_OK;
}

}  // extern "C"
