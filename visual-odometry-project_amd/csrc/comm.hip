// Shared-map exchange over RCCL for hosts that are not PyTorch: one fixed-size record per rank
//     [ T_cw 4x4 row-major (16) | n (1) | n landmarks x 3, n <= cap ]   float64
// all-gathered over xGMI.  The reference has no distributed code (its README: one thread); this is the one collective
// of the sequence-sharded layout.  RCCL is not linked: librccl.so.1 is opened when the first communicator is made -- the
// copy the process already holds if there is one (PyTorch ships its own under the same soname), /opt/rocm's otherwise --
// so that two RCCLs never meet in one process and a single-GPU user never loads the 570 MB library.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "vo_internal.h"

struct vo_comm {
  ncclComm_t comm = nullptr;
  int world = 0, rank = 0, device = 0;
};

namespace {

struct rccl_api {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
};

rccl_api& api() {
  static rccl_api a = [] {
    rccl_api r;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);     // the copy this process already uses
      if (r.lib) break;
    }
    if (!r.lib)
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (r.lib) break;
      }
    if (!r.lib) return r;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))dlsym(r.lib, "ncclAllGather");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.GetErrorString;
    return r;
  }();
  return a;
}

int need_rccl(vo_ctx* ctx) {
  if (!api().ok) return vo_set_error(ctx, VO_EHIP, "RCCL is not available: librccl.so.1 could not be opened (%s)", dlerror());
  return VO_OK;
}

}  // namespace

extern "C" {

int vo_comm_unique_id(vo_ctx* ctx, void* id128) {
  if (!ctx || !id128) return VO_EINVAL;
  VO_TRY(need_rccl(ctx));
  static_assert(sizeof(ncclUniqueId) == VO_COMM_ID_BYTES, "vo_hip.h: VO_COMM_ID_BYTES");
  ncclUniqueId id;
  const ncclResult_t r = api().GetUniqueId(&id);
  if (r != ncclSuccess) return vo_set_error(ctx, VO_EHIP, "ncclGetUniqueId: %s", api().GetErrorString(r));
  memcpy(id128, &id, sizeof(id));
  return VO_OK;
}

int vo_comm_create(vo_ctx* ctx, int world, int rank, const void* id128, vo_comm** out) {
  if (!ctx || !id128 || !out) return VO_EINVAL;
  *out = nullptr;
  VO_REQUIRE(ctx, world >= 1 && rank >= 0 && rank < world, "comm_create: bad world / rank");
  VO_TRY(need_rccl(ctx));
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  vo_comm* c = new (std::nothrow) vo_comm();
  if (!c) return VO_ENOMEM;
  const ncclResult_t r = api().CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) {
    delete c;
    return vo_set_error(ctx, VO_EHIP, "ncclCommInitRank(world %d, rank %d): %s", world, rank, api().GetErrorString(r));
  }
  c->world = world;
  c->rank = rank;
  c->device = ctx->device;
  *out = c;
  return VO_OK;
}

void vo_comm_destroy(vo_comm* c) {
  if (!c) return;
  if (c->comm && api().ok) (void)api().CommDestroy(c->comm);
  delete c;
}

int vo_comm_world(const vo_comm* c) { return c ? c->world : 0; }

int vo_allgather_state_dev(vo_ctx* ctx, vo_comm* c, const double* d_records, size_t doubles_per_rank, double* d_all,
                           void* stream) {
  if (!ctx || !c) return VO_EINVAL;
  VO_REQUIRE(ctx, d_records && d_all && doubles_per_rank > 0, "allgather_state: null pointer or empty record");
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
  const ncclResult_t r = api().AllGather(d_records, d_all, doubles_per_rank, ncclDouble, c->comm, st);
  if (r != ncclSuccess) return vo_set_error(ctx, VO_EHIP, "ncclAllGather: %s", api().GetErrorString(r));
  return VO_OK;
}

int vo_allgather_state(vo_ctx* ctx, vo_comm* c, const double* pose16, const double* landmarks, int n, int cap, double* all) {
  if (!ctx || !c) return VO_EINVAL;
  VO_REQUIRE(ctx, pose16 && all && n >= 0 && cap >= 0 && (n == 0 || landmarks), "allgather_state: bad arguments");
  const int m = n < cap ? n : cap;
  const size_t len = 17 + (size_t)3 * cap;
  VO_HIP_TRY(ctx, hipSetDevice(ctx->device));
  vo_buf* s = ctx->scratch;
  VO_TRY(vo_ensure(ctx, s[0], len * 8));
  VO_TRY(vo_ensure(ctx, s[1], len * 8 * (size_t)c->world));
  VO_TRY(vo_ensure_pinned(ctx, len * 8));
  double* h = (double*)ctx->h_pin;
  memset(h, 0, len * 8);
  memcpy(h, pose16, 128);
  h[16] = (double)m;
  if (m > 0) memcpy(h + 17, landmarks, (size_t)m * 24);
  hipStream_t st = ctx->stream;
  VO_HIP_TRY(ctx, hipMemcpyAsync(s[0].p, h, len * 8, hipMemcpyHostToDevice, st));
  VO_TRY(vo_allgather_state_dev(ctx, c, (const double*)s[0].p, len, (double*)s[1].p, st));
  VO_HIP_TRY(ctx, hipMemcpyAsync(all, s[1].p, len * 8 * (size_t)c->world, hipMemcpyDeviceToHost, st));
  VO_HIP_TRY(ctx, hipStreamSynchronize(st));
  return VO_OK;
}

}  // extern "C"
