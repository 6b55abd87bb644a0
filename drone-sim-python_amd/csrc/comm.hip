// The convergence exchange of the sharded fit as a C-ABI entry point (d2d_allreduce_stats), for hosts without
// torch.distributed: RCCL (the ROCm collective library over xGMI) is loaded at run time -- dlopen, so that libd2dhip.so itself
// carries no link-time dependency on it and single-GPU users never touch it.
#include <dlfcn.h>

#include <cstring>

#include "common.h"

namespace {
// the handful of RCCL declarations used (rccl.h: ncclUniqueId is 128 opaque bytes; ncclFloat64 = 8; ncclSum = 0, ncclMax = 2)
struct UniqueId { char internal[128]; };
typedef void *Comm;
typedef int (*GetUniqueIdFn)(UniqueId *);
typedef int (*CommInitRankFn)(Comm *, int, UniqueId, int);
typedef int (*CommDestroyFn)(Comm);
typedef int (*AllReduceFn)(const void *, void *, size_t, int, int, Comm, hipStream_t);
typedef int (*GroupFn)(void);
typedef const char *(*ErrStrFn)(int);
typedef int (*CommQueryFn)(const Comm, int *);      // ncclCommCount, ncclCommUserRank
struct Rccl {
  void *h = nullptr;
  GetUniqueIdFn get_id = nullptr;
  CommInitRankFn init_rank = nullptr;
  CommDestroyFn destroy = nullptr;
  AllReduceFn all_reduce = nullptr;
  GroupFn group_start = nullptr, group_end = nullptr;
  ErrStrFn err = nullptr;
  CommQueryFn count = nullptr, user_rank = nullptr;
};
Rccl g_rccl;

int load_rccl() {
  if (g_rccl.h) return D2D_OK;
  void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) { d2d_set_error("RCCL not found (dlopen librccl.so.1): %s", dlerror()); return D2D_ESTATE; }
  Rccl r;
  r.h = h;
  r.get_id = reinterpret_cast<GetUniqueIdFn>(dlsym(h, "ncclGetUniqueId"));
  r.init_rank = reinterpret_cast<CommInitRankFn>(dlsym(h, "ncclCommInitRank"));
  r.destroy = reinterpret_cast<CommDestroyFn>(dlsym(h, "ncclCommDestroy"));
  r.all_reduce = reinterpret_cast<AllReduceFn>(dlsym(h, "ncclAllReduce"));
  r.group_start = reinterpret_cast<GroupFn>(dlsym(h, "ncclGroupStart"));
  r.group_end = reinterpret_cast<GroupFn>(dlsym(h, "ncclGroupEnd"));
  r.err = reinterpret_cast<ErrStrFn>(dlsym(h, "ncclGetErrorString"));
  r.count = reinterpret_cast<CommQueryFn>(dlsym(h, "ncclCommCount"));
  r.user_rank = reinterpret_cast<CommQueryFn>(dlsym(h, "ncclCommUserRank"));
  if (!r.get_id || !r.init_rank || !r.destroy || !r.all_reduce || !r.group_start || !r.group_end || !r.count || !r.user_rank) {
    d2d_set_error("librccl is missing an expected symbol");
    return D2D_ESTATE;
  }
  g_rccl = r;
  return D2D_OK;
}
#define D2D_CHECK_NCCL(expr)                                                                                  \
  do {                                                                                                        \
    const int e_ = (expr);                                                                                    \
    if (e_ != 0) {                                                                                            \
      d2d_set_error("%s failed: %s", #expr, g_rccl.err ? g_rccl.err(e_) : "RCCL error");                      \
      return D2D_EHIP;                                                                                        \
    }                                                                                                         \
  } while (0)
}  // namespace

struct d2d_comm {
  Comm comm = nullptr;
  int rank = 0, world = 1;
};

extern "C" {

int d2d_comm_available(void) {
  // local only: can this process load RCCL and resolve every symbol the exchange uses?  (ncclCommInitRank blocks until all
  // ranks have joined, so every rank must know BEFORE d2d_comm_create that every other rank will get there)
  return load_rccl();
}

int d2d_comm_unique_id(void *id_out) {
  D2D_REQUIRE(id_out != nullptr, "d2d_comm_unique_id: id_out is NULL");
  if (int rc = load_rccl()) return rc;
  UniqueId id;
  D2D_CHECK_NCCL(g_rccl.get_id(&id));
  std::memcpy(id_out, id.internal, D2D_COMM_ID_BYTES);
  return D2D_OK;
}

int d2d_comm_create(d2d_ctx *ctx, const void *id, int rank, int world, d2d_comm **out) {
  D2D_REQUIRE(ctx && id && out, "d2d_comm_create: null argument");
  D2D_REQUIRE(world >= 1 && rank >= 0 && rank < world, "d2d_comm_create: rank %d not in [0, %d)", rank, world);
  if (int rc = load_rccl()) return rc;
  D2D_CHECK_HIP(hipSetDevice(ctx->device));
  UniqueId uid;
  std::memcpy(uid.internal, id, D2D_COMM_ID_BYTES);
  d2d_comm *c = new d2d_comm();
  c->rank = rank; c->world = world;
  const int e = g_rccl.init_rank(&c->comm, world, uid, rank);
  if (e != 0) {
    d2d_set_error("ncclCommInitRank failed: %s", g_rccl.err ? g_rccl.err(e) : "RCCL error");
    delete c;
    return D2D_EHIP;
  }
  *out = c;
  return D2D_OK;
}

int d2d_comm_destroy(d2d_comm *comm) {
  if (!comm) return D2D_OK;
  if (comm->comm && g_rccl.destroy) g_rccl.destroy(comm->comm);
  delete comm;
  return D2D_OK;
}

int d2d_comm_info(const d2d_comm *comm, int32_t *rank, int32_t *world) {
  D2D_REQUIRE(comm != nullptr, "d2d_comm_info: comm is NULL");
  // what the COMMUNICATOR says (ncclCommUserRank / ncclCommCount), not what the caller passed to d2d_comm_create: the caller's
  // values are only the cross-check -- a communicator that disagrees with them is not the one the caller thinks it has
  int r = -1, w = -1;
  D2D_CHECK_NCCL(g_rccl.user_rank(comm->comm, &r));
  D2D_CHECK_NCCL(g_rccl.count(comm->comm, &w));
  if (r != comm->rank || w != comm->world) {
    d2d_set_error("d2d_comm_info: RCCL reports rank %d of %d, d2d_comm_create was given rank %d of %d", r, w, comm->rank, comm->world);
    return D2D_ESTATE;
  }
  if (rank) *rank = r;
  if (world) *world = w;
  return D2D_OK;
}

int d2d_allreduce_stats(d2d_ctx *ctx, d2d_comm *comm, double *stats) {
  D2D_REQUIRE(ctx && comm && stats, "d2d_allreduce_stats: null argument");
  // (a host pointer would fail asynchronously inside the collective: refuse it here)
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, stats) != hipSuccess || attr.type != hipMemoryTypeDevice) {
    (void)hipGetLastError();
    d2d_set_error("d2d_allreduce_stats: stats must point to device memory (three doubles)");
    return D2D_EINVAL;
  }
  // one grouped exchange: sum of the costs and of the running counts, max of the gradient norms.  A failure between
  // group_start and group_end still closes the group: left open, it would swallow every later collective of this thread.
  D2D_CHECK_NCCL(g_rccl.group_start());
  int e = g_rccl.all_reduce(stats, stats, 1, 8 /* ncclFloat64 */, 0 /* ncclSum */, comm->comm, ctx->stream);
  if (e == 0) e = g_rccl.all_reduce(stats + 1, stats + 1, 1, 8, 2 /* ncclMax */, comm->comm, ctx->stream);
  if (e == 0) e = g_rccl.all_reduce(stats + 2, stats + 2, 1, 8, 0, comm->comm, ctx->stream);
  const int e_end = g_rccl.group_end();
  if (e == 0) e = e_end;
  if (e != 0) {
    d2d_set_error("d2d_allreduce_stats: ncclAllReduce failed: %s", g_rccl.err ? g_rccl.err(e) : "RCCL error");
    return D2D_EHIP;
  }
  return D2D_OK;
}

}  // extern "C"
