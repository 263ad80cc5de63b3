// ah_comm.hip — the path's ONE exchange step through the C ABI: RCCL over xGMI, on the context's compute stream.
//
// No reference analogue (arrow-go is single-process).  What a Go host needs to run configs C4 / C5 on the 8 GPUs of
// a node without Python: an all-reduce of the fused kernel's 16-byte {sum, count} partial (C4), and a
// ragged all-to-all + all-gather of O(groups) tuples for the key-hash-owner merge (C5).  One process per GPU; rank 0
// makes a unique id (ah_comm_unique_id), the host ships those 128 bytes to the other ranks by whatever it has (a file,
// a socket, the Go process launcher), every rank calls ah_comm_init.  Collectives are enqueued on the ah_ctx's compute
// stream behind the kernels that produced their inputs and return without waiting.
//
// RCCL is bound at first use with dlopen — librccl.so.1 as already loaded by the process if there is one (torch ships
// its own), else the system's — so libarrowhip.so itself carries no link-time dependency on it.
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>
#include "ah_common.h"

struct ah_comm {
  ah_ctx* ctx;
  ncclComm_t comm;
  int rank, world;
};

namespace {
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  char err[256] = {0};
};
Rccl g_rccl;

const char* load_rccl_once();
const char* load_rccl() {  // nullptr = ok, else why not; safe from any number of threads (Go runs ExecFns on whatever OS thread it likes)
  static std::once_flag once;
  static const char* why = nullptr;
  std::call_once(once, [] { why = load_rccl_once(); });
  return why;
}
const char* load_rccl_once() {
  const char* override_path = getenv("ARROWHIP_RCCL");
  const char* names[] = {override_path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (const char* n : names) {
    if (!n) continue;
    h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) { snprintf(g_rccl.err, sizeof g_rccl.err, "cannot load librccl.so (%s)", dlerror()); return g_rccl.err; }
#define AH_SYM(field, name) \
  *(void**)&g_rccl.field = dlsym(h, name); \
  if (!g_rccl.field) { snprintf(g_rccl.err, sizeof g_rccl.err, "librccl.so has no %s", name); return g_rccl.err; }
  AH_SYM(GetUniqueId, "ncclGetUniqueId") AH_SYM(CommInitRank, "ncclCommInitRank") AH_SYM(CommDestroy, "ncclCommDestroy")
  AH_SYM(AllReduce, "ncclAllReduce") AH_SYM(AllGather, "ncclAllGather") AH_SYM(Send, "ncclSend") AH_SYM(Recv, "ncclRecv")
  AH_SYM(GroupStart, "ncclGroupStart") AH_SYM(GroupEnd, "ncclGroupEnd") AH_SYM(GetErrorString, "ncclGetErrorString")
#undef AH_SYM
  g_rccl.lib = h;
  return nullptr;
}

int fail_nccl(ah_ctx* c, const char* what, ncclResult_t r) {
  return ah_fail(c, AH_EHIP, "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
}
#define AH_NCCL(c, call)                                 \
  do {                                                   \
    ncclResult_t r__ = (call);                           \
    if (r__ != ncclSuccess) return fail_nccl((c), #call, r__); \
  } while (0)

bool nccl_type(int type, ncclDataType_t* t) {
  switch (type) {
    case AH_INT64: *t = ncclInt64; return true;
    case AH_UINT64: *t = ncclUint64; return true;
    case AH_FLOAT64: *t = ncclFloat64; return true;
    case AH_INT32: *t = ncclInt32; return true;
    case AH_UINT32: *t = ncclUint32; return true;
    case AH_FLOAT32: *t = ncclFloat32; return true;
  }
  return false;
}
}  // namespace

AH_EXPORT int ah_comm_unique_id(void* id_host128) {
  if (!id_host128) return AH_EINVALID;
  if (load_rccl()) return AH_EHIP;
  ncclUniqueId id;
  if (g_rccl.GetUniqueId(&id) != ncclSuccess) return AH_EHIP;
  static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
  memcpy(id_host128, &id, sizeof(id));
  return AH_OK;
}

AH_EXPORT int ah_comm_init(ah_ctx* c, int rank, int world, const void* unique_id_host128, ah_comm** out) {
  AH_ENTER(c);
  if (!out) return ah_fail(c, AH_EINVALID, "comm_init: null out pointer");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world || !unique_id_host128) return ah_fail(c, AH_EINVALID, "comm_init: bad rank / world / id");
  if (const char* why = load_rccl()) return ah_fail(c, AH_EHIP, "comm_init: %s", why);
  ncclUniqueId id;
  memcpy(&id, unique_id_host128, sizeof(id));
  ncclComm_t comm;
  AH_NCCL(c, g_rccl.CommInitRank(&comm, world, id, rank));
  ah_comm* m = (ah_comm*)calloc(1, sizeof(ah_comm));
  if (!m) { g_rccl.CommDestroy(comm); return ah_fail(c, AH_EINVALID, "comm_init: out of memory"); }
  m->ctx = c; m->comm = comm; m->rank = rank; m->world = world;
  *out = m;
  return AH_OK;
}

AH_EXPORT int ah_comm_destroy(ah_comm* m) {
  if (!m) return AH_OK;
  (void)hipSetDevice(m->ctx->device);
  (void)hipStreamSynchronize(m->ctx->stream);
  if (g_rccl.CommDestroy) g_rccl.CommDestroy(m->comm);
  free(m);
  return AH_OK;
}

AH_EXPORT int ah_comm_rank(ah_comm* m) { return m ? m->rank : -1; }
AH_EXPORT int ah_comm_world(ah_comm* m) { return m ? m->world : -1; }

AH_EXPORT int ah_comm_allreduce_sum(ah_comm* m, int type, const void* send, void* recv, int64_t count) {
  if (!m) return AH_EINVALID;
  ah_ctx* c = m->ctx;
  AH_ENTER(c);
  ncclDataType_t t;
  if (!nccl_type(type, &t)) return ah_fail(c, AH_EINVALID, "allreduce: unsupported element type %d", type);
  if (count < 0 || (count > 0 && (!send || !recv))) return ah_fail(c, AH_EINVALID, "allreduce: bad buffer / count");
  if (count == 0) return AH_OK;
  AH_NCCL(c, g_rccl.AllReduce(send, recv, (size_t)count, t, ncclSum, m->comm, c->stream));
  return AH_OK;
}

AH_EXPORT int ah_comm_allgather(ah_comm* m, const void* send, void* recv, int64_t nbytes_per_rank) {
  if (!m) return AH_EINVALID;
  ah_ctx* c = m->ctx;
  AH_ENTER(c);
  if (nbytes_per_rank < 0 || (nbytes_per_rank > 0 && (!send || !recv))) return ah_fail(c, AH_EINVALID, "allgather: bad buffer / size");
  if (nbytes_per_rank == 0) return AH_OK;
  AH_NCCL(c, g_rccl.AllGather(send, recv, (size_t)nbytes_per_rank, ncclUint8, m->comm, c->stream));
  return AH_OK;
}

AH_EXPORT int ah_comm_alltoallv(ah_comm* m, const void* send, const int64_t* send_bytes_host, const int64_t* send_offs_host, void* recv,
                                const int64_t* recv_bytes_host, const int64_t* recv_offs_host) {
  if (!m) return AH_EINVALID;
  ah_ctx* c = m->ctx;
  AH_ENTER(c);
  if (!send_bytes_host || !send_offs_host || !recv_bytes_host || !recv_offs_host) return ah_fail(c, AH_EINVALID, "alltoallv: null size / offset array");
  for (int r = 0; r < m->world; r++)
    if (send_bytes_host[r] < 0 || recv_bytes_host[r] < 0 || send_offs_host[r] < 0 || recv_offs_host[r] < 0 ||
        (send_bytes_host[r] > 0 && !send) || (recv_bytes_host[r] > 0 && !recv))
      return ah_fail(c, AH_EINVALID, "alltoallv: bad size / offset for rank %d", r);
  // the own block never leaves the device
  if (send_bytes_host[m->rank] != recv_bytes_host[m->rank]) return ah_fail(c, AH_EINVALID, "alltoallv: own block sizes differ");
  if (send_bytes_host[m->rank] > 0)
    AH_HIP(c, hipMemcpyAsync((uint8_t*)recv + recv_offs_host[m->rank], (const uint8_t*)send + send_offs_host[m->rank],
                             (size_t)send_bytes_host[m->rank], hipMemcpyDeviceToDevice, c->stream));
  if (m->world == 1) return AH_OK;
  // direct exchange: all 7 xGMI links of a GPU carry a block at the same time (a ring would be per-link bound)
  AH_NCCL(c, g_rccl.GroupStart());
  for (int r = 0; r < m->world; r++) {
    if (r == m->rank) continue;
    if (send_bytes_host[r] > 0) {
      ncclResult_t e = g_rccl.Send((const uint8_t*)send + send_offs_host[r], (size_t)send_bytes_host[r], ncclUint8, r, m->comm, c->stream);
      if (e != ncclSuccess) { g_rccl.GroupEnd(); return fail_nccl(c, "ncclSend", e); }
    }
    if (recv_bytes_host[r] > 0) {
      ncclResult_t e = g_rccl.Recv((uint8_t*)recv + recv_offs_host[r], (size_t)recv_bytes_host[r], ncclUint8, r, m->comm, c->stream);
      if (e != ncclSuccess) { g_rccl.GroupEnd(); return fail_nccl(c, "ncclRecv", e); }
    }
  }
  AH_NCCL(c, g_rccl.GroupEnd());
  return AH_OK;
}
