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
#include <vector>
#include <rccl/rccl.h>
#include "ah_common.h"
#include "ah_ddsum.h"

struct ah_comm {
  ah_ctx* ctx;
  ncclComm_t comm;          // RCCL communicator, or nullptr for a host-transport communicator
  int rank, world;
  ah_transport transport;   // host-transport flavour (ah_comm_init_transport): the two exchanges over host memory
  bool has_transport;
  // grow-only blocks owned by the communicator: device temporaries of the C4 / C5 entry points, and pinned host staging
  uint8_t* arena; size_t arena_bytes;     // merge_groups: bucketing, exchange and re-aggregation
  uint8_t* arena2; size_t arena2_bytes;   // merge_groups: the gathered groups (reserved while the first still holds live data)
  uint8_t* stage; size_t stage_bytes;
  uint8_t* gather; size_t gather_bytes;   // comm_allgather_host over RCCL: the (world + 1) × count words of a size table
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

// ---- host-transport flavour: the same three collectives with the bytes carried by callbacks the host supplies (its launcher's
// sockets, MPI, a gloo group in the tests) instead of RCCL.  Device blocks are staged through pinned host memory; this is the
// flavour for hosts without RCCL between their processes — and for running N ranks on ONE GPU, which RCCL refuses.
namespace {
int stage_reserve(ah_comm* m, size_t nbytes, uint8_t** out) {
  ah_ctx* c = m->ctx;
  if (nbytes > m->stage_bytes) {
    if (m->stage) AH_HIP(c, hipHostFree(m->stage));
    m->stage = nullptr; m->stage_bytes = 0;
    const size_t want = (nbytes + 65535) & ~(size_t)65535;
    AH_HIP(c, hipHostMalloc((void**)&m->stage, want, hipHostMallocDefault));
    m->stage_bytes = want;
  }
  *out = m->stage;
  return AH_OK;
}
int arena_reserve_in(ah_comm* m, uint8_t** arena, size_t* have, size_t nbytes, uint8_t** out) {
  ah_ctx* c = m->ctx;
  if (nbytes > *have) {
    AH_HIP(c, hipStreamSynchronize(c->stream));
    if (*arena) AH_HIP(c, hipFree(*arena));
    *arena = nullptr; *have = 0;
    const size_t want = (nbytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    AH_HIP(c, hipMalloc((void**)arena, want));
    *have = want;
  }
  *out = *arena;
  return AH_OK;
}
int arena_reserve(ah_comm* m, size_t nbytes, uint8_t** out) { return arena_reserve_in(m, &m->arena, &m->arena_bytes, nbytes, out); }
int arena2_reserve(ah_comm* m, size_t nbytes, uint8_t** out) { return arena_reserve_in(m, &m->arena2, &m->arena2_bytes, nbytes, out); }
int transport_fail(ah_ctx* c, const char* what, int rc) { return ah_fail(c, AH_EHIP, "%s: the host transport returned %d", what, rc); }

int transport_allgather(ah_comm* m, const void* send, void* recv, int64_t nbytes) {
  ah_ctx* c = m->ctx;
  uint8_t* h;
  int rc = stage_reserve(m, (size_t)nbytes * (size_t)(m->world + 1), &h);
  if (rc != AH_OK) return rc;
  AH_HIP(c, hipMemcpyAsync(h, send, (size_t)nbytes, hipMemcpyDeviceToHost, c->stream));
  AH_HIP(c, hipStreamSynchronize(c->stream));
  const int trc = m->transport.allgather(m->transport.user, h, h + nbytes, nbytes);
  if (trc) return transport_fail(c, "allgather", trc);
  AH_HIP(c, hipMemcpyAsync(recv, h + nbytes, (size_t)nbytes * (size_t)m->world, hipMemcpyHostToDevice, c->stream));
  AH_HIP(c, hipStreamSynchronize(c->stream));   // the staging block is reused by the next call
  return AH_OK;
}
// every rank's vector, added in RANK order on every rank: the same bytes everywhere, run after run — also for Float64
int transport_allreduce(ah_comm* m, int type, const void* send, void* recv, int64_t count) {
  ah_ctx* c = m->ctx;
  const int w = ah_type_width(type);
  const int64_t nbytes = count * w;
  uint8_t* h;
  int rc = stage_reserve(m, (size_t)nbytes * (size_t)(m->world + 1), &h);
  if (rc != AH_OK) return rc;
  AH_HIP(c, hipMemcpyAsync(h, send, (size_t)nbytes, hipMemcpyDeviceToHost, c->stream));
  AH_HIP(c, hipStreamSynchronize(c->stream));
  const int trc = m->transport.allgather(m->transport.user, h, h + nbytes, nbytes);
  if (trc) return transport_fail(c, "allreduce", trc);
  uint8_t* all = h + nbytes;
  for (int64_t i = 0; i < count; i++) {
    switch (type) {
      case AH_FLOAT64: { double a = 0; for (int r = 0; r < m->world; r++) a += ((const double*)(all + (size_t)r * nbytes))[i]; ((double*)h)[i] = a; break; }
      case AH_FLOAT32: { float a = 0; for (int r = 0; r < m->world; r++) a += ((const float*)(all + (size_t)r * nbytes))[i]; ((float*)h)[i] = a; break; }
      case AH_INT64: case AH_UINT64: { uint64_t a = 0; for (int r = 0; r < m->world; r++) a += ((const uint64_t*)(all + (size_t)r * nbytes))[i]; ((uint64_t*)h)[i] = a; break; }
      default: { uint32_t a = 0; for (int r = 0; r < m->world; r++) a += ((const uint32_t*)(all + (size_t)r * nbytes))[i]; ((uint32_t*)h)[i] = a; break; }
    }
  }
  AH_HIP(c, hipMemcpyAsync(recv, h, (size_t)nbytes, hipMemcpyHostToDevice, c->stream));
  AH_HIP(c, hipStreamSynchronize(c->stream));
  return AH_OK;
}
int transport_alltoallv(ah_comm* m, const void* send, const int64_t* sb, const int64_t* so, void* recv, const int64_t* rb, const int64_t* ro) {
  ah_ctx* c = m->ctx;
  std::vector<int64_t> hso(m->world), hro(m->world);
  int64_t stot = 0, rtot = 0;
  for (int r = 0; r < m->world; r++) { hso[r] = stot; stot += sb[r]; }
  for (int r = 0; r < m->world; r++) { hro[r] = rtot; rtot += rb[r]; }
  uint8_t* h;
  int rc = stage_reserve(m, (size_t)(stot + rtot) + 64, &h);
  if (rc != AH_OK) return rc;
  for (int r = 0; r < m->world; r++)
    if (sb[r] > 0) AH_HIP(c, hipMemcpyAsync(h + hso[r], (const uint8_t*)send + so[r], (size_t)sb[r], hipMemcpyDeviceToHost, c->stream));
  AH_HIP(c, hipStreamSynchronize(c->stream));
  const int trc = m->transport.alltoallv(m->transport.user, h, sb, hso.data(), h + stot, rb, hro.data());
  if (trc) return transport_fail(c, "alltoallv", trc);
  for (int r = 0; r < m->world; r++)
    if (rb[r] > 0 && r != m->rank) AH_HIP(c, hipMemcpyAsync((uint8_t*)recv + ro[r], h + stot + hro[r], (size_t)rb[r], hipMemcpyHostToDevice, c->stream));
  AH_HIP(c, hipStreamSynchronize(c->stream));
  return AH_OK;
}
}  // namespace

AH_EXPORT int ah_comm_init_transport(ah_ctx* c, int rank, int world, const ah_transport* t, ah_comm** out) {
  AH_ENTER(c);
  if (!out) return ah_fail(c, AH_EINVALID, "comm_init_transport: null out pointer");
  *out = nullptr;
  if (world < 1 || rank < 0 || rank >= world || !t || !t->allgather || !t->alltoallv) return ah_fail(c, AH_EINVALID, "comm_init_transport: bad rank / world / callbacks");
  ah_comm* m = (ah_comm*)calloc(1, sizeof(ah_comm));
  if (!m) return ah_fail(c, AH_EINVALID, "comm_init_transport: out of memory");
  m->ctx = c; m->comm = nullptr; m->rank = rank; m->world = world; m->transport = *t; m->has_transport = true;
  *out = m;
  return AH_OK;
}

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
  if (m->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(m->comm);
  if (m->arena) (void)hipFree(m->arena);
  if (m->arena2) (void)hipFree(m->arena2);
  if (m->gather) (void)hipFree(m->gather);
  if (m->stage) (void)hipHostFree(m->stage);
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
  if (m->has_transport) return transport_allreduce(m, type, send, recv, count);
  AH_NCCL(c, g_rccl.AllReduce(send, recv, (size_t)count, t, ncclSum, m->comm, c->stream));
  return AH_OK;
}

AH_EXPORT int ah_comm_allgather(ah_comm* m, const void* send, void* recv, int64_t nbytes_per_rank) {
  if (!m) return AH_EINVALID;
  ah_ctx* c = m->ctx;
  AH_ENTER(c);
  if (nbytes_per_rank < 0 || (nbytes_per_rank > 0 && (!send || !recv))) return ah_fail(c, AH_EINVALID, "allgather: bad buffer / size");
  if (nbytes_per_rank == 0) return AH_OK;
  if (m->has_transport) return transport_allgather(m, send, recv, nbytes_per_rank);
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
  if (m->has_transport) return transport_alltoallv(m, send, send_bytes_host, send_offs_host, recv, recv_bytes_host, recv_offs_host);
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

// ---- owner bucketing of a rank's local groups (C5): two small launches and ONE read-back, whatever the world size --------------------
// (round 3's first version ran a compare, a count and four stream compactions per destination: 6·W launches and W host syncs.)
// The order of the tuples INSIDE a destination's block is whatever the atomics make it — harmless: a rank's keys are distinct,
// the owner's re-aggregation is order-free except for "first tuple of a key", which depends on the order of the source RANKS
// only, and the result is sorted by first row at the end.
namespace {
constexpr int kOwnBlock = 256, kOwnMaxWorld = 1024;
__device__ __forceinline__ unsigned owner_of_key(unsigned long long key, unsigned world) {   // = ah_hash_partition_u64: (hashInt(key) >> 40) mod world
  return (unsigned)((__builtin_bswap64(11400714785074694791ull * key) >> 40) % world);
}
__global__ __launch_bounds__(kOwnBlock) void owner_hist_kernel(const unsigned long long* __restrict__ keys, int64_t g, unsigned world,
                                                                unsigned long long* __restrict__ counts) {
  __shared__ unsigned s_c[kOwnMaxWorld];
  for (unsigned r = threadIdx.x; r < world; r += kOwnBlock) s_c[r] = 0;
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * kOwnBlock;
  for (int64_t i = (int64_t)blockIdx.x * kOwnBlock + threadIdx.x; i < g; i += stride) atomicAdd(&s_c[owner_of_key(keys[i], world)], 1u);
  __syncthreads();
  for (unsigned r = threadIdx.x; r < world; r += kOwnBlock) if (s_c[r]) atomicAdd(&counts[r], (unsigned long long)s_c[r]);
}
// block r = four columns of size[r] values back to back, starting at tuple base[r]
__global__ __launch_bounds__(kOwnBlock) void owner_scatter_kernel(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ sums,
                                                                   const long long* __restrict__ cnts, const long long* __restrict__ firsts, long long row_offset,
                                                                   int64_t g, unsigned world, const long long* __restrict__ base, const long long* __restrict__ size,
                                                                   unsigned long long* __restrict__ cursor, unsigned long long* __restrict__ sendbuf) {
  __shared__ unsigned s_c[kOwnMaxWorld];
  __shared__ unsigned long long s_base[kOwnMaxWorld];
  // one chunk of kOwnBlock · 4 groups per workgroup: count, reserve (one global atomic per owner and workgroup), place
  const int64_t lo = (int64_t)blockIdx.x * kOwnBlock * 4;
  for (unsigned r = threadIdx.x; r < world; r += kOwnBlock) s_c[r] = 0;
  __syncthreads();
  unsigned own[4], rk[4];
  unsigned long long k[4];
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int64_t i = lo + u * kOwnBlock + threadIdx.x;
    own[u] = 0; rk[u] = 0; k[u] = 0;
    if (i < g) { k[u] = keys[i]; own[u] = owner_of_key(k[u], world); rk[u] = atomicAdd(&s_c[own[u]], 1u); }
  }
  __syncthreads();
  for (unsigned r = threadIdx.x; r < world; r += kOwnBlock) s_base[r] = s_c[r] ? atomicAdd(&cursor[r], (unsigned long long)s_c[r]) : 0ull;
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; u++) {
    const int64_t i = lo + u * kOwnBlock + threadIdx.x;
    if (i >= g) continue;
    const unsigned r = own[u];
    const long long n_r = size[r];
    unsigned long long* blk = sendbuf + base[r] * 4;
    const long long e = (long long)s_base[r] + rk[u];
    blk[e] = k[u];
    blk[n_r + e] = sums[i];
    blk[2 * n_r + e] = (unsigned long long)cnts[i];
    blk[3 * n_r + e] = (unsigned long long)(firsts[i] + row_offset);
  }
}
}  // namespace

// ---- C4 / C5 as single calls: what a Go host runs per record-batch shard — no Python in between -----------------
// small host vector ↔ every rank (sizes): through the device for RCCL, directly for a host transport
static int comm_allgather_host(ah_comm* m, const int64_t* mine, int count, int64_t* all /* world × count */) {
  ah_ctx* c = m->ctx;
  const size_t nb = (size_t)count * 8;
  if (m->has_transport) {
    const int trc = m->transport.allgather(m->transport.user, mine, all, (int64_t)nb);
    return trc ? transport_fail(c, "allgather", trc) : AH_OK;
  }
  uint8_t* d = (uint8_t*)&c->dscalars[64];   // the popcount partials area (32 KiB): idle between kernels
  if ((size_t)(m->world + 1) * nb > 4096 * 8) {   // a world × world size table from 64 ranks on: a block of the communicator's own
    int grc = arena_reserve_in(m, &m->gather, &m->gather_bytes, (size_t)(m->world + 1) * nb, &d);
    if (grc != AH_OK) return grc;
  }
  AH_HIP(c, hipMemcpyAsync(d, mine, nb, hipMemcpyHostToDevice, c->stream));
  AH_NCCL(c, g_rccl.AllGather(d, d + nb, nb, ncclUint8, m->comm, c->stream));
  AH_HIP(c, hipMemcpyAsync(all, d + nb, nb * (size_t)m->world, hipMemcpyDeviceToHost, c->stream));
  AH_HIP(c, hipStreamSynchronize(c->stream));
  return AH_OK;
}

AH_EXPORT int ah_comm_cmp_filter_sum_i64(ah_comm* m, int cmpop, const int64_t* x, const uint8_t* valid, int64_t off, int64_t n_local,
                                         int64_t threshold, int64_t* out_sum_host, int64_t* out_count_host) {
  if (!m) return AH_EINVALID;
  ah_ctx* c = m->ctx;
  AH_ENTER(c);
  if (!out_sum_host || !out_count_host) return ah_fail(c, AH_EINVALID, "comm_cmp_filter_sum: null result pointer");
  int64_t* pair = (int64_t*)&c->dscalars[40];   // {sum, count}
  int rc = ah_cmp_filter_sum_i64_dev(c, cmpop, x, valid, off, n_local, threshold, pair);
  if (rc != AH_OK) return rc;
  // wrapping int64 sums are exact in any order: ONE all-reduce of 16 bytes is the path's whole exchange
  if ((rc = ah_comm_allreduce_sum(m, AH_INT64, pair, pair, 2)) != AH_OK) return rc;
  AH_HIP(c, hipMemcpyAsync(&c->pinned[16], pair, 16, hipMemcpyDeviceToHost, c->stream));
  AH_HIP(c, hipStreamSynchronize(c->stream));
  *out_sum_host = (int64_t) * (volatile uint64_t*)&c->pinned[16];
  *out_count_host = (int64_t) * (volatile uint64_t*)&c->pinned[17];
  return AH_OK;
}

AH_EXPORT int ah_comm_cmp_filter_sum_f64(ah_comm* m, int cmpop, const double* x, const uint8_t* valid, int64_t off, int64_t n_local,
                                         double threshold, double* out_sum_host, int64_t* out_count_host) {
  if (!m) return AH_EINVALID;
  ah_ctx* c = m->ctx;
  AH_ENTER(c);
  if (!out_sum_host || !out_count_host) return ah_fail(c, AH_EINVALID, "comm_cmp_filter_sum: null result pointer");
  if (m->world > 768) return ah_fail(c, AH_EINVALID, "comm_cmp_filter_sum: world too large");   // 40 bytes per rank in the 32 KiB area
  double* part = (double*)&c->dscalars[40];     // [40..43] {s, e, bs, be} of ah_ddsum.h, [44] count, [45] this rank's rounded sum (unused)
  int64_t* cnt = (int64_t*)&c->dscalars[44];
  int rc = ah_fused_f64_parts_dev(c, cmpop, x, valid, off, n_local, threshold, (double*)&c->dscalars[45], cnt, part);
  if (rc != AH_OK) return rc;
  // Float64: an all-reduce adds in whatever order the ring runs; every rank gets every rank's UN-ROUNDED accumulator instead
  // (40 bytes each), merges them in rank order in double-double and rounds ONCE — the same bytes on every rank and in every
  // run, and for every world size within 1 ULP of the exact sum over the undivided column like ah_cmp_filter_sum_f64 itself
  // (rounding every rank's sum first would cost up to world ULPs and would turn [1e308 | 1e308 | −1e308] into +inf);
  // non-finite rows follow the extended reals (ah_ddsum.h) however they fall over the ranks
  uint8_t* all = (uint8_t*)&c->dscalars[64];
  if ((rc = ah_comm_allgather(m, part, all, 40)) != AH_OK) return rc;
  std::vector<uint64_t> h((size_t)m->world * 5);
  AH_HIP(c, hipMemcpyAsync(h.data(), all, h.size() * 8, hipMemcpyDeviceToHost, c->stream));
  AH_HIP(c, hipStreamSynchronize(c->stream));
  ah_ddx acc;
  ah_ddx_init(acc);
  int64_t n = 0;
  for (int r = 0; r < m->world; r++) {
    ah_ddx p;
    memcpy(&p, &h[5 * (size_t)r], 32);
    ah_ddx_merge(acc, p);
    n += (int64_t)h[5 * (size_t)r + 4];
  }
  *out_sum_host = ah_ddx_result(acc);
  *out_count_host = n;
  return AH_OK;
}

// C5, plan A of SURVEY.md §8e: every rank has aggregated its own row shard (ah_hash_sum_*); this merges the per-rank groups:
//   1 bucket the local groups by OWNER = (hashInt(key) >> 40) mod world   (ah_hash_partition_u64 + one compare and four stream
//     compactions per destination — on the device)
//   2 ragged all-to-all of {key, sum, count, global first row} tuples: O(groups) bytes on the wire, never O(rows)
//   3 the owner re-aggregates the tuples of its keys: sum of the partial sums (Float64: the reproducible fixed-point sums of
//     ah_hash_sum_f64), sum of the counts, the first row of the FIRST tuple (source ranks arrive in ascending order and lower
//     ranks hold lower row ranges, so that is the smallest global row)
//   4 ragged all-gather of the owners' groups
//   5 ordered by global first row = what unique / dictionary_encode over the undivided column would give
//     (kernels/vector_hash.go:359-385, 721-741: first-seen order)
// All device buffers; out_* hold up to `capacity` groups.  *out_ngroups_host = number of global groups; AH_EINVALID with that
// number set when `capacity` is too small (nothing written).
//
// The NULL group (ah_hash_sum_* reports it as out_null_group: a group whose key slot holds 0 but which is not the key 0) has no key
// to be owned by: it is taken out of the local columns, every rank's null tuple travels in one small all-gather, every rank merges
// them in rank order (Float64: through the same fixed-point re-aggregation as the owners', so the bytes agree on all ranks), and
// the merged group joins the others before the final ordering by first row.  *out_null_group_host = its position, −1 if no rank had one.
AH_EXPORT int ah_comm_merge_groups(ah_comm* m, int is_f64, const uint64_t* keys, const void* sums, const int64_t* counts, const int64_t* first_rows,
                                   int64_t ngroups_local, int32_t null_group_local, int64_t row_offset, int64_t capacity, uint64_t* out_keys,
                                   void* out_sums, int64_t* out_counts, int64_t* out_first_rows, int64_t* out_ngroups_host,
                                   int32_t* out_null_group_host) {
  if (!m) return AH_EINVALID;
  ah_ctx* c = m->ctx;
  AH_ENTER(c);
  if (!out_ngroups_host) return ah_fail(c, AH_EINVALID, "merge_groups: null result pointer");   // (a caller bug every rank shares: the ranks run the same program)
  *out_ngroups_host = 0;
  if (out_null_group_host) *out_null_group_host = -1;
  const int W = m->world;
  int64_t g = ngroups_local;
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  int rc;
  // ---- 0: the null group leaves the local columns; {has one, sum bits, count, global first row, STATUS} of every rank to every rank.
  // Everything that can fail on ONE rank before the first collective — an argument only this rank got wrong, its arena, its copies —
  // is caught here and travels in the status word: the ranks fail TOGETHER after the all-gather instead of one returning early and
  // the others blocking in it.
  int64_t null_mine[5] = {0, 0, 0, 0, 0};
  auto step0 = [&]() -> int {
    if (g < 0 || capacity < 0) return ah_fail(c, AH_EINVALID, "merge_groups: negative count");
    if (g > 0 && (!keys || !sums || !counts || !first_rows)) return ah_fail(c, AH_EINVALID, "merge_groups: null input");
    if (null_group_local < -1 || (int64_t)null_group_local >= g) return ah_fail(c, AH_EINVALID, "merge_groups: null group %d of %lld groups", (int)null_group_local, (long long)g);
    if (null_group_local < 0) return AH_OK;
    const size_t at = (size_t)null_group_local * 8;
    null_mine[0] = 1;
    AH_HIP(c, hipMemcpyAsync(&null_mine[1], (const uint8_t*)sums + at, 8, hipMemcpyDeviceToHost, c->stream));
    AH_HIP(c, hipMemcpyAsync(&null_mine[2], (const uint8_t*)counts + at, 8, hipMemcpyDeviceToHost, c->stream));
    AH_HIP(c, hipMemcpyAsync(&null_mine[3], (const uint8_t*)first_rows + at, 8, hipMemcpyDeviceToHost, c->stream));
    AH_HIP(c, hipStreamSynchronize(c->stream));
    null_mine[3] += row_offset;
    // the other g − 1 groups, closed up, in the communicator's second block (free until step 4)
    uint8_t* cmp;
    int r0;
    if ((r0 = arena2_reserve(m, pad((size_t)g * 8) * 4 + 4096, &cmp)) != AH_OK) return r0;
    const void* cols[4] = {keys, sums, counts, first_rows};
    const void* closed[4];
    const size_t before = (size_t)null_group_local * 8, after = (size_t)(g - 1 - null_group_local) * 8;
    for (int k = 0; k < 4; k++) {
      uint8_t* dst = cmp + (size_t)k * pad((size_t)g * 8);
      if (before && (r0 = ah_copy_async(c, dst, cols[k], before)) != AH_OK) return r0;
      if (after && (r0 = ah_copy_async(c, dst + before, (const uint8_t*)cols[k] + before + 8, after)) != AH_OK) return r0;
      closed[k] = dst;
    }
    keys = (const uint64_t*)closed[0]; sums = closed[1]; counts = (const int64_t*)closed[2]; first_rows = (const int64_t*)closed[3];
    g -= 1;
    return AH_OK;
  };
  const int local_rc = step0();
  if (local_rc != AH_OK) { null_mine[0] = 0; null_mine[4] = local_rc; g = 0; }
  std::vector<int64_t> gathered((size_t)W * 5, 0);
  if (W == 1) memcpy(gathered.data(), null_mine, sizeof(null_mine));
  else if ((rc = comm_allgather_host(m, null_mine, 5, gathered.data())) != AH_OK) return rc;   // (the transport itself failing is every rank's failure)
  if (local_rc != AH_OK) return local_rc;   // this rank's own error, its message already set
  for (int r = 0; r < W; r++)
    if (gathered[(size_t)r * 5 + 4] != 0) return ah_fail(c, (int)gathered[(size_t)r * 5 + 4], "merge_groups: rank %d failed before the exchange (status %d)", r, (int)gathered[(size_t)r * 5 + 4]);
  std::vector<int64_t> null_all((size_t)W * 4, 0);
  for (int r = 0; r < W; r++) memcpy(&null_all[(size_t)r * 4], &gathered[(size_t)r * 5], 4 * sizeof(int64_t));
  int n_null_ranks = 0;
  for (int r = 0; r < W; r++) n_null_ranks += null_all[(size_t)r * 4] != 0;
  const int64_t has_null = n_null_ranks > 0 ? 1 : 0;
  // ---- 1: owners and the send blocks (block r = four columns of g_r values, back to back)
  if (W > kOwnMaxWorld) return ah_fail(c, AH_EINVALID, "merge_groups: world beyond %d", kOwnMaxWorld);
  uint8_t* a0;
  if ((rc = arena_reserve(m, pad((size_t)g * 32) + pad((size_t)W * 8) * 4 + 4096, &a0)) != AH_OK) return rc;
  size_t o = 0;
  auto take = [&](size_t b) { uint8_t* q = a0 + o; o += pad(b); return q; };
  uint8_t* gfirst = take(8);   // (anchor of the block's base for the re-reservation below)
  unsigned long long* d_counts = (unsigned long long*)take((size_t)W * 8);
  unsigned long long* d_cursor = (unsigned long long*)take((size_t)W * 8);
  long long* d_base = (long long*)take((size_t)W * 8);
  long long* d_size = (long long*)take((size_t)W * 8);
  uint8_t* sendbuf = take((size_t)g * 32);
  const size_t phase1 = o;
  std::vector<int64_t> scnt(W, 0);
  if (g > 0) {
    if (W == 1) {
      scnt[0] = g;
      if ((rc = ah_arithmetic_arr_scalar(c, AH_INT64, 0 /*ADD*/, first_rows, &row_offset, sendbuf + (size_t)3 * (size_t)g * 8, g)) != AH_OK) return rc;
      const void* cols[3] = {keys, sums, counts};
      for (int k = 0; k < 3; k++) if ((rc = ah_copy_async(c, sendbuf + (size_t)k * (size_t)g * 8, cols[k], (size_t)g * 8)) != AH_OK) return rc;
    } else {
      AH_HIP(c, hipMemsetAsync(d_counts, 0, pad((size_t)W * 8) * 2, c->stream));   // counts and cursors
      const unsigned hgrid = ah_stream_grid(c, ah_ceil_div(g, kOwnBlock), 4);
      owner_hist_kernel<<<hgrid, kOwnBlock, 0, c->stream>>>((const unsigned long long*)keys, g, (unsigned)W, d_counts);
      AH_LAUNCH_CHECK(c);
      AH_HIP(c, hipMemcpyAsync(scnt.data(), d_counts, (size_t)W * 8, hipMemcpyDeviceToHost, c->stream));
      AH_HIP(c, hipStreamSynchronize(c->stream));
      std::vector<long long> hb(W), hs(W);
      long long run = 0;
      for (int r = 0; r < W; r++) { hb[r] = run; hs[r] = (long long)scnt[r]; run += hs[r]; }
      AH_HIP(c, hipMemcpyAsync(d_base, hb.data(), (size_t)W * 8, hipMemcpyHostToDevice, c->stream));
      AH_HIP(c, hipMemcpyAsync(d_size, hs.data(), (size_t)W * 8, hipMemcpyHostToDevice, c->stream));
      owner_scatter_kernel<<<(unsigned)ah_ceil_div(g, (int64_t)kOwnBlock * 4), kOwnBlock, 0, c->stream>>>(
          (const unsigned long long*)keys, (const unsigned long long*)sums, (const long long*)counts, (const long long*)first_rows, (long long)row_offset, g, (unsigned)W,
          d_base, d_size, d_cursor, (unsigned long long*)sendbuf);
      AH_LAUNCH_CHECK(c);
      AH_HIP(c, hipStreamSynchronize(c->stream));   // hb / hs are stack vectors: the copies above must have read them
    }
  }
  // ---- 2: sizes, then the tuples
  std::vector<int64_t> table((size_t)W * W);
  if (W == 1) table[0] = scnt[0];
  else if ((rc = comm_allgather_host(m, scnt.data(), W, table.data())) != AH_OK) return rc;
  std::vector<int64_t> rcnt(W), sb(W), so(W), rb(W), ro(W);
  int64_t mrecv = 0, acc = 0;
  for (int s = 0; s < W; s++) { rcnt[s] = table[(size_t)s * W + m->rank]; ro[s] = mrecv * 32; rb[s] = rcnt[s] * 32; mrecv += rcnt[s]; }
  for (int r = 0; r < W; r++) { so[r] = acc * 32; sb[r] = scnt[r] * 32; acc += scnt[r]; }
  // the arena grows without keeping its contents: everything after the send blocks is placed in a second reservation that
  // includes them (same base unless it grew — so re-derive the pointers, and copy the send blocks if it moved)
  const size_t need2 = phase1 + pad((size_t)mrecv * 32) + pad((size_t)mrecv * 8) * 4 + pad(((size_t)mrecv + 1) * 8) * 7 + 4096;
  if (need2 > m->arena_bytes) {
    uint8_t* keep = nullptr;
    if (g > 0) {
      AH_HIP(c, hipMalloc((void**)&keep, (size_t)g * 32));
      if (hipMemcpyAsync(keep, sendbuf, (size_t)g * 32, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) {
        (void)hipFree(keep);
        return ah_fail(c, AH_EHIP, "merge_groups: copy failed");
      }
    }
    rc = arena_reserve(m, need2, &a0);
    if (rc == AH_OK && g > 0) {
      sendbuf = a0 + (sendbuf - (uint8_t*)gfirst) + 0;   // same offsets in the new block
      if (hipMemcpyAsync(sendbuf, keep, (size_t)g * 32, hipMemcpyDeviceToDevice, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
        rc = ah_fail(c, AH_EHIP, "merge_groups: copy failed");
    }
    if (keep) (void)hipFree(keep);
    if (rc != AH_OK) return rc;
  }
  o = phase1;
  uint8_t* recvbuf = take((size_t)mrecv * 32);
  if ((rc = ah_comm_alltoallv(m, sendbuf, sb.data(), so.data(), recvbuf, rb.data(), ro.data())) != AH_OK) return rc;
  // ---- 3: the owner's re-aggregation over its mrecv tuples (columns laid end to end, source ranks ascending)
  uint64_t* rk = (uint64_t*)take((size_t)mrecv * 8);
  uint8_t* rs = take((size_t)mrecv * 8);
  int64_t* rc_ = (int64_t*)take((size_t)mrecv * 8);
  int64_t* rf = (int64_t*)take((size_t)mrecv * 8);
  {
    int64_t at = 0;
    for (int s = 0; s < W; s++) {
      const size_t nb = (size_t)rcnt[s] * 8;
      const uint8_t* blk = recvbuf + ro[s];
      if (nb) {
        if ((rc = ah_copy_async(c, (uint8_t*)rk + (size_t)at * 8, blk, nb)) != AH_OK) return rc;
        if ((rc = ah_copy_async(c, rs + (size_t)at * 8, blk + nb, nb)) != AH_OK) return rc;
        if ((rc = ah_copy_async(c, (uint8_t*)rc_ + (size_t)at * 8, blk + 2 * nb, nb)) != AH_OK) return rc;
        if ((rc = ah_copy_async(c, (uint8_t*)rf + (size_t)at * 8, blk + 3 * nb, nb)) != AH_OK) return rc;
      }
      at += rcnt[s];
    }
  }
  uint64_t* ok = (uint64_t*)take(((size_t)mrecv + 1) * 8);
  uint8_t* osum = take(((size_t)mrecv + 1) * 8);
  int64_t* ocnt = (int64_t*)take(((size_t)mrecv + 1) * 8);
  int64_t* ofirst = (int64_t*)take(((size_t)mrecv + 1) * 8);
  uint64_t* ok2 = (uint64_t*)take(((size_t)mrecv + 1) * 8);
  int64_t* csum = (int64_t*)take(((size_t)mrecv + 1) * 8);
  int64_t* ofirst_rows = (int64_t*)take(((size_t)mrecv + 1) * 8);
  int64_t ng = 0, ng2 = 0;
  if (mrecv > 0) {
    if (is_f64) rc = ah_hash_sum_f64(c, rk, nullptr, 0, (const double*)rs, nullptr, 0, mrecv, ok, (double*)osum, ocnt, ofirst, &ng, nullptr);
    else rc = ah_hash_sum_i64(c, rk, nullptr, 0, (const int64_t*)rs, nullptr, 0, mrecv, ok, (int64_t*)osum, ocnt, ofirst, &ng, nullptr);
    if (rc != AH_OK) return rc;
    if ((rc = ah_hash_sum_i64(c, rk, nullptr, 0, rc_, nullptr, 0, mrecv, ok2, csum, ocnt, nullptr, &ng2, nullptr)) != AH_OK) return rc;
    if (ng2 != ng) return ah_fail(c, AH_EINVALID, "merge_groups: internal error (group counts differ)");
    if ((rc = ah_take_primitive(c, 8, rf, nullptr, 0, mrecv, 8, 1, ofirst, nullptr, 0, ng, 1, ofirst_rows, nullptr, nullptr, nullptr)) != AH_OK) return rc;
  }
  // ---- 4: every rank gets every owner's groups
  std::vector<int64_t> gcnt(W);
  const int64_t ng_local = ng;
  if (W == 1) gcnt[0] = ng;
  else if ((rc = comm_allgather_host(m, &ng_local, 1, gcnt.data())) != AH_OK) return rc;
  int64_t G = 0, mx = 0;
  for (int r = 0; r < W; r++) { G += gcnt[r]; mx = gcnt[r] > mx ? gcnt[r] : mx; }
  const int64_t GT = G + has_null;   // with the merged null group
  *out_ngroups_host = GT;
  if (GT > capacity) return ah_fail(c, AH_EINVALID, "merge_groups: %lld groups, the outputs hold %lld", (long long)GT, (long long)capacity);
  if (GT == 0) return AH_OK;
  if (!out_keys || !out_sums || !out_counts || !out_first_rows) return ah_fail(c, AH_EINVALID, "merge_groups: null output");
  // blocks of 4 columns × mx (padded) per rank; then the columns are laid end to end, rank order.  All of it in the communicator's
  // SECOND block: the first still holds the re-aggregated columns
  uint8_t* blk_local;
  if ((rc = arena2_reserve(m, pad((size_t)mx * 32) + pad((size_t)mx * 32 * (size_t)W) + pad((size_t)GT * 8) * 5 + pad((size_t)GT / 8 + 64) + pad((size_t)W * 8) * 8 + 4096, &blk_local)) != AH_OK) return rc;
  uint8_t* b0 = blk_local + pad((size_t)mx * 32);
  if (G > 0) {
    const void* cols[4] = {ok, osum, csum, ofirst_rows};
    for (int k = 0; k < 4 && ng > 0; k++)
      if ((rc = ah_copy_async(c, blk_local + (size_t)k * (size_t)mx * 8, cols[k], (size_t)ng * 8)) != AH_OK) return rc;
    if (W == 1) rc = ah_copy_async(c, b0, blk_local, (size_t)mx * 32);
    else rc = ah_comm_allgather(m, blk_local, b0, mx * 32);
    if (rc != AH_OK) return rc;
  }
  uint8_t* colbase = b0 + pad((size_t)mx * 32 * (size_t)W);
  uint8_t* col[4];
  for (int k = 0; k < 4; k++) col[k] = colbase + (size_t)k * pad((size_t)GT * 8);
  uint64_t* order = (uint64_t*)(colbase + 4 * pad((size_t)GT * 8));
  uint8_t* before_bits = (uint8_t*)order + pad((size_t)GT * 8);
  uint8_t* nullwork = before_bits + pad((size_t)GT / 8 + 64);   // 8 arrays of W words
  if (G > 0) {
    int64_t at = 0;
    for (int r = 0; r < W; r++) {
      const size_t nb = (size_t)gcnt[r] * 8;
      for (int k = 0; k < 4 && nb; k++)
        if ((rc = ah_copy_async(c, col[k] + (size_t)at * 8, b0 + (size_t)r * (size_t)mx * 32 + (size_t)k * (size_t)mx * 8, nb)) != AH_OK) return rc;
      at += gcnt[r];
    }
  }
  // ---- 4b: the merged null group becomes tuple G: {0, Σ sums, Σ counts, the smallest global first row}
  std::vector<int64_t> nk, ns, ncnt;   // (host staging: alive until the synchronisation at the end)
  int64_t null_tuple[4] = {0, 0, 0, 0};
  if (has_null) {
    int64_t first = INT64_MAX, cnt = 0;
    uint64_t isum = 0;
    for (int r = 0; r < W; r++) {
      const int64_t* t = &null_all[(size_t)r * 4];
      if (!t[0]) continue;
      nk.push_back(0); ns.push_back(t[1]); ncnt.push_back(t[2]);
      isum += (uint64_t)t[1];
      cnt += t[2];
      if (t[3] < first) first = t[3];
    }
    null_tuple[1] = (int64_t)isum; null_tuple[2] = cnt; null_tuple[3] = first;
    const size_t wpad = pad((size_t)W * 8);
    for (int k = 0; k < 4; k++)
      if (k != 1 || !is_f64 || n_null_ranks == 1) AH_HIP(c, hipMemcpyAsync(col[k] + (size_t)G * 8, &null_tuple[k], 8, hipMemcpyHostToDevice, c->stream));
    if (is_f64 && n_null_ranks == 1) {
      AH_HIP(c, hipMemcpyAsync(col[1] + (size_t)G * 8, &ns[0], 8, hipMemcpyHostToDevice, c->stream));   // one partial: its own bytes
    } else if (is_f64) {
      // several partial sums: the owners' re-aggregation (ah_hash_sum_f64 over tuples of one key), on every rank alike
      uint64_t* dk = (uint64_t*)nullwork;
      double* dsum = (double*)(nullwork + wpad);
      uint64_t* okk = (uint64_t*)(nullwork + 2 * wpad);
      double* osm = (double*)(nullwork + 3 * wpad);
      int64_t* oc = (int64_t*)(nullwork + 4 * wpad);
      int64_t* of = (int64_t*)(nullwork + 5 * wpad);
      AH_HIP(c, hipMemcpyAsync(dk, nk.data(), nk.size() * 8, hipMemcpyHostToDevice, c->stream));
      AH_HIP(c, hipMemcpyAsync(dsum, ns.data(), ns.size() * 8, hipMemcpyHostToDevice, c->stream));
      int64_t one = 0;
      // (an error return from here on first waits for the copies above: they read this frame's null_tuple / nk / ns)
      auto settle = [&](int code) { (void)hipStreamSynchronize(c->stream); return code; };
      if ((rc = ah_hash_sum_f64(c, dk, nullptr, 0, dsum, nullptr, 0, (int64_t)nk.size(), okk, osm, oc, of, &one, nullptr)) != AH_OK) return settle(rc);
      if (one != 1) return settle(ah_fail(c, AH_EINVALID, "merge_groups: internal error (null group merge)"));
      if ((rc = ah_copy_async(c, col[1] + (size_t)G * 8, osm, 8)) != AH_OK) return settle(rc);
    }
    // its position in first-seen order = the groups first seen before it (first rows are distinct: a row has one key)
    if (G > 0) {
      int64_t pos = 0;
      if ((rc = ah_comparison(c, AH_CMP_GT, AH_SHAPE_SA, AH_INT64, &null_tuple[3], col[3], before_bits, G, 0)) != AH_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
      if ((rc = ah_count_set_bits(c, before_bits, 0, G, &pos)) != AH_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
      if (out_null_group_host) *out_null_group_host = (int32_t)pos;
    } else if (out_null_group_host) {
      *out_null_group_host = 0;
    }
  }
  // ---- 5: global first-seen order
  if ((rc = ah_sort_indices(c, AH_INT64, col[3], nullptr, 0, GT, 0, 0, order)) != AH_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
  void* outs[4] = {out_keys, out_sums, out_counts, out_first_rows};
  for (int k = 0; k < 4; k++)
    if ((rc = ah_take_primitive(c, 8, col[k], nullptr, 0, GT, 8, 0, order, nullptr, 0, GT, 0, outs[k], nullptr, nullptr, nullptr)) != AH_OK) { (void)hipStreamSynchronize(c->stream); return rc; }
  AH_HIP(c, hipStreamSynchronize(c->stream));
  return AH_OK;
}
