// ah_ctx.hip — context, device buffers, pinned staging, copies on a side stream.
//
// Reference counterpart: there is none on device; on the host the analogous pieces
// are memory.Allocator (arrow/memory/allocator.go:23-27, 64-byte alignment :20) and
// memory.Set (arrow/memory/_lib/memory.c:20-27).  A Go `memory.Allocator` backed by
// ah_host_alloc_pinned gives Arrow buffers that hipMemcpyAsync can DMA directly.
#include <chrono>
#include "ah_common.h"

static const char* kVersion = "arrowhip 0.1 (gfx950)";

AH_EXPORT const char* ah_version(void) { return kVersion; }

AH_EXPORT int ah_device_count(int* n_host) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (n_host) *n_host = (e == hipSuccess) ? n : 0;
  return e == hipSuccess ? AH_OK : AH_EHIP;
}

static int ctx_init_common(ah_ctx* c) {
  AH_HIP(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
  AH_HIP(c, hipEventCreateWithFlags(&c->ev_copy, hipEventDisableTiming));
  AH_HIP(c, hipEventCreateWithFlags(&c->ev_compute, hipEventDisableTiming));
  AH_HIP(c, hipEventCreate(&c->t0));
  AH_HIP(c, hipEventCreate(&c->t1));
  AH_HIP(c, hipHostMalloc((void**)&c->pinned, 64 * sizeof(uint64_t), hipHostMallocDefault));
  AH_HIP(c, hipMalloc((void**)&c->dscalars, (64 + 4096) * sizeof(uint64_t)));
  AH_HIP(c, hipMemset(c->dscalars, 0, (64 + 4096) * sizeof(uint64_t)));   // (words [33] … [36] are kept zero between calls by the kernels that use them: ah_encode_first_look, e2_offs2_kernel)
  AH_HIP(c, hipHostMalloc((void**)&c->mailbox, 256, hipHostMallocCoherent | hipHostMallocMapped));   // [0..7] filter_count, [8..15] ah_mailbox_*, [16] stall reports (ah_scan.hip)
  memset(c->mailbox, 0, 256);
  hipDeviceProp_t prop;
  AH_HIP(c, hipGetDeviceProperties(&prop, c->device));
  c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  const char* e_nt = getenv("ARROWHIP_NT");
  c->tune_nt = e_nt ? atoi(e_nt) : 1;
  const char* e_bpc = getenv("ARROWHIP_BLOCKS_PER_CU");
  c->tune_blocks_per_cu = e_bpc ? atoi(e_bpc) : 0;  // 0 = each kernel's own default
  auto env_int = [](const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; };
  c->opt_take_binned = env_int("ARROWHIP_TAKE_BINNED", 1);
  c->opt_take_window_log2 = env_int("ARROWHIP_TAKE_WINDOW_LOG2", 22);
  c->opt_take_gather_wg = env_int("ARROWHIP_TAKE_GATHER_WG_PER_CU", 8);
  c->opt_groupby_partition = env_int("ARROWHIP_GROUPBY_PARTITION", 1);
  c->opt_groupby_keys = env_int("ARROWHIP_GROUPBY_KEYS", 1280);
  c->opt_hash_direct = env_int("ARROWHIP_HASH_DIRECT", 2);
  c->opt_encode_partition = env_int("ARROWHIP_ENCODE_PARTITION", 1);
  c->opt_encode_part_min = env_int("ARROWHIP_ENCODE_PART_MIN", 300000);
  c->opt_encode_early_look = env_int("ARROWHIP_ENCODE_EARLY_LOOK", 1);
  c->opt_encode_byte_map = env_int("ARROWHIP_ENCODE_BYTE_MAP", 1);
  c->opt_encode_part_slots = env_int("ARROWHIP_ENCODE_PART_SLOTS", 8192);
  c->opt_sort_msd = env_int("ARROWHIP_SORT_MSD", 1);
  c->opt_scan_segment_log2 = env_int("ARROWHIP_SCAN_SEGMENT_LOG2", 0);   // 0 = one segment: segments measured slower
  c->opt_take_gather_lds = env_int("ARROWHIP_TAKE_GATHER_LDS", 1);   // 1: the window's validity bits in LDS (ah_take_binned.hip)
  c->opt_take_vec_nt = env_int("ARROWHIP_TAKE_VEC_NT", 7);
  c->opt_encode_unperm_group = env_int("ARROWHIP_ENCODE_UNPERM_GROUP", 4);
  c->opt_encode_table_batch = env_int("ARROWHIP_ENCODE_TABLE_BATCH", 1);
  c->opt_encode_dict_compact = env_int("ARROWHIP_ENCODE_DICT_COMPACT", 1);
  c->opt_encode_resolve_wgs = env_int("ARROWHIP_ENCODE_RESOLVE_WGS", 1024);
  c->opt_encode_unperm2_group = env_int("ARROWHIP_ENCODE_UNPERM2_GROUP", 4);
  c->opt_groupby_scale_guess = env_int("ARROWHIP_GROUPBY_SCALE_GUESS", 1);
  c->opt_arith_xcd_map = env_int("ARROWHIP_ARITH_XCD_MAP", 0);
  c->opt_take_vec = env_int("ARROWHIP_TAKE_VEC", 1);   // 0: one row per lane always, 1: V rows per lane when the sample says clustered, 2: always
  c->opt_take_gather_load = env_int("ARROWHIP_TAKE_GATHER_LOAD", 0);   // 0 plain, 1 nontemporal, 2 L1-bypassing (sc1)
  return AH_OK;
}

static int ctx_create(int device_id, void* stream, bool borrow, ah_ctx** out) {
  if (!out) return AH_EINVALID;
  *out = nullptr;
  ah_ctx* c = (ah_ctx*)calloc(1, sizeof(ah_ctx));
  if (!c) return AH_EINVALID;
  c->device = device_id;
  hipError_t e = hipSetDevice(device_id);
  if (e != hipSuccess) { free(c); return AH_EHIP; }
  if (borrow) {
    c->stream = (hipStream_t)stream;
    c->owns_stream = false;
  } else {
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { free(c); return AH_EHIP; }
    c->owns_stream = true;
  }
  int rc = ctx_init_common(c);
  if (rc != AH_OK) { fprintf(stderr, "arrowhip: ctx init failed: %s\n", c->err); free(c); return rc; }
  // ah_filter_count leaves its tile prefixes for the fill that follows; only this context's own entry points drop them.  On a
  // stream shared with another producer (torch, a second ah_ctx) a foreign kernel may rewrite the mask in between, so there the
  // fill always recounts unless the caller vouches for the mask (ah_ctx_set_option "filter_cache" 1)
  c->opt_filter_cache = c->owns_stream ? 1 : 0;
  c->opt_take_hint_cache = c->owns_stream ? 1 : 0;   // (same rule: on a shared stream another producer may rewrite the index vector between two calls)
  c->opt_groupby_lean = 1;
  c->opt_scan_onepass = 1;
  c->opt_groupby_seed = 1;
  c->opt_groupby_reserve = 1;
  *out = c;
  return AH_OK;
}

AH_EXPORT int ah_ctx_create(int device_id, ah_ctx** out) { return ctx_create(device_id, nullptr, false, out); }
AH_EXPORT int ah_ctx_create_on_stream(int device_id, void* hip_stream, ah_ctx** out) {
  return ctx_create(device_id, hip_stream, true, out);
}

AH_EXPORT void ah_ctx_destroy(ah_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  (void)hipStreamSynchronize(c->copy_stream);
  ah_expr_cache_free(c);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->pinned) (void)hipHostFree(c->pinned);
  if (c->dscalars) (void)hipFree(c->dscalars);
  if (c->mailbox) (void)hipHostFree(c->mailbox);
  if (c->fcache.buf) (void)hipFree(c->fcache.buf);
  if (c->scan_recs) (void)hipFree(c->scan_recs);
  if (c->temp) (void)hipFree(c->temp);
  (void)hipEventDestroy(c->ev_copy);
  (void)hipEventDestroy(c->ev_compute);
  (void)hipEventDestroy(c->t0);
  (void)hipEventDestroy(c->t1);
  for (int i = 0; i < c->n_marks; i++) if (c->marks[i]) (void)hipEventDestroy(c->marks[i]);
  free(c->marks);
  (void)hipStreamDestroy(c->copy_stream);
  if (c->owns_stream) (void)hipStreamDestroy(c->stream);
  free(c);
}

// measurement / test switches of one context (none of them changes a result; DESIGN.md §6 lists them)
AH_EXPORT int ah_ctx_set_option(ah_ctx* c, const char* name, int64_t value) {
  AH_ENTER(c);
  if (!name) return ah_fail(c, AH_EINVALID, "set_option: null name");
  if (!strcmp(name, "nt")) c->tune_nt = (int)value;
  else if (!strcmp(name, "blocks_per_cu")) c->tune_blocks_per_cu = (int)value;
  else if (!strcmp(name, "take_binned")) c->opt_take_binned = (int)value;
  else if (!strcmp(name, "take_window_log2")) c->opt_take_window_log2 = (int)value;
  else if (!strcmp(name, "take_gather_wg_per_cu")) c->opt_take_gather_wg = (int)value;
  else if (!strcmp(name, "take_gather_lds")) c->opt_take_gather_lds = (int)value;
  else if (!strcmp(name, "take_vec")) c->opt_take_vec = (int)value;
  else if (!strcmp(name, "take_hint_cache")) { c->opt_take_hint_cache = (int)value; c->take_hint_valid = false; }
  else if (!strcmp(name, "take_vec_nt")) c->opt_take_vec_nt = (int)value;
  else if (!strcmp(name, "encode_unperm_group")) c->opt_encode_unperm_group = (int)value;
  else if (!strcmp(name, "encode_table_batch")) c->opt_encode_table_batch = (int)value;
  else if (!strcmp(name, "encode_dict_compact")) c->opt_encode_dict_compact = (int)value;
  else if (!strcmp(name, "encode_unperm2_group")) c->opt_encode_unperm2_group = (int)value;
  else if (!strcmp(name, "encode_resolve_wgs")) c->opt_encode_resolve_wgs = value < 1 ? 1 : (int)value;
  else if (!strcmp(name, "groupby_scale_guess")) c->opt_groupby_scale_guess = (int)value;
  else if (!strcmp(name, "arith_xcd_map")) c->opt_arith_xcd_map = (int)value;
  else if (!strcmp(name, "take_gather_load")) c->opt_take_gather_load = (int)value;
  else if (!strcmp(name, "groupby_partition")) c->opt_groupby_partition = (int)value;
  else if (!strcmp(name, "groupby_keys")) c->opt_groupby_keys = (int)value;
  else if (!strcmp(name, "hash_direct")) c->opt_hash_direct = (int)value;
  else if (!strcmp(name, "encode_partition")) c->opt_encode_partition = (int)value;
  else if (!strcmp(name, "encode_part_min")) c->opt_encode_part_min = (int)value;
  else if (!strcmp(name, "encode_early_look")) c->opt_encode_early_look = (int)value;
  else if (!strcmp(name, "encode_byte_map")) c->opt_encode_byte_map = (int)value;
  else if (!strcmp(name, "encode_part_slots")) c->opt_encode_part_slots = (int)value;
  else if (!strcmp(name, "sort_msd")) c->opt_sort_msd = (int)value;
  else if (!strcmp(name, "scan_onepass")) c->opt_scan_onepass = (int)value;
  else if (!strcmp(name, "groupby_lean")) c->opt_groupby_lean = (int)value;
  else if (!strcmp(name, "groupby_seed")) c->opt_groupby_seed = (int)value;
  else if (!strcmp(name, "groupby_reserve")) c->opt_groupby_reserve = (int)value;
  else if (!strcmp(name, "filter_cache")) { c->opt_filter_cache = value != 0; if (!value) c->fcache.valid = false; }
  else if (!strcmp(name, "scan_segment_log2")) c->opt_scan_segment_log2 = (int)value;
  else return ah_fail(c, AH_EINVALID, "set_option: unknown option '%s'", name);
  return AH_OK;
}

AH_EXPORT const char* ah_last_error(ah_ctx* c) { return c ? c->err : "null context"; }

// ---- a few device words the host must see before it can go on: posted to coherent pinned memory by a one-thread kernel and polled,
// instead of hipMemcpyAsync + hipStreamSynchronize (whose wake-up costs ≈ 15 µs: more than many of the passes it sits between)
namespace {
__global__ void mailbox_post_kernel(const unsigned long long* __restrict__ src, int nwords, const unsigned long long* __restrict__ src2, int nwords2,
                                    unsigned long long* mailbox, unsigned long long seq) {
  for (int i = 0; i < nwords; i++) __hip_atomic_store(&mailbox[i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  for (int i = 0; i < nwords2; i++) __hip_atomic_store(&mailbox[nwords + i], src2[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  __hip_atomic_store(&mailbox[7], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace
int ah_mailbox_read(ah_ctx* c, const unsigned long long* dev_words, int nwords, unsigned long long* out_host) {
  return ah_mailbox_read2(c, dev_words, nwords, nullptr, 0, out_host);
}
// A kernel that knows its own results posts them itself (ah_mailbox_post, ah_common.h) as its last act — no post kernel behind it:
// ah_mailbox_begin hands out the words and the sequence number to pass to that kernel, ah_mailbox_wait polls for them.
int ah_mailbox_begin(ah_ctx* c, unsigned long long** mb_out, unsigned long long* seq_out) {
  if (c->capturing) { c->capturing = 2; return ah_fail(c, AH_EINVALID, "this call returns a value to the host: it cannot be recorded into a graph"); }
  *mb_out = c->mailbox + 8;
  *seq_out = ++c->mailbox_seq;
  return AH_OK;
}
int ah_mailbox_wait(ah_ctx* c, unsigned long long seq, int nwords, unsigned long long* out_host) {
  if (nwords < 1 || nwords > 7) return ah_fail(c, AH_EINVALID, "mailbox_wait: 1..7 words");
  unsigned long long* mb = c->mailbox + 8;
  bool seen = false;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0;; spin++) {
    if (__atomic_load_n(&mb[7], __ATOMIC_ACQUIRE) == seq) { seen = true; break; }
    __builtin_ia32_pause();
    if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
  }
  if (!seen) {
    AH_HIP(c, hipStreamSynchronize(c->stream));
    if (__atomic_load_n(&mb[7], __ATOMIC_ACQUIRE) != seq) return ah_fail(c, AH_EHIP, "mailbox_wait: the kernel did not report");
  }
  for (int i = 0; i < nwords; i++) out_host[i] = __atomic_load_n(&mb[i], __ATOMIC_RELAXED);
  return AH_OK;
}
// words from two places (a count in one arena, a flag in another) in one post
int ah_mailbox_read2(ah_ctx* c, const unsigned long long* dev_words, int nwords, const unsigned long long* dev_words2, int nwords2,
                     unsigned long long* out_host) {
  if (nwords < 1 || nwords2 < 0 || nwords + nwords2 > 7) return ah_fail(c, AH_EINVALID, "mailbox_read: 1..7 words");
  if (c->capturing) { c->capturing = 2; return ah_fail(c, AH_EINVALID, "this call returns a value to the host: it cannot be recorded into a graph"); }
  unsigned long long* mb = c->mailbox + 8;   // the second half: the first belongs to ah_filter_count
  const unsigned long long seq = ++c->mailbox_seq;
  mailbox_post_kernel<<<1, 1, 0, c->stream>>>(dev_words, nwords, dev_words2, nwords2, mb, seq);
  AH_LAUNCH_CHECK(c);
  bool seen = false;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0;; spin++) {
    if (__atomic_load_n(&mb[7], __ATOMIC_ACQUIRE) == seq) { seen = true; break; }
    __builtin_ia32_pause();
    if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
  }
  if (!seen) {   // a long queue in front of the post, or a fault: the synchronisation tells which
    AH_HIP(c, hipStreamSynchronize(c->stream));
    if (__atomic_load_n(&mb[7], __ATOMIC_ACQUIRE) != seq) return ah_fail(c, AH_EHIP, "mailbox_read: the post kernel did not report");
  }
  for (int i = 0; i < nwords + nwords2; i++) out_host[i] = __atomic_load_n(&mb[i], __ATOMIC_RELAXED);
  return AH_OK;
}

int ah_scratch_reserve(ah_ctx* c, size_t nbytes, void** out) {
  if (nbytes > c->scratch_bytes) {
    // the old block may still be in use by enqueued kernels
    AH_HIP(c, hipStreamSynchronize(c->stream));
    if (c->scratch) AH_HIP(c, hipFree(c->scratch));
    c->scratch = nullptr;
    c->scratch_bytes = 0;
    size_t want = (nbytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    AH_HIP(c, hipMalloc(&c->scratch, want));
    c->scratch_bytes = want;
  }
  *out = c->scratch;
  return AH_OK;
}

int ah_temp_reserve(ah_ctx* c, size_t nbytes, void** out) {
  if (nbytes > c->temp_bytes) {
    AH_HIP(c, hipStreamSynchronize(c->stream));  // the old block may still be in use by enqueued kernels
    if (c->temp) AH_HIP(c, hipFree(c->temp));
    c->temp = nullptr;
    c->temp_bytes = 0;
    size_t want = (nbytes + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    if (hipMalloc(&c->temp, want) != hipSuccess) { (void)hipGetLastError(); return ah_fail(c, AH_EHIP, "out of device memory (%zu bytes of temporaries)", want); }
    c->temp_bytes = want;
  }
  *out = c->temp;
  return AH_OK;
}

AH_EXPORT int ah_buf_alloc(ah_ctx* c, size_t nbytes, void** dptr_host) {
  AH_ENTER_KEEP(c);
  if (!dptr_host) return ah_fail(c, AH_EINVALID, "ah_buf_alloc: null out pointer");
  *dptr_host = nullptr;
  if (nbytes == 0) nbytes = 1;
  // keep the 64-byte padding rule of arrow/memory/buffer.go:137-146 so 16-byte
  // vector accesses on the last elements stay in bounds
  nbytes = (nbytes + 63) & ~(size_t)63;
  AH_HIP(c, hipMalloc(dptr_host, nbytes));
  return AH_OK;
}

AH_EXPORT int ah_buf_free(ah_ctx* c, void* dptr) {
  AH_ENTER(c);
  if (dptr) AH_HIP(c, hipFree(dptr));
  return AH_OK;
}

AH_EXPORT int ah_host_alloc_pinned(ah_ctx* c, size_t nbytes, void** hptr_host) {
  AH_ENTER_KEEP(c);
  if (!hptr_host) return ah_fail(c, AH_EINVALID, "ah_host_alloc_pinned: null out pointer");
  if (nbytes == 0) nbytes = 1;
  nbytes = (nbytes + 63) & ~(size_t)63;
  AH_HIP(c, hipHostMalloc(hptr_host, nbytes, hipHostMallocDefault));
  return AH_OK;
}

AH_EXPORT int ah_host_free_pinned(ah_ctx* c, void* hptr) {
  AH_ENTER_KEEP(c);
  if (hptr) AH_HIP(c, hipHostFree(hptr));
  return AH_OK;
}

AH_EXPORT int ah_upload_async(ah_ctx* c, void* dptr, const void* hptr, size_t nbytes) {
  AH_ENTER_KEEP(c);
  if (ah_fcache_overlaps(c, dptr, nbytes)) c->fcache.valid = false;
  if (ah_take_hint_overlaps(c, dptr, nbytes)) c->take_hint_valid = false;
  if (nbytes == 0) return AH_OK;
  // the copy must not overtake compute work that still reads/writes dptr
  AH_HIP(c, hipEventRecord(c->ev_compute, c->stream));
  AH_HIP(c, hipStreamWaitEvent(c->copy_stream, c->ev_compute, 0));
  AH_HIP(c, hipMemcpyAsync(dptr, hptr, nbytes, hipMemcpyHostToDevice, c->copy_stream));
  AH_HIP(c, hipEventRecord(c->ev_copy, c->copy_stream));
  AH_HIP(c, hipStreamWaitEvent(c->stream, c->ev_copy, 0));
  return AH_OK;
}

AH_EXPORT int ah_download_async(ah_ctx* c, void* hptr, const void* dptr, size_t nbytes) {
  AH_ENTER_KEEP(c);
  if (nbytes == 0) return AH_OK;
  AH_HIP(c, hipEventRecord(c->ev_compute, c->stream));
  AH_HIP(c, hipStreamWaitEvent(c->copy_stream, c->ev_compute, 0));
  AH_HIP(c, hipMemcpyAsync(hptr, dptr, nbytes, hipMemcpyDeviceToHost, c->copy_stream));
  AH_HIP(c, hipEventRecord(c->ev_copy, c->copy_stream));
  AH_HIP(c, hipStreamWaitEvent(c->stream, c->ev_copy, 0));
  return AH_OK;
}

AH_EXPORT int ah_memset_async(ah_ctx* c, void* dptr, int byte_value, size_t nbytes) {
  AH_ENTER_KEEP(c);
  if (ah_fcache_overlaps(c, dptr, nbytes)) c->fcache.valid = false;
  if (ah_take_hint_overlaps(c, dptr, nbytes)) c->take_hint_valid = false;
  if (nbytes == 0) return AH_OK;
  AH_HIP(c, hipMemsetAsync(dptr, byte_value, nbytes, c->stream));
  return AH_OK;
}

namespace {
// Zero a byte range in ONE launch.  hipMemsetAsync splits a range that is not a multiple of its vector width into two fill kernels
// (bulk + tail): behind a host wait — the fill of a two-phase Filter, whose output validity is sized by the count — each of them
// is a launch the stream sits idle for (≈ 5 µs a piece in the kernel trace, profiles/r05_filter_cache_timeline.txt).
typedef unsigned zero_v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void zero_bytes_kernel(uint8_t* __restrict__ p, size_t n) {
  const size_t head = ((16 - ((uintptr_t)p & 15)) & 15) < n ? ((16 - ((uintptr_t)p & 15)) & 15) : n;
  const size_t body = (n - head) / 16, tail0 = head + body * 16;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < head) p[i] = 0;
  if (i < n - tail0) p[tail0 + i] = 0;
  zero_v4* q = (zero_v4*)(p + head);
  const zero_v4 z = {0u, 0u, 0u, 0u};
  for (size_t j = i; j < body; j += (size_t)gridDim.x * 256) q[j] = z;
}
}  // namespace

int ah_zero_bytes(ah_ctx* c, void* dptr, size_t nbytes) {
  if (nbytes == 0) return AH_OK;
  const size_t want = (nbytes / 16 + 255) / 256 + 1;
  zero_bytes_kernel<<<(unsigned)(want < 4096 ? want : 4096), 256, 0, c->stream>>>((uint8_t*)dptr, nbytes);
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

namespace {
// Device-to-device copy written like the streaming kernels of this library (16 bytes per lane and access, nontemporal, four
// accesses in flight per lane, exact grid): hipMemcpyDtoD reaches 4.8–5.1 TB/s of read + write on this part, this loop the
// same ≈ 6.3 TB/s plateau as the Add — so it is both what Concatenate moves chunks with and the measured streaming ceiling
// bench.py reports beside the 8 TB/s specification (`roofline.measured_copy_GB/s`).
constexpr int kCopyBlock = 256, kCopyUnroll = 4;
typedef unsigned copy_v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(kCopyBlock) void copy16_kernel(const copy_v4* __restrict__ src, copy_v4* __restrict__ dst, int64_t nvec) {
  const int64_t i = (int64_t)blockIdx.x * kCopyBlock * kCopyUnroll + threadIdx.x;
  if (i + (int64_t)(kCopyUnroll - 1) * kCopyBlock < nvec) {
    copy_v4 x[kCopyUnroll];
#pragma unroll
    for (int k = 0; k < kCopyUnroll; k++) x[k] = __builtin_nontemporal_load(&src[i + (int64_t)k * kCopyBlock]);
#pragma unroll
    for (int k = 0; k < kCopyUnroll; k++) __builtin_nontemporal_store(x[k], &dst[i + (int64_t)k * kCopyBlock]);
  } else {
#pragma unroll
    for (int k = 0; k < kCopyUnroll; k++) {
      const int64_t j = i + (int64_t)k * kCopyBlock;
      if (j < nvec) dst[j] = src[j];
    }
  }
}
}  // namespace

AH_EXPORT int ah_copy_async(ah_ctx* c, void* dst, const void* src, size_t nbytes) {
  AH_ENTER_KEEP(c);
  if (ah_fcache_overlaps(c, dst, nbytes)) c->fcache.valid = false;
  if (ah_take_hint_overlaps(c, dst, nbytes)) c->take_hint_valid = false;
  if (nbytes == 0) return AH_OK;
  if (!dst || !src) return ah_fail(c, AH_EINVALID, "copy: null buffer");
  const uintptr_t d = (uintptr_t)dst, s = (uintptr_t)src;
  const bool disjoint = d + nbytes <= s || s + nbytes <= d;
  if (nbytes >= ((size_t)1 << 20) && ((d | s) & 15) == 0 && disjoint) {
    const int64_t nvec = (int64_t)(nbytes / 16);
    copy16_kernel<<<(unsigned)ah_ceil_div(nvec, (int64_t)kCopyBlock * kCopyUnroll), kCopyBlock, 0, c->stream>>>((const copy_v4*)src, (copy_v4*)dst, nvec);
    AH_LAUNCH_CHECK(c);
    const size_t done = (size_t)nvec * 16;
    if (done < nbytes) AH_HIP(c, hipMemcpyAsync((uint8_t*)dst + done, (const uint8_t*)src + done, nbytes - done, hipMemcpyDeviceToDevice, c->stream));
    return AH_OK;
  }
  AH_HIP(c, hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, c->stream));
  return AH_OK;
}

// a kernel that gave up waiting for another workgroup (the one-pass scans' bounded look-back) says so in host-coherent memory
int ah_check_stall(ah_ctx* c) {
  if (!c->mailbox || !__atomic_load_n(&c->mailbox[16], __ATOMIC_ACQUIRE)) return AH_OK;
  __atomic_store_n(&c->mailbox[16], 0ull, __ATOMIC_RELAXED);
  return ah_fail(c, AH_EHIP, "cumulative_sum: a tile's look-back gave up waiting for a predecessor (results of that call are invalid)");
}

AH_EXPORT int ah_sync(ah_ctx* c) {
  AH_ENTER_KEEP(c);
  AH_HIP(c, hipStreamSynchronize(c->copy_stream));
  AH_HIP(c, hipStreamSynchronize(c->stream));
  return ah_check_stall(c);
}

AH_EXPORT int ah_timer_start(ah_ctx* c) {
  AH_ENTER_KEEP(c);
  AH_HIP(c, hipEventRecord(c->t0, c->stream));
  return AH_OK;
}

AH_EXPORT int ah_timer_stop(ah_ctx* c, float* ms_host) {
  AH_ENTER_KEEP(c);
  AH_HIP(c, hipEventRecord(c->t1, c->stream));
  AH_HIP(c, hipEventSynchronize(c->t1));
  float ms = 0.f;
  AH_HIP(c, hipEventElapsedTime(&ms, c->t0, c->t1));
  if (ms_host) *ms_host = ms;
  return AH_OK;
}

// numbered event slots on the compute stream: lets a harness bracket individual
// kernels inside a longer timed region without synchronising in between
AH_EXPORT int ah_event_record(ah_ctx* c, int slot) {
  AH_ENTER_KEEP(c);
  if (slot < 0 || slot >= (1 << 16)) return ah_fail(c, AH_EINVALID, "event_record: slot out of range");
  if (slot >= c->n_marks) {
    int n = c->n_marks ? c->n_marks : 64;
    while (n <= slot) n *= 2;
    hipEvent_t* m = (hipEvent_t*)realloc(c->marks, (size_t)n * sizeof(hipEvent_t));
    if (!m) return ah_fail(c, AH_EINVALID, "event_record: out of memory");
    for (int i = c->n_marks; i < n; i++) m[i] = nullptr;
    c->marks = m;
    c->n_marks = n;
  }
  if (!c->marks[slot]) AH_HIP(c, hipEventCreate(&c->marks[slot]));
  AH_HIP(c, hipEventRecord(c->marks[slot], c->stream));
  return AH_OK;
}

AH_EXPORT int ah_event_elapsed_ms(ah_ctx* c, int slot_a, int slot_b, float* ms_host) {
  AH_ENTER_KEEP(c);
  if (slot_a < 0 || slot_b < 0 || slot_a >= c->n_marks || slot_b >= c->n_marks || !c->marks[slot_a] || !c->marks[slot_b])
    return ah_fail(c, AH_EINVALID, "event_elapsed_ms: slot never recorded");
  AH_HIP(c, hipEventSynchronize(c->marks[slot_b]));
  float ms = 0.f;
  AH_HIP(c, hipEventElapsedTime(&ms, c->marks[slot_a], c->marks[slot_b]));
  if (ms_host) *ms_host = ms;
  return AH_OK;
}

// ArrowDeviceArray.sync_event for ARROW_DEVICE_ROCM is a hipEvent_t* (arrow/cdata/abi.h:104-128)
AH_EXPORT int ah_wait_event(ah_ctx* c, void* hip_event_ptr) {
  AH_ENTER_KEEP(c);
  if (!hip_event_ptr) return AH_OK;
  AH_HIP(c, hipStreamWaitEvent(c->stream, *(hipEvent_t*)hip_event_ptr, 0));
  return AH_OK;
}

AH_EXPORT int ah_device_id(ah_ctx* c) { return c ? c->device : -1; }


// ---- hipGraph capture of a sequence of calls ---------------------------------------------------------------------------------
// Small columns and chains of cheap kernels are LAUNCH-bound: a bitmap AND over 2^27 rows is 9 µs, most of it launch latency, and
// an Add → Compare → fused-sum chain over 2^16 rows spends more time between its kernels than in them.  Between ah_graph_begin
// and ah_graph_end the context's compute stream is in capture mode: the entry points called there record their launches (and
// memsets / device copies) into a graph instead of running them; ah_graph_launch replays the whole sequence with ONE submission.
// Capturable are the calls that neither wait for the device nor (re)allocate: the element-wise kernels, comparisons, bitmap ops,
// the *_dev flavours (sum, fused compare-filter-sum, filter, take), cumulative_sum without a null count — after one eager
// warm-up call of the same sequence, so that the scratch arenas have their size.  A call that would have to wait (anything with
// a *_host result) fails with the runtime's capture error and invalidates the capture; ah_graph_end then reports it.
struct ah_graph {
  ah_ctx* ctx;
  hipGraph_t graph;
  hipGraphExec_t exec;
  void* scratch;   // the arenas the recorded launches point into: a later call that grows one frees the block the graph still names
  void* temp;
};

AH_EXPORT int ah_graph_begin(ah_ctx* c) {
  AH_ENTER(c);
  if (c->capturing) return ah_fail(c, AH_EINVALID, "graph_begin: already capturing");
  // relaxed: begin, the recorded calls and end may each run on another OS thread (a goroutine migrates); the context serialises them
  AH_HIP(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed));
  c->capturing = 1;
  return AH_OK;
}

AH_EXPORT int ah_graph_end(ah_ctx* c, ah_graph** out) {
  if (!c) return AH_EINVALID;
  c->err[0] = 0;
  if (!out) return ah_fail(c, AH_EINVALID, "graph_end: null out pointer");
  *out = nullptr;
  if (!c->capturing) return ah_fail(c, AH_EINVALID, "graph_end: not capturing");
  const bool poisoned = c->capturing == 2;
  c->capturing = 0;
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(c->stream, &g);
  if (e == hipSuccess && g && poisoned) {
    (void)hipGraphDestroy(g);
    return ah_fail(c, AH_EINVALID, "graph_end: a recorded call needed a value on the host — nothing was run, the capture is dropped");
  }
  if (e != hipSuccess || !g) {
    (void)hipGetLastError();
    return ah_fail(c, AH_EHIP, "graph_end: the capture was invalidated (%s) — a captured call waited for the device or allocated", hipGetErrorString(e));
  }
  hipGraphExec_t x = nullptr;
  e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
  if (e != hipSuccess) { (void)hipGraphDestroy(g); return ah_fail(c, AH_EHIP, "graph_end: hipGraphInstantiate failed: %s", hipGetErrorString(e)); }
  ah_graph* r = (ah_graph*)calloc(1, sizeof(ah_graph));
  if (!r) { (void)hipGraphExecDestroy(x); (void)hipGraphDestroy(g); return ah_fail(c, AH_EINVALID, "graph_end: out of memory"); }
  r->ctx = c; r->graph = g; r->exec = x;
  r->scratch = c->scratch; r->temp = c->temp;
  *out = r;
  return AH_OK;
}

AH_EXPORT int ah_graph_launch(ah_ctx* c, ah_graph* g) {
  AH_ENTER(c);
  if (!g || g->ctx != c) return ah_fail(c, AH_EINVALID, "graph_launch: not a graph of this context");
  if (c->capturing) return ah_fail(c, AH_EINVALID, "graph_launch: the context is capturing");
  if (g->scratch != c->scratch || g->temp != c->temp)
    return ah_fail(c, AH_EINVALID, "graph_launch: a call since the recording grew one of the context's work areas (the recorded launches point into the old block) — record the sequence again");
  AH_HIP(c, hipGraphLaunch(g->exec, c->stream));
  return AH_OK;
}

AH_EXPORT int ah_graph_destroy(ah_graph* g) {
  if (!g) return AH_OK;
  (void)hipSetDevice(g->ctx->device);
  (void)hipStreamSynchronize(g->ctx->stream);
  (void)hipGraphExecDestroy(g->exec);
  (void)hipGraphDestroy(g->graph);
  free(g);
  return AH_OK;
}
