// ah_ingest.hip — overlapped host → HBM ingest: Arrow buffers in pinned host memory are cut into chunks, chunk k + 1 is on
// its way over PCIe (hipMemcpyAsync on a side stream) while chunk k is being computed on, and results that go back to the host
// leave on a third stream while chunk k + 1 computes.
//
// Reference counterpart: the executor does not hand a kernel the whole input either — it walks it span by span
// (ExecCtx.ChunkSize / NumParallel, arrow/compute/executor.go:46-64; the span iterator and the per-span kernel call,
// :658-702), with every buffer coming from a memory.Allocator (arrow/memory/allocator.go:20-27).  Here the allocator is
// ah_host_alloc_pinned (go/arrowhip/allocator.go), the span is a chunk of `chunk_bytes`, and what the span loop buys is not
// cache locality but overlap: a 1 GiB column crosses PCIe in ≈ 20 ms, the kernels over it take 0.15–0.5 ms, so an ingest that
// uploads everything, then computes, then downloads runs at the sum of the three; this one runs at the slowest of them.
//
// Ordering is per SLOT, not per stream: ah_upload_async / ah_download_async (ah_ctx.hip) fence the copy stream against the
// compute stream in both directions, which is right for one-shot transfers and wrong for a pipeline (upload k + 1 would wait
// for kernel k).  A slot is one chunk's device buffers plus three events:
//     uploaded   recorded on the h2d stream after the chunk's inputs arrived      → the compute stream waits for it
//     computed   recorded on the compute stream after the chunk's kernel          → the h2d stream waits for it before it
//                overwrites the slot's inputs; the d2h stream waits for it before it reads the slot's output
//     downloaded recorded on the d2h stream after the output left                 → the compute stream waits for it before
//                the next kernel overwrites the slot's output
// With `depth` slots (default 3) the three streams run `depth − 1` chunks apart at most.  Nothing here synchronises the host
// until the call's result is due.
#include <new>
#include <vector>
#include "ah_common.h"

struct ah_ingest {
  ah_ctx* ctx;
  size_t chunk_bytes;
  int depth;
  hipStream_t h2d, d2h;
  struct Slot {
    void* buf[3];        // two inputs and one output, chunk_bytes each
    uint8_t* bits;       // chunk_bytes / 8 + 64: a chunk's output validity (filter)
    hipEvent_t uploaded, computed, downloaded;
  } slots[8];
  void* partials;        // sum: 512 double-double partials per chunk, all reduced at the end
  size_t partials_bytes;
  // filter: the selection vector stays on the device between _count and _primitive
  uint8_t *fdata, *fvalid, *vvalid, *ovalid;
  size_t fdata_cap, fvalid_cap, vvalid_cap, ovalid_cap;
  bool has_fvalid;
  int64_t f_n, f_foff, f_bit0;   // rows, the caller's bit offset, the bit offset inside fdata / fvalid (= foff & 7)
  int f_null_sel;
  int64_t f_chunk_rows;
  std::vector<int64_t>* f_counts;   // survivors per chunk
  unsigned long long* dres;      // 64 bytes of device scalars
  unsigned long long* hres;      // 64 bytes pinned
};

namespace {

#define AHI(call) do { int rc__ = (call); if (rc__ != AH_OK) return rc__; } while (0)

int grow(ah_ctx* c, uint8_t** p, size_t* cap, size_t need) {
  if (need <= *cap) return AH_OK;
  AH_HIP(c, hipStreamSynchronize(c->stream));
  if (*p) AH_HIP(c, hipFree(*p));
  *p = nullptr; *cap = 0;
  const size_t want = (need + 4095) & ~(size_t)4095;
  AH_HIP(c, hipMalloc((void**)p, want));
  *cap = want;
  return AH_OK;
}

// inputs of chunk `k` into its slot: waits only for the kernel that last read this slot
int slot_upload(ah_ingest* g, int s, int which, const void* host, size_t nbytes, bool first_of_chunk) {
  ah_ctx* c = g->ctx;
  if (first_of_chunk) AH_HIP(c, hipStreamWaitEvent(g->h2d, g->slots[s].computed, 0));
  if (nbytes) AH_HIP(c, hipMemcpyAsync(g->slots[s].buf[which], host, nbytes, hipMemcpyHostToDevice, g->h2d));
  return AH_OK;
}
int slot_uploaded(ah_ingest* g, int s, bool writes_output) {   // → the compute stream may start on the slot
  ah_ctx* c = g->ctx;
  AH_HIP(c, hipEventRecord(g->slots[s].uploaded, g->h2d));
  AH_HIP(c, hipStreamWaitEvent(c->stream, g->slots[s].uploaded, 0));
  if (writes_output) AH_HIP(c, hipStreamWaitEvent(c->stream, g->slots[s].downloaded, 0));
  return AH_OK;
}
int slot_computed(ah_ingest* g, int s) {
  AH_HIP(g->ctx, hipEventRecord(g->slots[s].computed, g->ctx->stream));
  return AH_OK;
}
int slot_download(ah_ingest* g, int s, void* host, const void* dev, size_t nbytes) {
  ah_ctx* c = g->ctx;
  AH_HIP(c, hipStreamWaitEvent(g->d2h, g->slots[s].computed, 0));
  if (nbytes) AH_HIP(c, hipMemcpyAsync(host, dev, nbytes, hipMemcpyDeviceToHost, g->d2h));
  AH_HIP(c, hipEventRecord(g->slots[s].downloaded, g->d2h));
  return AH_OK;
}
int drain(ah_ingest* g) {
  ah_ctx* c = g->ctx;
  AH_HIP(c, hipStreamSynchronize(g->h2d));
  AH_HIP(c, hipStreamSynchronize(c->stream));
  AH_HIP(c, hipStreamSynchronize(g->d2h));
  return AH_OK;
}
// a failed call must not leave copies in flight into buffers the caller is about to free
int fail_drained(ah_ingest* g, int rc) {
  char keep[sizeof(g->ctx->err)];
  memcpy(keep, g->ctx->err, sizeof keep);
  (void)drain(g);
  memcpy(g->ctx->err, keep, sizeof keep);
  return rc;
}

}  // namespace

AH_EXPORT int ah_ingest_create(ah_ctx* c, size_t chunk_bytes, int depth, ah_ingest** out) {
  AH_ENTER(c);
  if (!out) return ah_fail(c, AH_EINVALID, "ingest_create: null out pointer");
  *out = nullptr;
  if (chunk_bytes == 0) chunk_bytes = (size_t)32 << 20;   // 32 MiB ≈ 0.6 ms of PCIe: long enough to hide a launch, short enough to overlap
  if (depth == 0) depth = 3;
  if (depth < 2 || depth > 8) return ah_fail(c, AH_EINVALID, "ingest_create: depth must be 2..8");
  if (chunk_bytes < 4096 || (chunk_bytes & 4095)) return ah_fail(c, AH_EINVALID, "ingest_create: chunk_bytes must be a multiple of 4096");
  ah_ingest* g = new (std::nothrow) ah_ingest();
  if (!g) return ah_fail(c, AH_EINVALID, "ingest_create: out of memory");
  memset((void*)g, 0, sizeof(*g));
  g->ctx = c; g->chunk_bytes = chunk_bytes; g->depth = depth;
  g->f_counts = new (std::nothrow) std::vector<int64_t>();
  hipError_t e = hipStreamCreateWithFlags(&g->h2d, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&g->d2h, hipStreamNonBlocking);
  for (int s = 0; s < depth && e == hipSuccess; s++) {
    for (int b = 0; b < 3 && e == hipSuccess; b++) e = hipMalloc(&g->slots[s].buf[b], chunk_bytes + 64);
    if (e == hipSuccess) e = hipMalloc((void**)&g->slots[s].bits, chunk_bytes / 8 + 64);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->slots[s].uploaded, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->slots[s].computed, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&g->slots[s].downloaded, hipEventDisableTiming);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&g->dres, 64);
  if (e == hipSuccess) e = hipHostMalloc((void**)&g->hres, 64, hipHostMallocDefault);
  if (e != hipSuccess || !g->f_counts) {
    (void)hipGetLastError();
    ah_ingest_destroy(g);
    return ah_fail(c, AH_EHIP, "ingest_create: %s", e != hipSuccess ? hipGetErrorString(e) : "out of memory");
  }
  *out = g;
  return AH_OK;
}

AH_EXPORT int ah_ingest_destroy(ah_ingest* g) {
  if (!g) return AH_OK;
  ah_ctx* c = g->ctx;
  (void)hipSetDevice(c->device);
  if (g->h2d) (void)hipStreamSynchronize(g->h2d);
  (void)hipStreamSynchronize(c->stream);
  if (g->d2h) (void)hipStreamSynchronize(g->d2h);
  for (int s = 0; s < 8; s++) {
    for (int b = 0; b < 3; b++) if (g->slots[s].buf[b]) (void)hipFree(g->slots[s].buf[b]);
    if (g->slots[s].bits) (void)hipFree(g->slots[s].bits);
    if (g->slots[s].uploaded) (void)hipEventDestroy(g->slots[s].uploaded);
    if (g->slots[s].computed) (void)hipEventDestroy(g->slots[s].computed);
    if (g->slots[s].downloaded) (void)hipEventDestroy(g->slots[s].downloaded);
  }
  for (void* p : {(void*)g->partials, (void*)g->fdata, (void*)g->fvalid, (void*)g->vvalid, (void*)g->ovalid, (void*)g->dres})
    if (p) (void)hipFree(p);
  if (g->hres) (void)hipHostFree(g->hres);
  if (g->h2d) (void)hipStreamDestroy(g->h2d);
  if (g->d2h) (void)hipStreamDestroy(g->d2h);
  delete g->f_counts;
  delete g;
  return AH_OK;
}

// ---- Sum: chunks in, every chunk's workgroup partials kept, ONE final reduction — the same double-double tree as the resident
// kernel, so the sum of a column that arrives in pieces is rounded once (arrow/math/float64.go:34-47 semantics: validity ignored)
static int ingest_sum(ah_ingest* g, int is_f64, const void* host, size_t len, void* res_host) {
  ah_ctx* c = g->ctx;
  if (!res_host) return ah_fail(c, AH_EINVALID, "ingest_sum: null result pointer");
  memset(res_host, 0, 8);
  if (len == 0) return AH_OK;
  if (!host) return ah_fail(c, AH_EINVALID, "ingest_sum: null buffer");
  const size_t rows_per_chunk = g->chunk_bytes / 8;
  const size_t nchunks = (len + rows_per_chunk - 1) / rows_per_chunk;
  constexpr int kMaxPartials = 512;
  const size_t part_bytes = ah_sum_partial_bytes(is_f64);
  AHI(grow(c, (uint8_t**)&g->partials, &g->partials_bytes, nchunks * kMaxPartials * part_bytes));
  std::vector<int> nparts(nchunks, 0);
  int total_parts = 0;
  // a whole Float64 column of ≤ 31 rows is summed left to right by one lane, as both reference paths do (ah_sum.hip)
  const bool short_f64 = is_f64 && len <= 31;
  if (short_f64) {
    int rc = slot_upload(g, 0, 0, host, len * 8, true);
    if (rc == AH_OK) rc = slot_uploaded(g, 0, false);
    if (rc == AH_OK) rc = ah_sum_short_f64(c, g->slots[0].buf[0], len, g->dres);
    if (rc == AH_OK) rc = slot_computed(g, 0);
    if (rc != AH_OK) return fail_drained(g, rc);
  }
  for (size_t k = 0; k < nchunks && !short_f64; k++) {
    const int s = (int)(k % g->depth);
    const size_t r0 = k * rows_per_chunk, rows = len - r0 < rows_per_chunk ? len - r0 : rows_per_chunk;
    int rc = slot_upload(g, s, 0, (const uint8_t*)host + r0 * 8, rows * 8, true);
    if (rc == AH_OK) rc = slot_uploaded(g, s, false);
    // every chunk's partials are packed back to back so that the final kernel sees one array
    if (rc == AH_OK) rc = ah_sum_chunk_partials(c, is_f64, g->slots[s].buf[0], rows, (uint8_t*)g->partials + (size_t)total_parts * part_bytes,
                                                kMaxPartials, &nparts[k]);
    if (rc == AH_OK) rc = slot_computed(g, s);
    if (rc != AH_OK) return fail_drained(g, rc);
    total_parts += nparts[k];
  }
  int rc = short_f64 ? AH_OK : ah_sum_finish_partials(c, is_f64, g->partials, total_parts, g->dres);
  if (rc != AH_OK) return fail_drained(g, rc);
  AH_HIP(c, hipMemcpyAsync(g->hres, g->dres, 8, hipMemcpyDeviceToHost, c->stream));
  AHI(drain(g));
  memcpy(res_host, g->hres, 8);
  return AH_OK;
}

AH_EXPORT int ah_ingest_sum_float64(ah_ingest* g, const double* host, size_t len, double* res_host) {
  if (!g) return AH_EINVALID;
  AH_ENTER(g->ctx);
  return ingest_sum(g, 1, host, len, res_host);
}
AH_EXPORT int ah_ingest_sum_int64(ah_ingest* g, const int64_t* host, size_t len, int64_t* res_host) {
  if (!g) return AH_EINVALID;
  AH_ENTER(g->ctx);
  return ingest_sum(g, 0, host, len, res_host);
}

// ---- element-wise binary arithmetic (unchecked: base_arithmetic.cc:441-475 semantics, as ah_arithmetic_binary): both operands
// up, the result down, chunk by chunk
AH_EXPORT int ah_ingest_arithmetic_binary(ah_ingest* g, int type, int8_t op, const void* l_host, const void* r_host, void* out_host, int64_t len) {
  if (!g) return AH_EINVALID;
  ah_ctx* c = g->ctx;
  AH_ENTER(c);
  const int w = ah_type_width(type);
  if (!w) return ah_fail(c, AH_EINVALID, "ingest_arithmetic: unsupported type %d", type);
  if (len < 0) return ah_fail(c, AH_EINVALID, "ingest_arithmetic: negative length");
  if (len == 0) return AH_OK;
  if (!l_host || !r_host || !out_host) return ah_fail(c, AH_EINVALID, "ingest_arithmetic: null buffer");
  const int64_t rows_per_chunk = (int64_t)(g->chunk_bytes / (size_t)w);
  const int64_t nchunks = ah_ceil_div(len, rows_per_chunk);
  for (int64_t k = 0; k < nchunks; k++) {
    const int s = (int)(k % g->depth);
    const int64_t r0 = k * rows_per_chunk, rows = len - r0 < rows_per_chunk ? len - r0 : rows_per_chunk;
    const size_t nb = (size_t)rows * (size_t)w, off = (size_t)r0 * (size_t)w;
    int rc = slot_upload(g, s, 0, (const uint8_t*)l_host + off, nb, true);
    if (rc == AH_OK) rc = slot_upload(g, s, 1, (const uint8_t*)r_host + off, nb, false);
    if (rc == AH_OK) rc = slot_uploaded(g, s, true);
    if (rc == AH_OK) rc = ah_arithmetic_binary(c, type, op, g->slots[s].buf[0], g->slots[s].buf[1], g->slots[s].buf[2], rows);
    if (rc == AH_OK) rc = slot_computed(g, s);
    if (rc == AH_OK) rc = slot_download(g, s, (uint8_t*)out_host + off, g->slots[s].buf[2], nb);
    if (rc != AH_OK) return fail_drained(g, rc);
  }
  return drain(g);
}

// ---- Filter, two-phase like ah_filter_count / ah_filter_primitive (the host allocates the output in between).  The selection
// vector is 1/64 of the values: it goes up whole in the count call and STAYS on the device; the fill call streams the values.
AH_EXPORT int ah_ingest_filter_count(ah_ingest* g, const uint8_t* fdata_host, const uint8_t* fvalid_host, int64_t foff, int64_t n, int null_sel,
                                     int64_t* n_out_host) {
  if (!g) return AH_EINVALID;
  ah_ctx* c = g->ctx;
  AH_ENTER(c);
  if (!n_out_host) return ah_fail(c, AH_EINVALID, "ingest_filter_count: null result pointer");
  if (n < 0 || foff < 0) return ah_fail(c, AH_EINVALID, "ingest_filter_count: negative length/offset");
  *n_out_host = 0;
  g->f_n = -1;
  if (n == 0) { g->f_n = 0; g->f_counts->clear(); return AH_OK; }
  if (!fdata_host) return ah_fail(c, AH_EINVALID, "ingest_filter_count: null filter data");
  const int64_t byte0 = foff >> 3, nbytes = ((foff + n + 7) >> 3) - byte0;
  AHI(grow(c, &g->fdata, &g->fdata_cap, (size_t)nbytes + 64));
  AH_HIP(c, hipMemcpyAsync(g->fdata, fdata_host + byte0, (size_t)nbytes, hipMemcpyHostToDevice, g->h2d));
  g->has_fvalid = fvalid_host != nullptr;
  if (fvalid_host) {
    AHI(grow(c, &g->fvalid, &g->fvalid_cap, (size_t)nbytes + 64));
    AH_HIP(c, hipMemcpyAsync(g->fvalid, fvalid_host + byte0, (size_t)nbytes, hipMemcpyHostToDevice, g->h2d));
  }
  AH_HIP(c, hipEventRecord(g->slots[0].uploaded, g->h2d));
  AH_HIP(c, hipStreamWaitEvent(c->stream, g->slots[0].uploaded, 0));
  g->f_bit0 = foff & 7; g->f_foff = foff; g->f_null_sel = null_sel;
  // survivors per chunk (chunks of chunk_bytes / 8 rows whatever the value width: narrower values just fill less of a slot)
  g->f_chunk_rows = (int64_t)(g->chunk_bytes / 8);
  const int64_t nchunks = ah_ceil_div(n, g->f_chunk_rows);
  g->f_counts->assign((size_t)nchunks, 0);
  int64_t total = 0;
  for (int64_t k = 0; k < nchunks; k++) {
    const int64_t r0 = k * g->f_chunk_rows, rows = n - r0 < g->f_chunk_rows ? n - r0 : g->f_chunk_rows;
    int64_t cnt = 0;
    int rc = ah_filter_count(c, g->fdata, g->has_fvalid ? g->fvalid : nullptr, g->f_bit0 + r0, rows, null_sel, &cnt);
    if (rc != AH_OK) return fail_drained(g, rc);
    (*g->f_counts)[(size_t)k] = cnt;
    total += cnt;
  }
  g->f_n = n;
  *n_out_host = total;
  return AH_OK;
}

AH_EXPORT int ah_ingest_filter_primitive(ah_ingest* g, int byte_width, const void* values_host, const uint8_t* vvalid_host, int64_t voff, int64_t n,
                                         int64_t n_out, void* out_values_host, uint8_t* out_valid_host, int64_t* out_null_count_host) {
  if (!g) return AH_EINVALID;
  ah_ctx* c = g->ctx;
  AH_ENTER(c);
  if (out_null_count_host) *out_null_count_host = 0;
  if (byte_width != 1 && byte_width != 2 && byte_width != 4 && byte_width != 8) return ah_fail(c, AH_EINVALID, "filter: invalid values byte width %d", byte_width);
  if (n < 0 || voff < 0) return ah_fail(c, AH_EINVALID, "ingest_filter: negative length/offset");
  if (g->f_n != n) return ah_fail(c, AH_EINVALID, "ingest_filter: call ah_ingest_filter_count for this mask first (it keeps the mask on the device)");
  if (n == 0) return AH_OK;
  if (!values_host || (n_out > 0 && !out_values_host)) return ah_fail(c, AH_EINVALID, "ingest_filter: null buffer");
  int64_t total = 0;
  for (int64_t cnt : *g->f_counts) total += cnt;
  if (total != n_out) return ah_fail(c, AH_EINVALID, "filter: n_out=%lld does not match the selection count %lld", (long long)n_out, (long long)total);
  const bool want_valid = out_valid_host != nullptr;
  const int64_t vbit0 = voff & 7;
  if (vvalid_host && want_valid) {   // value validity: 1/64 of the values, whole, before the first chunk
    const int64_t byte0 = voff >> 3, nbytes = ((voff + n + 7) >> 3) - byte0;
    AHI(grow(c, &g->vvalid, &g->vvalid_cap, (size_t)nbytes + 64));
    AH_HIP(c, hipMemcpyAsync(g->vvalid, vvalid_host + byte0, (size_t)nbytes, hipMemcpyHostToDevice, g->h2d));
  }
  if (want_valid) {
    AHI(grow(c, &g->ovalid, &g->ovalid_cap, (size_t)((n_out + 7) >> 3) + 64));
    AH_HIP(c, hipMemsetAsync(g->ovalid, 0, (size_t)((n_out + 7) >> 3) + 8, c->stream));
  }
  const uint8_t* vv = (vvalid_host && want_valid) ? g->vvalid : nullptr;
  const uint8_t* fv = g->has_fvalid ? g->fvalid : nullptr;
  const int64_t nchunks = (int64_t)g->f_counts->size();
  int64_t pos = 0;   // output rows so far
  for (int64_t k = 0; k < nchunks; k++) {
    const int s = (int)(k % g->depth);
    const int64_t r0 = k * g->f_chunk_rows, rows = n - r0 < g->f_chunk_rows ? n - r0 : g->f_chunk_rows;
    const int64_t cnt = (*g->f_counts)[(size_t)k];
    int rc = AH_OK;
    if (cnt > 0) {   // a chunk nothing is selected from never crosses PCIe
      rc = slot_upload(g, s, 0, (const uint8_t*)values_host + (size_t)r0 * (size_t)byte_width, (size_t)rows * (size_t)byte_width, true);
      if (rc == AH_OK) rc = slot_uploaded(g, s, true);
      if (rc == AH_OK)
        rc = ah_filter_primitive(c, byte_width, g->slots[s].buf[0], vv, vbit0 + r0, g->fdata, fv, g->f_bit0 + r0, rows, g->f_null_sel, cnt,
                                 g->slots[s].buf[2], want_valid ? g->slots[s].bits : nullptr, nullptr);
      // the chunk's validity bits start at bit 0 of the slot; they belong at bit `pos` of the call's output bitmap
      if (rc == AH_OK && want_valid) rc = ah_copy_bitmap(c, g->slots[s].bits, 0, cnt, g->ovalid, pos, 0);
      if (rc == AH_OK) rc = slot_computed(g, s);
      if (rc == AH_OK) rc = slot_download(g, s, (uint8_t*)out_values_host + (size_t)pos * (size_t)byte_width, g->slots[s].buf[2], (size_t)cnt * (size_t)byte_width);
    }
    if (rc != AH_OK) return fail_drained(g, rc);
    pos += cnt;
  }
  if (want_valid) {
    int rc = ah_popcount_async(c, g->ovalid, 0, n_out, g->dres);
    if (rc != AH_OK) return fail_drained(g, rc);
    AH_HIP(c, hipMemcpyAsync(g->hres, g->dres, 8, hipMemcpyDeviceToHost, c->stream));
    AH_HIP(c, hipMemcpyAsync(out_valid_host, g->ovalid, (size_t)((n_out + 7) >> 3), hipMemcpyDeviceToHost, c->stream));
  }
  AHI(drain(g));
  if (out_null_count_host) *out_null_count_host = want_valid ? n_out - (int64_t)g->hres[0] : 0;
  return AH_OK;
}

// ---- the slot protocol itself, for hosts that run their own kernels over their own chunking (the Go shim's stage()) -----------
AH_EXPORT int ah_ingest_depth(ah_ingest* g) { return g ? g->depth : -1; }
AH_EXPORT size_t ah_ingest_chunk_bytes(ah_ingest* g) { return g ? g->chunk_bytes : 0; }
AH_EXPORT void* ah_ingest_slot_buffer(ah_ingest* g, int slot, int which) {
  if (!g || slot < 0 || slot >= g->depth || which < 0 || which > 2) return nullptr;
  return g->slots[slot].buf[which];
}
AH_EXPORT int ah_ingest_slot_upload(ah_ingest* g, int slot, int which, size_t dst_offset, const void* hptr, size_t nbytes, int first_of_chunk) {
  if (!g) return AH_EINVALID;
  AH_ENTER_KEEP(g->ctx);
  if (slot < 0 || slot >= g->depth || which < 0 || which > 2 || dst_offset + nbytes > g->chunk_bytes) return ah_fail(g->ctx, AH_EINVALID, "ingest_slot_upload: bad slot / buffer / range");
  if (first_of_chunk) AH_HIP(g->ctx, hipStreamWaitEvent(g->h2d, g->slots[slot].computed, 0));
  if (nbytes) AH_HIP(g->ctx, hipMemcpyAsync((uint8_t*)g->slots[slot].buf[which] + dst_offset, hptr, nbytes, hipMemcpyHostToDevice, g->h2d));
  return AH_OK;
}
AH_EXPORT int ah_ingest_slot_ready(ah_ingest* g, int slot, int writes_output) {
  if (!g) return AH_EINVALID;
  AH_ENTER_KEEP(g->ctx);
  if (slot < 0 || slot >= g->depth) return ah_fail(g->ctx, AH_EINVALID, "ingest_slot_ready: bad slot");
  return slot_uploaded(g, slot, writes_output != 0);
}
AH_EXPORT int ah_ingest_slot_release(ah_ingest* g, int slot) {
  if (!g) return AH_EINVALID;
  AH_ENTER_KEEP(g->ctx);
  if (slot < 0 || slot >= g->depth) return ah_fail(g->ctx, AH_EINVALID, "ingest_slot_release: bad slot");
  return slot_computed(g, slot);
}
AH_EXPORT int ah_ingest_slot_download(ah_ingest* g, int slot, int which, size_t src_offset, void* hptr, size_t nbytes) {
  if (!g) return AH_EINVALID;
  AH_ENTER_KEEP(g->ctx);
  if (slot < 0 || slot >= g->depth || which < 0 || which > 2 || src_offset + nbytes > g->chunk_bytes) return ah_fail(g->ctx, AH_EINVALID, "ingest_slot_download: bad slot / buffer / range");
  return slot_download(g, slot, hptr, (const uint8_t*)g->slots[slot].buf[which] + src_offset, nbytes);
}
AH_EXPORT int ah_ingest_wait(ah_ingest* g) {
  if (!g) return AH_EINVALID;
  AH_ENTER_KEEP(g->ctx);
  return drain(g);
}

// pin / unpin memory the host already owns (a Go []byte for the duration of one cgo call): hipMemcpyAsync from pageable memory
// is staged by the runtime and does not overlap anything
AH_EXPORT int ah_host_register(ah_ctx* c, void* hptr, size_t nbytes) {
  AH_ENTER_KEEP(c);
  if (!hptr || !nbytes) return ah_fail(c, AH_EINVALID, "host_register: null buffer");
  AH_HIP(c, hipHostRegister(hptr, nbytes, hipHostRegisterDefault));
  return AH_OK;
}
AH_EXPORT int ah_host_unregister(ah_ctx* c, void* hptr) {
  AH_ENTER_KEEP(c);
  if (!hptr) return AH_OK;
  AH_HIP(c, hipHostUnregister(hptr));
  return AH_OK;
}
