/*
 * cabi_driver.c — a compiled C caller that plays the Go executor against libarrowhip.so (TEST INFRASTRUCTURE).
 *
 * No Go toolchain exists in the build image, so go/arrowhip/register.go cannot run.  What CAN be tested without Go is the
 * contract its ExecFns rely on, from a foreign (non-Python) caller:
 *   · the exact call SEQUENCES of register.go — stage operands (alloc + upload) → kernel → size the output through the
 *     "allocator" (two-phase for array_filter: count, allocate, fill) → download — for add, greater, array_filter,
 *     array_take, dictionary_encode   (exec/kernel.go:457-499 NullHandling / MemAlloc, :617 ArrayKernelExec,
 *     :660-661 scalar defaults, :724-725 vector defaults);
 *   · that a context works from ANY thread: compute/exec.go:165 runs a kernel on a fresh goroutine, and goroutines hop OS
 *     threads between cgo calls — here every STEP of every ExecFn (one to three C-ABI calls) runs on a NEW pthread;
 *   · two ExecFns at the same time on two contexts (scalar executors are pooled per call, executor.go:867-873).
 * Inputs are generated here (a 64-bit LCG), inputs and results are written to <outdir>/ as raw little-endian files, and
 * tests/test_cabi_driver.py compares them with the CPU oracle.
 *
 *   gcc -O2 -std=c11 -pthread -Iinclude tests/cabi_driver.c -Larrow_go_amd -larrowhip -Wl,-rpath,$PWD/arrow_go_amd -o <exe>
 *   <exe> <outdir> <rows>
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "arrowhip.h"

static const char* g_out;
static int64_t g_n;

#define CHECK(ctx, call)                                                                              \
  do {                                                                                                \
    int rc__ = (call);                                                                                \
    if (rc__ != AH_OK) {                                                                              \
      fprintf(stderr, "cabi_driver: %s -> %d (%s)\n", #call, rc__, (ctx) ? ah_last_error(ctx) : ""); \
      exit(3);                                                                                        \
    }                                                                                                 \
  } while (0)

/* every step on a thread of its own: the goroutine that runs an ExecFn may be on another OS thread at every cgo call */
typedef void (*step_fn)(void*);
typedef struct { step_fn f; void* st; } hop_arg;
static void* hop_main(void* p) { hop_arg* a = (hop_arg*)p; a->f(a->st); return NULL; }
static unsigned long g_threads_used = 0;
static void hop(step_fn f, void* st) {
  hop_arg a = {f, st};
  pthread_t t;
  if (pthread_create(&t, NULL, hop_main, &a) != 0) { perror("pthread_create"); exit(4); }
  pthread_join(t, NULL);
  __atomic_add_fetch(&g_threads_used, 1, __ATOMIC_RELAXED);
}

static void dump(const char* name, const void* p, size_t nbytes) {
  char path[1024];
  snprintf(path, sizeof path, "%s/%s", g_out, name);
  FILE* f = fopen(path, "wb");
  if (!f || (nbytes && fwrite(p, 1, nbytes, f) != nbytes)) { perror(path); exit(5); }
  fclose(f);
}
static uint64_t lcg(uint64_t* s) { *s = *s * 6364136223846793005ull + 1442695040888963407ull; return *s >> 11; }
static size_t bytes_for_bits(int64_t n) { return (size_t)((n + 7) / 8); }

/* ---- "memory.Allocator": zero-filled host buffers, 64-byte aligned and padded (arrow/memory/allocator.go:20-27) -------- */
static void* go_allocate(size_t nbytes) {
  void* p = NULL;
  size_t cap = (nbytes + 63) & ~(size_t)63;
  if (cap == 0) cap = 64;
  if (posix_memalign(&p, 64, cap) != 0) exit(6);
  memset(p, 0, cap);
  return p;
}

/* ---- x.stage(): alloc + upload of one operand ----------------------------------------------------------------------- */
typedef struct { ah_ctx* ctx; const void* host; size_t nbytes; void* dev; } stage_st;
static void step_stage(void* p) {
  stage_st* s = (stage_st*)p;
  CHECK(s->ctx, ah_buf_alloc(s->ctx, s->nbytes + 64, &s->dev));
  CHECK(s->ctx, ah_upload_async(s->ctx, s->dev, s->host, s->nbytes));
  CHECK(s->ctx, ah_sync(s->ctx));
}
static void* stage(ah_ctx* ctx, const void* host, size_t nbytes) {
  stage_st s = {ctx, host, nbytes, NULL};
  hop(step_stage, &s);
  return s.dev;
}
typedef struct { ah_ctx* ctx; void* host; const void* dev; size_t nbytes; } dl_st;
static void step_download(void* p) {
  dl_st* s = (dl_st*)p;
  CHECK(s->ctx, ah_download_async(s->ctx, s->host, s->dev, s->nbytes));
  CHECK(s->ctx, ah_sync(s->ctx));
}
static void download(ah_ctx* ctx, void* host, const void* dev, size_t nbytes) {
  dl_st s = {ctx, host, dev, nbytes};
  hop(step_download, &s);
}
typedef struct { ah_ctx* ctx; void* dev; } free_st;
static void step_free(void* p) { free_st* s = (free_st*)p; CHECK(s->ctx, ah_buf_free(s->ctx, s->dev)); }
static void dfree(ah_ctx* ctx, void* dev) { free_st s = {ctx, dev}; hop(step_free, &s); }
typedef struct { ah_ctx* ctx; size_t nbytes; void* dev; } alloc_st;
static void step_alloc(void* p) { alloc_st* s = (alloc_st*)p; CHECK(s->ctx, ah_buf_alloc(s->ctx, s->nbytes, &s->dev)); }
static void* dalloc(ah_ctx* ctx, size_t nbytes) { alloc_st s = {ctx, nbytes, NULL}; hop(step_alloc, &s); return s.dev; }

/* ---- binaryExec (register.go): add_unchecked(int64, int64) ------------------------------------------------------------ */
typedef struct { ah_ctx* ctx; void *l, *r, *o; int64_t n; } add_st;
static void step_add(void* p) { add_st* s = (add_st*)p; CHECK(s->ctx, ah_arithmetic_binary(s->ctx, AH_INT64, 0 /*ADD*/, s->l, s->r, s->o, s->n)); }
static void exec_add(ah_ctx* ctx, const int64_t* a, const int64_t* b, int64_t* out /* preallocated, zeroed: MemPrealloc */, int64_t n) {
  add_st s = {ctx, stage(ctx, a, (size_t)n * 8), stage(ctx, b, (size_t)n * 8), dalloc(ctx, (size_t)n * 8 + 64), n};
  hop(step_add, &s);
  download(ctx, out, s.o, (size_t)n * 8);
  dfree(ctx, s.l); dfree(ctx, s.r); dfree(ctx, s.o);
}

/* ---- compareExec: greater(int64 array, int64 scalar) into a bitmap at bit offset out_off -------------------------------- */
typedef struct { ah_ctx* ctx; void* l; const int64_t* scalar; void* o; int64_t n; int off; } cmp_st;
static void step_cmp(void* p) { cmp_st* s = (cmp_st*)p; CHECK(s->ctx, ah_comparison(s->ctx, AH_CMP_GT, AH_SHAPE_AS, AH_INT64, s->l, s->scalar, (uint8_t*)s->o, s->n, s->off)); }
static void exec_greater(ah_ctx* ctx, const int64_t* a, int64_t thr, uint8_t* out_bits, int64_t n, int out_off) {
  const size_t ob = bytes_for_bits(out_off + n);
  cmp_st s = {ctx, stage(ctx, a, (size_t)n * 8), &thr, stage(ctx, out_bits, ob) /* the device copy starts from the executor's bytes */, n, out_off};
  hop(step_cmp, &s);
  download(ctx, out_bits, s.o, ob);
  dfree(ctx, s.l); dfree(ctx, s.o);
}

/* ---- filterExec: array_filter(int64 values with nulls, boolean filter) — count, allocate, fill -------------------------- */
typedef struct { ah_ctx* ctx; void *fd, *fv; int64_t foff, n; int null_sel; int64_t n_out; } fcount_st;
static void step_fcount(void* p) { fcount_st* s = (fcount_st*)p; CHECK(s->ctx, ah_filter_count(s->ctx, (uint8_t*)s->fd, (uint8_t*)s->fv, s->foff, s->n, s->null_sel, &s->n_out)); }
typedef struct { ah_ctx* ctx; void *v, *vv, *fd, *fv, *o, *ov; int64_t voff, foff, n, n_out, nulls; int null_sel; } ffill_st;
static void step_ffill(void* p) {
  ffill_st* s = (ffill_st*)p;
  CHECK(s->ctx, ah_filter_primitive(s->ctx, 8, s->v, (uint8_t*)s->vv, s->voff, (uint8_t*)s->fd, (uint8_t*)s->fv, s->foff, s->n, s->null_sel, s->n_out, s->o,
                                    (uint8_t*)s->ov, &s->nulls));
}
static void exec_filter(ah_ctx* ctx, const int64_t* vals, const uint8_t* vvalid, int64_t voff, const uint8_t* fdata, const uint8_t* fvalid, int64_t foff,
                        int64_t n, int null_sel, int64_t** out_vals, uint8_t** out_valid, int64_t* n_out, int64_t* nulls) {
  void* dv = stage(ctx, vals, (size_t)n * 8);
  void* dvv = stage(ctx, vvalid, bytes_for_bits(voff + n));
  void* dfd = stage(ctx, fdata, bytes_for_bits(foff + n));
  void* dfv = fvalid ? stage(ctx, fvalid, bytes_for_bits(foff + n)) : NULL;
  fcount_st c = {ctx, dfd, dfv, foff, n, null_sel, 0};
  hop(step_fcount, &c);
  /* between the two phases the Go kernel allocates: device scratch for the result, then — in finishVector — ctx.Allocate */
  ffill_st f = {ctx, dv, dvv, dfd, dfv, dalloc(ctx, (size_t)c.n_out * 8 + 64), dalloc(ctx, bytes_for_bits(c.n_out) + 64), voff, foff, n, c.n_out, 0, null_sel};
  hop(step_ffill, &f);
  *out_vals = (int64_t*)go_allocate((size_t)c.n_out * 8);
  *out_valid = (uint8_t*)go_allocate(bytes_for_bits(c.n_out));
  download(ctx, *out_vals, f.o, (size_t)c.n_out * 8);
  download(ctx, *out_valid, f.ov, bytes_for_bits(c.n_out));
  *n_out = c.n_out; *nulls = f.nulls;
  dfree(ctx, dv); dfree(ctx, dvv); dfree(ctx, dfd); if (dfv) dfree(ctx, dfv); dfree(ctx, f.o); dfree(ctx, f.ov);
}

/* ---- takeExec: array_take(int64 values with nulls, int32 indices with nulls) ---------------------------------------------- */
typedef struct { ah_ctx* ctx; void *v, *vv, *i, *iv, *o, *ov; int64_t nvalues, nidx, nulls, bad; int rc; } take_st;
static void step_take(void* p) {
  take_st* s = (take_st*)p;
  s->rc = ah_take_primitive(s->ctx, 8, s->v, (uint8_t*)s->vv, 0, s->nvalues, 4, 1, s->i, (uint8_t*)s->iv, 0, s->nidx, 1, s->o, (uint8_t*)s->ov, &s->nulls, &s->bad);
}
static int exec_take(ah_ctx* ctx, const int64_t* vals, const uint8_t* vvalid, int64_t nvalues, const int32_t* idx, const uint8_t* ivalid, int64_t nidx,
                     int64_t* out_vals, uint8_t* out_valid, int64_t* nulls, char* errbuf, size_t errcap) {
  take_st s = {ctx, stage(ctx, vals, (size_t)nvalues * 8), stage(ctx, vvalid, bytes_for_bits(nvalues)), stage(ctx, idx, (size_t)nidx * 4),
               stage(ctx, ivalid, bytes_for_bits(nidx)), dalloc(ctx, (size_t)nidx * 8 + 64), dalloc(ctx, bytes_for_bits(nidx) + 64), nvalues, nidx, 0, 0, 0};
  hop(step_take, &s);
  if (s.rc == AH_OK) {
    download(ctx, out_vals, s.o, (size_t)nidx * 8);
    download(ctx, out_valid, s.ov, bytes_for_bits(nidx));
    *nulls = s.nulls;
  } else {
    snprintf(errbuf, errcap, "%d:%s", s.rc, ah_last_error(ctx));   /* the Go shim wraps this into arrow.ErrIndex */
  }
  dfree(ctx, s.v); dfree(ctx, s.vv); dfree(ctx, s.i); dfree(ctx, s.iv); dfree(ctx, s.o); dfree(ctx, s.ov);
  return s.rc;
}

/* ---- hashExec: dictionary_encode(int64 keys with nulls), NullEncodingMask ------------------------------------------------- */
typedef struct { ah_ctx* ctx; void *k, *kv, *ids, *idv, *dict; int64_t n, ndict; int32_t null_id; } enc_st;
static void step_encode(void* p) {
  enc_st* s = (enc_st*)p;
  CHECK(s->ctx, ah_hash_u64_encode(s->ctx, (const uint64_t*)s->k, (uint8_t*)s->kv, 0, s->n, 0, (int32_t*)s->ids, (uint8_t*)s->idv, (uint64_t*)s->dict, &s->ndict, &s->null_id));
}
static void exec_dictionary_encode(ah_ctx* ctx, const int64_t* keys, const uint8_t* kvalid, int64_t n, int32_t* out_ids, uint8_t* out_ids_valid,
                                   uint64_t** out_dict, int64_t* ndict) {
  enc_st s = {ctx, stage(ctx, keys, (size_t)n * 8), stage(ctx, kvalid, bytes_for_bits(n)), dalloc(ctx, (size_t)n * 4 + 64), dalloc(ctx, bytes_for_bits(n) + 64),
              dalloc(ctx, (size_t)(n + 1) * 8 + 64), n, 0, -1};
  hop(step_encode, &s);
  *out_dict = (uint64_t*)go_allocate((size_t)s.ndict * 8);
  download(ctx, out_ids, s.ids, (size_t)n * 4);
  download(ctx, out_ids_valid, s.idv, bytes_for_bits(n));
  download(ctx, *out_dict, s.dict, (size_t)s.ndict * 8);
  *ndict = s.ndict;
  dfree(ctx, s.k); dfree(ctx, s.kv); dfree(ctx, s.ids); dfree(ctx, s.idv); dfree(ctx, s.dict);
}

/* ---- two ExecFns at once, each on its own context -------------------------------------------------------------------- */
typedef struct { ah_ctx* ctx; const int64_t *a, *b; int64_t* out; int64_t n; } par_add;
static void* par_add_main(void* p) { par_add* s = (par_add*)p; for (int rep = 0; rep < 3; rep++) exec_add(s->ctx, s->a, s->b, s->out, s->n); return NULL; }
typedef struct { ah_ctx* ctx; const int64_t* a; uint8_t* bits; int64_t n; } par_cmp;
static void* par_cmp_main(void* p) { par_cmp* s = (par_cmp*)p; for (int rep = 0; rep < 3; rep++) exec_greater(s->ctx, s->a, 0, s->bits, s->n, 5); return NULL; }

typedef struct { int device; ah_ctx* ctx; } create_st;
static void step_create(void* p) { create_st* s = (create_st*)p; CHECK((ah_ctx*)NULL, ah_ctx_create(s->device, &s->ctx)); }
static void step_destroy(void* p) { ah_ctx_destroy((ah_ctx*)p); }

/* ---- `<exe> bench <log2 rows>`: the two-phase Filter call (count, then fill) timed from C — what a compiled host pays between the
 * two dependent launches, without an interpreter in the loop (bench.py drives the same two calls through Python's ctypes) */
static int bench_filter(int lg) {
  const int64_t n = (int64_t)1 << lg;
  ah_ctx* ctx;
  CHECK((ah_ctx*)NULL, ah_ctx_create(0, &ctx));
  uint64_t seed = 12345;
  int64_t* a = (int64_t*)go_allocate((size_t)n * 8);
  uint8_t* fd = (uint8_t*)go_allocate(bytes_for_bits(n) + 64); uint8_t* vv = (uint8_t*)go_allocate(bytes_for_bits(n) + 64);
  for (int64_t i = 0; i < n; i++) a[i] = (int64_t)lcg(&seed);
  for (size_t i = 0; i < bytes_for_bits(n); i++) { fd[i] = (uint8_t)lcg(&seed); vv[i] = (uint8_t)(lcg(&seed) | lcg(&seed) | lcg(&seed)); }   /* s = 0.5, 87 % valid */
  void *dv, *dfd, *dvv, *dout, *dov;
  CHECK(ctx, ah_buf_alloc(ctx, (size_t)n * 8 + 64, &dv)); CHECK(ctx, ah_buf_alloc(ctx, bytes_for_bits(n) + 64, &dfd));
  CHECK(ctx, ah_buf_alloc(ctx, bytes_for_bits(n) + 64, &dvv)); CHECK(ctx, ah_buf_alloc(ctx, (size_t)n * 8 + 64, &dout));
  CHECK(ctx, ah_buf_alloc(ctx, bytes_for_bits(n) + 64, &dov));
  CHECK(ctx, ah_upload_async(ctx, dv, a, (size_t)n * 8)); CHECK(ctx, ah_upload_async(ctx, dfd, fd, bytes_for_bits(n)));
  CHECK(ctx, ah_upload_async(ctx, dvv, vv, bytes_for_bits(n))); CHECK(ctx, ah_sync(ctx));
  int64_t n_out = 0;
  const int reps = 20;
  float ms_both = 0, ms_fill = 0, ms_count = 0;
  for (int pass = 0; pass < 2; pass++) {   /* pass 0 warms up */
    CHECK(ctx, ah_event_record(ctx, 1));
    for (int r = 0; r < reps; r++) {
      CHECK(ctx, ah_filter_count(ctx, (uint8_t*)dfd, NULL, 0, n, AH_DROP_NULLS, &n_out));
      CHECK(ctx, ah_filter_primitive(ctx, 8, dv, (uint8_t*)dvv, 0, (uint8_t*)dfd, NULL, 0, n, AH_DROP_NULLS, n_out, dout, (uint8_t*)dov, NULL));
    }
    CHECK(ctx, ah_event_record(ctx, 2));
    CHECK(ctx, ah_event_elapsed_ms(ctx, 1, 2, &ms_both));
    CHECK(ctx, ah_event_record(ctx, 3));
    for (int r = 0; r < reps; r++)
      CHECK(ctx, ah_filter_primitive(ctx, 8, dv, (uint8_t*)dvv, 0, (uint8_t*)dfd, NULL, 0, n, AH_DROP_NULLS, n_out, dout, (uint8_t*)dov, NULL));
    CHECK(ctx, ah_event_record(ctx, 4));
    CHECK(ctx, ah_event_elapsed_ms(ctx, 3, 4, &ms_fill));
    CHECK(ctx, ah_event_record(ctx, 5));
    for (int r = 0; r < reps; r++) CHECK(ctx, ah_filter_count(ctx, (uint8_t*)dfd, NULL, 0, n, AH_DROP_NULLS, &n_out));
    CHECK(ctx, ah_event_record(ctx, 6));
    CHECK(ctx, ah_event_elapsed_ms(ctx, 5, 6, &ms_count));
  }
  const double traffic = (8 + 0.125 + 0.125) * (double)n + (8 + 0.125) * (double)n_out;
  printf("{\"caller\": \"C (tests/cabi_driver.c)\", \"rows\": %lld, \"selected\": %.4f, \"count_and_fill_ms\": %.4f, \"fill_only_ms\": %.4f, \"count_only_ms\": %.4f, "
         "\"traffic_GB/s\": %.1f, \"frac_of_8TB/s\": %.4f}\n",
         (long long)n, (double)n_out / (double)n, ms_both / reps, ms_fill / reps, ms_count / reps, traffic / (ms_both / reps) / 1e6, traffic / (ms_both / reps) / 1e6 / 8000.0);
  ah_ctx_destroy(ctx);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 3 && strcmp(argv[1], "bench") == 0) return bench_filter(atoi(argv[2]));
  if (argc < 3) { fprintf(stderr, "usage: %s <outdir> <rows> | bench <log2 rows>\n", argv[0]); return 2; }
  g_out = argv[1];
  g_n = atoll(argv[2]);
  const int64_t n = g_n;
  create_st c1 = {0, NULL}, c2 = {0, NULL};
  hop(step_create, &c1);   /* created on one thread, used on many others, destroyed on yet another */
  hop(step_create, &c2);
  uint64_t seed = 0x9E3779B97F4A7C15ull;
  int64_t* a = (int64_t*)go_allocate((size_t)n * 8); int64_t* b = (int64_t*)go_allocate((size_t)n * 8); int64_t* keys = (int64_t*)go_allocate((size_t)n * 8);
  int32_t* idx = (int32_t*)go_allocate((size_t)n * 4);
  const int64_t voff = 3, foff = 13;
  uint8_t* vvalid = (uint8_t*)go_allocate(bytes_for_bits(voff + n) + 8); uint8_t* fdata = (uint8_t*)go_allocate(bytes_for_bits(foff + n) + 8);
  uint8_t* fvalid = (uint8_t*)go_allocate(bytes_for_bits(foff + n) + 8); uint8_t* ivalid = (uint8_t*)go_allocate(bytes_for_bits(n) + 8);
  uint8_t* v0 = (uint8_t*)go_allocate(bytes_for_bits(n) + 8);
  for (int64_t i = 0; i < n; i++) {
    a[i] = (int64_t)(lcg(&seed) << 11) ^ (int64_t)lcg(&seed);
    b[i] = (int64_t)(lcg(&seed) << 11) ^ (int64_t)lcg(&seed);
    keys[i] = (int64_t)(lcg(&seed) % 5003) * 1000003;
    idx[i] = (int32_t)(lcg(&seed) % (uint64_t)n);
  }
  for (size_t i = 0; i < bytes_for_bits(voff + n); i++) vvalid[i] = (uint8_t)(lcg(&seed) | lcg(&seed) | lcg(&seed));   /* ≈ 87 % ones */
  for (size_t i = 0; i < bytes_for_bits(foff + n); i++) { fdata[i] = (uint8_t)lcg(&seed); fvalid[i] = (uint8_t)(lcg(&seed) | lcg(&seed) | lcg(&seed)); }
  for (size_t i = 0; i < bytes_for_bits(n); i++) { ivalid[i] = (uint8_t)(lcg(&seed) | lcg(&seed) | lcg(&seed)); v0[i] = (uint8_t)(lcg(&seed) | lcg(&seed) | lcg(&seed)); }
  ivalid[(n / 2) >> 3] |= (uint8_t)(1u << ((n / 2) & 7));   /* the slot the bounds-check case below makes invalid-range must be a VALID index slot */
  dump("a.bin", a, (size_t)n * 8); dump("b.bin", b, (size_t)n * 8); dump("keys.bin", keys, (size_t)n * 8); dump("idx.bin", idx, (size_t)n * 4);
  dump("vvalid.bin", vvalid, bytes_for_bits(voff + n)); dump("fdata.bin", fdata, bytes_for_bits(foff + n)); dump("fvalid.bin", fvalid, bytes_for_bits(foff + n));
  dump("ivalid.bin", ivalid, bytes_for_bits(n)); dump("v0.bin", v0, bytes_for_bits(n));

  /* add */
  int64_t* sum = (int64_t*)go_allocate((size_t)n * 8);
  exec_add(c1.ctx, a, b, sum, n);
  dump("add.bin", sum, (size_t)n * 8);
  /* greater at out.Offset = 5: the executor's bytes around the range (here 0xA5 / 0x5A patterns) must survive */
  const int goff = 5;
  uint8_t* gt = (uint8_t*)go_allocate(bytes_for_bits(goff + n));
  memset(gt, 0xA5, bytes_for_bits(goff + n));
  dump("gt_before.bin", gt, bytes_for_bits(goff + n));
  exec_greater(c1.ctx, a, 12345, gt, n, goff);
  dump("gt.bin", gt, bytes_for_bits(goff + n));
  /* array_filter: Drop and Emit */
  for (int null_sel = 0; null_sel < 2; null_sel++) {
    int64_t* fo; uint8_t* fov; int64_t n_out, nulls;
    exec_filter(c1.ctx, a, vvalid, voff, fdata, fvalid, foff, n, null_sel, &fo, &fov, &n_out, &nulls);
    char name[64];
    snprintf(name, sizeof name, "filter%d_vals.bin", null_sel); dump(name, fo, (size_t)n_out * 8);
    snprintf(name, sizeof name, "filter%d_valid.bin", null_sel); dump(name, fov, bytes_for_bits(n_out));
    int64_t meta[2] = {n_out, nulls};
    snprintf(name, sizeof name, "filter%d_meta.bin", null_sel); dump(name, meta, sizeof meta);
    free(fo); free(fov);
  }
  /* array_take, then the same with one index out of range: the error text is the reference's */
  int64_t* to = (int64_t*)go_allocate((size_t)n * 8); uint8_t* tov = (uint8_t*)go_allocate(bytes_for_bits(n));
  int64_t tnulls = 0;
  char err[600] = "";
  if (exec_take(c1.ctx, a, v0, n, idx, ivalid, n, to, tov, &tnulls, err, sizeof err) != AH_OK) { fprintf(stderr, "cabi_driver: take failed: %s\n", err); return 3; }
  dump("take_vals.bin", to, (size_t)n * 8); dump("take_valid.bin", tov, bytes_for_bits(n)); dump("take_meta.bin", &tnulls, 8);
  {
    const int64_t pos = n / 2;
    const int32_t keep = idx[pos];
    idx[pos] = (int32_t)n + 7;
    const int rc = exec_take(c1.ctx, a, v0, n, idx, ivalid, n, to, tov, &tnulls, err, sizeof err);
    idx[pos] = keep;
    char line[700];
    snprintf(line, sizeof line, "%d|%s|%lld", rc, err, (long long)(n + 7));
    dump("take_error.txt", line, strlen(line));
  }
  /* dictionary_encode */
  int32_t* ids = (int32_t*)go_allocate((size_t)n * 4); uint8_t* idv = (uint8_t*)go_allocate(bytes_for_bits(n));
  uint64_t* dict; int64_t ndict;
  exec_dictionary_encode(c1.ctx, keys, v0, n, ids, idv, &dict, &ndict);
  dump("enc_ids.bin", ids, (size_t)n * 4); dump("enc_ids_valid.bin", idv, bytes_for_bits(n)); dump("enc_dict.bin", dict, (size_t)ndict * 8);
  /* two executors at the same time, a context each */
  int64_t* sum2 = (int64_t*)go_allocate((size_t)n * 8);
  uint8_t* gt2 = (uint8_t*)go_allocate(bytes_for_bits(5 + n));
  par_add pa = {c1.ctx, a, b, sum2, n};
  par_cmp pc = {c2.ctx, a, gt2, n};
  pthread_t t1, t2;
  pthread_create(&t1, NULL, par_add_main, &pa);
  pthread_create(&t2, NULL, par_cmp_main, &pc);
  pthread_join(t1, NULL); pthread_join(t2, NULL);
  dump("par_add.bin", sum2, (size_t)n * 8); dump("par_gt.bin", gt2, bytes_for_bits(5 + n));
  hop(step_destroy, c1.ctx);
  hop(step_destroy, c2.ctx);
  printf("cabi_driver ok: %lld rows, %lu steps each on its own pthread\n", (long long)n, g_threads_used);
  return 0;
}
