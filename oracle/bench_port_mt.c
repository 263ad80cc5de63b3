/*
 * bench_port_mt.c — CPU baselines for configs C3 and C5 (BASELINE.json configs[2], [4]): the oracle's C ports of the reference's
 * Go kernels, timed on the GPU box's host cores beside the GPU numbers.  TEST INFRASTRUCTURE (baseline measurement only; run by
 * bench.py's cpu_baseline leg, nothing else).
 *
 *   Filter   orc_filter_primitive   = PrimitiveFilter, arrow/compute/internal/kernels/vector_selection.go:267-395 (Int64 values
 *            with 10 % nulls, a Bernoulli(0.5) mask, Drop)
 *   Take     orc_take_primitive     = PrimitiveTake,   vector_selection.go:878-988 (Int64 values and Int32 indices, uniformly
 *            random, 10 % nulls on both sides, bounds check on)
 *   encode   orc_hash_u64_encode    = dictionary_encode over hashing.Table[uint64], internal/hashing/xxh3_memo_table_types.go:283-294,
 *            kernels/vector_hash.go:359-385 (2^16 distinct Int64 keys)
 *   hash_sum orc_hash_sum_f64       = that encode + row-order accumulation into the group of each row (the definition of C5's local
 *            aggregate; the reference has no group-by kernel)
 *
 * Threads.  Filter and Take are cut into contiguous shards, one per thread, every thread producing its own output — what a Go user
 * gets from a chunked column and ExecCtx.NumParallel (arrow/compute/executor.go:47-50).  The hash kernels are timed on ONE core
 * only: the reference feeds all chunks of a column through one memo table, one after the other (vector_hash.go: the kernel state
 * is shared by the spans), so more cores do not make its dictionary_encode faster.
 *
 *   cc -O2 -pthread bench_port_mt.c -ldl -o _ref/bench_port_mt ;  _ref/bench_port_mt <dir of liboracle.so> <log2 rows> <threads>…
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int (*filter_t)(int, const void*, const uint8_t*, int64_t, const uint8_t*, const uint8_t*, int64_t, int64_t, int, void*, uint8_t*, int64_t*, int64_t*);
typedef int (*take_t)(int, const void*, const uint8_t*, int64_t, int64_t, int, int, const void*, const uint8_t*, int64_t, int64_t, int, void*, uint8_t*,
                      int64_t*, int64_t*);
typedef int (*encode_t)(const uint64_t*, const uint8_t*, int64_t, int64_t, int, int32_t*, uint8_t*, uint64_t*, int64_t*, int32_t*);
typedef int (*hsum_t)(const uint64_t*, const uint8_t*, int64_t, const double*, const uint8_t*, int64_t, int64_t, uint64_t*, double*, int64_t*, int64_t*,
                      int64_t*, int32_t*);

static filter_t f_filter; static take_t f_take; static encode_t f_encode; static hsum_t f_hsum;
static size_t g_n; static int g_threads, g_kind;
static int64_t* g_vals; static uint8_t *g_vvalid, *g_mask, *g_ivalid;
static int32_t* g_idx;
static int64_t* g_out; static uint8_t* g_ovalid;
static int64_t g_outlen[1024];
static pthread_barrier_t g_bar;

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd(void) { uint64_t x = rng_state; x ^= x << 13; x ^= x >> 7; x ^= x << 17; return rng_state = x; }
static void fill_bits(uint8_t* b, size_t n, double p) {
  memset(b, 0, n / 8 + 8);
  const uint64_t thr = (uint64_t)(p * 18446744073709551615.0);
  for (size_t i = 0; i < n; i++) if (rnd() < thr) b[i >> 3] |= (uint8_t)(1u << (i & 7));
}

static void* worker(void* arg) {
  const int id = (int)(intptr_t)arg;
  size_t lo = g_n / g_threads * id, hi = id == g_threads - 1 ? g_n : g_n / g_threads * (id + 1);
  lo &= ~(size_t)63; if (id != g_threads - 1) hi &= ~(size_t)63;   /* shards start on whole bitmap words */
  for (int rep = 0; rep < 4; rep++) {   /* rep 0 = warm-up */
    pthread_barrier_wait(&g_bar);
    int64_t nulls = 0, bad = -1;
    if (g_kind == 0) f_filter(8, g_vals + lo, g_vvalid + lo / 8, 0, g_mask + lo / 8, NULL, 0, (int64_t)(hi - lo), 0, g_out + lo, g_ovalid + lo / 8, &g_outlen[id], &nulls);
    else f_take(8, g_vals, g_vvalid, 0, (int64_t)g_n, 4, 1, g_idx + lo, g_ivalid + lo / 8, 0, (int64_t)(hi - lo), 1, g_out + lo, g_ovalid + lo / 8, &nulls, &bad);
    pthread_barrier_wait(&g_bar);
  }
  return 0;
}

int main(int argc, char** argv) {
  const char* dir = argc > 1 ? argv[1] : ".";
  const int lg = argc > 2 ? atoi(argv[2]) : 24;
  char p1[512];
  snprintf(p1, sizeof p1, "%s/liboracle.so", dir);
  void* h = dlopen(p1, RTLD_NOW);
  if (!h) { fprintf(stderr, "bench_port_mt: %s\n", dlerror()); return 1; }
  f_filter = (filter_t)dlsym(h, "orc_filter_primitive");
  f_take = (take_t)dlsym(h, "orc_take_primitive");
  f_encode = (encode_t)dlsym(h, "orc_hash_u64_encode");
  f_hsum = (hsum_t)dlsym(h, "orc_hash_sum_f64");
  if (!f_filter || !f_take || !f_encode || !f_hsum) { fprintf(stderr, "bench_port_mt: missing symbol\n"); return 1; }
  if (lg < 10 || lg > 28) { fprintf(stderr, "bench_port_mt: log2 rows must be 10..28\n"); return 1; }
  g_n = (size_t)1 << lg;
  g_vals = aligned_alloc(64, g_n * 8); g_out = aligned_alloc(64, g_n * 8); g_idx = aligned_alloc(64, g_n * 4);
  g_vvalid = aligned_alloc(64, g_n / 8 + 64); g_mask = aligned_alloc(64, g_n / 8 + 64); g_ivalid = aligned_alloc(64, g_n / 8 + 64);
  g_ovalid = aligned_alloc(64, g_n / 8 + 64);
  for (size_t i = 0; i < g_n; i++) { g_vals[i] = (int64_t)rnd(); g_idx[i] = (int32_t)(rnd() % g_n); g_out[i] = 0; }
  fill_bits(g_vvalid, g_n, 0.9); fill_bits(g_mask, g_n, 0.5); fill_bits(g_ivalid, g_n, 0.9);
  memset(g_ovalid, 0, g_n / 8 + 64);
  printf("{\"rows\": %zu", g_n);
  /* ---- C3: Filter and Take on 1 … N threads */
  const char* names[] = {"filter_int64_nulls10_sel0.50", "take_int64_random_i32_nulls10"};
  for (int ai = 3; ai < argc; ai++) {
    g_threads = atoi(argv[ai]);
    if (g_threads < 1 || g_threads > 1024 || (size_t)g_threads * 64 > g_n) continue;
    for (g_kind = 0; g_kind < 2; g_kind++) {
      pthread_t th[1024];
      pthread_barrier_init(&g_bar, 0, g_threads + 1);
      for (int t = 0; t < g_threads; t++) pthread_create(&th[t], 0, worker, (void*)(intptr_t)t);
      double best = 1e30;
      for (int rep = 0; rep < 4; rep++) {
        pthread_barrier_wait(&g_bar);
        const double t0 = now();
        pthread_barrier_wait(&g_bar);
        const double dt = now() - t0;
        if (rep > 0 && dt < best) best = dt;
      }
      for (int t = 0; t < g_threads; t++) pthread_join(th[t], 0);
      pthread_barrier_destroy(&g_bar);
      int64_t kept = 0;
      for (int t = 0; t < g_threads; t++) kept += g_outlen[t];
      /* algorithmic bytes as bench.py's GPU lines count them */
      const double bytes = g_kind == 0 ? (8 + 0.25) * (double)g_n + (8 + 0.125) * (double)kept : (20 + 0.375) * (double)g_n;
      printf(", \"%s_threads%d\": {\"ms\": %.3f, \"GB/s\": %.3f, \"input_GB/s\": %.3f, \"Mrows/s\": %.1f}", names[g_kind], g_threads, best * 1e3, bytes / best / 1e9,
             8.0 * (double)g_n / best / 1e9, (double)g_n / best / 1e6);
    }
  }
  /* ---- C5: dictionary_encode and hash + sum, 2^16 distinct keys, one core (the reference's memo table is one sequential structure) */
  {
    uint64_t* keys = (uint64_t*)g_out;   /* reuse */
    for (size_t i = 0; i < g_n; i++) keys[i] = (rnd() & 0xFFFFu) * 0x9E3779B97F4A7C15ull;
    double* fv = (double*)g_vals;
    for (size_t i = 0; i < g_n; i++) fv[i] = (double)(int64_t)(rnd() >> 40) * 0.125 - 1000.0;
    int32_t* ids = (int32_t*)g_idx;
    uint64_t* dict = aligned_alloc(64, ((size_t)1 << 16) * 8 + 4096);
    double* sums = aligned_alloc(64, ((size_t)1 << 16) * 8 + 4096);
    int64_t* counts = aligned_alloc(64, ((size_t)1 << 16) * 8 + 4096);
    int64_t* firsts = aligned_alloc(64, ((size_t)1 << 16) * 8 + 4096);
    int64_t nd = 0, ng = 0; int32_t nullid = -1;
    double best_e = 1e30, best_h = 1e30;
    for (int rep = 0; rep < 3; rep++) {
      double t0 = now();
      f_encode(keys, NULL, 0, (int64_t)g_n, 0, ids, NULL, dict, &nd, &nullid);
      double dt = now() - t0;
      if (rep > 0 && dt < best_e) best_e = dt;
      t0 = now();
      f_hsum(keys, NULL, 0, fv, NULL, 0, (int64_t)g_n, dict, sums, counts, firsts, &ng, &nullid);
      dt = now() - t0;
      if (rep > 0 && dt < best_h) best_h = dt;
    }
    if (nd != ng || nd > 65536) { fprintf(stderr, "bench_port_mt: encode / hash_sum disagree (%lld vs %lld groups)\n", (long long)nd, (long long)ng); return 2; }
    printf(", \"dictionary_encode_int64_2^16_keys_threads1\": {\"ms\": %.3f, \"GB/s\": %.3f, \"Mrows/s\": %.1f, \"keys\": %lld}", best_e * 1e3,
           12.0 * (double)g_n / best_e / 1e9, (double)g_n / best_e / 1e6, (long long)nd);
    printf(", \"hash_sum_float64_2^16_groups_threads1\": {\"ms\": %.3f, \"GB/s\": %.3f, \"Mrows/s\": %.1f, \"groups\": %lld}", best_h * 1e3,
           16.0 * (double)g_n / best_h / 1e9, (double)g_n / best_h / 1e6, (long long)ng);
  }
  printf("}\n");
  return 0;
}
