/*
 * bench_ref_mt.c — the reference's AVX2 kernels on N host threads ("1 thread and
 * N = nproc threads, each over a contiguous 1/N shard with a final combine — what a user gets by
 * chunking + NumParallel").  1 GiB columns (2^27 rows): Float64 Sum, Int64 Add arr+arr,
 * greater(int64, scalar).  TEST INFRASTRUCTURE (baseline measurement only).
 *   cc -O2 -pthread bench_ref_mt.c -ldl -o _ref/bench_ref_mt ;  _ref/bench_ref_mt _ref <threads>…
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef void (*sumf_t)(double*, size_t, double*);
typedef void (*arith_t)(int, int8_t, const void*, const void*, void*, int);
typedef void (*cmp_t)(int, const void*, const void*, void*, int, int);

static sumf_t f_sum; static arith_t f_add; static cmp_t f_gt;
static double* g_x; static int64_t *g_a, *g_b, *g_c; static uint8_t* g_m;
static size_t g_n; static int g_threads, g_kind;
static pthread_barrier_t g_bar;
static double g_partial[1024];

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

static void* worker(void* arg) {
  int id = (int)(intptr_t)arg;
  size_t lo = g_n / g_threads * id, hi = id == g_threads - 1 ? g_n : g_n / g_threads * (id + 1);
  lo &= ~(size_t)63; if (id != g_threads - 1) hi &= ~(size_t)63;   /* shards start on 64-row (8-byte bitmap) boundaries */
  for (int rep = 0; rep < 4; rep++) {   /* rep 0 = warm-up (page faults, clocks) */
    pthread_barrier_wait(&g_bar);
    /* the leaves take a 32-bit length (_lib/base_arithmetic.cc:238): a shard is ≤ 2^27 rows here */
    if (g_kind == 0) f_sum(g_x + lo, hi - lo, &g_partial[id]);
    else if (g_kind == 1) f_add(9 /*INT64*/, 0 /*ADD*/, g_a + lo, g_b + lo, g_c + lo, (int)(hi - lo));
    else f_gt(9, g_a + lo, g_b /*scalar*/, g_m + lo / 8, (int)(hi - lo), 0);
    pthread_barrier_wait(&g_bar);
  }
  return 0;
}

int main(int argc, char** argv) {
  const char* dir = argc > 1 ? argv[1] : "_ref";
  char p1[512];
  snprintf(p1, sizeof p1, "%s/libref_avx2.so", dir);
  void* a = dlopen(p1, RTLD_NOW);
  if (!a) { fprintf(stderr, "bench_ref_mt: %s\n", dlerror()); return 1; }
  f_sum = (sumf_t)dlsym(a, "sum_float64_avx2");
  f_add = (arith_t)dlsym(a, "arithmetic_binary_avx2");
  f_gt = (cmp_t)dlsym(a, "comparison_greater_arr_scalar_avx2");
  if (!f_sum || !f_add || !f_gt) { fprintf(stderr, "bench_ref_mt: missing symbol\n"); return 1; }
  g_n = (size_t)1 << 27;
  g_x = aligned_alloc(64, g_n * 8); g_a = aligned_alloc(64, g_n * 8); g_b = aligned_alloc(64, g_n * 8); g_c = aligned_alloc(64, g_n * 8);
  g_m = aligned_alloc(64, g_n / 8);
  for (size_t i = 0; i < g_n; i++) { g_x[i] = (double)(i & 1023); g_a[i] = (int64_t)(i * 2654435761u) - (1ll << 31); g_b[i] = (int64_t)i; g_c[i] = 0; }
  memset(g_m, 0, g_n / 8);
  const char* names[] = {"Float64_Sum_1GiB", "Int64_Add_1GiB", "Int64_greater_scalar_1GiB"};
  const double bytes[] = {8.0, 24.0, 8.125};
  printf("{");
  int first = 1;
  for (int ai = 2; ai < argc; ai++) {
    g_threads = atoi(argv[ai]);
    if (g_threads < 1 || g_threads > 1024) continue;
    for (g_kind = 0; g_kind < 3; g_kind++) {
      pthread_t th[1024];
      pthread_barrier_init(&g_bar, 0, g_threads + 1);
      for (int t = 0; t < g_threads; t++) pthread_create(&th[t], 0, worker, (void*)(intptr_t)t);
      double best = 1e30;
      for (int rep = 0; rep < 4; rep++) {
        pthread_barrier_wait(&g_bar);
        double t0 = now();
        pthread_barrier_wait(&g_bar);
        double dt = now() - t0;
        if (rep > 0 && dt < best) best = dt;
      }
      for (int t = 0; t < g_threads; t++) pthread_join(th[t], 0);
      pthread_barrier_destroy(&g_bar);
      double total = 0;  /* the final combine of the shard partials */
      for (int t = 0; t < g_threads; t++) total += g_partial[t];
      printf("%s\"%s_threads%d\": {\"ms\": %.3f, \"GB/s\": %.1f}", first ? "" : ", ", names[g_kind], g_threads, best * 1e3, g_n * bytes[g_kind] / best / 1e9);
      first = 0;
      if (g_kind == 0 && total < 0) return 2;
    }
  }
  printf("}\n");
  return 0;
}
