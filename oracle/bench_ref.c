/*
 * bench_ref.c — config C1 on the CPU: BenchmarkFloat64Funcs_Sum_8192 (arrow/math/float64_test.go:72-86)
 * re-timed on this host with the reference's own kernels, in C so that no FFI overhead is in
 * the loop: sum_float64_avx2 (the AVX2 path's machine code) and sum_float64_x86 (the C source
 * compiled strict-sequential = the noasm order, arrow/math/float64.go:41-47).
 * TEST INFRASTRUCTURE (baseline measurement only).   cc -O2 bench_ref.c -ldl -o _ref/bench_ref
 */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

typedef void (*sumf_t)(double*, size_t, double*);
typedef void (*sumi_t)(long long*, size_t, long long*);

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }

int main(int argc, char** argv) {
  const char* dir = argc > 1 ? argv[1] : "_ref";
  char p1[512], p2[512];
  snprintf(p1, sizeof p1, "%s/libref_avx2.so", dir);
  snprintf(p2, sizeof p2, "%s/libref_c.so", dir);
  void* a = dlopen(p1, RTLD_NOW); void* c = dlopen(p2, RTLD_NOW);
  if (!a || !c) { fprintf(stderr, "bench_ref: %s\n", dlerror()); return 1; }
  sumf_t favx = (sumf_t)dlsym(a, "sum_float64_avx2"), fseq = (sumf_t)dlsym(c, "sum_float64_x86");
  sumi_t iavx = (sumi_t)dlsym(a, "sum_int64_avx2"), iseq = (sumi_t)dlsym(c, "sum_int64_x86");
  size_t sizes[] = {256, 1024, 8192, 1000000};
  printf("{");
  for (int s = 0; s < 4; s++) {
    size_t n = sizes[s];
    double* x = malloc(n * 8); long long* xi = malloc(n * 8);
    for (size_t i = 0; i < n; i++) { x[i] = (double)i; xi[i] = (long long)i; }   /* makeArrayFloat64: buf[i] = i */
    long iters = (long)(2e8 / (double)n) + 10;
    double r; long long ri; volatile double sink = 0;
    struct { const char* name; int kind; } k[] = {{"Float64_Sum_avx2", 0}, {"Float64_Sum_noasm_order", 1}, {"Int64_Sum_avx2", 2}, {"Int64_Sum_noasm_order", 3}};
    for (int j = 0; j < 4; j++) {
      double t0 = now();
      for (long it = 0; it < iters; it++) {
        switch (k[j].kind) { case 0: favx(x, n, &r); sink += r; break; case 1: fseq(x, n, &r); sink += r; break;
                             case 2: iavx(xi, n, &ri); sink += ri; break; default: iseq(xi, n, &ri); sink += ri; }
      }
      double dt = (now() - t0) / iters;
      printf("%s\"%s_%zu\": {\"ns_per_op\": %.1f, \"MB/s\": %.2f}", (s || j) ? ", " : "", k[j].name, n, dt * 1e9, n * 8 / dt / 1e6);
    }
    free(x); free(xi);
  }
  printf("}\n");
  return 0;
}
