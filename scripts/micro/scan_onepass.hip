// One-pass cumulative_sum (Int64, unchecked, no nulls) with LARGE tiles — a measurement of the design DESIGN.md §3.4 argues about:
// decoupled look-back where a workgroup parks 128 KiB (or 256 KiB) of the column in registers, so that a generation of resident
// tiles is 32–64 MiB of streaming (5–10 µs) against the ≈ 3 µs cross-XCD hand-offs on its critical path.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 scan_onepass.hip -o /tmp/scan_onepass && /tmp/scan_onepass [log2 rows]
// Prints ms per call for the variants (VPT = 16-byte vectors per lane) and checks the result against a CPU prefix sum.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

// records: per tile 4 words {marker|lo32 of aggregate, marker|hi32, marker'|lo32 of inclusive, marker'|hi32}; marker = epoch·4 + state
__device__ __forceinline__ void rec_store(u64* rec, unsigned marker, u64 v) {
  __hip_atomic_store(&rec[0], ((u64)marker << 32) | (v & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(&rec[1], ((u64)marker << 32) | (v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool rec_load(const u64* rec, unsigned marker, u64* v) {
  const u64 a = __hip_atomic_load(&rec[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const u64 b = __hip_atomic_load(&rec[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((unsigned)(a >> 32) != marker || (unsigned)(b >> 32) != marker) return false;
  *v = (a & 0xffffffffull) | (b << 32);
  return true;
}

// 64-lane inclusive scan of a 64-bit value with DPP moves (row_shr 1, 2, 4, 8 inside the rows of 16; row_bcast 15 / 31 across
// them) instead of six ds_bpermute round trips through the LDS crossbar per half.  A lane without a source keeps 0.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ u64 dpp_move(u64 v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, ROW_MASK, 0xF, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, ROW_MASK, 0xF, false);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 wave_incl_scan(u64 v) {
  v += dpp_move<0x111, 0xF>(v);   // row_shr:1
  v += dpp_move<0x112, 0xF>(v);   // row_shr:2
  v += dpp_move<0x114, 0xF>(v);   // row_shr:4
  v += dpp_move<0x118, 0xF>(v);   // row_shr:8
  v += dpp_move<0x142, 0xA>(v);   // row_bcast:15 → rows 1 and 3
  v += dpp_move<0x143, 0xC>(v);   // row_bcast:31 → rows 2 and 3
  return v;
}
__device__ __forceinline__ u64 read_lane63(u64 v) {
  return ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63) << 32) | (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
}

template <int VPT, int WPE, int kThreads>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void scan_kernel(const u64* __restrict__ in, u64* __restrict__ out, int64_t n, u64 start, u64* __restrict__ recs,
                                                         unsigned* __restrict__ ticket, unsigned epoch) {
  constexpr int TILE = kThreads * VPT * 2;   // rows
  __shared__ u64 s_wave[kThreads / 64];
  __shared__ u64 s_prefix;
  __shared__ unsigned s_tile;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const unsigned m_agg = epoch * 4u + 1u, m_inc = epoch * 4u + 2u;
  const int64_t ntiles = (n + TILE - 1) / TILE;
  for (;;) {
    if (t == 0) s_tile = atomicAdd(ticket, 1u);
    __syncthreads();
    const int64_t tile = s_tile;
    if (tile >= ntiles) return;
    const int64_t wbase = tile * TILE + (int64_t)wave * (64 * VPT * 2);
    u64x2 x[VPT];
#pragma unroll
    for (int k = 0; k < VPT; k++) {
      const int64_t e = wbase + ((int64_t)k * 64 + lane) * 2;
      if (e + 2 <= n) x[k] = __builtin_nontemporal_load((const u64x2*)(in + e));
      else { x[k].x = e < n ? in[e] : 0; x[k].y = 0; }
    }
    // per-vector sums → 64-lane inclusive scans chained over the VPT vectors of the wave
    u64 run = 0;          // sum of the wave's vectors in front
    // in place: x[k] becomes the inclusive prefix of its two rows inside the wave chunk (no second array: ≤ 64 registers = two
    // workgroups per CU, so that one streams while the other waits for its look-back)
#pragma unroll
    for (int k = 0; k < VPT; k++) {
      const u64 s = x[k].x + x[k].y, inc = wave_incl_scan(s);
      const u64 pre = run + inc - s;
      run += read_lane63(inc);
      x[k].x += pre;
      x[k].y += x[k].x;
    }
    if (lane == 0) s_wave[wave] = run;
    __syncthreads();
    u64 wpre = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kThreads / 64; w++) { const u64 v = s_wave[w]; if (w < wave) wpre += v; total += v; }
    // publish the aggregate, look back (wave 0), publish the inclusive prefix.  (All sixteen waves looking back together — 1024
    // predecessors in one round — was measured: 0.477 ms against 0.447: the extra polling and barriers cost more than the rounds.)
    if (wave == 0) {
      u64* mine = recs + (size_t)tile * 4;
      u64 excl = start;
      if (tile == 0) {
        if (lane == 0) rec_store(mine + 2, m_inc, start + total);
      } else {
        if (lane == 0) rec_store(mine, m_agg, total);
        u64 acc = 0;
        for (int64_t back = tile - 1;; back -= 64) {
          const int64_t p = back - lane;
          u64 v = 0;
          int kind = 0;   // 0 nothing there (before tile 0), 1 aggregate, 2 inclusive
          if (p >= 0) {
            const u64* r = recs + (size_t)p * 4;
            for (;;) {
              if (rec_load(r + 2, m_inc, &v)) { kind = 2; break; }
              if (rec_load(r, m_agg, &v)) { kind = 1; break; }
              __builtin_amdgcn_s_sleep(1);
            }
          }
          const u64 incm = __ballot(kind == 2);
          const int stop = incm ? __builtin_ctzll(incm) : 64;   // nearest predecessor with an inclusive prefix
          u64 contrib = (lane <= stop && kind != 0) ? v : 0;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) contrib += __shfl_down(contrib, o, 64);
          acc += __shfl(contrib, 0, 64);
          if (incm || back - 63 <= 0) break;
        }
        excl = acc;   // (holds `start` through tile 0's inclusive record)
        if (lane == 0) rec_store(mine + 2, m_inc, excl + total);
      }
      if (lane == 0) s_prefix = excl;
    }
    __syncthreads();
    const u64 base = s_prefix + wpre;
#pragma unroll
    for (int k = 0; k < VPT; k++) {
      const int64_t e = wbase + ((int64_t)k * 64 + lane) * 2;
      u64x2 y;
      y.x = base + x[k].x;
      y.y = base + x[k].y;
      if (e + 2 <= n) __builtin_nontemporal_store(y, (u64x2*)(out + e));
      else if (e < n) out[e] = y.x;
    }
    __syncthreads();   // s_tile / s_wave are reused
  }
}

template <int VPT, int WPE, int kThreads>
float run(const u64* din, u64* dout, int64_t n, u64* recs, unsigned* ticket, int wg_per_cu, unsigned* epoch, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const int64_t ntiles = (n + kThreads * VPT * 2 - 1) / (kThreads * VPT * 2);
  int64_t grid = 256 * wg_per_cu;
  if (grid > ntiles) grid = ntiles;
  float best = 1e9f;
  for (int r = 0; r < reps + 1; r++) {
    CK(hipMemsetAsync(ticket, 0, 4, 0));
    (*epoch)++;
    CK(hipEventRecord(a, 0));
    scan_kernel<VPT, WPE, kThreads><<<(unsigned)grid, kThreads, 0, 0>>>(din, dout, n, 7ull, recs, ticket, *epoch);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (r > 0 && ms < best) best = ms;
  }
  return best;
}

int main(int argc, char** argv) {
  const int lg = argc > 1 ? atoi(argv[1]) : 27;
  const int64_t n = ((int64_t)1 << lg) - 3;
  std::vector<u64> h((size_t)n);
  u64 s = 88172645463325252ull;
  for (int64_t i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[(size_t)i] = s; }
  u64 *din, *dout, *recs; unsigned* ticket;
  CK(hipMalloc(&din, (size_t)n * 8 + 64)); CK(hipMalloc(&dout, (size_t)n * 8 + 64));
  CK(hipMalloc(&recs, (size_t)(n / 2048 + 2) * 32)); CK(hipMalloc(&ticket, 64));
  CK(hipMemset(recs, 0, (size_t)(n / 2048 + 2) * 32));
  CK(hipMemcpy(din, h.data(), (size_t)n * 8, hipMemcpyHostToDevice));
  unsigned epoch = 0;
  std::vector<u64> ref((size_t)n), got((size_t)n);
  { u64 acc = 7; for (int64_t i = 0; i < n; i++) { acc += h[(size_t)i]; ref[(size_t)i] = acc; } }
  auto check = [&](const char* what) {
    CK(hipMemcpy(got.data(), dout, (size_t)n * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n; i++) if (got[(size_t)i] != ref[(size_t)i]) { printf("%s: MISMATCH at %lld\n", what, (long long)i); return; }
    printf("%s: ok\n", what);
  };
  printf("1024 threads x VPT  8 (128 KiB tiles) 1 wg/cu: %.4f ms\n", run<8, 4, 1024>(din, dout, n, recs, ticket, 1, &epoch, 5));
  check("1024 x 8");
  printf("1024 threads x VPT 10 (160 KiB tiles) 1 wg/cu: %.4f ms\n", run<10, 4, 1024>(din, dout, n, recs, ticket, 1, &epoch, 5));
  check("1024 x 10");
  printf("1024 threads x VPT 12 (192 KiB tiles) 1 wg/cu: %.4f ms\n", run<12, 4, 1024>(din, dout, n, recs, ticket, 1, &epoch, 5));
  check("1024 x 12");
  printf("1024 threads x VPT  6 ( 96 KiB tiles) 1 wg/cu: %.4f ms\n", run<6, 4, 1024>(din, dout, n, recs, ticket, 1, &epoch, 5));
  printf(" 512 threads x VPT  8 ( 64 KiB tiles) 2 wg/cu: %.4f ms\n", run<8, 4, 512>(din, dout, n, recs, ticket, 2, &epoch, 5));
  printf(" 512 threads x VPT 12 ( 96 KiB tiles) 2 wg/cu: %.4f ms\n", run<12, 4, 512>(din, dout, n, recs, ticket, 2, &epoch, 5));
  check("512 x 12");
  for (int lg2 = 20; lg2 <= 26; lg2 += 3) {   // smaller columns (the records and tickets are per call)
    const int64_t m = ((int64_t)1 << lg2) - 5;
    printf("2^%d rows, 1024 x VPT 8: %.4f ms\n", lg2, run<8, 4, 1024>(din, dout, m, recs, ticket, 1, &epoch, 5));
  }
  return 0;
}
