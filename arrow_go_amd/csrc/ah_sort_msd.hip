// ah_sort_msd.hip — the `rest` range of sort_indices for LARGE inputs with (mostly) distinct keys: two MSD partition
// passes into ≈ n / 64 buckets, then every bucket sorted by one wave in registers.
//
// Same contract as the LSD passes of ah_sort.hip (arraySortOneColumnRange, kernels/vector_sort_internal.go:252-273 →
// slices.SortStableFunc): the rows ordered by key, ties in input order.  Input: (order-preserving 64-bit key, row)
// pairs in row order, as pass (1) of ah_sort.hip leaves them.  Eight stable 8-bit LSD passes move ≈ 290 B/row at
// ≈ 2.4 TB/s (15 ms for 2^27 Int64 rows): each pass pays for its stability with an 8-ballot match-any per row.  Here
// nothing has to be stable, because the last step compares (key, row) pairs and rows are distinct:
//
//   0 map        a strided sample of 2^18 keys, sorted (the LSD passes, on 1/512 of the data) → 4095 quantile splitters: key →
//                bucket = interval · B/4096 + linear position inside the interval — a monotone map that gives every bucket
//                about the same number of rows whatever the distribution (doubles drawn from a normal distribution fill
//                20 of the 1024 top-bit digits and leave the rest empty)
//   1 bucket     one streaming pass: bucket id of every row (binary search over the splitters in LDS)
//   2 level 1    partition by bucket >> log2(NB2) into NB1 ≤ 1024 parents           (hist → offsets → LDS-staged scatter,
//   3 level 2    every parent partitioned by bucket & (NB2 − 1), NB2 ≤ 2048           the machinery of ah_bins.h; ranks inside
//                — its tiles never cross a parent boundary                            a tile from LDS atomics: NOT stable)
//   4 local      one wave per bucket (≈ 64 rows expected, 512 at most): bitonic network over (key, row) in registers
//
// ≈ 100 B/row of streaming traffic.  Buckets of 513 … 8192 rows go to a workgroup each (LDS); a bucket above that (many equal keys) makes the whole attempt void: the caller regenerates the pairs and runs the LSD passes, which
// accept anything.  The result does not depend on which path ran (a stable sort has exactly one answer).
#include "ah_common.h"
#include "ah_bins.h"
#include "ah_msd.h"

namespace {

constexpr int kLocalMax = 512;                  // rows one wave sorts (64 expected; 8 per lane at most)
constexpr int kBigMax = 8192, kBigList = 1024;  // rows one workgroup sorts in LDS; such buckets per call

struct MsMap {                 // u = (key & mask) << lshift: the varying bits, most significant at bit 63
  unsigned long long mask;
  int lshift;
  int fkind, descending;       // float column (4 / 8 bytes, 0 = integers) and its order: how to turn a key back into a value
};
// Position of u between two break points, 0 … 1.  Linear in the KEY — except across zero of a float column: there the key space
// passes through every exponent (−2^-1000 … +2^-1000 lie between the two break points) while the rows, dense in VALUE near zero,
// all sit at the two ends; so a piece whose ends differ in sign is interpolated in value space.
__device__ __forceinline__ double ms_value(unsigned long long u, const MsMap& m) {
  if (m.fkind == 8) {
    const unsigned long long k = m.descending ? ~u : u;
    return __builtin_bit_cast(double, (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k);
  }
  const unsigned k = m.descending ? ~(unsigned)(u >> 32) : (unsigned)(u >> 32);
  return (double)__builtin_bit_cast(float, (k >> 31) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ double ms_frac(unsigned long long u, unsigned long long pa, unsigned long long pb, const MsMap& m) {
  if (pb <= pa) return 0.0;
  if (m.fkind && m.lshift == (m.fkind == 8 ? 0 : 32) && ((pa ^ pb) >> 63)) {
    const double xa = ms_value(pa, m), xb = ms_value(pb, m);
    if (xb != xa && xa - xa == 0.0 && xb - xb == 0.0) return (ms_value(u, m) - xa) / (xb - xa);   // both ends finite (±inf are keys too)
  }
  return (double)(u - pa) / (double)(pb - pa);
}
constexpr int kSplit = 4096;                    // intervals of the bucket map
constexpr int64_t kSample = (int64_t)1 << 18;   // 64 sampled keys per interval

// ---- 0: the map --------------------------------------------------------------------------------------------------------
// Sample quantiles: the sorted sample cut into 4096 equal parts gives splitters that hold ≈ n / 4096 rows between neighbours
// whatever the distribution; inside an interval the bucket is linear in the key.  (A table over the top 12 key bits was tried
// first: fine for uniform and log-normal columns, but a normal distribution's density changes 400-fold inside the binade
// [2, 4) and 500 of 2^18 buckets came out above 256 rows.)
__global__ __launch_bounds__(256) void ms_sample_kernel(const unsigned long long* __restrict__ keys, int64_t n, int64_t stride, int64_t count, MsMap m,
                                                         unsigned long long* __restrict__ sample, unsigned* __restrict__ sample_rows) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= count) return;
  const int64_t i = j * stride;
  sample[j] = i < n ? (keys[i] & m.mask) << m.lshift : ~0ull;
  sample_rows[j] = (unsigned)j;
}
// The break points of the map: P[0] = smallest key, P[1 + i] = sorted sample, P[S + 1] = largest key.  split[k] = P[64 k + 1]
// (k = 1 … 4095), split[0] = P[0], split[4096] = P[S + 1]: neighbours hold ≈ n / 4096 rows between them.  Inside an interval the
// bucket is linear in the key — unless the interval's own 64 sample points say that is far from true (flag[k]): the interval
// that contains zero of a float column spans every exponent below its ends, the outermost intervals hold the tails.  Rows of
// flagged intervals (a fraction of a per cent) look their position up among those 64 points (ms_bucket_kernel).
struct MsPoints {
  const unsigned long long* sorted;
  int64_t count;                 // S
  unsigned long long umin, umax;
  __device__ __forceinline__ unsigned long long at(int64_t i) const { return i <= 0 ? umin : (i > count ? umax : sorted[i - 1]); }
  __device__ __forceinline__ int64_t first(int k) const { return k == 0 ? 0 : 64 * (int64_t)k + 1; }          // P index of split[k]
  __device__ __forceinline__ int64_t last(int k) const { return k == kSplit - 1 ? count + 1 : 64 * (int64_t)(k + 1) + 1; }
};
constexpr int kFlagWords = kSplit / 32;
__global__ __launch_bounds__(256) void ms_split_kernel(MsPoints p, unsigned long long* __restrict__ split, unsigned* __restrict__ flags) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k > kSplit) return;
  split[k] = k == kSplit ? p.umax : p.at(p.first(k));
  if (k == kSplit) return;
  const int64_t i0 = p.first(k), i1 = p.last(k);
  const unsigned long long a = p.at(i0), b = p.at(i1);
  bool flag = k == 0 || k == kSplit - 1;
  if (!flag && b > a) {
    const double w = (double)(b - a), g = (double)(i1 - i0);
    for (int64_t i = i0 + 1; i < i1; i++) {
      const double dev = (double)(p.at(i) - a) / w - (double)(i - i0) / g;
      flag = flag || dev > 0.25 || dev < -0.25;   // sampling noise of 64 points stays below ≈ 0.2
    }
  }
  if (flag) atomicOr(&flags[k >> 5], 1u << (k & 31));
}

// A key repeated so often that its bucket would outgrow a workgroup's LDS shows in the sorted sample as a run of equal values:
// `run` consecutive sample points ≈ run · n / S rows.  Found before the expensive passes, so that a low-cardinality column costs
// the attempt 0.3 ms instead of 3.
__global__ __launch_bounds__(256) void ms_dupes_kernel(const unsigned long long* __restrict__ sorted, int64_t count, int64_t run, unsigned* __restrict__ found) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i + run < count && sorted[i] == sorted[i + run]) *found = 1u;
}

// guide[c] = largest k with split[k] ≤ c · 2^52 (0 if none): a key whose top 12 bits are c has its interval in
// [guide[c], guide[c + 1]] — one or two candidates for evenly spread keys instead of twelve halving steps
__global__ __launch_bounds__(256) void ms_guide_kernel(const unsigned long long* __restrict__ split, unsigned short* __restrict__ guide) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c > 4096) return;
  if (c == 4096) { guide[c] = (unsigned short)(kSplit - 1); return; }
  const unsigned long long v = (unsigned long long)c << 52;
  int lo = 0;
  for (int step = kSplit >> 1; step > 0; step >>= 1) lo += split[lo + step] <= v ? step : 0;
  guide[c] = (unsigned short)lo;
}

// ---- 1: bucket ids ----------------------------------------------------------------------------------------------------
// 1024 threads: the 41 KB of tables are shared by 16 waves, two workgroups fill a CU (with 256 threads three of them did: 12 waves
// on dependent LDS reads — 0.83 ms for 2^27 rows)
__global__ __launch_bounds__(kThreads) void ms_bucket_kernel(const unsigned long long* __restrict__ keys, int64_t n, MsMap m, MsPoints p,
                                                         const unsigned long long* __restrict__ split, const unsigned* __restrict__ flags,
                                                         const unsigned short* __restrict__ guide, unsigned per, unsigned* __restrict__ bucket) {
  __shared__ unsigned long long s_s[kSplit + 1];
  __shared__ unsigned s_f[kFlagWords];
  __shared__ unsigned short s_g[4097];
  for (int i = threadIdx.x; i <= kSplit; i += kThreads) { s_s[i] = split[i]; s_g[i] = guide[i]; }
  for (int i = threadIdx.x; i < kFlagWords; i += kThreads) s_f[i] = flags[i];
  __syncthreads();
  constexpr int U = 4;
  const int64_t stride = (int64_t)gridDim.x * kThreads * U;
  for (int64_t base = (int64_t)blockIdx.x * kThreads * U + threadIdx.x; base < n; base += stride) {
    unsigned long long u[U];
    unsigned lo[U], hi[U];
#pragma unroll
    for (int q = 0; q < U; q++) {
      const int64_t i = base + q * kThreads;
      u[q] = i < n ? (__builtin_nontemporal_load(&keys[i]) & m.mask) << m.lshift : s_s[0];
    }
    // largest k with split[k] ≤ u, between the guide's two candidates; the four searches of a lane side by side
#pragma unroll
    for (int q = 0; q < U; q++) { lo[q] = s_g[u[q] >> 52]; hi[q] = s_g[(u[q] >> 52) + 1]; }
    bool more = true;
    while (more) {
      more = false;
#pragma unroll
      for (int q = 0; q < U; q++) {
        if (lo[q] < hi[q]) {
          const unsigned mid = (lo[q] + hi[q] + 1) >> 1;
          if (s_s[mid] <= u[q]) lo[q] = mid; else hi[q] = mid - 1;
          more = more || lo[q] < hi[q];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < U; q++) {
      const int64_t i = base + q * kThreads;
      if (i >= n) continue;
      double f;   // position inside the interval in units of its probability mass, 0 ≤ f ≤ 1
      if ((s_f[lo[q] >> 5] >> (lo[q] & 31)) & 1u) {
        // descend through the interval's sample points by halving the RANK range.  Down to pieces of 16 points always (fewer
        // would follow the sample's noise: the gaps between neighbouring order statistics vary 5-fold); further only where the
        // piece's middle point lies far from the middle of its key range, i.e. where the key scale really changes inside it
        int64_t a = p.first((int)lo[q]), b = p.last((int)lo[q]);
        const double g = (double)(b - a);
        const int64_t a0 = a;
        unsigned long long pa = p.at(a), pb = p.at(b);
        while (b - a > 1) {
          const int64_t mid = (a + b) >> 1;
          const unsigned long long pm = p.at(mid);
          if (b - a <= 16) {
            const double d = ms_frac(pm, pa, pb, m) - (double)(mid - a) / (double)(b - a);
            if (d < 0.3 && d > -0.3) break;
          }
          if (pm <= u[q]) { a = mid; pa = pm; } else { b = mid; pb = pm; }
        }
        f = ((double)(a - a0) + ms_frac(u[q], pa, pb, m) * (double)(b - a)) / g;
      } else {
        // (u − x) · 1/(y − x): the hardware reciprocal is a few ulp off, but a CONSTANT factor per interval keeps the map monotone,
        // which is all the sort needs; the correctly rounded division was a third of this kernel's instructions
        const unsigned long long x = s_s[lo[q]], y = s_s[lo[q] + 1];
        f = y > x ? (double)(u[q] - x) * __builtin_amdgcn_rcp((double)(y - x)) : 0.0;
      }
      f = f >= 0.0 ? f : 0.0;   // (also catches a NaN)
      unsigned in = (unsigned)(f * (double)per);
      in = in < per ? in : per - 1;
      __builtin_nontemporal_store(lo[q] * per + in, &bucket[i]);
    }
  }
}

// ---- hist: per (tile, digit) counts.  digit = (bucket >> shift) & mask --------------------------------------------------
__global__ __launch_bounds__(kThreads) void ms_hist_kernel(const unsigned* __restrict__ bucket, int64_t n, const unsigned* __restrict__ pstart, int nparents,
                                                            int shift, unsigned mask, int nb, unsigned* __restrict__ cnt) {
  __shared__ unsigned s_h[kMaxNb2];
  __shared__ unsigned s_cnt[kThreads], s_start[kThreads], s_wsum[kThreads / 64];
  __shared__ int s_pick;
  const TileRange r = ms_tile(pstart, nparents, n, s_cnt, s_start, s_wsum, &s_pick);
  if (r.parent < 0) return;
  for (int b = threadIdx.x; b < nb; b += kThreads) s_h[b] = 0;
  __syncthreads();
  unsigned bk[kMsRows];
#pragma unroll
  for (int u = 0; u < kMsRows; u++) { const int64_t i = r.lo + u * kThreads + threadIdx.x; bk[u] = i < r.hi ? __builtin_nontemporal_load(&bucket[i]) : 0u; }
#pragma unroll
  for (int u = 0; u < kMsRows; u++) { const int64_t i = r.lo + u * kThreads + threadIdx.x; if (i < r.hi) atomicAdd(&s_h[(bk[u] >> shift) & mask], 1u); }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += kThreads) cnt[r.id * nb + b] = s_h[b];
}

// ---- scatter: the tile staged in digit order in LDS, written as runs ----------------------------------------------------
// WITH_BUCKET: the bucket ids travel with the pairs (level 1); level 2 drops them
template <bool WITH_BUCKET>
__global__ __launch_bounds__(kThreads) void ms_scatter_kernel(const unsigned long long* __restrict__ keys, const unsigned* __restrict__ rows,
                                                               const unsigned* __restrict__ bucket, int64_t n, const unsigned* __restrict__ pstart,
                                                               int nparents, int shift, unsigned mask, int nb, const unsigned* __restrict__ toffs,
                                                               unsigned long long* __restrict__ out_keys, unsigned* __restrict__ out_rows,
                                                               unsigned* __restrict__ out_bucket) {
  __shared__ unsigned s_cnt[kMaxNb2], s_start[kMaxNb2], s_goff[kMaxNb2], s_wsum[kThreads / 64];
  __shared__ unsigned s_a[kThreads], s_b[kThreads];
  __shared__ unsigned long long s_stage[kMsTile];
  __shared__ uint16_t s_bin[kMsTile];
  __shared__ int s_pick;
  __shared__ unsigned s_carry;
  const TileRange r = ms_tile(pstart, nparents, n, s_a, s_b, s_wsum, &s_pick);
  if (r.parent < 0) return;
  const int t = threadIdx.x;
  for (int b = t; b < nb; b += kThreads) s_cnt[b] = 0;
  unsigned long long k[kMsRows];
  unsigned rw[kMsRows], bk[kMsRows], dg[kMsRows], rank[kMsRows];
  bool live[kMsRows];
#pragma unroll
  for (int u = 0; u < kMsRows; u++) {
    const int64_t i = r.lo + u * kThreads + t;
    live[u] = i < r.hi;
    k[u] = live[u] ? __builtin_nontemporal_load(&keys[i]) : 0ull;
    rw[u] = live[u] ? __builtin_nontemporal_load(&rows[i]) : 0u;
    bk[u] = live[u] ? __builtin_nontemporal_load(&bucket[i]) : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kMsRows; u++) {
    dg[u] = (bk[u] >> shift) & mask;
    rank[u] = live[u] ? atomicAdd(&s_cnt[dg[u]], 1u) : 0u;
  }
  __syncthreads();
  // exclusive scan over nb ≤ 2048 digit counts, 1024 at a time
  unsigned carry = 0;
  for (int h = 0; h * kThreads < nb; h++) {
    s_a[t] = t + h * kThreads < nb ? s_cnt[t + h * kThreads] : 0u;
    __syncthreads();
    block_excl_scan(s_a, s_b, s_wsum, kThreads);
    if (t + h * kThreads < nb) {
      const unsigned st = carry + s_b[t];
      s_start[t + h * kThreads] = st;
      s_goff[t + h * kThreads] = toffs[r.id * nb + t + h * kThreads] - st;   // global position = s_goff[digit] + staged position
    }
    if (t == kThreads - 1) s_carry = s_b[t] + s_a[t];
    __syncthreads();
    carry += s_carry;
    __syncthreads();
  }
  const int tile_n = (int)(r.hi - r.lo);
#pragma unroll
  for (int u = 0; u < kMsRows; u++)
    if (live[u]) { const unsigned q = s_start[dg[u]] + rank[u]; s_stage[q] = k[u]; s_bin[q] = (uint16_t)dg[u]; }
  __syncthreads();
  int64_t dst[kMsRows];
#pragma unroll
  for (int u = 0; u < kMsRows; u++) {
    const int q = u * kThreads + t;
    dst[u] = q < tile_n ? (int64_t)s_goff[s_bin[q]] + q : -1;
    if (dst[u] >= 0) out_keys[dst[u]] = s_stage[q];
  }
  __syncthreads();
  unsigned* s_stage32 = reinterpret_cast<unsigned*>(s_stage);
#pragma unroll
  for (int u = 0; u < kMsRows; u++)
    if (live[u]) { const unsigned q = s_start[dg[u]] + rank[u]; s_stage32[q] = rw[u]; if (WITH_BUCKET) s_stage32[kMsTile + q] = bk[u]; }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < kMsRows; u++)
    if (dst[u] >= 0) {
      out_rows[dst[u]] = s_stage32[u * kThreads + t];
      if (WITH_BUCKET) out_bucket[dst[u]] = s_stage32[kMsTile + u * kThreads + t];
    }
}

// ---- 4: one wave sorts one bucket ----------------------------------------------------------------------------------------
struct KR {
  unsigned long long k;
  unsigned r;
  static __device__ __forceinline__ bool less(const KR& a, const KR& b) { return a.k < b.k || (a.k == b.k && a.r < b.r); }
  static __device__ __forceinline__ KR xchg(const KR& a, int j, int lane) {   // the pair lane ^ j holds
    KR o;
    o.k = ((unsigned long long)ms_xor_lane((unsigned)(a.k >> 32), j, lane) << 32) | ms_xor_lane((unsigned)a.k, j, lane);
    o.r = ms_xor_lane(a.r, j, lane);
    return o;
  }
};
__device__ __forceinline__ bool kr_less(const KR& a, const KR& b) { return KR::less(a, b); }
// One 64-bit word per element where the bucket allows it.  The network is all vector instructions (SQ counters: the vector
// unit is saturated, LDS and memory idle), and a (key, row) pair costs three exchanged words, a two-level comparison and
// three selects per stage.  The keys of a bucket lie close together — it is 1 / 2^21 of the column's quantile range — so:
//   tier 1: all keys within 2^36 of lane 0's → (key − key₀ + 2^36) << 27 | row, unique, sorted as plain integers;
//   tier 2: within 2^56 → … << 7 | position in the bucket, and the rows are fetched by position afterwards (ds_bpermute);
//           two equal keys would come out in bucket order instead of row order — seen by comparing neighbours, and then,
//           as for everything wider, the pairs go through the network (equal keys mean a dense bucket: tier 1 took those).
struct C64 {
  unsigned long long c;
  static __device__ __forceinline__ bool less(const C64& a, const C64& b) { return a.c < b.c; }
  static __device__ __forceinline__ C64 xchg(const C64& a, int j, int lane) {
    C64 o;
    o.c = ((unsigned long long)ms_xor_lane((unsigned)(a.c >> 32), j, lane) << 32) | ms_xor_lane((unsigned)a.c, j, lane);
    return o;
  }
};
constexpr int kMsRowBits = 27;   // the MSD path takes ≤ 2^27 rows (ah_sort_msd_temp_bytes)
template <int SLOTS>
__device__ __forceinline__ void ms_sort_bucket(const unsigned long long* __restrict__ keys, const unsigned* rows, unsigned s, int m,
                                               int lane, unsigned* out_rows, unsigned long long* __restrict__ out64) {
  KR x[SLOTS];
#pragma unroll
  for (int sl = 0; sl < SLOTS; sl++) {   // slots past the bucket hold the largest pair
    const int i = sl * 64 + lane;
    x[sl].k = i < m ? keys[s + i] : ~0ull;
    x[sl].r = i < m ? rows[s + i] : ~0u;
  }
  auto emit = [&](int i, unsigned row) { if (i < m) { if (out64) out64[s + i] = row; else out_rows[s + i] = row; } };
  if (SLOTS <= 2) {
    const unsigned long long k0 = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(x[0].k >> 32)) << 32) |
                                  __builtin_amdgcn_readfirstlane((unsigned)x[0].k);   // lane 0 of slot 0 is inside the bucket (m ≥ 1)
    bool fits1 = true, fits2 = true;
#pragma unroll
    for (int sl = 0; sl < SLOTS; sl++) {
      const bool in = sl * 64 + lane < m;
      fits1 = fits1 && (!in || x[sl].k - k0 + (1ull << 36) < (1ull << 37));
      fits2 = fits2 && (!in || x[sl].k - k0 + (1ull << 56) < (1ull << 57));
    }
    if (__all(fits1)) {
      C64 y[SLOTS];
#pragma unroll
      for (int sl = 0; sl < SLOTS; sl++)
        y[sl].c = sl * 64 + lane < m ? ((x[sl].k - k0 + (1ull << 36)) << kMsRowBits) | x[sl].r : ~0ull;
      ms_bitonic<C64, SLOTS>(y, lane);
#pragma unroll
      for (int sl = 0; sl < SLOTS; sl++) emit(sl * 64 + lane, (unsigned)y[sl].c & ((1u << kMsRowBits) - 1u));
      return;
    }
    if (__all(fits2)) {
      C64 y[SLOTS];
#pragma unroll
      for (int sl = 0; sl < SLOTS; sl++)
        y[sl].c = sl * 64 + lane < m ? ((x[sl].k - k0 + (1ull << 56)) << 7) | (unsigned)(sl * 64 + lane) : ~0ull;
      ms_bitonic<C64, SLOTS>(y, lane);
      bool tie = false;
#pragma unroll
      for (int sl = 0; sl < SLOTS; sl++) {
        unsigned long long nxt = __shfl_down(y[sl].c, 1, 64);
        if (sl + 1 < SLOTS) {
          const unsigned long long head = ((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(y[sl + 1 < SLOTS ? sl + 1 : sl].c >> 32)) << 32) |
                                          __builtin_amdgcn_readfirstlane((unsigned)y[sl + 1 < SLOTS ? sl + 1 : sl].c);
          nxt = lane == 63 ? head : nxt;
        } else if (lane == 63) {
          nxt = ~0ull;
        }
        tie = tie || (sl * 64 + lane + 1 < m && (y[sl].c >> 7) == (nxt >> 7));
      }
      if (!__any(tie)) {
#pragma unroll
        for (int sl = 0; sl < SLOTS; sl++) {
          const unsigned at = (unsigned)y[sl].c & 127u;
          unsigned row = __shfl(x[0].r, (int)(at & 63u), 64);
          if (SLOTS > 1) { const unsigned row1 = __shfl(x[SLOTS > 1 ? 1 : 0].r, (int)(at & 63u), 64); row = (at >> 6) ? row1 : row; }
          emit(sl * 64 + lane, row);
        }
        return;
      }
    }
  }
  ms_bitonic<KR, SLOTS>(x, lane);
#pragma unroll
  for (int sl = 0; sl < SLOTS; sl++) emit(sl * 64 + lane, x[sl].r);
}

__global__ __launch_bounds__(256) void ms_local_kernel(const unsigned long long* __restrict__ keys, const unsigned* rows,
                                                        const unsigned* __restrict__ bstart, int64_t nbuckets, unsigned* out_rows,
                                                        unsigned long long* __restrict__ out64, unsigned* __restrict__ oversize,
                                                        unsigned* __restrict__ big_list) {
  const int lane = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= nbuckets) return;
  const unsigned s = bstart[b], e = bstart[b + 1];
  const int m = (int)(e - s);
  if (m <= 0) return;
  if (m > kLocalMax) {   // beyond one wave: a workgroup takes it (ms_big_kernel) — or, above its LDS, nobody does and the attempt is void
    if (lane == 0) {
      if (m > kBigMax) atomicMax(&oversize[0], (unsigned)m);
      else { const unsigned at = atomicAdd(&oversize[1], 1u); if (at < (unsigned)kBigList) big_list[at] = (unsigned)b; }
    }
    return;
  }
  if (m == 1) { if (lane == 0) { if (out64) out64[s] = rows[s]; else out_rows[s] = rows[s]; } return; }
  if (m <= 64) ms_sort_bucket<1>(keys, rows, s, m, lane, out_rows, out64);
  else if (m <= 128) ms_sort_bucket<2>(keys, rows, s, m, lane, out_rows, out64);
  else if (m <= 256) ms_sort_bucket<4>(keys, rows, s, m, lane, out_rows, out64);   // the tail of the size distribution: a few buckets in a million
  else ms_sort_bucket<8>(keys, rows, s, m, lane, out_rows, out64);
}

// A bucket of 513 … 8192 rows (the rows beyond the sample's extremes, a run of equal keys): one workgroup, bitonic network in LDS.
// A handful per call at most; the list holds 1024.
__global__ __launch_bounds__(kThreads) void ms_big_kernel(const unsigned long long* __restrict__ keys, const unsigned* rows,
                                                           const unsigned* __restrict__ bstart, const unsigned* __restrict__ oversize,
                                                           const unsigned* __restrict__ big_list, unsigned* out_rows,
                                                           unsigned long long* __restrict__ out64) {
  __shared__ unsigned long long s_k[kBigMax];
  __shared__ unsigned s_r[kBigMax];
  if (blockIdx.x >= oversize[1] || blockIdx.x >= (unsigned)kBigList) return;
  const unsigned b = big_list[blockIdx.x], s = bstart[b];
  const int m = (int)(bstart[b + 1] - s);
  int p2 = 1024;
  while (p2 < m) p2 <<= 1;
  for (int i = threadIdx.x; i < p2; i += kThreads) { s_k[i] = i < m ? keys[s + i] : ~0ull; s_r[i] = i < m ? rows[s + i] : ~0u; }
  __syncthreads();
  for (int k = 2; k <= p2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < p2; i += kThreads) {
        const int q = i ^ j;
        if (q > i) {
          const KR x{s_k[i], s_r[i]}, y{s_k[q], s_r[q]};
          const bool asc = (i & k) == 0;
          if (kr_less(y, x) == asc) { s_k[i] = y.k; s_r[i] = y.r; s_k[q] = x.k; s_r[q] = x.r; }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < m; i += kThreads) { if (out64) out64[s + i] = s_r[i]; else out_rows[s + i] = s_r[i]; }
}

}  // namespace

static void ms_plan(int64_t n, int* lb_out, int* lb2_out) {
  int lb = 0;
  while (((int64_t)64 << lb) < n) lb++;          // buckets: ≈ 64 rows each
  if (lb > 21) lb = 21;
  *lb_out = lb;
  *lb2_out = lb >= 20 ? lb - 10 : lb - lb / 2;    // NB2 = 2^lb2 ≤ 2048, NB1 = 2^(lb − lb2) ≤ 1024
}
static size_t ms_pad(size_t b) { return (b + 255) & ~(size_t)255; }

// bytes of temporaries ah_sort_rest_msd wants for n pairs (0: it will not run).  2^27 rows = 2^21 buckets of 64: the largest table
// the two partition levels address (1024 × 2048); below 2^22 rows the fixed costs (sorting the sample) eat the gain
size_t ah_sort_msd_temp_bytes(int64_t n) {
  if (n < ((int64_t)1 << 22) || n > ((int64_t)1 << 27)) return 0;
  int lb, lb2;
  ms_plan(n, &lb, &lb2);
  const int nb2 = 1 << lb2, nb1 = 1 << (lb - lb2);
  const int64_t ntiles = ah_ceil_div(n, kMsTile), ngrp = ah_ceil_div(ntiles, kGroupTiles), nvt = ((ntiles + nb1 + 7) / 8) * 8;
  return ms_pad((size_t)n * 4) * 2 + ms_pad((size_t)kSample * 8) * 2 + ms_pad((size_t)kSample * 4) * 2 + ms_pad((size_t)(kSplit + 1) * 8) + ms_pad(kFlagWords * 4) + ms_pad(4097 * 2) +
         ms_pad((size_t)ntiles * nb1 * 4) * 2 + ms_pad((size_t)ngrp * nb1 * 4) + ms_pad((size_t)(nb1 + 1) * 4) + ms_pad((size_t)nvt * nb2 * 4) * 2 +
         ms_pad(((size_t)1 << lb) * 4 + 4) + ms_pad((size_t)kBigList * 4) + 256;
}

// keys / rows: the `rest` range (n pairs, row order).  alt_keys / alt_rows: same-sized scratch.  tmp: ah_sort_msd_temp_bytes(n) bytes.
// *used = 1: the rows are in sorted order in `rows` — or, if out64 is given, widened in out64[0 .. n) and NOT in `rows`.
// *used = 0: gave up half-way, keys / rows / alt_* are clobbered.  *used = −1: not attempted, nothing touched.
int ah_sort_rest_msd(ah_ctx* c, unsigned long long* keys, unsigned* rows, unsigned long long* alt_keys, unsigned* alt_rows, int64_t n,
                     unsigned long long varying, unsigned long long kmin, unsigned long long kmax, int float_bytes, int descending, void* tmp,
                     unsigned long long* out64, int* used) {
  *used = -1;
  if (!tmp || varying == 0 || ah_sort_msd_temp_bytes(n) == 0) return AH_OK;
  unsigned* out_rows = rows;   // the last step rewrites every bucket in place
  auto pad = ms_pad;
  int lb, lb2;
  ms_plan(n, &lb, &lb2);
  const int nb2 = 1 << lb2, nb1 = 1 << (lb - lb2);
  const unsigned nbuckets = 1u << lb;
  const int hb = 63 - __builtin_clzll(varying);
  MsMap map{hb == 63 ? ~0ull : ((1ull << (hb + 1)) - 1), 63 - hb, float_bytes, descending};
  const int64_t ntiles = ah_ceil_div(n, kMsTile), ngrp = ah_ceil_div(ntiles, kGroupTiles), nvt = ((ntiles + nb1 + 7) / 8) * 8;
  const unsigned grid1 = (unsigned)(((ntiles + 7) / 8) * 8);
  const int64_t sample_n = n < kSample ? n : kSample;
  uint8_t* base = (uint8_t*)tmp;
  int rc;
  size_t off = 0;
  auto take = [&](size_t b) { uint8_t* q = base + off; off += pad(b); return q; };
  unsigned* bucket = (unsigned*)take((size_t)n * 4);
  unsigned* bucket2 = (unsigned*)take((size_t)n * 4);
  unsigned long long* sample = (unsigned long long*)take((size_t)kSample * 8);
  unsigned long long* sample_alt = (unsigned long long*)take((size_t)kSample * 8);
  unsigned* sample_rows = (unsigned*)take((size_t)kSample * 4);
  unsigned* sample_rows_alt = (unsigned*)take((size_t)kSample * 4);
  unsigned long long* split = (unsigned long long*)take((size_t)(kSplit + 1) * 8);
  unsigned* flags = (unsigned*)take(kFlagWords * 4);
  unsigned short* guide = (unsigned short*)take(4097 * 2);
  unsigned* cnt1 = (unsigned*)take((size_t)ntiles * nb1 * 4);
  unsigned* toffs1 = (unsigned*)take((size_t)ntiles * nb1 * 4);
  unsigned* gsum = (unsigned*)take((size_t)ngrp * nb1 * 4);
  unsigned* pstart = (unsigned*)take((size_t)(nb1 + 1) * 4);
  unsigned* cnt2 = (unsigned*)take((size_t)nvt * nb2 * 4);
  unsigned* toffs2 = (unsigned*)take((size_t)nvt * nb2 * 4);
  unsigned* bstart = (unsigned*)take(((size_t)nbuckets + 1) * 4);
  unsigned* big_list = (unsigned*)take((size_t)kBigList * 4);
  unsigned* oversize = (unsigned*)&c->dscalars[26];   // [0] largest bucket nobody can sort, [1] buckets listed for ms_big_kernel
  AH_HIP(c, hipMemsetAsync(oversize, 0, 8, c->stream));
  // 0: the map
  ms_sample_kernel<<<(unsigned)ah_ceil_div(sample_n, 256), 256, 0, c->stream>>>(keys, n, n / sample_n, sample_n, map, sample, sample_rows);
  AH_LAUNCH_CHECK(c);
  unsigned long long* sorted = nullptr;
  if ((rc = ah_sort_pairs_lsd(c, sample, sample_rows, sample_alt, sample_rows_alt, sample_n, (unsigned*)cnt2, (unsigned*)toffs2, &sorted)) != AH_OK) return rc;
  {
    // equal keys filling more than half of what a workgroup can sort (kBigMax rows): give up now
    const int64_t run = (int64_t)(kBigMax / 2) * sample_n / n;
    ms_dupes_kernel<<<(unsigned)ah_ceil_div(sample_n, 256), 256, 0, c->stream>>>(sorted, sample_n, run < 1 ? 1 : run, oversize);
    AH_LAUNCH_CHECK(c);
    AH_HIP(c, hipMemcpyAsync(&c->pinned[12], oversize, 8, hipMemcpyDeviceToHost, c->stream));
    AH_HIP(c, hipStreamSynchronize(c->stream));
    if (((volatile unsigned*)&c->pinned[12])[0] != 0) {
      if (getenv("ARROWHIP_DEBUG_MSD")) fprintf(stderr, "msd: n=%lld: the sample shows a key with more than %d rows, not attempted\n", (long long)n, kBigMax / 2);
      *used = -1;   // nothing was clobbered yet
      return AH_OK;
    }
  }
  *used = 0;   // from here on the pairs are being moved
  MsPoints pts{sorted, sample_n, (kmin & map.mask) << map.lshift, (kmax & map.mask) << map.lshift};
  AH_HIP(c, hipMemsetAsync(flags, 0, kFlagWords * 4, c->stream));
  ms_split_kernel<<<(kSplit + 256) / 256, 256, 0, c->stream>>>(pts, split, flags);
  AH_LAUNCH_CHECK(c);
  // 1: bucket ids
  ms_guide_kernel<<<17, 256, 0, c->stream>>>(split, guide);
  AH_LAUNCH_CHECK(c);
  ms_bucket_kernel<<<ah_stream_grid(c, ah_ceil_div(n, kThreads * 4), 2), kThreads, 0, c->stream>>>(keys, n, map, pts, split, flags, guide, nbuckets / kSplit, bucket);
  AH_LAUNCH_CHECK(c);
  // 2: level 1
  ms_hist_kernel<<<grid1, kThreads, 0, c->stream>>>(bucket, n, nullptr, 1, lb2, (unsigned)(nb1 - 1), nb1, cnt1);
  AH_LAUNCH_CHECK(c);
  colsum_kernel<<<(unsigned)ngrp, kMaxBins, 0, c->stream>>>(cnt1, nb1, ntiles, gsum);
  AH_LAUNCH_CHECK(c);
  bin_prefix_kernel<<<1, kMaxBins, 0, c->stream>>>(gsum, nb1, ngrp, n, pstart);
  AH_LAUNCH_CHECK(c);
  tile_offs_kernel<<<(unsigned)ngrp, kMaxBins, 0, c->stream>>>(cnt1, gsum, nb1, ntiles, toffs1);
  AH_LAUNCH_CHECK(c);
  ms_scatter_kernel<true><<<grid1, kThreads, 0, c->stream>>>(keys, rows, bucket, n, nullptr, 1, lb2, (unsigned)(nb1 - 1), nb1, toffs1, alt_keys,
                                                                        alt_rows, bucket2);
  AH_LAUNCH_CHECK(c);
  // 3: level 2, parent by parent
  ms_hist_kernel<<<(unsigned)nvt, kThreads, 0, c->stream>>>(bucket2, n, pstart, nb1, 0, (unsigned)(nb2 - 1), nb2, cnt2);
  AH_LAUNCH_CHECK(c);
  ms_offs2_kernel<<<(unsigned)nb1, kThreads, 0, c->stream>>>(cnt2, pstart, nb1, nb2, toffs2, bstart, n);
  AH_LAUNCH_CHECK(c);
  ms_scatter_kernel<false><<<(unsigned)nvt, kThreads, 0, c->stream>>>(alt_keys, alt_rows, bucket2, n, pstart, nb1, 0, (unsigned)(nb2 - 1), nb2, toffs2, keys,
                                                                      rows, nullptr);
  AH_LAUNCH_CHECK(c);
  // 4: buckets
  ms_local_kernel<<<(unsigned)ah_ceil_div((int64_t)nbuckets, 4), 256, 0, c->stream>>>(keys, rows, bstart, (int64_t)nbuckets, out_rows, out64, oversize, big_list);
  AH_LAUNCH_CHECK(c);
  ms_big_kernel<<<kBigList, kThreads, 0, c->stream>>>(keys, rows, bstart, oversize, big_list, out_rows, out64);
  AH_LAUNCH_CHECK(c);
  AH_HIP(c, hipMemcpyAsync(&c->pinned[12], oversize, 8, hipMemcpyDeviceToHost, c->stream));
  AH_HIP(c, hipStreamSynchronize(c->stream));
  const unsigned too_big = ((volatile unsigned*)&c->pinned[12])[0], listed = ((volatile unsigned*)&c->pinned[12])[1];
  if (getenv("ARROWHIP_DEBUG_MSD")) fprintf(stderr, "msd: n=%lld lb=%d nb1=%d nb2=%d largest unsortable bucket=%u, buckets sorted by a workgroup=%u\n", (long long)n, lb, nb1, nb2, too_big, listed);
  if (too_big != 0 || listed > (unsigned)kBigList) return AH_OK;   // a bucket beyond one wave's reach: the caller's LSD passes redo the range
  *used = 1;
  return AH_OK;
}
