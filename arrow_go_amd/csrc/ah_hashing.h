// ah_hashing.h — pieces shared by ah_hash.hip (unique / dictionary_encode / the id-based group-by) and
// ah_groupby.hip (the partition-first group-by): the reference's integer hash, first-seen ranking over an n-bit
// "first occurrence" bitmap, and the 128-bit fixed-point accumulation that makes Float64 group sums reproducible.
#pragma once
#include "ah_common.h"

namespace {

constexpr int kBlock = 256;
constexpr unsigned long long kEmpty = ~0ull;
constexpr unsigned kNoRow = ~0u;

__device__ __forceinline__ uint64_t hash_int(uint64_t v) {  // hash_funcs.go:60-67, alg 0
  return __builtin_bswap64(11400714785074694791ull * v);
}

// per word: exclusive popcount prefix INSIDE its rank tile; per tile: total.  A rank tile is kRankTileWords = 256 words — one
// workgroup of this kernel — so that 2^26 rows are 4096 tiles and scan_kernel below finishes in ONE round (32-word tiles: 32 768
// tiles, eight rounds, 18 µs on a path that is a chain of small launches).
constexpr int kRankTileLog2 = 8, kRankTileWords = 1 << kRankTileLog2;
static inline int64_t rank_tiles(int64_t nwords) { return (nwords + kRankTileWords - 1) >> kRankTileLog2; }
__global__ __launch_bounds__(kBlock) void word_prefix_kernel(const unsigned long long* __restrict__ firsts, int64_t nwords,
                                                              unsigned* __restrict__ wordprefix, int* __restrict__ tilecnt) {
  static_assert(kBlock == kRankTileWords, "one workgroup per rank tile");
  __shared__ int s_w[kBlock / 64];
  int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int v = w < nwords ? __popcll(firsts[w]) : 0;
  int inc = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(inc, o, 64);
    if (lane >= o) inc += t;
  }
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int k = 0; k < kBlock / 64; k++) { if (k < wave) base += s_w[k]; total += s_w[k]; }
  if (w < nwords) wordprefix[w] = (unsigned)(base + inc - v);
  if (threadIdx.x == 0) tilecnt[blockIdx.x] = total;
}

// exclusive scan of the tile totals, one workgroup: 4096 entries per round (one 16-byte load per thread, a wave scan, 16 wave
// totals through LDS).  (The first version gave each thread ntiles / 1024 consecutive entries to walk: 64 dependent
// 4-byte loads per thread at 2^26 rows = 62 µs of pure latency; this one takes ≈ 10 µs.)
__global__ __launch_bounds__(1024) void scan_kernel(const int* __restrict__ counts, int64_t ntiles,
                                                     int64_t* __restrict__ offsets, unsigned long long* __restrict__ total) {
  __shared__ int64_t wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int64_t carry = 0;
  for (int64_t base = 0; base < ntiles; base += 4096) {
    const int64_t i0 = base + (int64_t)tid * 4;
    int c[4] = {0, 0, 0, 0};
    if (i0 + 4 <= ntiles) {
      const ah_vec16<int> v = *reinterpret_cast<const ah_vec16<int>*>(counts + i0);   // element-aligned 16-byte load
      c[0] = v.v[0]; c[1] = v.v[1]; c[2] = v.v[2]; c[3] = v.v[3];
    } else {
      for (int k = 0; k < 4; k++) if (i0 + k < ntiles) c[k] = counts[i0 + k];
    }
    const int64_t s = (int64_t)c[0] + c[1] + c[2] + c[3];
    int64_t inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int64_t t = __shfl_up(inc, o, 64);
      if (lane >= o) inc += t;
    }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    int64_t wbase = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
      const int64_t t = wave_tot[k];
      if (k < wave) wbase += t;
      tot += t;
    }
    int64_t run = carry + wbase + inc - s;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (i0 + k < ntiles) offsets[i0 + k] = run;
      run += c[k];
    }
    carry += tot;
    __syncthreads();   // wave_tot is rewritten in the next round
  }
  if (tid == 0) *total = (unsigned long long)carry;
}

__device__ __forceinline__ unsigned rank_of_row(unsigned fr, const unsigned long long* __restrict__ firsts,
                                                const unsigned* __restrict__ wordprefix, const int64_t* __restrict__ tileoff) {
  unsigned w = fr >> 6;
  unsigned long long below = firsts[w] & ((1ull << (fr & 63)) - 1);
  return (unsigned)(tileoff[w >> kRankTileLog2] + wordprefix[w] + __popcll(below));
}

// ---- deterministic Float64 group sums: fixed point --------------------------------------------------------------
// fp64 atomic adds make a group's sum depend on the order the hardware happens to perform them in.  Integer addition
// is associative, so every value is converted to a 128-bit fixed-point number  q = trunc(x · 2^sh),  sh = 94 − emax,
// emax = exponent of the largest finite |x| (|q| < 2^95; 2^30 rows cannot overflow 2^127), accumulated with
// 64-bit integer atomics (low word with carry into the high word — each addend derives its own carry from the value
// its atomic returned, so any interleaving gives the same 128 bits), and rounded to double ONCE at the end.
// Result: identical bytes run to run, on any launch geometry.  ±inf / NaN addends are tallied as three flag bits per
// group and give the IEEE result of any order: NaN if a NaN or both infinities were seen, else the infinity.
//
// WHICH emax.  A 53-bit significand shifted left by e − emax + 42 bits loses nothing while e ≥ emax − 42.  So every
// call also finds the smallest biased exponent among its non-zero finite values (range[1] = 0x7ff − that exponent, 0 = none
// seen: inverted so that both words are atomicMax targets, and one 32-bit register per lane in the passes that carry it):
//   · the column spans ≤ 42 binades ("narrow", the usual case): ONE scale for the call, no addend is truncated, the
//     128-bit sum is the exact sum and the result is its correctly rounded double — for every group, whatever its magnitude;
//   · wider ("wide": an outlier or sentinel such as 1e300 next to ordinary values): one scale would truncate every
//     addend below 2^(emax − 94) to zero, i.e. one group's outlier would wipe out all other groups' sums.  The scale is
//     then PER GROUP (gmax[g] = largest finite |x| of group g, found by a pass of its own on the id-based path — the
//     partition-first paths hand the call over): error ≤ ½ulp(Σ_g) + n_g·2^(emax_g − 94) ≤ n_g·2^−94·Σ_g|x| — inside the
//     n_g·ε·Σ_g|x| of the sequential row-order definition, now with the GROUP's own Σ|x|.
struct FxAcc {  // global accumulators of one call (device pointers); null for integer sums
  unsigned long long* lo;
  unsigned long long* hi;
  unsigned* flags;
  const unsigned long long* absmax;  // range[0] = bit pattern of the largest finite |x|, range[1] = 0x7ff − smallest biased exponent of a non-zero finite |x| (denormals: 1)
  const unsigned long long* gmax;    // per-group largest finite |x| (wide columns); null: the call's one scale
};
__device__ __forceinline__ int fx_shift(unsigned long long absmax_bits) {
  const int e = (int)((absmax_bits >> 52) & 0x7ff);
  return 94 - ((e ? e : 1) - 1023);
}
// does one scale for the whole call truncate an addend?  (range as in FxAcc::absmax)
__host__ __device__ __forceinline__ bool fx_wide(unsigned long long max_bits, unsigned long long inv_min_exp) {
  if (!inv_min_exp) return false;   // no non-zero finite value at all
  const int emax = (int)((max_bits >> 52) & 0x7ff), emin = 0x7ff - (int)inv_min_exp;
  return (emax ? emax : 1) - emin > 42;
}
// 0x7ff − biased exponent of a non-zero finite |x| given as its bit pattern (a denormal counts as exponent 1, like fx_split)
__device__ __forceinline__ unsigned fx_inv_exp(unsigned long long abs_bits) {
  const unsigned e = (unsigned)(abs_bits >> 52);
  return 0x7ffu - (e ? e : 1u);
}
// sets bit 1 of *flag (the partition-first paths' "redo the call on the id-based path" word) for a wide column
__global__ void fx_range_check_kernel(const unsigned long long* __restrict__ range, unsigned* __restrict__ flag) {
  if (fx_wide(range[0], range[1])) atomicOr(flag, 2u);
}
__device__ __forceinline__ bool fx_finite(double x) { return ((__builtin_bit_cast(unsigned long long, x) >> 52) & 0x7ff) != 0x7ff; }
__device__ __forceinline__ unsigned fx_flag(double x) { return x != x ? 1u : (x > 0 ? 2u : 4u); }   // NaN, +inf, −inf
// q = trunc(|x| · 2^sh) as a 128-bit integer, negated for x < 0.  Done on the bits of x: |x| = m · 2^(e − 1075) with the
// 53-bit significand m, so q = m shifted by s = e − 1075 + sh (≤ 42 for |x| ≤ absmax: q < 2^95; right shifts truncate).
// (The first version went through ldexp / trunc / floor and two double → uint64 conversions: ≈ 4× the instructions,
// and the group-by kernels are instruction-bound.)
__device__ __forceinline__ void fx_split(double x, int sh, unsigned long long* lo, unsigned long long* hi) {
  // without branches (a divergent if / else costs the wave both sides plus the exec-mask bookkeeping): exactly one of the
  // two shift counts is non-zero; m < 2^53, so a right shift by 63 stands for every right shift ≥ 53, and the high word
  // (m >> (64 − sl), nothing for sl ≤ 11) is (m >> 1) >> (63 − sl) for every sl in [0, 42]
  const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
  const int e = (int)((b >> 52) & 0x7ff);
  const unsigned long long m = (b & 0xfffffffffffffull) | (e ? 1ull << 52 : 0ull);
  const int s = (e ? e : 1) - 1075 + sh;
  const int sl = s > 0 ? s : 0;
  int sr = s < 0 ? -s : 0;
  sr = sr < 63 ? sr : 63;
  unsigned long long l = (m << sl) >> sr;
  unsigned long long u = (m >> 1) >> (63 - sl);
  // two's complement of the 128-bit magnitude for x < 0: q = (q ^ S) − S with S = the sign, extended to 128 bits
  const unsigned __int128 S = (unsigned __int128)(__int128)((long long)b >> 63);
  const unsigned __int128 q = ((((unsigned __int128)u << 64) | l) ^ S) - S;
  *lo = (unsigned long long)q;
  *hi = (unsigned long long)(q >> 64);
}
template <typename P>   // P = pointer into LDS or global memory
__device__ __forceinline__ void fx_add(P lo_arr, P hi_arr, size_t g, unsigned long long lo, unsigned long long hi) {
  const unsigned long long old = atomicAdd(&lo_arr[g], lo);
  const unsigned long long carry = old + lo < old ? 1ull : 0ull;
  if (hi + carry) atomicAdd(&hi_arr[g], hi + carry);
}
__device__ __forceinline__ double fx_to_double(unsigned long long lo, unsigned long long hi, int sh) {
  const bool neg = (long long)hi < 0;
  if (neg) { lo = ~lo + 1; hi = ~hi + (lo == 0 ? 1 : 0); }
  double d;
  if (hi == 0) {
    d = (double)lo;                                    // u64 → f64 is correctly rounded
  } else {
    const int lz = __clzll((long long)hi);             // hi != 0: 0..63
    unsigned long long top = lz ? (hi << lz) | (lo >> (64 - lz)) : hi;
    const unsigned long long rest = lz ? lo << lz : lo;
    top |= rest ? 1ull : 0ull;                         // sticky bit: below the 53 bits the conversion keeps
    d = ldexp((double)top, 64 - lz);
  }
  d = ldexp(d, -sh);
  return neg ? -d : d;
}
// out[0] = largest finite |x| (bits), out[1] = largest fx_inv_exp of the non-zero finite values; both zeroed by the caller.
// Streams like the Sum kernel: 16 bytes per lane and load (two values), four loads in flight, a grid-stride loop over ≈ 2 workgroups
// per CU (8-byte loads, eight in flight: 132 µs for 2^26 values = 4.1 TB/s; this form runs at the reduction kernels' rate).
__global__ __launch_bounds__(kBlock) void absmax_kernel(const unsigned long long* __restrict__ vals, const uint8_t* __restrict__ vvalid, int64_t voff,
                                                         int64_t n, unsigned long long* __restrict__ out) {
  unsigned long long m = 0;
  unsigned im = 0;
  constexpr int U = 4;
  const int64_t nvec = n >> 1;
  // one value: a finite non-zero |x| inside the current [min exponent, max] changes nothing and skips the validity read
  auto one = [&](unsigned long long raw, int64_t i) {
    const unsigned long long b = raw & 0x7fffffffffffffffull;   // |x| of finite values order like their bit patterns
    if ((b >> 52) != 0x7ff && b != 0 && (b > m || fx_inv_exp(b) > im) && ah_bit(vvalid, voff + i)) {
      m = b > m ? b : m;
      im = fx_inv_exp(b) > im ? fx_inv_exp(b) : im;
    }
  };
  const int64_t stride = (int64_t)gridDim.x * kBlock * U;
  for (int64_t base = (int64_t)blockIdx.x * kBlock * U + threadIdx.x; base < nvec; base += stride) {
    ah_vec16<unsigned long long> x[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t j = base + (int64_t)u * kBlock;
      if (j < nvec) x[u] = ah_ld16_nt<unsigned long long>(vals + 2 * j);   // element-aligned 16-byte load (a slice need not be 16-byte aligned) — through the
                                                                             // native vector type: the struct load compiled to two 8-byte loads (ah_common.h)
      else { x[u].v[0] = 0; x[u].v[1] = 0; }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t j = base + (int64_t)u * kBlock;
      one(x[u].v[0], 2 * j);
      one(x[u].v[1], 2 * j + 1);
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) one(vals[n - 1], n - 1);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long t = __shfl_down(m, o, 64);
    const unsigned ti = __shfl_down(im, o, 64);
    m = t > m ? t : m;
    im = ti > im ? ti : im;
  }
  __shared__ unsigned long long s_m[kBlock / 64];
  __shared__ unsigned s_im[kBlock / 64];
  if ((threadIdx.x & 63) == 0) { s_m[threadIdx.x >> 6] = m; s_im[threadIdx.x >> 6] = im; }
  __syncthreads();
  if (threadIdx.x == 0) {   // one atomic pair per workgroup: same-address atomics cost ≈ 12 ns each, serialised
    for (int w = 1; w < kBlock / 64; w++) { m = s_m[w] > m ? s_m[w] : m; im = s_im[w] > im ? s_im[w] : im; }
    if (m) atomicMax(&out[0], m);
    if (im) atomicMax(&out[1], (unsigned long long)im);
  }
}
__global__ __launch_bounds__(kBlock) void fx_finalize_kernel(FxAcc acc, int64_t ngroups, double* __restrict__ out_sums) {
  const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (g >= ngroups) return;
  const unsigned f = acc.flags[g];
  double r;
  if (f) r = (f & 1u) || (f & 6u) == 6u ? __builtin_nan("") : ((f & 2u) ? __builtin_inf() : -__builtin_inf());
  else r = fx_to_double(acc.lo[g], acc.hi[g], fx_shift(acc.gmax ? acc.gmax[g] : *acc.absmax));
  out_sums[g] = r;
}

}  // namespace
