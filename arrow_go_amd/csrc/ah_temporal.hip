// Unit changes of temporal columns — ShiftTime (arrow/compute/internal/kernels/cast_temporal.go:35-104), the leaf of
// every timestamp → timestamp, duration → duration, time32 ↔ time64 and date32 ↔ date64 cast (cast_temporal.go:240-420)
// and of the implicit casts DispatchBest inserts when two temporal operands differ in unit (arithmetic.go:130-131,
// scalar_comparisons.go commonTemporal).
//
// One streaming pass: read InT (int32 / int64), multiply or divide by a constant, write OutT.  The reference walks the
// column twice as wide as it has to when it checks (a bitmap reader beside the value loop); here the check rides in the
// same pass and costs nothing unless it fires:
//   multiply, checked: a VALID value outside [MinInt64/factor, MaxInt64/factor] fails — the bound is on int64 even when
//                      OutT is int32 (:53), so an int32 product still wraps silently, exactly as in the reference;
//   divide,   checked: a VALID value that is not a multiple fails (InT(out)·InT(factor) ≠ v, :85/:96 — which also
//                      catches a quotient that does not fit an int32 OutT).
// Every slot is converted, valid or not (the loops at :42-44, :61, :84 do not look at validity for the arithmetic).
// The failure reports the FIRST offending row's value, as the reference's loop would: atomicMin on the row index (the
// rare path), read back by the host.
//
// Bound: HBM.  Algorithmic bytes per row = sizeof(InT) + sizeof(OutT) (+ 1/8 validity when checking a column with nulls).
#include "ah_common.h"

namespace {

constexpr int kBlock = 256;

enum { SHIFT_CONVERT = 0, SHIFT_MULTIPLY = 1, SHIFT_DIVIDE = 2 };

// F: the divisor as a compile-time constant for the factors the unit table produces (10^3, 10^6, 10^9, 86 400 000) — a
// multiply-high instead of the generic 64-bit division sequence, which made the int64 divide ALU-bound (4.6 TB/s);
// 0 = take it from the argument.
template <typename InT, typename OutT, int OP, int64_t F>
__device__ __forceinline__ OutT shift_one(InT v, int64_t factor, int64_t lo, int64_t hi, bool& bad) {
  using UOut = typename std::make_unsigned<OutT>::type;
  using UIn = typename std::make_unsigned<InT>::type;
  if constexpr (OP == SHIFT_CONVERT) {
    bad = false;
    return (OutT)v;
  } else if constexpr (OP == SHIFT_MULTIPLY) {
    bad = (int64_t)v < lo || (int64_t)v > hi;
    return (OutT)((UOut)(OutT)v * (UOut)(OutT)factor);  // OutT(v) * OutT(factor), wrapping like Go
  } else {
    const InT f = F ? (InT)F : (InT)factor;
    const OutT q = (OutT)(v / f);                        // truncated quotient, then narrowed
    bad = (InT)((UIn)(InT)q * (UIn)f) != v;
    return q;
  }
}

// Lane shape: K rows per access so that the WIDER of the two sides moves 16 bytes per lane (the narrow side then moves 8 —
// still one contiguous run per wave), U such accesses per lane a workgroup-tile apart, both loads issued before any
// arithmetic.  (A first version gave each lane 4 consecutive int64 = two 16-byte accesses 32 bytes apart: every wave
// instruction touched each cache line half — 5.4 TB/s where the 16-byte shape of the other streaming kernels gets 6+.)
template <typename InT, typename OutT>
struct ShiftShape {
  static constexpr int K = 16 / (int)(sizeof(InT) > sizeof(OutT) ? sizeof(InT) : sizeof(OutT));
  static constexpr int U = 2;
  typedef InT VIn __attribute__((ext_vector_type(K)));
  typedef OutT VOut __attribute__((ext_vector_type(K)));
};

template <typename InT, typename OutT, int OP, bool CHECK, int64_t F = 0>
__global__ __launch_bounds__(kBlock) void shift_time_kernel(const InT* __restrict__ in, const uint8_t* __restrict__ valid, int64_t voff, int64_t n,
                                                            int64_t factor, int64_t lo, int64_t hi, OutT* __restrict__ out,
                                                            unsigned long long* __restrict__ first_bad, int aligned) {
  using S = ShiftShape<InT, OutT>;
  constexpr int K = S::K;
  typedef typename S::VIn VIn;
  typedef typename S::VOut VOut;
  const int64_t base0 = (int64_t)blockIdx.x * (kBlock * K * S::U) + (int64_t)threadIdx.x * K;
  const int64_t base1 = base0 + kBlock * K;
  if (base0 >= n) return;
  auto load = [&](int64_t base) {
    VIn x;
    if (base + K <= n && aligned) {
      x = __builtin_nontemporal_load((const VIn*)(in + base));
    } else {
#pragma unroll
      for (int k = 0; k < K; k++) x[k] = base + k < n ? in[base + k] : (InT)0;
    }
    return x;
  };
  auto finish = [&](int64_t base, VIn x) {
    VOut y;
    unsigned bad = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
      bool b;
      y[k] = shift_one<InT, OutT, OP, F>(x[k], factor, lo, hi, b);
      bad |= (unsigned)b << k;
    }
    if constexpr (CHECK) {
      if (bad) {  // rare: look at validity only now; the lowest failing VALID row of this lane competes for "first"
#pragma unroll
        for (int k = K - 1; k >= 0; k--) {
          const int64_t r = base + k;
          bool live = ((bad >> k) & 1) && r < n;
          if (live && valid) {
            const int64_t bit = voff + r;
            live = ((valid[bit >> 3] >> (bit & 7)) & 1) != 0;
          }
          if (live) atomicMin(first_bad, (unsigned long long)r);
        }
      }
    }
    if (base + K <= n && aligned) {
      __builtin_nontemporal_store(y, (VOut*)(out + base));
    } else {
#pragma unroll
      for (int k = 0; k < K; k++)
        if (base + k < n) out[base + k] = y[k];
    }
  };
  const VIn x0 = load(base0);
  if (base1 < n) {
    const VIn x1 = load(base1);
    finish(base0, x0);
    finish(base1, x1);
  } else {
    finish(base0, x0);
  }
}

template <typename InT, typename OutT>
void launch_shift(ah_ctx* c, int op, bool check, const void* in, const uint8_t* valid, int64_t voff, int64_t n, int64_t factor, int64_t lo,
                  int64_t hi, void* out, unsigned long long* first_bad) {
  using S = ShiftShape<InT, OutT>;
  const int64_t tile = (int64_t)kBlock * S::K * S::U;
  const unsigned grid = (unsigned)((n + tile - 1) / tile);
  const int aligned = ((uintptr_t)in % (S::K * sizeof(InT)) == 0) && ((uintptr_t)out % (S::K * sizeof(OutT)) == 0);
  const InT* pi = (const InT*)in;
  OutT* po = (OutT*)out;
#define AH_SHIFT(OP)                                                                                                                   \
  do {                                                                                                                                 \
    if (check) shift_time_kernel<InT, OutT, OP, true><<<grid, kBlock, 0, c->stream>>>(pi, valid, voff, n, factor, lo, hi, po, first_bad, aligned); \
    else shift_time_kernel<InT, OutT, OP, false><<<grid, kBlock, 0, c->stream>>>(pi, valid, voff, n, factor, lo, hi, po, first_bad, aligned);      \
  } while (0)
#define AH_SHIFT_DIV(F)                                                                                                                       \
  do {                                                                                                                                         \
    if (check) shift_time_kernel<InT, OutT, SHIFT_DIVIDE, true, F><<<grid, kBlock, 0, c->stream>>>(pi, valid, voff, n, factor, lo, hi, po, first_bad, aligned); \
    else shift_time_kernel<InT, OutT, SHIFT_DIVIDE, false, F><<<grid, kBlock, 0, c->stream>>>(pi, valid, voff, n, factor, lo, hi, po, first_bad, aligned);      \
  } while (0)
  if (op == SHIFT_CONVERT) AH_SHIFT(SHIFT_CONVERT);
  else if (op == SHIFT_MULTIPLY) AH_SHIFT(SHIFT_MULTIPLY);
  else if (factor == 1000) AH_SHIFT_DIV(1000);
  else if (factor == 1000000) AH_SHIFT_DIV(1000000);
  else if (factor == 1000000000) AH_SHIFT_DIV(1000000000);
  else if (factor == 86400000) AH_SHIFT_DIV(86400000);
  else AH_SHIFT(SHIFT_DIVIDE);
#undef AH_SHIFT_DIV
#undef AH_SHIFT
}

}  // namespace

AH_EXPORT int ah_shift_time(ah_ctx* c, int in_bits, int out_bits, int op, int64_t factor, int check, const void* values, const uint8_t* valid,
                            int64_t off, int64_t n, void* out, int64_t* bad_value) {
  AH_ENTER(c);
  if (n < 0 || off < 0) return ah_fail(c, AH_EINVALID, "shift_time: negative length/offset");
  if ((in_bits != 32 && in_bits != 64) || (out_bits != 32 && out_bits != 64)) return ah_fail(c, AH_EINVALID, "shift_time: widths are 32 or 64 bits");
  if (op != AH_SHIFT_MULTIPLY && op != AH_SHIFT_DIVIDE) return ah_fail(c, AH_EINVALID, "shift_time: unknown op %d", op);
  if (factor < 1) return ah_fail(c, AH_EINVALID, "shift_time: factor must be >= 1");
  if (op == AH_SHIFT_DIVIDE && in_bits == 32 && factor > 0x7fffffffLL) return ah_fail(c, AH_EINVALID, "shift_time: divisor does not fit the 32-bit input");
  if (bad_value) *bad_value = 0;
  if (n == 0) return AH_OK;
  const int kop = factor == 1 ? SHIFT_CONVERT : op == AH_SHIFT_MULTIPLY ? SHIFT_MULTIPLY : SHIFT_DIVIDE;  // :41-45: a factor of 1 only converts
  const bool chk = check != 0 && kop != SHIFT_CONVERT;
  const int64_t hi = INT64_MAX / factor, lo = INT64_MIN / factor;
  unsigned long long* first_bad = (unsigned long long*)c->dscalars;
  if (chk) AH_HIP(c, hipMemsetAsync(first_bad, 0xff, sizeof(unsigned long long), c->stream));
  if (in_bits == 32 && out_bits == 32) launch_shift<int32_t, int32_t>(c, kop, chk, values, valid, off, n, factor, lo, hi, out, first_bad);
  else if (in_bits == 32) launch_shift<int32_t, int64_t>(c, kop, chk, values, valid, off, n, factor, lo, hi, out, first_bad);
  else if (out_bits == 32) launch_shift<int64_t, int32_t>(c, kop, chk, values, valid, off, n, factor, lo, hi, out, first_bad);
  else launch_shift<int64_t, int64_t>(c, kop, chk, values, valid, off, n, factor, lo, hi, out, first_bad);
  AH_HIP(c, hipGetLastError());
  if (!chk) return AH_OK;
  AH_HIP(c, hipMemcpyAsync(c->pinned, first_bad, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
  AH_HIP(c, hipStreamSynchronize(c->stream));
  const unsigned long long row = *(volatile unsigned long long*)c->pinned;
  if (row == ~0ULL) return AH_OK;
  int64_t v = 0;
  if (in_bits == 32) {
    int32_t v32 = 0;
    AH_HIP(c, hipMemcpy(&v32, (const int32_t*)values + row, 4, hipMemcpyDeviceToHost));
    v = v32;
  } else {
    AH_HIP(c, hipMemcpy(&v, (const int64_t*)values + row, 8, hipMemcpyDeviceToHost));
  }
  if (bad_value) *bad_value = v;
  return ah_fail(c, AH_EINVALID, op == AH_SHIFT_MULTIPLY ? "would result in out of bounds timestamp: %lld" : "would lose data: %lld", (long long)v);
}
