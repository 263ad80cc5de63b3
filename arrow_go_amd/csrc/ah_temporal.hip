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
constexpr int kRows = 4;  // rows per lane: 16 bytes of int32, 32 bytes of int64

enum { SHIFT_CONVERT = 0, SHIFT_MULTIPLY = 1, SHIFT_DIVIDE = 2 };

template <typename T>
using Vec4 = T __attribute__((ext_vector_type(4)));

template <typename InT, typename OutT, int OP>
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
    const InT f = (InT)factor;
    const OutT q = (OutT)(v / f);                        // truncated quotient, then narrowed
    bad = (InT)((UIn)(InT)q * (UIn)f) != v;
    return q;
  }
}

template <typename InT, typename OutT, int OP, bool CHECK>
__global__ __launch_bounds__(kBlock) void shift_time_kernel(const InT* __restrict__ in, const uint8_t* __restrict__ valid, int64_t voff, int64_t n,
                                                            int64_t factor, int64_t lo, int64_t hi, OutT* __restrict__ out,
                                                            unsigned long long* __restrict__ first_bad, int aligned) {
  const int64_t base = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kRows;
  if (base >= n) return;
  InT v0 = 0, v1 = 0, v2 = 0, v3 = 0;
  const bool full = base + kRows <= n;
  if (full && aligned) {
    const Vec4<InT> x = __builtin_nontemporal_load((const Vec4<InT>*)(in + base));
    v0 = x.x; v1 = x.y; v2 = x.z; v3 = x.w;
  } else {
    v0 = in[base];
    if (base + 1 < n) v1 = in[base + 1];
    if (base + 2 < n) v2 = in[base + 2];
    if (base + 3 < n) v3 = in[base + 3];
  }
  bool b0, b1, b2, b3;
  const OutT o0 = shift_one<InT, OutT, OP>(v0, factor, lo, hi, b0);
  const OutT o1 = shift_one<InT, OutT, OP>(v1, factor, lo, hi, b1);
  const OutT o2 = shift_one<InT, OutT, OP>(v2, factor, lo, hi, b2);
  const OutT o3 = shift_one<InT, OutT, OP>(v3, factor, lo, hi, b3);
  if constexpr (CHECK) {
    if (b0 | b1 | b2 | b3) {  // rare: look at validity only now
      auto live = [&](int k) {
        const int64_t r = base + k;
        if (r >= n) return false;
        if (!valid) return true;
        const int64_t bit = voff + r;
        return ((valid[bit >> 3] >> (bit & 7)) & 1) != 0;
      };
      int first = -1;
      if (b3 && live(3)) first = 3;
      if (b2 && live(2)) first = 2;
      if (b1 && live(1)) first = 1;
      if (b0 && live(0)) first = 0;
      if (first >= 0) atomicMin(first_bad, (unsigned long long)(base + first));
    }
  }
  if (full && aligned) {
    Vec4<OutT> y;
    y.x = o0; y.y = o1; y.z = o2; y.w = o3;
    __builtin_nontemporal_store(y, (Vec4<OutT>*)(out + base));
  } else {
    out[base] = o0;
    if (base + 1 < n) out[base + 1] = o1;
    if (base + 2 < n) out[base + 2] = o2;
    if (base + 3 < n) out[base + 3] = o3;
  }
}

template <typename InT, typename OutT>
void launch_shift(ah_ctx* c, int op, bool check, const void* in, const uint8_t* valid, int64_t voff, int64_t n, int64_t factor, int64_t lo,
                  int64_t hi, void* out, unsigned long long* first_bad) {
  const unsigned grid = (unsigned)((n + (int64_t)kBlock * kRows - 1) / ((int64_t)kBlock * kRows));
  const int aligned = ((uintptr_t)in % (kRows * sizeof(InT)) == 0) && ((uintptr_t)out % (kRows * sizeof(OutT)) == 0);
  const InT* pi = (const InT*)in;
  OutT* po = (OutT*)out;
#define AH_SHIFT(OP)                                                                                                                   \
  do {                                                                                                                                 \
    if (check) shift_time_kernel<InT, OutT, OP, true><<<grid, kBlock, 0, c->stream>>>(pi, valid, voff, n, factor, lo, hi, po, first_bad, aligned); \
    else shift_time_kernel<InT, OutT, OP, false><<<grid, kBlock, 0, c->stream>>>(pi, valid, voff, n, factor, lo, hi, po, first_bad, aligned);      \
  } while (0)
  if (op == SHIFT_CONVERT) AH_SHIFT(SHIFT_CONVERT);
  else if (op == SHIFT_MULTIPLY) AH_SHIFT(SHIFT_MULTIPLY);
  else AH_SHIFT(SHIFT_DIVIDE);
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
