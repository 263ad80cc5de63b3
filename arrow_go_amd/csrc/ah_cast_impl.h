// ah_cast_impl.h — templates shared by ah_cast.hip and ah_cast_wide.hip: two translation units so that
// the ~180 kernel instantiations (10 x 10 type pairs x check variants) compile in parallel.
#pragma once
#include <limits>
#include <type_traits>
#include "ah_common.h"

namespace ah_cast_impl {
namespace {  // internal linkage: each translation unit instantiates its own pairs

constexpr int kBlock = 256;
constexpr int kUnroll = 4;

template <typename T, int E>
struct alignas(sizeof(T)) VecE {
  T v[E];
};

// raw carrier of N bytes for nontemporal accesses (the column is streamed once; keep it out of L2)
template <int N> struct Raw;
template <> struct Raw<1> { using type = uint8_t; };
template <> struct Raw<2> { using type = uint16_t; };
template <> struct Raw<4> { using type = uint32_t; };
template <> struct Raw<8> { using type = uint64_t; };
template <> struct Raw<16> { typedef unsigned type __attribute__((ext_vector_type(4))); };

template <typename V, bool NT>
__device__ __forceinline__ V load_vec(const void* p) {
  if constexpr (NT) {
    using R = typename Raw<sizeof(V)>::type;
    return __builtin_bit_cast(V, __builtin_nontemporal_load((const R*)p));
  } else if constexpr (sizeof(V) == 16) {
    // element-aligned 16 bytes (an Arrow slice): through a native vector of the struct's alignment — the struct load itself compiles
    // to 12 + 4 or 8 + 8 bytes (ah_common.h, ah_ld16)
    using RU = ah_raw16<typename std::remove_cv<typename std::remove_reference<decltype(((V*)nullptr)->v[0])>::type>::type>;
    const RU raw = *(const RU*)p;
    return __builtin_bit_cast(V, raw);
  } else {
    return *(const V*)p;
  }
}
template <typename V, bool NT>
__device__ __forceinline__ void store_vec(void* p, const V& v) {
  if constexpr (NT) {
    using R = typename Raw<sizeof(V)>::type;
    __builtin_nontemporal_store(__builtin_bit_cast(R, v), (R*)p);
  } else if constexpr (sizeof(V) == 16) {
    using RU = ah_raw16<typename std::remove_cv<typename std::remove_reference<decltype(((V*)nullptr)->v[0])>::type>::type>;
    *(RU*)p = __builtin_bit_cast(RU, v);
  } else {
    *(V*)p = v;
  }
}

template <typename OUT>
__device__ __forceinline__ OUT float_to_int(double v) {
  unsigned long long bits;
  if (v != v) {
    bits = 0;
  } else if (std::is_same<OUT, uint64_t>::value && v >= 9223372036854775808.0) {
    bits = v >= 18446744073709551616.0 ? ~0ull : (unsigned long long)(long long)(v - 9223372036854775808.0) + 0x8000000000000000ull;
  } else if (v >= 9223372036854775808.0) {
    bits = 0x7fffffffffffffffull;
  } else if (v <= -9223372036854775808.0) {
    bits = 0x8000000000000000ull;
  } else {
    bits = (unsigned long long)(long long)v;
  }
  return (OUT)bits;
}

template <typename IN, typename OUT>
__device__ __forceinline__ OUT convert(IN v) {
  if constexpr (std::is_floating_point<IN>::value && std::is_integral<OUT>::value) return float_to_int<OUT>((double)v);
  else return (OUT)v;  // int → int: modular; int → float, float → float: round to nearest even
}

// CHK: 0 none · 1 integer input must lie in [lo, hi] · 2 float input must be integral and inside OUT's range
template <typename IN, typename OUT, int CHK>
__device__ __forceinline__ bool is_bad(IN v, IN lo, IN hi) {
  if constexpr (CHK == 1) {
    return v < lo || v > hi;
  } else if constexpr (CHK == 2) {
    using L = std::numeric_limits<OUT>;
    const double d = (double)v;
    const double lo_d = (double)L::min();                                  // −2^(k−1) or 0: exact
    const double hi_excl = (double)(L::max() / 2 + 1) * 2.0;               // 2^(k−1) or 2^k: exact
    return !(d == __builtin_trunc(d) && d >= lo_d && d < hi_excl);          // NaN fails every comparison
  } else {
    return false;
  }
}

// NT: both buffers are aligned to their per-lane vector → nontemporal vector accesses
template <typename IN, typename OUT, int CHK, bool NT>
__global__ __launch_bounds__(kBlock) void cast_kernel(const IN* __restrict__ in, const uint8_t* __restrict__ valid, int64_t off, int64_t n,
                                                       OUT* __restrict__ out, IN lo, IN hi, unsigned long long* __restrict__ first_bad) {
  constexpr int W = sizeof(IN) > sizeof(OUT) ? sizeof(IN) : sizeof(OUT);
  constexpr int E = 16 / W;  // elements per lane-vector: the wider side moves 16 bytes
  using VI = VecE<IN, E>;
  using VO = VecE<OUT, E>;
  const int64_t j0 = (int64_t)blockIdx.x * kBlock * kUnroll + threadIdx.x;
  VI x[kUnroll];
#pragma unroll
  for (int k = 0; k < kUnroll; k++) {
    const int64_t e0 = (j0 + (int64_t)k * kBlock) * E;
    if (e0 + E <= n) {
      x[k] = load_vec<VI, NT>(in + e0);
    } else {
#pragma unroll
      for (int e = 0; e < E; e++) x[k].v[e] = e0 + e < n ? in[e0 + e] : (IN)0;
    }
  }
  unsigned long long bad_at = ~0ull;
#pragma unroll
  for (int k = 0; k < kUnroll; k++) {
    const int64_t e0 = (j0 + (int64_t)k * kBlock) * E;
    if (e0 >= n) continue;
    VO o;
    unsigned bad = 0;
#pragma unroll
    for (int e = 0; e < E; e++) {
      o.v[e] = convert<IN, OUT>(x[k].v[e]);
      if (CHK != 0 && is_bad<IN, OUT, CHK>(x[k].v[e], lo, hi)) bad |= 1u << e;
    }
    if (CHK != 0 && bad) {
      const int cnt = n - e0 >= E ? E : (int)(n - e0);
      if (cnt < E) bad &= (1u << cnt) - 1u;
      if (valid) bad &= (unsigned)ah_load_bits64(valid, off + e0, cnt);  // only valid slots can fail
      if (bad) {
        const unsigned long long at = (unsigned long long)e0 + (unsigned)(__ffs((int)bad) - 1);
        if (at < bad_at) bad_at = at;
      }
    }
    if (e0 + E <= n) {
      store_vec<VO, NT>(out + e0, o);
    } else {
#pragma unroll
      for (int e = 0; e < E; e++) if (e0 + e < n) out[e0 + e] = o.v[e];
    }
  }
  if (CHK != 0 && bad_at != ~0ull) atomicMin(first_bad, bad_at);
}

template <typename OUT>
__global__ __launch_bounds__(kBlock) void bool_to_num_kernel(const uint8_t* __restrict__ bits, int64_t off, int64_t n, OUT* __restrict__ out) {
  constexpr int E = 16 / sizeof(OUT);
  using VO = VecE<OUT, E>;
  const int64_t e0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * E;
  if (e0 >= n) return;
  const int cnt = n - e0 >= E ? E : (int)(n - e0);
  const unsigned b = (unsigned)ah_load_bits64(bits, off + e0, cnt);
  VO o;
#pragma unroll
  for (int e = 0; e < E; e++) o.v[e] = (b >> e) & 1u ? (OUT)1 : (OUT)0;
  if (cnt == E) {
    if (((uintptr_t)(out + e0) & 15) == 0) store_vec<VO, true>(out + e0, o);
    else *(VO*)(out + e0) = o;
  } else {
    for (int e = 0; e < cnt; e++) out[e0 + e] = o.v[e];
  }
}

template <typename IN, typename OUT, int CHK>
int launch_cast(ah_ctx* c, const void* in, const uint8_t* valid, int64_t off, int64_t n, void* out, IN lo, IN hi) {
  constexpr int W = sizeof(IN) > sizeof(OUT) ? sizeof(IN) : sizeof(OUT);
  constexpr int E = 16 / W;
  const int64_t per_block = (int64_t)kBlock * kUnroll * E;
  const unsigned grid = (unsigned)ah_ceil_div(n, per_block);
  unsigned long long* first_bad = (unsigned long long*)&c->dscalars[12];
  const bool nt = c->tune_nt && ((uintptr_t)in % (E * sizeof(IN))) == 0 && ((uintptr_t)out % (E * sizeof(OUT))) == 0;
  if (nt) cast_kernel<IN, OUT, CHK, true><<<grid, kBlock, 0, c->stream>>>((const IN*)in, valid, off, n, (OUT*)out, lo, hi, first_bad);
  else cast_kernel<IN, OUT, CHK, false><<<grid, kBlock, 0, c->stream>>>((const IN*)in, valid, off, n, (OUT*)out, lo, hi, first_bad);
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

template <typename T> struct Lim { static constexpr bool is_int = std::is_integral<T>::value; };

// bounds of a checked int → int / int → float cast, in the INPUT type (getSafeMinMaxSigned/Unsigned,
// helpers.go:496-543; checkIntToFloatTrunc, numeric_cast.go:698-729).  need = false: every input fits.
template <typename IN, typename OUT>
void int_bounds(bool* need, IN* lo, IN* hi) {
  using LI = std::numeric_limits<IN>;
  if constexpr (std::is_integral<OUT>::value) {
    using LO = std::numeric_limits<OUT>;
    // compare through __int128 so that every signed/unsigned pairing is exact
    __int128 lo_w = (__int128)LI::min() > (__int128)LO::min() ? (__int128)LI::min() : (__int128)LO::min();
    __int128 hi_w = (__int128)LI::max() < (__int128)LO::max() ? (__int128)LI::max() : (__int128)LO::max();
    *lo = (IN)lo_w;
    *hi = (IN)hi_w;
    *need = !((__int128)LI::min() >= lo_w && (__int128)LI::max() <= hi_w);
  } else {
    const int digits = std::numeric_limits<OUT>::digits;  // 24 / 53
    if ((int)sizeof(IN) * 8 - (std::is_signed<IN>::value ? 1 : 0) <= digits) { *need = false; *lo = LI::min(); *hi = LI::max(); return; }
    const unsigned long long limit = 1ull << digits;
    *hi = (IN)limit;
    *lo = std::is_signed<IN>::value ? (IN)(-(long long)limit) : (IN)0;
    *need = true;
  }
}

template <typename IN, typename OUT>
int cast_pair(ah_ctx* c, const void* in, const uint8_t* valid, int64_t off, int64_t n, void* out, int allow_int_overflow,
              int allow_float_truncate, int* chk_out, IN* lo_out, IN* hi_out) {
  *chk_out = 0;
  if constexpr (std::is_integral<IN>::value) {
    const bool checked = std::is_integral<OUT>::value ? !allow_int_overflow : !allow_float_truncate;
    bool need = false;
    IN lo = 0, hi = 0;
    if (checked) int_bounds<IN, OUT>(&need, &lo, &hi);
    if (need) {
      *chk_out = 1; *lo_out = lo; *hi_out = hi;
      return launch_cast<IN, OUT, 1>(c, in, valid, off, n, out, lo, hi);
    }
    return launch_cast<IN, OUT, 0>(c, in, valid, off, n, out, (IN)0, (IN)0);
  } else {
    if constexpr (std::is_integral<OUT>::value) {
      if (!allow_float_truncate) {
        *chk_out = 2;
        return launch_cast<IN, OUT, 2>(c, in, valid, off, n, out, (IN)0, (IN)0);
      }
    }
    return launch_cast<IN, OUT, 0>(c, in, valid, off, n, out, (IN)0, (IN)0);
  }
}

inline const char* type_name(int t) {
  switch (t) {
    case AH_UINT8: return "uint8"; case AH_INT8: return "int8"; case AH_UINT16: return "uint16"; case AH_INT16: return "int16";
    case AH_UINT32: return "uint32"; case AH_INT32: return "int32"; case AH_UINT64: return "uint64"; case AH_INT64: return "int64";
    case AH_FLOAT32: return "float32"; case AH_FLOAT64: return "float64";
  }
  return "?";
}

template <typename IN>
int cast_from(ah_ctx* c, int out_type, const void* in, const uint8_t* valid, int64_t off, int64_t n, void* out, int aio, int aft) {
  int chk = 0;
  IN lo = 0, hi = 0;
  int rc;
  const bool any_check = std::is_integral<IN>::value ? true : !aft;
  if (any_check) AH_HIP(c, hipMemsetAsync(&c->dscalars[12], 0xFF, sizeof(uint64_t), c->stream));
  switch (out_type) {
#define AH_CAST_TO(ID, OUT) case ID: rc = cast_pair<IN, OUT>(c, in, valid, off, n, out, aio, aft, &chk, &lo, &hi); break;
    AH_CAST_TO(AH_UINT8, uint8_t) AH_CAST_TO(AH_INT8, int8_t) AH_CAST_TO(AH_UINT16, uint16_t) AH_CAST_TO(AH_INT16, int16_t)
    AH_CAST_TO(AH_UINT32, uint32_t) AH_CAST_TO(AH_INT32, int32_t) AH_CAST_TO(AH_UINT64, uint64_t) AH_CAST_TO(AH_INT64, int64_t)
    AH_CAST_TO(AH_FLOAT32, float) AH_CAST_TO(AH_FLOAT64, double)
#undef AH_CAST_TO
    default: return ah_fail(c, AH_ENOTIMPL, "cast: unsupported target type %d", out_type);
  }
  if (rc != AH_OK || chk == 0) return rc;
  // a checked cast: was there an offender?
  AH_HIP(c, hipMemcpyAsync(c->pinned, &c->dscalars[12], sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  AH_HIP(c, hipStreamSynchronize(c->stream));
  const unsigned long long at = *(volatile unsigned long long*)c->pinned;
  if (at == ~0ull) return AH_OK;
  IN v;
  AH_HIP(c, hipMemcpyAsync(c->pinned, (const IN*)in + at, sizeof(IN), hipMemcpyDeviceToHost, c->stream));
  AH_HIP(c, hipStreamSynchronize(c->stream));
  memcpy(&v, (const void*)c->pinned, sizeof(IN));
  if constexpr (std::is_integral<IN>::value) {
    if (std::is_signed<IN>::value)
      return ah_fail(c, AH_EINVALID, "integer value %lld not in range: %lld to %lld", (long long)v, (long long)lo, (long long)hi);  // helpers.go:591-594
    return ah_fail(c, AH_EINVALID, "integer value %llu not in range: %llu to %llu", (unsigned long long)v, (unsigned long long)lo,
                   (unsigned long long)hi);
  } else {
    const double d = (double)v;  // numeric_cast.go:614-617, Go's %f
    if (d != d) return ah_fail(c, AH_EINVALID, "float value NaN was truncated converting to %s", type_name(out_type));
    if (__builtin_isinf(d)) return ah_fail(c, AH_EINVALID, "float value %sInf was truncated converting to %s", d > 0 ? "+" : "-", type_name(out_type));
    return ah_fail(c, AH_EINVALID, "float value %f was truncated converting to %s", d, type_name(out_type));
  }
}

}  // namespace
}  // namespace ah_cast_impl
