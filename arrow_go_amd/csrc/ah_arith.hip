// ah_arith.hip — element-wise ADD / SUB / MUL (array∘array, array∘scalar,
// scalar∘array), ABS / NEGATE / SIGN, and the checked integer variants.
//
// Replaces: _arithmetic_binary_avx2, _arithmetic_arr_scalar_avx2,
//   _arithmetic_scalar_arr_avx2, _arithmetic_unary_same_types_avx2
//   (kernels/base_arithmetic_avx2_amd64.go:27-60; C truth
//   kernels/_lib/base_arithmetic.cc:52-273,441-483; Go loop
//   kernels/base_arithmetic.go:110-134), reached from compute.Add/Subtract/Multiply
//   through ScalarBinary (kernels/helpers.go:193-236); and the pure-Go checked path
//   kernels/base_arithmetic.go:249-286 through ScalarBinaryNotNull
//   (kernels/helpers.go:284-380).
//
// Roofline: HBM, 3·w bytes per row (2·w for a scalar operand), no reuse.  One
// 16-byte vector per lane per operand (global_load_dwordx4 / global_store_dwordx4,
// nontemporal), kUnroll vectors in flight per lane, grid-stride.  Integer ops are
// done in the unsigned type of the same width (two's-complement wraparound, as the
// C source does for MUL :107-124 and the SIMD lanes do for ADD/SUB).
#include <type_traits>
#include "ah_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kUnroll = 4;

template <typename T>
using Vec16 = T __attribute__((ext_vector_type(16 / sizeof(T))));

enum { OP_ADD = 0, OP_SUB = 1, OP_MUL = 2, OP_ABS = 3, OP_NEG = 4, OP_SIGN = 5 };

template <typename T, int OP>
__device__ __forceinline__ T apply_binary(T a, T b) {
  if (OP == OP_ADD) return (T)(a + b);
  if (OP == OP_SUB) return (T)(a - b);
  return (T)(a * b);
}

// unary ops need signedness: ST is the logical (possibly signed / float) type
template <typename ST, int OP>
__device__ __forceinline__ ST apply_unary(ST x) {
  if constexpr (__is_floating_point(ST)) {
    if (OP == OP_ABS) return __builtin_fabs(x);  // clears the sign bit (base_arithmetic.cc:139-146)
    if (OP == OP_NEG) return -x;
    return __builtin_isnan(x) ? x : (x == 0 ? (ST)0 : (__builtin_signbit(x) ? (ST)-1 : (ST)1));
  } else if constexpr ((ST)-1 > (ST)0) {  // unsigned
    if (OP == OP_ABS) return x;
    if (OP == OP_NEG) return (ST)(~x + 1);
    return (ST)(x > 0 ? 1 : 0);
  } else {
    using U = typename std::make_unsigned<ST>::type;
    if (OP == OP_ABS) {
      U m = x < 0 ? (U)~(U)0 : (U)0;
      return (ST)(((U)x + m) ^ m);
    }
    if (OP == OP_NEG) return (ST)((U)0 - (U)x);
    return (ST)(x > 0 ? 1 : (x ? -1 : 0));
  }
}

template <typename V, bool NT>
__device__ __forceinline__ V vload(const V* p) {
  if (NT) return __builtin_nontemporal_load(p);
  return *p;
}
template <typename V, bool NT>
__device__ __forceinline__ void vstore(V* p, V v) {
  if (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// SHAPE: 0 = l[i] op r[i], 1 = l[i] op s, 2 = s op r[i]   (array operand in `a`,
// for shape 0 the second array in `b`).  ALIGNED: all pointers 16-byte aligned →
// ext-vector accesses; otherwise element-aligned 16-byte structs (Arrow slices).
template <typename T, int OP, int SHAPE, bool ALIGNED, bool NT>
__global__ __launch_bounds__(kBlock) void binary_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                         T* __restrict__ out, int64_t len, T scalar, unsigned xmap) {
  constexpr int V = 16 / sizeof(T);
  using VT = typename std::conditional<ALIGNED, Vec16<T>, ah_vec16<T>>::type;
  const int64_t nvec = len / V;
  const VT* av = (const VT*)a;
  const VT* bv = (const VT*)b;
  VT* ov = (VT*)out;
  const int64_t stride = (int64_t)gridDim.x * kBlock * kUnroll;
  // xmap (measurement switch arith_xcd_map): block b runs on XCD b & 7 — give every XCD one contiguous eighth of the columns
  const unsigned bid = xmap ? (blockIdx.x & 7u) * xmap + (blockIdx.x >> 3) : blockIdx.x;
  int64_t i = (int64_t)bid * kBlock * kUnroll + threadIdx.x;
  auto compute = [&](const VT& x, const VT& y) {
    VT o;
#pragma unroll
    for (int e = 0; e < V; e++) {
      T xe, ye;
      if constexpr (ALIGNED) { xe = x[e]; ye = SHAPE == 0 ? y[e] : scalar; }
      else { xe = x.v[e]; ye = SHAPE == 0 ? y.v[e] : scalar; }
      T r = SHAPE == 2 ? apply_binary<T, OP>(ye, xe) : apply_binary<T, OP>(xe, ye);
      if constexpr (ALIGNED) o[e] = r; else o.v[e] = r;
    }
    return o;
  };
  for (; i + (int64_t)(kUnroll - 1) * kBlock < nvec; i += stride) {
    VT x[kUnroll], y[kUnroll] = {};
#pragma unroll
    for (int k = 0; k < kUnroll; k++) {
      if constexpr (ALIGNED) {
        x[k] = vload<VT, NT>(&av[i + (int64_t)k * kBlock]);
        if (SHAPE == 0) y[k] = vload<VT, NT>(&bv[i + (int64_t)k * kBlock]);
      } else {
        x[k] = NT ? ah_ld16_nt<T>((const T*)(av + (i + (int64_t)k * kBlock))) : ah_ld16<T>((const T*)(av + (i + (int64_t)k * kBlock)));
        if (SHAPE == 0) y[k] = NT ? ah_ld16_nt<T>((const T*)(bv + (i + (int64_t)k * kBlock))) : ah_ld16<T>((const T*)(bv + (i + (int64_t)k * kBlock)));
      }
    }
#pragma unroll
    for (int k = 0; k < kUnroll; k++) {
      VT o = compute(x[k], y[k]);
      if constexpr (ALIGNED) vstore<VT, NT>(&ov[i + (int64_t)k * kBlock], o);
      else if constexpr (NT) ah_st16_nt<T>((T*)(ov + (i + (int64_t)k * kBlock)), o);
      else ah_st16<T>((T*)(ov + (i + (int64_t)k * kBlock)), o);
    }
  }
#pragma unroll
  for (int k = 0; k < kUnroll; k++) {
    int64_t j = i + (int64_t)k * kBlock;
    if (j < nvec) {
      VT x = av[j], y = x;
      if (SHAPE == 0) y = bv[j];
      ov[j] = compute(x, y);
    }
  }
  // scalar tail (< V elements) — block 0
  if (blockIdx.x == 0) {
    int64_t j = nvec * V + threadIdx.x;
    if (j < len) {
      T xe = a[j], ye = SHAPE == 0 ? b[j] : scalar;
      out[j] = SHAPE == 2 ? apply_binary<T, OP>(ye, xe) : apply_binary<T, OP>(xe, ye);
    }
  }
}

template <typename ST, int OP, bool ALIGNED, bool NT>
__global__ __launch_bounds__(kBlock) void unary_kernel(const ST* __restrict__ a, ST* __restrict__ out, int64_t len) {
  constexpr int V = 16 / sizeof(ST);
  using VT = typename std::conditional<ALIGNED, Vec16<ST>, ah_vec16<ST>>::type;
  const int64_t nvec = len / V;
  const VT* av = (const VT*)a;
  VT* ov = (VT*)out;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
    VT x, o;
    if constexpr (ALIGNED) x = vload<VT, NT>(&av[i]); else x = ah_ld16<ST>((const ST*)(av + i));
#pragma unroll
    for (int e = 0; e < V; e++) {
      if constexpr (ALIGNED) o[e] = apply_unary<ST, OP>(x[e]);
      else o.v[e] = apply_unary<ST, OP>(x.v[e]);
    }
    if constexpr (ALIGNED) vstore<VT, NT>(&ov[i], o); else ah_st16<ST>((ST*)(ov + i), o);
  }
  if (blockIdx.x == 0) {
    int64_t j = nvec * V + threadIdx.x;
    if (j < len) out[j] = apply_unary<ST, OP>(a[j]);
  }
}

template <typename T, int OP, int SHAPE>
int launch_binary(ah_ctx* c, const void* a, const void* b, void* out, int64_t len, T scalar) {
  constexpr int V = 16 / sizeof(T);
  bool aligned = (((uintptr_t)a | (uintptr_t)out | (SHAPE == 0 ? (uintptr_t)b : 0)) & 15) == 0;
  int64_t iters = ah_ceil_div(len / V + 1, (int64_t)kBlock * kUnroll);
  unsigned grid = ah_stream_grid(c, iters, /*default_bpc=*/0);
  unsigned xmap = 0;
  if (c->opt_arith_xcd_map && grid >= 64) { grid = (grid + 7u) & ~7u; xmap = grid >> 3; }
  const T* pa = (const T*)a; const T* pb = (const T*)b; T* po = (T*)out;
  if (aligned) {
    if (c->tune_nt) binary_kernel<T, OP, SHAPE, true, true><<<grid, kBlock, 0, c->stream>>>(pa, pb, po, len, scalar, xmap);
    else binary_kernel<T, OP, SHAPE, true, false><<<grid, kBlock, 0, c->stream>>>(pa, pb, po, len, scalar, xmap);
  } else {   // an element-aligned slice: the same 16-byte accesses and hints at the elements' alignment
    if (c->tune_nt) binary_kernel<T, OP, SHAPE, false, true><<<grid, kBlock, 0, c->stream>>>(pa, pb, po, len, scalar, xmap);
    else binary_kernel<T, OP, SHAPE, false, false><<<grid, kBlock, 0, c->stream>>>(pa, pb, po, len, scalar, xmap);
  }
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

template <typename T>
int dispatch_binary_op(ah_ctx* c, int op, int shape, const void* l, const void* r, void* out, int64_t len) {
  // for shapes AS/SA the scalar operand is HOST memory
  T scalar = 0;
  const void* a = l; const void* b = r;
  if (shape == AH_SHAPE_AS) { memcpy(&scalar, r, sizeof(T)); b = nullptr; }
  if (shape == AH_SHAPE_SA) { memcpy(&scalar, l, sizeof(T)); a = r; b = nullptr; }
#define AH_SHAPE_SWITCH(OPC)                                                              \
  switch (shape) {                                                                        \
    case AH_SHAPE_AA: return launch_binary<T, OPC, 0>(c, a, b, out, len, scalar);         \
    case AH_SHAPE_AS: return launch_binary<T, OPC, 1>(c, a, b, out, len, scalar);         \
    case AH_SHAPE_SA: return launch_binary<T, OPC, 2>(c, a, b, out, len, scalar);         \
  }
  switch (op) {
    case AH_OP_ADD: case AH_OP_ADD_CHECKED: AH_SHAPE_SWITCH(OP_ADD) break;
    case AH_OP_SUB: case AH_OP_SUB_CHECKED: AH_SHAPE_SWITCH(OP_SUB) break;
    case AH_OP_MUL: case AH_OP_MUL_CHECKED: AH_SHAPE_SWITCH(OP_MUL) break;
  }
#undef AH_SHAPE_SWITCH
  return ah_fail(c, AH_ENOTIMPL, "arithmetic: unsupported op %d / shape %d", op, shape);
}

int arith_binary(ah_ctx* c, int type, int op, int shape, const void* l, const void* r, void* out, int64_t len) {
  if (len < 0) return ah_fail(c, AH_EINVALID, "arithmetic: negative length");
  if (len == 0) return AH_OK;
  int w = ah_type_width(type);
  if (!w) return ah_fail(c, AH_ENOTIMPL, "arithmetic: unsupported type id %d", type);
  const void* arr0 = shape == AH_SHAPE_SA ? r : l;
  if ((((uintptr_t)arr0 | (uintptr_t)out | (shape == AH_SHAPE_AA ? (uintptr_t)r : 0)) & (uintptr_t)(w - 1)) != 0)
    return ah_fail(c, AH_EINVALID, "arithmetic: buffer not element-aligned");
  switch (type) {
    case AH_UINT8: case AH_INT8: return dispatch_binary_op<uint8_t>(c, op, shape, l, r, out, len);
    case AH_UINT16: case AH_INT16: return dispatch_binary_op<uint16_t>(c, op, shape, l, r, out, len);
    case AH_UINT32: case AH_INT32: return dispatch_binary_op<uint32_t>(c, op, shape, l, r, out, len);
    case AH_UINT64: case AH_INT64: return dispatch_binary_op<uint64_t>(c, op, shape, l, r, out, len);
    case AH_FLOAT32: return dispatch_binary_op<float>(c, op, shape, l, r, out, len);
    case AH_FLOAT64: return dispatch_binary_op<double>(c, op, shape, l, r, out, len);
  }
  return ah_fail(c, AH_ENOTIMPL, "arithmetic: unsupported type id %d", type);
}

template <typename ST>
int dispatch_unary(ah_ctx* c, int op, const void* in, void* out, int64_t len) {
  constexpr int V = 16 / sizeof(ST);
  bool aligned = (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
  unsigned grid = ah_stream_grid(c, ah_ceil_div(len / V + 1, kBlock), /*default_bpc=*/0);
  const ST* a = (const ST*)in; ST* o = (ST*)out;
#define AH_UNARY(OPC)                                                                          \
  if (aligned) unary_kernel<ST, OPC, true, true><<<grid, kBlock, 0, c->stream>>>(a, o, len);   \
  else unary_kernel<ST, OPC, false, false><<<grid, kBlock, 0, c->stream>>>(a, o, len);
  switch (op) {
    case AH_OP_ABS: AH_UNARY(OP_ABS) break;
    case AH_OP_NEGATE: AH_UNARY(OP_NEG) break;
    case AH_OP_SIGN: AH_UNARY(OP_SIGN) break;
    default: return ah_fail(c, AH_ENOTIMPL, "arithmetic_unary: unsupported op %d", op);
  }
#undef AH_UNARY
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

// ---- checked integer ops ---------------------------------------------------------
// flag word: bit 0 set ⇔ some tested slot overflowed by the reference's carry test.
template <typename ST, int OP>
__device__ __forceinline__ ST checked_one(ST a, ST b, bool valid, bool& ovf) {
  using U = typename std::make_unsigned<ST>::type;
  constexpr bool kSigned = (ST)-1 < (ST)0;
  constexpr int bits = sizeof(ST) * 8;
  if (OP == OP_MUL) {
    // mulWithOverflow (base_arithmetic.go:84-106), every slot (ScalarBinary), null payloads included
    constexpr ST tmin = kSigned ? (ST)((U)1 << (bits - 1)) : (ST)0;
    constexpr ST tmax = kSigned ? (ST)(~((U)1 << (bits - 1))) : (ST)~(U)0;
    bool o = false;
    if (a > 0) { if (b > 0) { if (a > (ST)(tmax / b)) o = true; } else { if (b < (ST)(tmin / a)) o = true; } }
    else if (b > 0) { if (a < (ST)(tmin / b)) o = true; }
    else { if (a != 0 && b < (ST)(tmax / a)) o = true; }
    ovf |= o;
    return o ? (ST)0 : (ST)((U)a * (U)b);
  }
  if (!valid) return (ST)0;  // helpers.go:303-306: null slots hold the zero value
  U ua = (U)a, ub = (U)b, o, cy;
  if (OP == OP_ADD) { o = (U)(ua + ub); cy = (U)((ua & ub) | ((ua | ub) & (U)~o)); }
  else { o = (U)(ua - ub); cy = (U)(((U)~ua & ub) | ((U) ~(ua ^ ub) & o)); }
  // `carry > 0` after an ARITHMETIC shift by bits-2 for signed T, logical shift by bits-1 for
  // unsigned T (base_arithmetic.go:250-262): signed ⇒ top carry bit clear ∧ next bit set
  bool top = (cy >> (bits - 1)) & 1, next = (cy >> (bits - 2)) & 1;
  ovf |= kSigned ? (!top && next) : top;
  return (ST)o;
}

// One 16-byte vector per lane per operand, validity as V bits per lane out of the bitmaps
// (two aligned 8-byte loads at most), result zeroed under nulls, 16-byte store.
// 16-byte aligned operands stream through nontemporal vector accesses; `aligned` is wave-uniform
template <typename ST>
__device__ __forceinline__ ah_vec16<ST> load16(const ST* base, int64_t i, bool aligned) {
  ah_vec16<ST> v;
  if (aligned) {
    const Vec16<ST> t = __builtin_nontemporal_load((const Vec16<ST>*)base + i);
    __builtin_memcpy(&v, &t, 16);
  } else {
    v = ah_ld16<ST>(base + i * (int64_t)(16 / sizeof(ST)));
  }
  return v;
}
template <typename ST>
__device__ __forceinline__ void store16(ST* base, int64_t i, const ah_vec16<ST>& v, bool aligned) {
  if (aligned) {
    Vec16<ST> t;
    __builtin_memcpy(&t, &v, 16);
    __builtin_nontemporal_store(t, (Vec16<ST>*)base + i);
  } else {
    ah_st16<ST>(base + i * (int64_t)(16 / sizeof(ST)), v);
  }
}

template <typename ST, int OP /*OP_ADD, OP_SUB, OP_MUL*/, int SHAPE>
__global__ __launch_bounds__(kBlock) void checked_kernel(const ST* __restrict__ l, const uint8_t* __restrict__ lv, int64_t loff,
                                                          const ST* __restrict__ r, const uint8_t* __restrict__ rv, int64_t roff,
                                                          ST scalar, ST* __restrict__ out, int64_t len, unsigned* __restrict__ flag, int aligned) {
  constexpr int V = 16 / sizeof(ST);
  using VT = ah_vec16<ST>;
  bool ovf = false;
  const int64_t nvec = len / V;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
    VT a, b, o;
    if (SHAPE != 2) a = load16<ST>(l, i, aligned);
    if (SHAPE != 1) b = load16<ST>(r, i, aligned);
    unsigned vbits = (1u << V) - 1;
    if (OP != OP_MUL) {
      if (SHAPE != 2 && lv) vbits &= (unsigned)ah_load_bits64(lv, loff + i * V, V);
      if (SHAPE != 1 && rv) vbits &= (unsigned)ah_load_bits64(rv, roff + i * V, V);
    }
#pragma unroll
    for (int e = 0; e < V; e++)
      o.v[e] = checked_one<ST, OP>(SHAPE == 2 ? scalar : a.v[e], SHAPE == 1 ? scalar : b.v[e], (vbits >> e) & 1, ovf);
    store16<ST>(out, i, o, aligned);
  }
  if (blockIdx.x == 0) {  // < V trailing elements
    int64_t j = nvec * V + threadIdx.x;
    if (j < len) {
      bool valid = (SHAPE == 2 || ah_bit(lv, loff + j)) && (SHAPE == 1 || ah_bit(rv, roff + j));
      out[j] = checked_one<ST, OP>(SHAPE == 2 ? scalar : l[j], SHAPE == 1 ? scalar : r[j], valid, ovf);
    }
  }
  if (__any(ovf) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

template <typename ST>
int dispatch_checked(ah_ctx* c, int op, int shape, const void* l, const uint8_t* lv, int64_t loff, const void* r,
                     const uint8_t* rv, int64_t roff, void* out, int64_t len, unsigned* flag) {
  ST scalar = 0;
  if (shape == AH_SHAPE_AS) memcpy(&scalar, r, sizeof(ST));
  if (shape == AH_SHAPE_SA) memcpy(&scalar, l, sizeof(ST));
  unsigned grid = ah_stream_grid(c, ah_ceil_div(len / (16 / (int64_t)sizeof(ST)) + 1, kBlock), /*default_bpc=*/0);
  const ST* pl = (const ST*)l; const ST* pr = (const ST*)r; ST* po = (ST*)out;
  const int aligned = c->tune_nt && ((((uintptr_t)out) | (shape != AH_SHAPE_SA ? (uintptr_t)l : 0) | (shape != AH_SHAPE_AS ? (uintptr_t)r : 0)) & 15) == 0;
#define AH_CHK(OPC)                                                                                                       \
  switch (shape) {                                                                                                        \
    case AH_SHAPE_AA: checked_kernel<ST, OPC, 0><<<grid, kBlock, 0, c->stream>>>(pl, lv, loff, pr, rv, roff, scalar, po, len, flag, aligned); break; \
    case AH_SHAPE_AS: checked_kernel<ST, OPC, 1><<<grid, kBlock, 0, c->stream>>>(pl, lv, loff, nullptr, nullptr, 0, scalar, po, len, flag, aligned); break; \
    case AH_SHAPE_SA: checked_kernel<ST, OPC, 2><<<grid, kBlock, 0, c->stream>>>(nullptr, nullptr, 0, pr, rv, roff, scalar, po, len, flag, aligned); break; \
  }
  switch (op) {
    case AH_OP_ADD_CHECKED: AH_CHK(OP_ADD) break;
    case AH_OP_SUB_CHECKED: AH_CHK(OP_SUB) break;
    case AH_OP_MUL_CHECKED: AH_CHK(OP_MUL) break;
  }
#undef AH_CHK
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

}  // namespace

AH_EXPORT int ah_arithmetic_binary(ah_ctx* c, int type, int8_t op, const void* l, const void* r, void* out, int64_t len) {
  AH_ENTER(c);
  return arith_binary(c, type, op, AH_SHAPE_AA, l, r, out, len);
}
AH_EXPORT int ah_arithmetic_arr_scalar(ah_ctx* c, int type, int8_t op, const void* l, const void* r_host, void* out, int64_t len) {
  AH_ENTER(c);
  if (!r_host) return ah_fail(c, AH_EINVALID, "arithmetic_arr_scalar: null scalar");
  return arith_binary(c, type, op, AH_SHAPE_AS, l, r_host, out, len);
}
AH_EXPORT int ah_arithmetic_scalar_arr(ah_ctx* c, int type, int8_t op, const void* l_host, const void* r, void* out, int64_t len) {
  AH_ENTER(c);
  if (!l_host) return ah_fail(c, AH_EINVALID, "arithmetic_scalar_arr: null scalar");
  return arith_binary(c, type, op, AH_SHAPE_SA, l_host, r, out, len);
}

AH_EXPORT int ah_arithmetic_unary(ah_ctx* c, int type, int8_t op, const void* in, void* out, int64_t len) {
  AH_ENTER(c);
  if (len < 0) return ah_fail(c, AH_EINVALID, "arithmetic_unary: negative length");
  if (len == 0) return AH_OK;
  switch (type) {
    case AH_UINT8: return dispatch_unary<uint8_t>(c, op, in, out, len);
    case AH_INT8: return dispatch_unary<int8_t>(c, op, in, out, len);
    case AH_UINT16: return dispatch_unary<uint16_t>(c, op, in, out, len);
    case AH_INT16: return dispatch_unary<int16_t>(c, op, in, out, len);
    case AH_UINT32: return dispatch_unary<uint32_t>(c, op, in, out, len);
    case AH_INT32: return dispatch_unary<int32_t>(c, op, in, out, len);
    case AH_UINT64: return dispatch_unary<uint64_t>(c, op, in, out, len);
    case AH_INT64: return dispatch_unary<int64_t>(c, op, in, out, len);
    case AH_FLOAT32: return dispatch_unary<float>(c, op, in, out, len);
    case AH_FLOAT64: return dispatch_unary<double>(c, op, in, out, len);
  }
  return ah_fail(c, AH_ENOTIMPL, "arithmetic_unary: unsupported type id %d", type);
}

AH_EXPORT int ah_arithmetic_checked(ah_ctx* c, int type, int8_t op, int shape,
                                    const void* l, const uint8_t* lvalid, int64_t loff,
                                    const void* r, const uint8_t* rvalid, int64_t roff,
                                    int scalar_valid, void* out, int64_t len) {
  AH_ENTER(c);
  if (len < 0) return ah_fail(c, AH_EINVALID, "arithmetic_checked: negative length");
  if (len == 0) return AH_OK;
  if (shape < AH_SHAPE_AA || shape > AH_SHAPE_SA) return ah_fail(c, AH_EINVALID, "arithmetic_checked: bad shape %d", shape);
  // floats: checked == unchecked SIMD kernels (base_arithmetic_amd64.go:109-117)
  if (type == AH_FLOAT32 || type == AH_FLOAT64) return arith_binary(c, type, op, shape, l, r, out, len);
  if (op != AH_OP_ADD_CHECKED && op != AH_OP_SUB_CHECKED && op != AH_OP_MUL_CHECKED)
    return ah_fail(c, AH_ENOTIMPL, "arithmetic_checked: unsupported op %d", op);
  int w = ah_type_width(type);
  if (!w) return ah_fail(c, AH_ENOTIMPL, "arithmetic_checked: unsupported type id %d", type);
  if (op != AH_OP_MUL_CHECKED && shape != AH_SHAPE_AA && !scalar_valid) {
    // null scalar: output stays as allocated = zero (helpers.go:312-314,341-343)
    AH_HIP(c, hipMemsetAsync(out, 0, (size_t)len * w, c->stream));
    return AH_OK;
  }
  unsigned* flag = (unsigned*)c->dscalars;
  AH_HIP(c, hipMemsetAsync(flag, 0, sizeof(unsigned), c->stream));
  int rc;
  switch (type) {
    case AH_UINT8: rc = dispatch_checked<uint8_t>(c, op, shape, l, lvalid, loff, r, rvalid, roff, out, len, flag); break;
    case AH_INT8: rc = dispatch_checked<int8_t>(c, op, shape, l, lvalid, loff, r, rvalid, roff, out, len, flag); break;
    case AH_UINT16: rc = dispatch_checked<uint16_t>(c, op, shape, l, lvalid, loff, r, rvalid, roff, out, len, flag); break;
    case AH_INT16: rc = dispatch_checked<int16_t>(c, op, shape, l, lvalid, loff, r, rvalid, roff, out, len, flag); break;
    case AH_UINT32: rc = dispatch_checked<uint32_t>(c, op, shape, l, lvalid, loff, r, rvalid, roff, out, len, flag); break;
    case AH_INT32: rc = dispatch_checked<int32_t>(c, op, shape, l, lvalid, loff, r, rvalid, roff, out, len, flag); break;
    case AH_UINT64: rc = dispatch_checked<uint64_t>(c, op, shape, l, lvalid, loff, r, rvalid, roff, out, len, flag); break;
    case AH_INT64: rc = dispatch_checked<int64_t>(c, op, shape, l, lvalid, loff, r, rvalid, roff, out, len, flag); break;
    default: return ah_fail(c, AH_ENOTIMPL, "arithmetic_checked: unsupported type id %d", type);
  }
  if (rc != AH_OK) return rc;
  AH_HIP(c, hipMemcpyAsync(c->pinned, flag, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
  AH_HIP(c, hipStreamSynchronize(c->stream));
  if (*(volatile unsigned*)c->pinned & 1u) return ah_fail(c, AH_EOVERFLOW, "overflow");
  return AH_OK;
}
