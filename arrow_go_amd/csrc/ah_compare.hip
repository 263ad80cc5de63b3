// ah_compare.hip — element-wise comparison → LSB-first packed bitmap.
//
// Replaces: the 12 _comparison_{equal,not_equal,greater,greater_equal}_{arr_arr,
//   arr_scalar,scalar_arr}_avx2 leaves (kernels/scalar_comparison_avx2_amd64.go;
//   C truth kernels/_lib/scalar_comparison.cc:63-256; Go path
//   kernels/scalar_comparisons.go:51-218) behind compute's "equal", "not_equal",
//   "greater", "greater_equal" (+ "less"/"less_equal" by operand swap,
//   compute/scalar_compare.go:73-99).
//
// Roofline: HBM read, w bytes in + 1/8 byte out per row (2·w + 1/8 for
// array∘array).  Each lane loads one 16-byte vector (V = 16/w elements), forms
// its V result bits, and the wave merges them to 32-bit words with log2(32/V)
// shuffles, so one lane in every 32/V stores a whole dword: the bitmap is written
// with 4-byte stores, never bit by bit.  Bits outside [prefix, prefix+length) of
// the output are preserved: the ≤7-bit prefix and the ragged tail are
// read-modify-written by exactly one lane each (the reference does the same with
// set_bit_to, scalar_comparison.cc:59-61,71-81,91-95).
#include <type_traits>
#include "ah_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kUnroll = 4;

template <typename T>
using Vec16 = T __attribute__((ext_vector_type(16 / sizeof(T))));

template <typename T, int OP>
__device__ __forceinline__ bool cmp(T a, T b) {
  if (OP == AH_CMP_EQ) return a == b;
  if (OP == AH_CMP_NE) return a != b;
  if (OP == AH_CMP_GT) return a > b;
  return a >= b;
}

struct alignas(1) U32Unaligned { uint8_t b[4]; };

__device__ __forceinline__ void store_u32_any(uint8_t* p, uint32_t w) {
  U32Unaligned u;
  u.b[0] = (uint8_t)w; u.b[1] = (uint8_t)(w >> 8); u.b[2] = (uint8_t)(w >> 16); u.b[3] = (uint8_t)(w >> 24);
  *(U32Unaligned*)p = u;
}

// write `nvalid` (1..32) low bits of w at body byte position p, preserving the rest
__device__ __forceinline__ void store_bits(uint8_t* p, uint32_t w, int nvalid) {
  if (nvalid >= 32) { store_u32_any(p, w); return; }
  int nbytes = nvalid >> 3, rem = nvalid & 7;
  for (int k = 0; k < nbytes; k++) p[k] = (uint8_t)(w >> (8 * k));
  if (rem) {
    uint8_t m = (uint8_t)((1u << rem) - 1);
    p[nbytes] = (uint8_t)((p[nbytes] & ~m) | ((w >> (8 * nbytes)) & m));
  }
}

// SHAPE 0: l[i] OP r[i]; 1: l[i] OP s; 2: s OP r[i] (array operand passed as `a`)
template <typename T, int OP, int SHAPE, bool ALIGNED, bool NT>
__global__ __launch_bounds__(kBlock) void compare_kernel(const T* __restrict__ a, const T* __restrict__ b, T scalar,
                                                          uint8_t* __restrict__ out, int64_t length, int prefix) {
  constexpr int V = 16 / sizeof(T);
  constexpr int LPW = 32 / V;  // lanes per output dword
  using VT = typename std::conditional<ALIGNED, Vec16<T>, ah_vec16<T>>::type;
  auto pred = [&](T x, T y) { return SHAPE == 2 ? cmp<T, OP>(scalar, x) : cmp<T, OP>(x, SHAPE == 0 ? y : scalar); };

  // ≤7-bit prefix inside out[0] — one lane
  int64_t prefix_len = 0;
  if (prefix) {
    prefix_len = length < 8 - prefix ? length : 8 - prefix;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      uint8_t byte = out[0];
      for (int k = 0; k < (int)prefix_len; k++) {
        uint8_t m = (uint8_t)(1u << (prefix + k));
        bool r = pred(a[k], SHAPE == 0 ? b[k] : scalar);
        byte = (uint8_t)((byte & ~m) | (r ? m : 0));
      }
      out[0] = byte;
    }
  }
  const int64_t nb = length - prefix_len;  // body elements == body bits
  if (nb <= 0) return;
  uint8_t* ob = out + (prefix ? 1 : 0);
  const T* ab = a + prefix_len;
  const T* bb = SHAPE == 0 ? b + prefix_len : nullptr;
  const VT* av = (const VT*)ab;
  const VT* bv = (const VT*)bb;
  const int64_t nvec_full = nb / V;
  const int64_t nvec = (nb + V - 1) / V;
  const int64_t n_iters = (nvec + (int64_t)kBlock * kUnroll - 1) / ((int64_t)kBlock * kUnroll);
  const int lane = threadIdx.x & 63;

  auto mask_of = [&](const VT& xv, const VT& yv) {
    uint32_t m = 0;
#pragma unroll
    for (int e = 0; e < V; e++) {
      T xe, ye;
      if constexpr (ALIGNED) { xe = xv[e]; ye = yv[e]; } else { xe = xv.v[e]; ye = yv.v[e]; }
      m |= (uint32_t)pred(xe, ye) << e;
    }
    return m;
  };
  // merge the V-bit lane masks into dwords (lane L gets the bits of lanes L .. L+LPW−1) and store
  auto emit = [&](int64_t j, uint32_t m) {
#pragma unroll
    for (int s = 1; s < LPW; s <<= 1) m |= (uint32_t)__shfl_down((int)m, s, 64) << (V * s);
    if ((lane % LPW) == 0) {
      const int64_t bit0 = j * V;  // first body bit of this dword (multiple of 32)
      if (bit0 < nb) {
        const int64_t left = nb - bit0;
        store_bits(ob + (bit0 >> 3), m, left >= 32 ? 32 : (int)left);
      }
    }
  };

  for (int64_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
    const int64_t base = it * kBlock * kUnroll + threadIdx.x;
    if ((it + 1) * (int64_t)kBlock * kUnroll <= nvec_full) {
      // the whole tile is full vectors: all loads first, then straight-line code (two separate paths on purpose —
      // one loop with a per-vector "is it loaded already" choice made the compiler index the register arrays
      // dynamically and spill them to scratch: 3.8 TB/s instead of 6+ on 1- and 2-byte types)
      VT x[kUnroll], y[kUnroll] = {};
#pragma unroll
      for (int k = 0; k < kUnroll; k++) {
        if constexpr (ALIGNED && NT) {
          x[k] = __builtin_nontemporal_load(&av[base + (int64_t)k * kBlock]);
          if (SHAPE == 0) y[k] = __builtin_nontemporal_load(&bv[base + (int64_t)k * kBlock]);
        } else if constexpr (ALIGNED) {
          x[k] = av[base + (int64_t)k * kBlock];
          if (SHAPE == 0) y[k] = bv[base + (int64_t)k * kBlock];
        } else {
          x[k] = ah_ld16<T>((const T*)(av + (base + (int64_t)k * kBlock)));
          if (SHAPE == 0) y[k] = ah_ld16<T>((const T*)(bv + (base + (int64_t)k * kBlock)));
        }
      }
#pragma unroll
      for (int k = 0; k < kUnroll; k++) emit(base + (int64_t)k * kBlock, mask_of(x[k], y[k]));
    } else {
      for (int k = 0; k < kUnroll; k++) {
        const int64_t j = base + (int64_t)k * kBlock;
        uint32_t m = 0;
        if (j < nvec_full) {
          VT xv = av[j], yv = xv;
          if (SHAPE == 0) yv = bv[j];
          m = mask_of(xv, yv);
        } else if (j == nvec_full) {  // ragged last vector: element-wise, in bounds
          const int rem = (int)(nb - nvec_full * V);
          for (int e = 0; e < rem; e++) m |= (uint32_t)pred(ab[j * V + e], SHAPE == 0 ? bb[j * V + e] : scalar) << e;
        }
        emit(j, m);
      }
    }
  }
}

template <typename T, int OP, int SHAPE>
int launch_compare(ah_ctx* c, const void* a, const void* b, T scalar, uint8_t* out, int64_t length, int prefix) {
  constexpr int V = 16 / sizeof(T);
  int64_t prefix_len = prefix ? (length < 8 - prefix ? length : 8 - prefix) : 0;
  const T* pa = (const T*)a; const T* pb = (const T*)b;
  bool aligned = ((((uintptr_t)(pa + prefix_len)) | (SHAPE == 0 ? (uintptr_t)(pb + prefix_len) : 0)) & 15) == 0;
  int64_t nvec = (length - prefix_len + V - 1) / V;
  unsigned grid = ah_stream_grid(c, ah_ceil_div(nvec > 0 ? nvec : 1, (int64_t)kBlock * kUnroll), /*default_bpc=*/0);
  if (aligned) {
    if (c->tune_nt) compare_kernel<T, OP, SHAPE, true, true><<<grid, kBlock, 0, c->stream>>>(pa, pb, scalar, out, length, prefix);
    else compare_kernel<T, OP, SHAPE, true, false><<<grid, kBlock, 0, c->stream>>>(pa, pb, scalar, out, length, prefix);
  } else {
    compare_kernel<T, OP, SHAPE, false, false><<<grid, kBlock, 0, c->stream>>>(pa, pb, scalar, out, length, prefix);
  }
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

template <typename T>
int dispatch_compare(ah_ctx* c, int cmpop, int shape, const void* l, const void* r, uint8_t* out, int64_t length, int prefix) {
  T scalar = 0;
  const void* a = l; const void* b = r;
  if (shape == AH_SHAPE_AS) { memcpy(&scalar, r, sizeof(T)); b = nullptr; }
  if (shape == AH_SHAPE_SA) { memcpy(&scalar, l, sizeof(T)); a = r; b = nullptr; }
#define AH_CMP_SHAPES(OPC)                                                                     \
  switch (shape) {                                                                             \
    case AH_SHAPE_AA: return launch_compare<T, OPC, 0>(c, a, b, scalar, out, length, prefix);  \
    case AH_SHAPE_AS: return launch_compare<T, OPC, 1>(c, a, b, scalar, out, length, prefix);  \
    case AH_SHAPE_SA: return launch_compare<T, OPC, 2>(c, a, b, scalar, out, length, prefix);  \
  }
  switch (cmpop) {
    case AH_CMP_EQ: AH_CMP_SHAPES(AH_CMP_EQ) break;
    case AH_CMP_NE: AH_CMP_SHAPES(AH_CMP_NE) break;
    case AH_CMP_GT: AH_CMP_SHAPES(AH_CMP_GT) break;
    case AH_CMP_GE: AH_CMP_SHAPES(AH_CMP_GE) break;
  }
#undef AH_CMP_SHAPES
  return ah_fail(c, AH_EINVALID, "comparison: bad op %d / shape %d", cmpop, shape);
}

}  // namespace

AH_EXPORT int ah_comparison(ah_ctx* c, int cmpop, int shape, int type, const void* l, const void* r,
                            uint8_t* out_bits, int64_t length, int out_bit_offset) {
  AH_ENTER(c);
  if (length < 0 || out_bit_offset < 0) return ah_fail(c, AH_EINVALID, "comparison: negative length/offset");
  if (length == 0) return AH_OK;
  int prefix = out_bit_offset % 8;  // scalar_comparison.cc:71
  int w = ah_type_width(type);
  if (!w) return ah_fail(c, AH_ENOTIMPL, "comparison: unsupported type id %d", type);
  const void* arr0 = shape == AH_SHAPE_SA ? r : l;
  if ((((uintptr_t)arr0 | (shape == AH_SHAPE_AA ? (uintptr_t)r : 0)) & (uintptr_t)(w - 1)) != 0)
    return ah_fail(c, AH_EINVALID, "comparison: buffer not element-aligned");
  switch (type) {
    case AH_UINT8: return dispatch_compare<uint8_t>(c, cmpop, shape, l, r, out_bits, length, prefix);
    case AH_INT8: return dispatch_compare<int8_t>(c, cmpop, shape, l, r, out_bits, length, prefix);
    case AH_UINT16: return dispatch_compare<uint16_t>(c, cmpop, shape, l, r, out_bits, length, prefix);
    case AH_INT16: return dispatch_compare<int16_t>(c, cmpop, shape, l, r, out_bits, length, prefix);
    case AH_UINT32: return dispatch_compare<uint32_t>(c, cmpop, shape, l, r, out_bits, length, prefix);
    case AH_INT32: return dispatch_compare<int32_t>(c, cmpop, shape, l, r, out_bits, length, prefix);
    case AH_UINT64: return dispatch_compare<uint64_t>(c, cmpop, shape, l, r, out_bits, length, prefix);
    case AH_INT64: return dispatch_compare<int64_t>(c, cmpop, shape, l, r, out_bits, length, prefix);
    case AH_FLOAT32: return dispatch_compare<float>(c, cmpop, shape, l, r, out_bits, length, prefix);
    case AH_FLOAT64: return dispatch_compare<double>(c, cmpop, shape, l, r, out_bits, length, prefix);
  }
  return ah_fail(c, AH_ENOTIMPL, "comparison: unsupported type id %d", type);
}
