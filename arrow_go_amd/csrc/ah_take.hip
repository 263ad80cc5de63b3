// ah_take.hip — Take: gather of fixed-width values by an integer index vector.
//
// Replaces: kernels.PrimitiveTake (kernels/vector_selection.go:1162-1192) →
//   takeIdxDispatch (:1144-1160) → primitiveTakeImpl[IdxT, ValT] (:878-988) and
//   checkIndexBounds (kernels/helpers.go:929-981), behind compute's
//   "take"/"array_take" (compute/selection.go:93-114,206-237).
//
// Contract: out[i] = values[idx[i]] iff the index slot is valid
// and the addressed value is valid; otherwise payload 0 and validity 0.  Indices of
// any integer width; after the bounds check they are reinterpreted as unsigned of
// the same width (:1147-1158).  Only VALID index slots are bounds-checked; the error
// names the first offender in index order.  The sorted / reverse-sorted heuristics
// of the reference (:734-876) change loop shape only, never results — not needed.
//
// Roofline: HBM, 4 (int32 index) + 8 (gathered value) + 8 (store) = 20 algorithmic
// bytes per output row; random gathers additionally pay the 64-byte sector
// granularity.  One output row per lane per step, kUnroll independent gathers in
// flight per lane; index loads and value stores are fully coalesced; the output
// validity word for 64 rows is one wave ballot, stored by lane 0.
#include <type_traits>
#include "ah_common.h"

namespace {

constexpr int kBlock = 256;
#ifndef AH_TAKE_UNROLL
#define AH_TAKE_UNROLL 4
#endif
constexpr int kUnroll = AH_TAKE_UNROLL;

template <int W> struct UIntOf;
template <> struct UIntOf<1> { using type = uint8_t; };
template <> struct UIntOf<2> { using type = uint16_t; };
template <> struct UIntOf<4> { using type = uint32_t; };
template <> struct UIntOf<8> { using type = uint64_t; };
// 16- and 32-byte values — Decimal128 / Decimal256 and FixedSizeBinary of those widths (FSBImpl, vector_selection.go:1997, takes any
// width byte by byte; here the widths that are whole 16-byte accesses): moved as 2 / 4 × 64-bit vectors, the plain gather kernel only
template <> struct UIntOf<16> { using type = unsigned long long __attribute__((ext_vector_type(2))); };
template <> struct UIntOf<32> { using type = unsigned long long __attribute__((ext_vector_type(4))); };

// NT: the index vector and the output are streamed once — nontemporal, so that they do not push the gathered values' lines out of L2
template <int W, typename IdxT, bool HAS_VALID, bool NT>
__global__ __launch_bounds__(kBlock) void take_kernel(const void* __restrict__ values_v, const uint8_t* __restrict__ vvalid, int64_t voff,
                                                       uint64_t nvalues, const IdxT* __restrict__ idx, const uint8_t* __restrict__ ivalid,
                                                       int64_t ioff, int64_t nidx, void* __restrict__ out_v, uint8_t* __restrict__ out_valid,
                                                       unsigned long long* __restrict__ first_bad, unsigned long long* __restrict__ valid_total) {
  using T = typename UIntOf<W>::type;
  using UIdx = typename std::make_unsigned<IdxT>::type;
  const T* __restrict__ values = (const T*)values_v;
  T* __restrict__ out = (T*)out_v;
  const int lane = threadIdx.x & 63;
  const int64_t n_iters = (nidx + (int64_t)kBlock * kUnroll - 1) / ((int64_t)kBlock * kUnroll);
  for (int64_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
    const int64_t base = it * kBlock * kUnroll + threadIdx.x;
    // two memory round trips, not four: {index, index-validity byte} go out together, then {value, value-validity byte}
    // (each pair used to be a dependent chain: with 10 % nulls the identity take ran at 2.2 TB/s against 5.2 without)
    uint64_t u[kUnroll];
    bool ok[kUnroll];
    IdxT raw[kUnroll];
    uint8_t ib[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; k++) {
      const int64_t i = base + (int64_t)k * kBlock;
      raw[k] = 0;
      ib[k] = 0xff;
      if (i < nidx) {
        raw[k] = NT ? __builtin_nontemporal_load(&idx[i]) : idx[i];
        if (ivalid != nullptr) ib[k] = ivalid[(ioff + i) >> 3];
      }
    }
#pragma unroll
    for (int k = 0; k < kUnroll; k++) {
      const int64_t i = base + (int64_t)k * kBlock;
      ok[k] = false;
      u[k] = 0;
      if (i < nidx && ((ib[k] >> ((ioff + i) & 7)) & 1)) {
        const IdxT s = raw[k];
        u[k] = (uint64_t)(UIdx)s;  // reinterpret as unsigned of the same width
        bool oob = (std::is_signed<IdxT>::value && s < 0) || u[k] >= nvalues;  // helpers.go:937-939
        if (oob) atomicMin(first_bad, (unsigned long long)i);
        else ok[k] = true;
      }
    }
    T v[kUnroll];
    uint8_t vb[kUnroll];
#pragma unroll
    for (int k = 0; k < kUnroll; k++) {
      v[k] = 0;
      vb[k] = 0xff;
      if (ok[k]) {
        v[k] = values[u[k]];
        if (HAS_VALID && vvalid != nullptr) vb[k] = vvalid[(voff + (int64_t)u[k]) >> 3];
      }
    }
    if (HAS_VALID) {
#pragma unroll
      for (int k = 0; k < kUnroll; k++)
        if (!((vb[k] >> ((voff + (int64_t)u[k]) & 7)) & 1)) { ok[k] = false; v[k] = 0; }
    }
#pragma unroll
    for (int k = 0; k < kUnroll; k++) {
      const int64_t i = base + (int64_t)k * kBlock;
      if (i < nidx) {
        if (NT) __builtin_nontemporal_store(v[k], &out[i]);
        else out[i] = v[k];
      }
      if (HAS_VALID) {
        uint64_t word = __ballot(ok[k]);
        const int64_t i0 = i - lane;  // first row of this wave's 64 (multiple of 64)
        if (lane == 0 && i0 < nidx) {
          int64_t left = nidx - i0;
          uint8_t* p = out_valid + (i0 >> 3);
          if (left >= 64) {
            // i0 is a multiple of 64, so p is as aligned as out_valid itself: one 8-byte store (unaligned global access is
            // legal on gfx950) instead of eight 1-byte stores — with nulls the identity take ran at 2.2 TB/s against 5.2 without
            *(uint64_t*)p = word;
          } else {
            int nbytes = (int)((left + 7) >> 3);
            for (int bb = 0; bb < nbytes; bb++) p[bb] = (uint8_t)(word >> (8 * bb));
          }
        }
      }
    }
  }
}


// ---- clustered indices: V = 16 / W adjacent output rows per lane -----------------------------------------------------------------
// The kernel above moves W bytes per lane and instruction; for 8-byte values that is half of what the memory pipeline takes per
// request, and sorted / identity / reversed index vectors stopped at 4.3–4.9 TB/s where 16-byte accesses stream at 6.3.  Here a lane
// owns V consecutive output rows: their indices arrive in one load, the V results leave in ONE 16-byte store, and when the V
// indices all name values inside one 16-byte window [base, base + V) — consecutive ascending (identity, sorted runs, slices) or
// descending (a reversed column), repeats of a sorted column included — the V values arrive in ONE 16-byte load (element-aligned,
// which gfx950 allows) and each row picks its element.  Anything else falls back to V separate gathers, so results never depend
// on the choice (the reference's isSorted / primitiveTakeImplSorted split, vector_selection.go:734-876, is the same kind of
// decision: loop shape only).  Chosen when the 64 × 255 neighbour sample of ah_take_binned.hip says "clustered".
template <typename IdxT, int V>
struct alignas(sizeof(IdxT)) IdxVec { IdxT v[V]; };
// the same 16 (or V·sizeof(IdxT)) bytes as a native vector of element alignment: what __builtin_nontemporal_load/store accept.  The
// columns of a clustered take are streamed once — without the hint the index vector and the output evict the values' lines from L2
template <typename T, int V>
using TakeRaw __attribute__((aligned(sizeof(T)))) = T __attribute__((ext_vector_type(V)));
// NT is a TEMPLATE parameter: behind a run-time (or merely inlined) switch the hinted and the plain access of one address and one
// type are merged into a plain one — the hints of this kernel were lost that way twice (ISA check: `nt` on the loads and stores)
template <typename S, typename T, int V, bool NT>
__device__ __forceinline__ S take_ld(const T* p) {
  static_assert(sizeof(S) == sizeof(T) * V, "carrier size");
  TakeRaw<T, V> raw;
  if constexpr (NT) raw = __builtin_nontemporal_load(reinterpret_cast<const TakeRaw<T, V>*>(p));
  else raw = *reinterpret_cast<const TakeRaw<T, V>*>(p);   // (one access of V elements: a load of the carrier struct is split)
  S r;
#pragma unroll
  for (int j = 0; j < V; j++) r.v[j] = raw[j];
  return r;
}
template <typename S, typename T, int V, bool NT>
__device__ __forceinline__ void take_st(T* p, const S& x) {
  if constexpr (NT) __builtin_nontemporal_store(__builtin_bit_cast(TakeRaw<T, V>, x), reinterpret_cast<TakeRaw<T, V>*>(p));
  else *reinterpret_cast<TakeRaw<T, V>*>(p) = __builtin_bit_cast(TakeRaw<T, V>, x);
}
#ifndef AH_TAKE_VEC_STEPS
#define AH_TAKE_VEC_STEPS 1
#endif
constexpr int kVecSteps = AH_TAKE_VEC_STEPS;   // steps per workgroup

template <int CTRL>
__device__ __forceinline__ unsigned take_dpp(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xF, 0xF, false); }

template <int W, typename IdxT, bool HAS_VALID, int NT>
__global__ __launch_bounds__(kBlock) void take_vec_kernel(const void* __restrict__ values_v, const uint8_t* __restrict__ vvalid, int64_t voff,
                                                           uint64_t nvalues, const IdxT* __restrict__ idx, const uint8_t* __restrict__ ivalid,
                                                           int64_t ioff, int64_t nidx, void* __restrict__ out_v, uint8_t* __restrict__ out_valid,
                                                           unsigned long long* __restrict__ first_bad) {
  using T = typename UIntOf<W>::type;
  using UIdx = typename std::make_unsigned<IdxT>::type;
  constexpr int V = 16 / W;          // rows per lane (2 or 4)
  const int64_t vbytes = (voff + (int64_t)nvalues + 7) >> 3;   // bytes of the value validity bitmap
  constexpr bool nt_idx = NT & 1, nt_val = NT & 2, nt_out = NT & 4;   // (a run-time switch would be merged away: the two loads of one address fold into a plain one)
  constexpr int K = 4;               // groups per lane per step: 4 index loads, then 4 value loads in flight
  constexpr unsigned kAll = (1u << V) - 1u;
  const T* __restrict__ values = (const T*)values_v;
  T* __restrict__ out = (T*)out_v;
  const int lane = threadIdx.x & 63;
  const int64_t ngroups = (nidx + V - 1) / V;
  const int64_t n_iters = (ngroups + (int64_t)kBlock * K - 1) / ((int64_t)kBlock * K);
  // kVecSteps steps per workgroup, the NEXT step's index vectors requested before this step's gathers go out.  Measured: 4 steps
  // per workgroup 5.1 TB/s on the identity vector, ONE step per workgroup (exact grid, the default) 5.4 — as for the element-wise
  // kernels, the hardware's workgroup dispatch is the better pipeline; the loop stays for the grid-capped case (ARROWHIP_BLOCKS_PER_CU)
  IdxVec<IdxT, V> ivn[K];
  auto load_idx = [&](int64_t it2) {
    const int64_t gb = it2 * kBlock * K + threadIdx.x;
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int64_t r = (gb + (int64_t)k * kBlock) * V;
      if (r + V <= nidx) {
        ivn[k] = take_ld<IdxVec<IdxT, V>, IdxT, V, nt_idx>(idx + r);
      } else {
#pragma unroll
        for (int j = 0; j < V; j++) ivn[k].v[j] = r + j < nidx ? idx[r + j] : (IdxT)0;
      }
    }
  };
  if ((int64_t)blockIdx.x < n_iters) load_idx(blockIdx.x);
  for (int64_t it = blockIdx.x; it < n_iters; it += gridDim.x) {
    const int64_t gbase = it * kBlock * K + threadIdx.x;
    IdxVec<IdxT, V> iv[K];
    unsigned ibits[K];   // the V index-validity bits of the lane's rows
#pragma unroll
    for (int k = 0; k < K; k++) iv[k] = ivn[k];
    if (it + gridDim.x < n_iters) load_idx(it + gridDim.x);
#pragma unroll
    for (int k = 0; k < K; k++) {
      // the wave's 64·V rows start at a multiple of 64·V: V scalar-loaded words, the lane's V bits sit in word (lane·V) >> 6
      const int64_t r0 = (gbase + (int64_t)k * kBlock - lane) * V;
      uint64_t mine = 0;
#pragma unroll
      for (int w = 0; w < V; w++) {
        const uint64_t word = ah_wave_bits64(ivalid, ioff + r0 + 64 * w, nidx - (r0 + 64 * w));   // 0 past the end, ones without a bitmap
        if (((lane * V) >> 6) == w) mine = word;
      }
      ibits[k] = (unsigned)(mine >> ((lane * V) & 63)) & kAll;
    }
    uint64_t u[K][V], lo[K];
    unsigned okb[K];     // rows with a valid, in-range index
    bool merged[K];      // all V raw indices lie in [lo, lo + V) and that window is inside the column
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int64_t r = (gbase + (int64_t)k * kBlock) * V;
      okb[k] = 0;
      uint64_t mn = ~0ull, mx = 0;
      bool neg = false;
#pragma unroll
      for (int j = 0; j < V; j++) {
        const IdxT s = iv[k].v[j];
        u[k][j] = (uint64_t)(UIdx)s;   // reinterpret as unsigned of the same width
        neg = neg || (std::is_signed<IdxT>::value && s < 0);
        mn = u[k][j] < mn ? u[k][j] : mn;
        mx = u[k][j] > mx ? u[k][j] : mx;
        if ((ibits[k] >> j) & 1u) {
          const bool oob = (std::is_signed<IdxT>::value && s < 0) || u[k][j] >= nvalues;   // helpers.go:937-939
          if (oob) atomicMin(first_bad, (unsigned long long)(r + j));
          else okb[k] |= 1u << j;
        }
      }
      // (the slots under a null index take part with whatever they hold — an identity vector stays one window per lane whatever its
      // validity; their rows are cleared below)
      lo[k] = mn;
      merged[k] = !neg && mx - mn < (uint64_t)V && mn + V <= nvalues && r < nidx;
    }
    ah_vec16<T> x[K];
    unsigned vbits[K];   // value validity of the V rows
#pragma unroll
    for (int k = 0; k < K; k++) {
      vbits[k] = kAll;
      if (merged[k]) {
        const ah_vec16<T> t = take_ld<ah_vec16<T>, T, V, nt_val>(values + lo[k]);
        unsigned b = 0xffffu;
        const int64_t p = voff + (int64_t)lo[k];   // validity bits p … p + V − 1: one byte, or two neighbours
        if (HAS_VALID && vvalid != nullptr) {
          const int64_t by = p >> 3;
          if (by + 1 < vbytes) {   // one unaligned 2-byte load (byte loads cost a request each: 8 per lane and step were the nulls path's gap)
            uint16_t h;
            __builtin_memcpy(&h, vvalid + by, 2);
            b = h;
          } else {
            b = (unsigned)vvalid[by] | ((unsigned)vvalid[(p + V - 1) >> 3] << 8);
          }
        }
        unsigned vb = 0;
#pragma unroll
        for (int j = 0; j < V; j++) {
          const unsigned d = (unsigned)(u[k][j] - lo[k]);   // 0 … V − 1
          T e = t.v[0];
#pragma unroll
          for (int q = 1; q < V; q++) e = d == (unsigned)q ? t.v[q] : e;
          x[k].v[j] = e;
          vb |= ((b >> ((unsigned)(p & 7) + d)) & 1u) << j;
        }
        if (HAS_VALID) vbits[k] = vb;
      } else {
#pragma unroll
        for (int j = 0; j < V; j++) {
          x[k].v[j] = 0;
          if ((okb[k] >> j) & 1u) {
            x[k].v[j] = values[u[k][j]];
            if (HAS_VALID && vvalid != nullptr) {
              const int64_t p = voff + (int64_t)u[k][j];
              if (!((vvalid[p >> 3] >> (p & 7)) & 1u)) vbits[k] &= ~(1u << j);
            }
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int64_t g = gbase + (int64_t)k * kBlock, r = g * V;
      const unsigned good = HAS_VALID ? (okb[k] & vbits[k]) : okb[k];
#pragma unroll
      for (int j = 0; j < V; j++)
        if (!((good >> j) & 1u)) x[k].v[j] = 0;   // a null output keeps payload 0 (:959-973)
      if (r + V <= nidx) {
        take_st<ah_vec16<T>, T, V, nt_out>(out + r, x[k]);
      } else {
#pragma unroll
        for (int j = 0; j < V; j++) if (r + j < nidx) out[r + j] = x[k].v[j];
      }
      if (HAS_VALID) {
        // a validity BYTE = the V bits of 8 / V neighbouring lanes: OR them together inside the quad (DPP, no LDS, no ballots), the
        // first lane of each group stores the byte — consecutive bytes from consecutive lane groups, one store instruction per step
        unsigned m = good << (V * (lane & (8 / V - 1)));
        m |= take_dpp<0xB1>(m);                    // quad_perm [1,0,3,2]
        if (V == 2) m |= take_dpp<0x4E>(m);        // quad_perm [2,3,0,1]
        if ((lane & (8 / V - 1)) == 0 && r < nidx) out_valid[r >> 3] = (uint8_t)m;
      }
    }
  }
}

// first_bad = "none", the popcount's and the sample's words: one launch in front of a Take (three memsets before).
// (Counting the output's valid rows inside the clustered kernel — one add per wave into a table of partial counts — was built and
// measured: a workgroup of that kernel lives for ONE step, a wave cannot retire before its atomic is acknowledged, and the kernel went
// from 487 to 540 µs with 256, with 2^14 and with line-spread counters alike; the popcount pass over the finished bitmap stays.)
__global__ void take_prep_kernel(unsigned long long* __restrict__ first_bad, unsigned long long* __restrict__ valid_total, unsigned long long* __restrict__ hits) {
  *first_bad = ~0ull; *valid_total = 0; *hits = 0;
}

template <int W, typename IdxT>
int launch_take(ah_ctx* c, const void* values, const uint8_t* vvalid, int64_t voff, int64_t nvalues, const void* idx,
                const uint8_t* ivalid, int64_t ioff, int64_t nidx, void* out_values, uint8_t* out_valid,
                unsigned long long* first_bad, unsigned long long* valid_total) {
  if constexpr (W == 4 || W == 8) {
    if (c->take_clustered_hint) {   // set by the neighbour sample of ah_take_binned_try: adjacent index slots mostly name adjacent values
      constexpr int V = 16 / W;
      const unsigned vgrid = ah_stream_grid(c, ah_ceil_div(ah_ceil_div(ah_ceil_div(nidx, V), (int64_t)kBlock * 4), kVecSteps), 0);
      auto go = [&](auto nt_tag) {
        constexpr int NT = decltype(nt_tag)::value;
        if (out_valid)
          take_vec_kernel<W, IdxT, true, NT><<<vgrid, kBlock, 0, c->stream>>>(values, vvalid, voff, (uint64_t)nvalues, (const IdxT*)idx, ivalid, ioff,
                                                                              nidx, out_values, out_valid, first_bad);
        else
          take_vec_kernel<W, IdxT, false, NT><<<vgrid, kBlock, 0, c->stream>>>(values, nullptr, voff, (uint64_t)nvalues, (const IdxT*)idx, ivalid, ioff,
                                                                               nidx, out_values, nullptr, first_bad);
      };
      // option take_vec_nt (measurement switch): 7 = every stream nontemporal (default), 5 = index + output, 4 = output only, 0 = none
      switch (c->opt_take_vec_nt) {
        case 0: go(std::integral_constant<int, 0>{}); break;
        case 4: go(std::integral_constant<int, 4>{}); break;
        case 5: go(std::integral_constant<int, 5>{}); break;
        default: go(std::integral_constant<int, 7>{}); break;
      }
      AH_LAUNCH_CHECK(c);
      return AH_OK;
    }
  }
  unsigned grid = ah_stream_grid(c, ah_ceil_div(nidx, (int64_t)kBlock * kUnroll));
  const bool nt = c->opt_take_vec_nt != 0 && nidx * (int64_t)W >= ((int64_t)8 << 20);   // small outputs stay in L2 for their consumer
  auto go = [&](auto nt_tag) {
    constexpr bool NT = decltype(nt_tag)::value;
    if (out_valid)
      take_kernel<W, IdxT, true, NT><<<grid, kBlock, 0, c->stream>>>(values, vvalid, voff, (uint64_t)nvalues, (const IdxT*)idx, ivalid, ioff,
                                                                     nidx, out_values, out_valid, first_bad, valid_total);
    else
      take_kernel<W, IdxT, false, NT><<<grid, kBlock, 0, c->stream>>>(values, nullptr, voff, (uint64_t)nvalues, (const IdxT*)idx, ivalid, ioff,
                                                                      nidx, out_values, nullptr, first_bad, valid_total);
  };
  if (nt) go(std::true_type{}); else go(std::false_type{});
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

template <int W>
int dispatch_idx(ah_ctx* c, int iw, int is_signed, const void* values, const uint8_t* vvalid, int64_t voff, int64_t nvalues,
                 const void* idx, const uint8_t* ivalid, int64_t ioff, int64_t nidx, void* out_values, uint8_t* out_valid,
                 unsigned long long* first_bad, unsigned long long* valid_total) {
#define AH_TAKE(IT) return launch_take<W, IT>(c, values, vvalid, voff, nvalues, idx, ivalid, ioff, nidx, out_values, out_valid, first_bad, valid_total)
  switch (iw) {
    case 1: if (is_signed) AH_TAKE(int8_t); else AH_TAKE(uint8_t);
    case 2: if (is_signed) AH_TAKE(int16_t); else AH_TAKE(uint16_t);
    case 4: if (is_signed) AH_TAKE(int32_t); else AH_TAKE(uint32_t);
    case 8: if (is_signed) AH_TAKE(int64_t); else AH_TAKE(uint64_t);
  }
#undef AH_TAKE
  return ah_fail(c, AH_EINDEX, "invalid indices byte width");  // vector_selection.go:1157
}

// Boolean values (booleanTakeImpl, vector_selection.go:990-1074): out data bit i = value bit idx[i],
// validity as for the other widths; a null output keeps data bit 0 (fresh zeroed buffer in the
// reference).  64 rows per wave step, both output words come from ballots.
template <typename IdxT>
__global__ __launch_bounds__(kBlock) void take_bool_kernel(const uint8_t* __restrict__ data, const uint8_t* __restrict__ vvalid, int64_t voff,
                                                            uint64_t nvalues, const IdxT* __restrict__ idx, const uint8_t* __restrict__ ivalid,
                                                            int64_t ioff, int64_t nidx, uint8_t* __restrict__ out_data,
                                                            uint8_t* __restrict__ out_valid, unsigned long long* __restrict__ first_bad) {
  using UIdx = typename std::make_unsigned<IdxT>::type;
  const int lane = threadIdx.x & 63;
  const int64_t nchunks = (nidx + 63) >> 6;
  const int64_t wave_stride = (int64_t)gridDim.x * (kBlock / 64);
  for (int64_t c = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); c < nchunks; c += wave_stride) {
    const int64_t i = c * 64 + lane;
    bool ok = false, bit = false;
    if (i < nidx && ah_bit(ivalid, ioff + i)) {
      const IdxT s = idx[i];
      const uint64_t u = (uint64_t)(UIdx)s;
      if ((std::is_signed<IdxT>::value && s < 0) || u >= nvalues) atomicMin(first_bad, (unsigned long long)i);
      else if (ah_bit(vvalid, voff + (int64_t)u)) { ok = true; bit = ah_bit(data, voff + (int64_t)u); }
    }
    const unsigned long long dword = __ballot(ok && bit), vword = __ballot(ok);
    if (lane == 0) {
      const int64_t left = nidx - c * 64;
      const int nbytes = left >= 64 ? 8 : (int)((left + 7) >> 3);
      for (int b = 0; b < nbytes; b++) {
        out_data[c * 8 + b] = (uint8_t)(dword >> (8 * b));
        if (out_valid) out_valid[c * 8 + b] = (uint8_t)(vword >> (8 * b));
      }
    }
  }
}

// Slots of ANY byte width (FSBImpl, vector_selection.go:1997-2031: `copy(buf, valueData[start:start+valueSize])` per row) — what the
// 1 / 2 / 4 / 8 / 16 / 32-byte kernels above do not take (the reference's own test column is binary(3)).  A row per lane, the slot copied
// byte by byte; the 64 rows of a wave step share one validity word (ballot).  Not a tuned path: odd widths are rare, the rule is the same.
template <typename IdxT>
__global__ __launch_bounds__(kBlock) void take_bytes_kernel(int w, const uint8_t* __restrict__ values, const uint8_t* __restrict__ vvalid, int64_t voff,
                                                             uint64_t nvalues, const IdxT* __restrict__ idx, const uint8_t* __restrict__ ivalid,
                                                             int64_t ioff, int64_t nidx, uint8_t* __restrict__ out, uint8_t* __restrict__ out_valid,
                                                             unsigned long long* __restrict__ first_bad) {
  using UIdx = typename std::make_unsigned<IdxT>::type;
  const int lane = threadIdx.x & 63;
  const int64_t nchunks = (nidx + 63) >> 6;
  const int64_t wave_stride = (int64_t)gridDim.x * (kBlock / 64);
  for (int64_t c = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); c < nchunks; c += wave_stride) {
    const int64_t i = c * 64 + lane;
    bool ok = false;
    uint64_t u = 0;
    if (i < nidx && ah_bit(ivalid, ioff + i)) {
      const IdxT s = idx[i];
      u = (uint64_t)(UIdx)s;
      if ((std::is_signed<IdxT>::value && s < 0) || u >= nvalues) atomicMin(first_bad, (unsigned long long)i);   // helpers.go:937-939
      else ok = ah_bit(vvalid, voff + (int64_t)u);
    }
    if (i < nidx) {
      uint8_t* dst = out + i * (int64_t)w;
      const uint8_t* src = values + u * (uint64_t)w;
      for (int b = 0; b < w; b++) dst[b] = ok ? src[b] : (uint8_t)0;   // a null output keeps the zero of a fresh buffer
    }
    const unsigned long long vword = __ballot(ok);
    if (out_valid && lane == 0) {
      const int64_t left = nidx - c * 64;
      const int nbytes = left >= 64 ? 8 : (int)((left + 7) >> 3);
      for (int b = 0; b < nbytes; b++) out_valid[c * 8 + b] = (uint8_t)(vword >> (8 * b));
    }
  }
}

// *_dev flavour: {position of the first offending index or UINT64_MAX, output null count} stay in device memory
__global__ void take_status_kernel(const unsigned long long* __restrict__ first_bad, const unsigned long long* __restrict__ nvalid, int has_valid,
                                   int64_t nidx, unsigned long long* __restrict__ status) {
  status[0] = *first_bad;
  status[1] = has_valid ? (unsigned long long)nidx - *nvalid : 0ull;
}

int take_primitive_core(ah_ctx* c, int byte_width, const void* values, const uint8_t* vvalid, int64_t voff, int64_t nvalues, int idx_byte_width,
                        int idx_signed, const void* idx, const uint8_t* ivalid, int64_t ioff, int64_t nidx, void* out_values, uint8_t* out_valid,
                        int64_t* out_null_count_host, int64_t* bad_index_host, uint64_t* status_dev);

// the sample this call took (or reused) stays readable for the NEXT Take, unless this call's own outputs overlap the index vector
static inline void take_leave_hint(ah_ctx* c, const void* idx, int byte_width, int64_t nidx, const void* out_values, const void* out_valid) {
  if (!c->opt_take_hint_cache || c->take_hint_idx != idx) return;
  c->take_hint_valid = true;
  if (ah_take_hint_overlaps(c, out_values, (size_t)nidx * (size_t)byte_width) || (out_valid && ah_take_hint_overlaps(c, out_valid, (size_t)(nidx + 7) / 8)))
    c->take_hint_valid = false;
}

}  // namespace

AH_EXPORT int ah_take_primitive(ah_ctx* c, int byte_width, const void* values, const uint8_t* vvalid, int64_t voff,
                                int64_t nvalues, int idx_byte_width, int idx_signed, const void* idx,
                                const uint8_t* ivalid, int64_t ioff, int64_t nidx, int bounds_check,
                                void* out_values, uint8_t* out_valid, int64_t* out_null_count_host,
                                int64_t* bad_index_host) {
  const bool hint_live = c && c->take_hint_valid;   // read before AH_ENTER drops it: this Take is the call the sample was left for
  AH_ENTER(c);
  c->take_hint_live = hint_live;
  (void)bounds_check;  // the check is fused into the gather and always on (header)
  return take_primitive_core(c, byte_width, values, vvalid, voff, nvalues, idx_byte_width, idx_signed, idx, ivalid, ioff, nidx, out_values, out_valid,
                             out_null_count_host, bad_index_host, nullptr);
}

AH_EXPORT int ah_take_primitive_dev(ah_ctx* c, int byte_width, const void* values, const uint8_t* vvalid, int64_t voff,
                                    int64_t nvalues, int idx_byte_width, int idx_signed, const void* idx,
                                    const uint8_t* ivalid, int64_t ioff, int64_t nidx, void* out_values, uint8_t* out_valid,
                                    uint64_t* status_dev) {
  const bool hint_live = c && c->take_hint_valid;   // read before AH_ENTER drops it: this Take is the call the sample was left for
  AH_ENTER(c);
  c->take_hint_live = hint_live;
  if (!status_dev) return ah_fail(c, AH_EINVALID, "take: null status pointer");
  return take_primitive_core(c, byte_width, values, vvalid, voff, nvalues, idx_byte_width, idx_signed, idx, ivalid, ioff, nidx, out_values, out_valid,
                             nullptr, nullptr, status_dev);
}

namespace {

int take_primitive_core(ah_ctx* c, int byte_width, const void* values, const uint8_t* vvalid, int64_t voff, int64_t nvalues, int idx_byte_width,
                        int idx_signed, const void* idx, const uint8_t* ivalid, int64_t ioff, int64_t nidx, void* out_values, uint8_t* out_valid,
                        int64_t* out_null_count_host, int64_t* bad_index_host, uint64_t* status_dev) {
  if (nidx < 0 || nvalues < 0 || voff < 0 || ioff < 0) return ah_fail(c, AH_EINVALID, "take: negative length/offset");
  if (out_null_count_host) *out_null_count_host = 0;
  if (nidx == 0) {
    if (status_dev) {
      AH_HIP(c, hipMemsetAsync(status_dev, 0xFF, 8, c->stream));
      AH_HIP(c, hipMemsetAsync(status_dev + 1, 0, 8, c->stream));
    }
    return AH_OK;
  }
  if (!idx || !out_values || (!values && nvalues > 0)) return ah_fail(c, AH_EINVALID, "take: null buffer");
  if (!out_valid && (vvalid || ivalid)) {
    // the caller decided there are no nulls (PrimitiveTake :1176 uses the null COUNTS);
    // validity inputs are then ignored exactly like the reference's no-null path
    vvalid = nullptr;
    ivalid = nullptr;
  }
  unsigned long long* first_bad = (unsigned long long*)&c->dscalars[1];
  unsigned long long* valid_total = (unsigned long long*)&c->dscalars[2];
  take_prep_kernel<<<1, 1, 0, c->stream>>>(first_bad, valid_total, (unsigned long long*)&c->dscalars[3]);
  AH_LAUNCH_CHECK(c);
  int rc, binned = 0;
  c->take_clustered_hint = c->opt_take_vec == 2;
  // random indices into a column beyond the caches: bin → gather in L2-sized windows → unpermute (ah_take_binned.hip)
  rc = ah_take_binned_try(c, byte_width, values, vvalid, voff, nvalues, idx_byte_width, idx_signed, idx, ivalid, ioff, nidx, out_values, out_valid,
                          first_bad, &binned);
  if (rc != AH_OK) return rc;
  // 16- and 32-byte slots move as 16-byte vectors: a buffer that is only 8-byte aligned (a sliced FixedSizeBinary / Decimal column, a
  // zero-copy import) takes the arbitrary-width kernel instead of relying on the hardware's unaligned-access mode
  const bool wide_misaligned = (byte_width == 16 || byte_width == 32) && ((((uintptr_t)values) | ((uintptr_t)out_values)) & 15) != 0;
  if (!binned) switch (wide_misaligned ? 0 : byte_width) {
    case 1: rc = dispatch_idx<1>(c, idx_byte_width, idx_signed, values, vvalid, voff, nvalues, idx, ivalid, ioff, nidx, out_values, out_valid, first_bad, valid_total); break;
    case 2: rc = dispatch_idx<2>(c, idx_byte_width, idx_signed, values, vvalid, voff, nvalues, idx, ivalid, ioff, nidx, out_values, out_valid, first_bad, valid_total); break;
    case 4: rc = dispatch_idx<4>(c, idx_byte_width, idx_signed, values, vvalid, voff, nvalues, idx, ivalid, ioff, nidx, out_values, out_valid, first_bad, valid_total); break;
    case 8: rc = dispatch_idx<8>(c, idx_byte_width, idx_signed, values, vvalid, voff, nvalues, idx, ivalid, ioff, nidx, out_values, out_valid, first_bad, valid_total); break;
    case 16: rc = dispatch_idx<16>(c, idx_byte_width, idx_signed, values, vvalid, voff, nvalues, idx, ivalid, ioff, nidx, out_values, out_valid, first_bad, valid_total); break;
    case 32: rc = dispatch_idx<32>(c, idx_byte_width, idx_signed, values, vvalid, voff, nvalues, idx, ivalid, ioff, nidx, out_values, out_valid, first_bad, valid_total); break;
    default: {
      if (byte_width < 1 || byte_width > 4096) return ah_fail(c, AH_EINVALID, "invalid values byte width for take");  // :1189
      const unsigned grid = ah_stream_grid(c, ah_ceil_div(ah_ceil_div(nidx, 64), kBlock / 64), 8);
#define AH_TBY(IT) take_bytes_kernel<IT><<<grid, kBlock, 0, c->stream>>>(byte_width, (const uint8_t*)values, vvalid, voff, (uint64_t)nvalues, (const IT*)idx, ivalid, ioff, nidx, (uint8_t*)out_values, out_valid, first_bad); break
      switch (idx_byte_width * 2 + (idx_signed ? 1 : 0)) {
        case 2: AH_TBY(uint8_t); case 3: AH_TBY(int8_t); case 4: AH_TBY(uint16_t); case 5: AH_TBY(int16_t);
        case 8: AH_TBY(uint32_t); case 9: AH_TBY(int32_t); case 16: AH_TBY(uint64_t); case 17: AH_TBY(int64_t);
        default: return ah_fail(c, AH_EINDEX, "invalid indices byte width");
      }
#undef AH_TBY
      AH_LAUNCH_CHECK(c);
      rc = AH_OK;
    }
  }
  if (rc != AH_OK) return rc;
  if (status_dev) {   // no round trip: the caller looks at the two words when (and if) it wants to
    if (out_valid && (rc = ah_popcount_async(c, out_valid, 0, nidx, valid_total)) != AH_OK) return rc;
    take_status_kernel<<<1, 1, 0, c->stream>>>(first_bad, valid_total, out_valid != nullptr, nidx, (unsigned long long*)status_dev);
    AH_LAUNCH_CHECK(c);
    take_leave_hint(c, idx, byte_width, nidx, out_values, out_valid);
    return AH_OK;
  }
  if (out_valid && out_null_count_host) {
    // the count of the output's valid rows and the two words for the host from the same two launches
    unsigned long long w[2];
    if ((rc = ah_popcount_post(c, out_valid, 0, nidx, valid_total, first_bad, w)) != AH_OK) return rc;
    c->pinned[0] = w[0]; c->pinned[1] = w[1];
  } else if ((rc = ah_mailbox_read(c, (const unsigned long long*)&c->dscalars[1], 2, (unsigned long long*)c->pinned)) != AH_OK) return rc;
  uint64_t bad_pos = *(volatile uint64_t*)&c->pinned[0];
  uint64_t nvalid = *(volatile uint64_t*)&c->pinned[1];
  if (bad_pos != ~0ull) {
    // fetch the offending index value for the message ("%d out of bounds", helpers.go:950)
    uint64_t raw = 0;
    AH_HIP(c, hipMemcpy(&raw, (const uint8_t*)idx + bad_pos * (uint64_t)idx_byte_width, (size_t)idx_byte_width, hipMemcpyDeviceToHost));
    int64_t val;
    switch (idx_byte_width) {
      case 1: val = idx_signed ? (int64_t)(int8_t)raw : (int64_t)(uint8_t)raw; break;
      case 2: val = idx_signed ? (int64_t)(int16_t)raw : (int64_t)(uint16_t)raw; break;
      case 4: val = idx_signed ? (int64_t)(int32_t)raw : (int64_t)(uint32_t)raw; break;
      default: val = (int64_t)raw; break;
    }
    if (bad_index_host) *bad_index_host = val;
    if (idx_signed || idx_byte_width < 8) return ah_fail(c, AH_EINDEX, "%lld out of bounds", (long long)val);
    return ah_fail(c, AH_EINDEX, "%llu out of bounds", (unsigned long long)raw);
  }
  if (out_null_count_host) *out_null_count_host = out_valid ? nidx - (int64_t)nvalid : 0;
  take_leave_hint(c, idx, byte_width, nidx, out_values, out_valid);
  return AH_OK;
}

}  // namespace

AH_EXPORT int ah_take_boolean(ah_ctx* c, const uint8_t* data, const uint8_t* vvalid, int64_t voff, int64_t nvalues, int idx_byte_width,
                              int idx_signed, const void* idx, const uint8_t* ivalid, int64_t ioff, int64_t nidx, int bounds_check,
                              uint8_t* out_data, uint8_t* out_valid, int64_t* out_null_count_host, int64_t* bad_index_host) {
  AH_ENTER(c);
  (void)bounds_check;
  if (nidx < 0 || nvalues < 0 || voff < 0 || ioff < 0) return ah_fail(c, AH_EINVALID, "take: negative length/offset");
  if (out_null_count_host) *out_null_count_host = 0;
  if (nidx == 0) return AH_OK;
  if (!idx || !out_data || (!data && nvalues > 0)) return ah_fail(c, AH_EINVALID, "take: null buffer");
  if (!out_valid && (vvalid || ivalid)) { vvalid = nullptr; ivalid = nullptr; }
  unsigned long long* first_bad = (unsigned long long*)&c->dscalars[1];
  unsigned long long* valid_total = (unsigned long long*)&c->dscalars[2];
  AH_HIP(c, hipMemsetAsync(first_bad, 0xFF, sizeof(*first_bad), c->stream));
  AH_HIP(c, hipMemsetAsync(valid_total, 0, sizeof(*valid_total), c->stream));
  const unsigned grid = ah_stream_grid(c, ah_ceil_div(ah_ceil_div(nidx, 64), kBlock / 64), 8);
#define AH_TB(IT) take_bool_kernel<IT><<<grid, kBlock, 0, c->stream>>>(data, vvalid, voff, (uint64_t)nvalues, (const IT*)idx, ivalid, ioff, nidx, out_data, out_valid, first_bad); break
  switch (idx_byte_width * 2 + (idx_signed ? 1 : 0)) {
    case 2: AH_TB(uint8_t); case 3: AH_TB(int8_t); case 4: AH_TB(uint16_t); case 5: AH_TB(int16_t);
    case 8: AH_TB(uint32_t); case 9: AH_TB(int32_t); case 16: AH_TB(uint64_t); case 17: AH_TB(int64_t);
    default: return ah_fail(c, AH_EINDEX, "invalid indices byte width");
  }
#undef AH_TB
  AH_LAUNCH_CHECK(c);
  if (out_valid && out_null_count_host) {
    int rc = ah_popcount_async(c, out_valid, 0, nidx, valid_total);
    if (rc != AH_OK) return rc;
  }
  { int mrc = ah_mailbox_read(c, (const unsigned long long*)&c->dscalars[1], 2, (unsigned long long*)c->pinned); if (mrc != AH_OK) return mrc; }
  const uint64_t bad_pos = *(volatile uint64_t*)&c->pinned[0];
  const uint64_t nvalid = *(volatile uint64_t*)&c->pinned[1];
  if (bad_pos != ~0ull) {
    uint64_t raw = 0;
    AH_HIP(c, hipMemcpy(&raw, (const uint8_t*)idx + bad_pos * (uint64_t)idx_byte_width, (size_t)idx_byte_width, hipMemcpyDeviceToHost));
    int64_t val;
    switch (idx_byte_width) {
      case 1: val = idx_signed ? (int64_t)(int8_t)raw : (int64_t)(uint8_t)raw; break;
      case 2: val = idx_signed ? (int64_t)(int16_t)raw : (int64_t)(uint16_t)raw; break;
      case 4: val = idx_signed ? (int64_t)(int32_t)raw : (int64_t)(uint32_t)raw; break;
      default: val = (int64_t)raw; break;
    }
    if (bad_index_host) *bad_index_host = val;
    if (idx_signed || idx_byte_width < 8) return ah_fail(c, AH_EINDEX, "%lld out of bounds", (long long)val);
    return ah_fail(c, AH_EINDEX, "%llu out of bounds", (unsigned long long)raw);
  }
  if (out_null_count_host) *out_null_count_host = out_valid ? nidx - (int64_t)nvalid : 0;
  return AH_OK;
}
