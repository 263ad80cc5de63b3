// ah_varlen.hip — Take (and, through take indices, Filter) of binary / string columns: row §8(f)-4.
//
// Replaces VarBinaryImpl (arrow/compute/internal/kernels/vector_selection.go:1925-1992) driven by
// takeExec / filterExec (:1460-1598, 1821-1923) behind "array_take" / "array_filter" for
// Binary, String (int32 offsets) and LargeBinary, LargeString (int64 offsets):
//   a selected VALID slot appends the output offset and the value's bytes; every other emitted slot
//   (null value, null index, filter-null under EmitNulls) appends only the offset — a zero-length
//   null; output offsets start at 0; "binary output offset overflow" (checkBinaryTakeOffset
//   :1241-1247) when the bytes no longer fit the offset type; index bounds as in helpers.go:929-981.
// The reference appends value after value into growing builders.  Here the output size is data-
// dependent, so — like ah_filter_count / ah_filter_primitive — the work is two calls with the
// caller's allocation in between:
//   ah_take_binary_offsets  lengths gathered through the indices (+ validity word per 64 rows from a
//                           ballot, + fused bounds check) → scan (ah_scan.hip's kernels) → output
//                           offsets, total bytes, null count;
//   ah_take_binary_data     a wave owns 64 output rows: (source, destination, length) per lane,
//                           compacted through LDS; four 16-lane groups then copy four rows at a
//                           time, 4 (unaligned) bytes per lane.
// Filter = ah_filter_to_indices (GetTakeIndices, :102-236) + these two: filter(values, mask) and
// take(values, indices_of(mask)) select the same slots with the same validity, which is also how
// the reference filters record batches (compute/selection.go:687).
// Traffic: 2·w_off + w_idx read and w_off (+1/8) written per row in the first call (+ the scan's
// 24 B/row), the value bytes once in and once out in the second.
#include <type_traits>
#include "ah_common.h"

namespace {

constexpr int kBlock = 256;

template <typename IdxT> __device__ __forceinline__ unsigned long long as_unsigned(IdxT v) { return (unsigned long long)(typename std::make_unsigned<IdxT>::type)v; }

// lens[i] = byte length of output row i (0 for a null), validity word per 64 rows, first out-of-bounds position
template <typename OffT, typename IdxT>
__global__ __launch_bounds__(kBlock) void lens_kernel(const OffT* __restrict__ offsets, const uint8_t* __restrict__ vvalid, int64_t voff,
                                                       unsigned long long nvalues, const IdxT* __restrict__ idx,
                                                       const uint8_t* __restrict__ ivalid, int64_t ioff, int64_t n,
                                                       long long* __restrict__ lens, uint8_t* __restrict__ out_valid,
                                                       unsigned long long* __restrict__ first_bad) {
  const int lane = threadIdx.x & 63;
  const int64_t nchunks = (n + 63) >> 6;
  const int64_t wave_stride = (int64_t)gridDim.x * (kBlock / 64);
  for (int64_t c = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6); c < nchunks; c += wave_stride) {
    const int64_t i = c * 64 + lane;
    bool ok = false;
    long long len = 0;
    if (i < n && ah_bit(ivalid, ioff + i)) {
      const IdxT s = idx[i];
      const unsigned long long u = as_unsigned<IdxT>(s);
      if ((std::is_signed<IdxT>::value && s < 0) || u >= nvalues) {
        atomicMin(first_bad, (unsigned long long)i);  // helpers.go:937-939: only valid index slots are checked
      } else if (ah_bit(vvalid, voff + (int64_t)u)) {
        ok = true;
        len = (long long)offsets[voff + (int64_t)u + 1] - (long long)offsets[voff + (int64_t)u];
      }
    }
    if (i < n) lens[i] = len;
    if (out_valid) {
      const unsigned long long word = __ballot(ok);
      if (lane == 0) {
        const int64_t left = n - c * 64;
        const int nbytes = left >= 64 ? 8 : (int)((left + 7) >> 3);
        uint8_t* p = out_valid + c * 8;
        for (int b = 0; b < nbytes; b++) p[b] = (uint8_t)(word >> (8 * b));
      }
    }
  }
}

// out_offsets[0] = 0, out_offsets[i + 1] = incl[i]; int32 offsets: flag an overflow
template <typename OffT>
__global__ __launch_bounds__(kBlock) void offsets_kernel(const long long* __restrict__ incl, int64_t n, OffT* __restrict__ out_offsets,
                                                          unsigned* __restrict__ overflow) {
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i <= n; i += stride) {
    const long long v = i == 0 ? 0 : incl[i - 1];
    if (sizeof(OffT) == 4 && v > 2147483647ll) atomicOr(overflow, 1u);
    out_offsets[i] = (OffT)v;
  }
}

struct __attribute__((packed)) U32u { unsigned v; };  // 4 bytes at any address (gfx950 global memory runs in unaligned-access mode)

template <typename OffT, typename IdxT>
__global__ __launch_bounds__(kBlock) void copy_kernel(const OffT* __restrict__ offsets, const uint8_t* __restrict__ data, int64_t voff,
                                                       const IdxT* __restrict__ idx, int64_t n, const OffT* __restrict__ out_offsets,
                                                       uint8_t* __restrict__ out_data) {
  struct Row { long long src, dst, len; };
  __shared__ Row s_rows[kBlock / 64][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int group = lane >> 4, sub = lane & 15;  // four 16-lane groups copy four rows at a time, 4 bytes per lane
  const int64_t nchunks = (n + 63) >> 6;
  const int64_t wave_stride = (int64_t)gridDim.x * (kBlock / 64);
  for (int64_t c = (int64_t)blockIdx.x * (kBlock / 64) + wave; c < nchunks; c += wave_stride) {
    const int64_t i = c * 64 + lane;
    long long src = 0, dst = 0, len = 0;
    if (i < n) {
      dst = (long long)out_offsets[i];
      len = (long long)out_offsets[i + 1] - dst;
      if (len > 0) src = (long long)offsets[voff + (int64_t)as_unsigned<IdxT>(idx[i])];  // len > 0 ⇒ a valid, in-bounds slot
    }
    // rows that have bytes, compacted into LDS in row order
    const unsigned long long todo = __ballot(len > 0);
    const int count = __popcll(todo);
    if (len > 0) s_rows[wave][__popcll(todo & (lane == 0 ? 0ull : (~0ull >> (64 - lane))))] = Row{src, dst, len};
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // LDS ops of a wave are in order; keep the compiler from reordering
    __builtin_amdgcn_wave_barrier();
    for (int k = group; k < count; k += 4) {
      volatile const Row* rp = &s_rows[wave][k];  // written by other lanes of this wave
      Row r;
      r.src = rp->src; r.dst = rp->dst; r.len = rp->len;
      const uint8_t* sp = data + r.src;
      uint8_t* dp = out_data + r.dst;
      long long b = (long long)sub * 4;
      for (; b + 4 <= r.len; b += 64) ((U32u*)(dp + b))->v = ((const U32u*)(sp + b))->v;
      const long long tail = r.len & ~3ll;  // the last len mod 4 bytes
      if (sub < (int)(r.len - tail)) dp[tail + sub] = sp[tail + sub];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // the next chunk overwrites s_rows
    __builtin_amdgcn_wave_barrier();
  }
}


template <typename OffT, typename IdxT>
int run_offsets(ah_ctx* c, const void* offsets, const uint8_t* vvalid, int64_t voff, int64_t nvalues, const void* idx, const uint8_t* ivalid,
                int64_t ioff, int64_t n, void* out_offsets, uint8_t* out_valid, long long* lens, long long* incl) {
  const unsigned grid = ah_stream_grid(c, ah_ceil_div(ah_ceil_div(n, 64), kBlock / 64), 8);
  lens_kernel<OffT, IdxT><<<grid, kBlock, 0, c->stream>>>((const OffT*)offsets, vvalid, voff, (unsigned long long)nvalues, (const IdxT*)idx, ivalid,
                                                         ioff, n, lens, out_valid, (unsigned long long*)&c->dscalars[1]);
  AH_LAUNCH_CHECK(c);
  int rc = ah_cumulative_sum(c, AH_INT64, lens, nullptr, 0, n, nullptr, 0, 0, incl, nullptr, nullptr);
  if (rc != AH_OK) return rc;
  offsets_kernel<OffT><<<ah_stream_grid(c, ah_ceil_div(n + 1, kBlock), 8), kBlock, 0, c->stream>>>(incl, n, (OffT*)out_offsets,
                                                                                                  (unsigned*)&c->dscalars[3]);
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

template <typename OffT>
int offsets_idx(ah_ctx* c, int iw, int is_signed, const void* offsets, const uint8_t* vvalid, int64_t voff, int64_t nvalues, const void* idx,
                const uint8_t* ivalid, int64_t ioff, int64_t n, void* out_offsets, uint8_t* out_valid, long long* lens, long long* incl) {
#define AH_VL(IT) return run_offsets<OffT, IT>(c, offsets, vvalid, voff, nvalues, idx, ivalid, ioff, n, out_offsets, out_valid, lens, incl)
  switch (iw) {
    case 1: if (is_signed) AH_VL(int8_t); else AH_VL(uint8_t);
    case 2: if (is_signed) AH_VL(int16_t); else AH_VL(uint16_t);
    case 4: if (is_signed) AH_VL(int32_t); else AH_VL(uint32_t);
    case 8: if (is_signed) AH_VL(int64_t); else AH_VL(uint64_t);
  }
#undef AH_VL
  return ah_fail(c, AH_EINDEX, "invalid indices byte width");
}

template <typename OffT>
int copy_idx(ah_ctx* c, int iw, const void* offsets, const uint8_t* data, int64_t voff, const void* idx, int64_t n, const void* out_offsets,
             uint8_t* out_data) {
  const unsigned grid = ah_stream_grid(c, ah_ceil_div(ah_ceil_div(n, 64), kBlock / 64), 16);
#define AH_CP(IT) copy_kernel<OffT, IT><<<grid, kBlock, 0, c->stream>>>((const OffT*)offsets, data, voff, (const IT*)idx, n, (const OffT*)out_offsets, out_data); break
  switch (iw) {  // the unsigned reinterpretation is all the copy needs
    case 1: AH_CP(uint8_t);
    case 2: AH_CP(uint16_t);
    case 4: AH_CP(uint32_t);
    case 8: AH_CP(uint64_t);
    default: return ah_fail(c, AH_EINDEX, "invalid indices byte width");
  }
#undef AH_CP
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

}  // namespace

AH_EXPORT int ah_take_binary_offsets(ah_ctx* c, int offset_width, const void* offsets, const uint8_t* vvalid, int64_t voff, int64_t nvalues,
                                     int idx_byte_width, int idx_signed, const void* idx, const uint8_t* ivalid, int64_t ioff, int64_t nidx,
                                     int bounds_check, void* out_offsets, uint8_t* out_valid, int64_t* out_null_count_host,
                                     int64_t* out_total_bytes_host, int64_t* bad_index_host) {
  AH_ENTER(c);
  (void)bounds_check;  // fused and always on, like ah_take_primitive
  if (nidx < 0 || nvalues < 0 || voff < 0 || ioff < 0) return ah_fail(c, AH_EINVALID, "take: negative length/offset");
  if (offset_width != 4 && offset_width != 8) return ah_fail(c, AH_EINVALID, "take: binary offsets are 4 or 8 bytes wide");
  if (out_null_count_host) *out_null_count_host = 0;
  if (out_total_bytes_host) *out_total_bytes_host = 0;
  if (!out_offsets || !offsets) return ah_fail(c, AH_EINVALID, "take: null buffer");
  if (nidx == 0) {  // a lone closing offset
    AH_HIP(c, hipMemsetAsync(out_offsets, 0, (size_t)offset_width, c->stream));
    return AH_OK;
  }
  if (!idx) return ah_fail(c, AH_EINVALID, "take: null buffer");
  if (!out_valid && (vvalid || ivalid)) { vvalid = nullptr; ivalid = nullptr; }  // the caller's null counts say: no nulls (:1176)
  // temporaries in the context's temp arena (the scan called below uses the scratch arena)
  const size_t col = (((size_t)nidx * 8) + 255) & ~(size_t)255;
  void* arena;
  int rc = ah_temp_reserve(c, 2 * col, &arena);
  if (rc != AH_OK) return rc;
  long long* lens = (long long*)arena;
  long long* incl = (long long*)((uint8_t*)arena + col);
  AH_HIP(c, hipMemsetAsync(&c->dscalars[1], 0xFF, sizeof(uint64_t), c->stream));  // first bad position
  AH_HIP(c, hipMemsetAsync(&c->dscalars[2], 0, 2 * sizeof(uint64_t), c->stream)); // valid count, overflow flag
  rc = offset_width == 4 ? offsets_idx<int32_t>(c, idx_byte_width, idx_signed, offsets, vvalid, voff, nvalues, idx, ivalid, ioff, nidx, out_offsets, out_valid, lens, incl)
                         : offsets_idx<int64_t>(c, idx_byte_width, idx_signed, offsets, vvalid, voff, nvalues, idx, ivalid, ioff, nidx, out_offsets, out_valid, lens, incl);
  if (rc != AH_OK) return rc;
  if (out_valid && out_null_count_host) {
    rc = ah_popcount_async(c, out_valid, 0, nidx, (unsigned long long*)&c->dscalars[2]);
    if (rc != AH_OK) return rc;
  }
  AH_HIP(c, hipMemcpyAsync(c->pinned, &c->dscalars[1], 3 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  AH_HIP(c, hipMemcpyAsync(&c->pinned[3], incl + nidx - 1, 8, hipMemcpyDeviceToHost, c->stream));
  AH_HIP(c, hipStreamSynchronize(c->stream));
  const uint64_t bad_pos = *(volatile uint64_t*)&c->pinned[0];
  const uint64_t nvalid = *(volatile uint64_t*)&c->pinned[1];
  const unsigned overflow = *(volatile unsigned*)&c->pinned[2];
  const int64_t total = *(volatile int64_t*)&c->pinned[3];
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
  if (overflow & 1u) return ah_fail(c, AH_EINVALID, "binary output offset overflow");  // :1245
  if (out_null_count_host) *out_null_count_host = out_valid ? nidx - (int64_t)nvalid : 0;
  if (out_total_bytes_host) *out_total_bytes_host = total;
  return AH_OK;
}

AH_EXPORT int ah_take_binary_data(ah_ctx* c, int offset_width, const void* offsets, const uint8_t* data, int64_t voff, int idx_byte_width,
                                  const void* idx, int64_t nidx, const void* out_offsets, uint8_t* out_data) {
  AH_ENTER(c);
  if (nidx < 0 || voff < 0) return ah_fail(c, AH_EINVALID, "take: negative length/offset");
  if (offset_width != 4 && offset_width != 8) return ah_fail(c, AH_EINVALID, "take: binary offsets are 4 or 8 bytes wide");
  if (nidx == 0) return AH_OK;
  if (!offsets || !idx || !out_offsets) return ah_fail(c, AH_EINVALID, "take: null buffer");
  return offset_width == 4 ? copy_idx<int32_t>(c, idx_byte_width, offsets, data, voff, idx, nidx, out_offsets, out_data)
                           : copy_idx<int64_t>(c, idx_byte_width, offsets, data, voff, idx, nidx, out_offsets, out_data);
}
