// ah_bitmap.hip — validity/boolean bitmap utilities with arbitrary bit offsets.
//
// Replaces: bitutil.BitmapAnd/Or/Xor/AndNot/Xnor (arrow/bitutil/bitmaps.go:592-637;
//   aligned SIMD leaf _bitmap_aligned_*_avx2, arrow/bitutil/bitmap_ops_avx2_amd64.go:
//   26-52, C truth arrow/bitutil/_lib/bitmap_ops.c:24-46; unaligned Go word loop
//   bitmaps.go:568-582) — the null-propagation step of every scalar kernel call
//   (compute/executor.go:237-349) and the data path of "and"/"or"/"xor"/"and_not"
//   (kernels/scalar_boolean.go:67-300);
//   bitutil.CountSetBits (arrow/bitutil/bitutil.go:89-130);
//   bitutil.CopyBitmap / InvertBitmap (bitmaps.go:418-493); SetBitsTo
//   (bitutil.go:158-204); the Kleene word kernels (scalar_boolean.go:29-65).
//
// One scheme serves the aligned and the unaligned case: the OUTPUT is cut into
// 8-byte-aligned 64-bit words; each lane owns one output word, funnel-shifts the
// matching 64 input bits out of (at most) two aligned input words per operand,
// applies the op and stores the word whole — except the first/last word of the
// range, whose out-of-range bits are preserved with a masked read-modify-write by
// the single lane that owns it.  HBM-bound: 3/8 byte per row for a binary op.
#include "ah_common.h"

namespace {

constexpr int kBlock = 256;

struct OutSpan {
  uint64_t* base;   // 8-byte aligned word holding the first output bit
  int first_bit;    // position of the first output bit inside base[0] (0..63)
  int64_t nwords;   // words touched
};

__host__ __device__ inline OutSpan make_span(uint8_t* out, int64_t ooff, int64_t nbits) {
  uintptr_t addr = (uintptr_t)out + (uintptr_t)(ooff >> 3);
  uintptr_t base = addr & ~(uintptr_t)7;
  OutSpan s;
  s.base = (uint64_t*)base;
  s.first_bit = (int)((addr - base) * 8 + (ooff & 7));
  s.nwords = (s.first_bit + nbits + 63) / 64;
  return s;
}

template <int OP>
__device__ __forceinline__ uint64_t bitop(uint64_t a, uint64_t b) {
  if (OP == AH_BIT_AND) return a & b;
  if (OP == AH_BIT_OR) return a | b;
  if (OP == AH_BIT_XOR) return a ^ b;
  if (OP == AH_BIT_AND_NOT) return a & ~b;
  if (OP == AH_BIT_XNOR) return ~(a ^ b);
  if (OP == 100) return a;   // copy
  if (OP == 101) return ~a;  // invert
  return b;                  // 102: fill (b holds the pattern)
}

// word t of the span covers logical bits [64t - first_bit, 64t - first_bit + 64)
// clipped to [0, nbits); returns the clip as (lo logical index, count, shift in word)
__device__ __forceinline__ void word_range(const OutSpan& s, int64_t t, int64_t nbits, int64_t* lo, int* cnt, int* sh) {
  int64_t start = t * 64 - s.first_bit;
  int64_t end = start + 64;
  int64_t l = start < 0 ? 0 : start;
  int64_t e = end > nbits ? nbits : end;
  *lo = l;
  *cnt = (int)(e - l);
  *sh = (int)(l - start);
}

template <int OP>
__global__ __launch_bounds__(kBlock) void bitmap_op_kernel(const uint8_t* __restrict__ l, int64_t loff,
                                                            const uint8_t* __restrict__ r, int64_t roff,
                                                            uint8_t* out, int64_t ooff, int64_t nbits, uint64_t fill) {
  OutSpan s = make_span(out, ooff, nbits);
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < s.nwords; t += stride) {
    int64_t lo; int cnt, sh;
    word_range(s, t, nbits, &lo, &cnt, &sh);
    if (cnt <= 0) continue;
    uint64_t a = OP == 102 ? 0 : ah_load_bits64(l, loff + lo, cnt);
    uint64_t b = (OP >= 100) ? fill : ah_load_bits64(r, roff + lo, cnt);
    uint64_t v = bitop<OP>(a, b) << sh;
    uint64_t mask = (cnt >= 64 ? ~0ull : ((1ull << cnt) - 1)) << sh;
    if (mask == ~0ull) s.base[t] = v;
    else s.base[t] = (s.base[t] & ~mask) | (v & mask);
  }
}

__global__ __launch_bounds__(kBlock) void popcount_kernel(const uint8_t* __restrict__ bits, int64_t off, int64_t nbits,
                                                           unsigned long long* __restrict__ partials) {
  // read-side span: same word cutting, no writes; one partial per workgroup (a single
  // global counter would serialise ~12 ns per same-address atomic)
  OutSpan s = make_span((uint8_t*)bits, off, nbits);
  const uint64_t* w = (const uint64_t*)s.base;
  uint64_t acc = 0;
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < s.nwords; t += stride) {
    uint64_t v = w[t];
    if (t == 0) v &= ~0ull << s.first_bit;
    if (t == s.nwords - 1) {
      int endbit = (int)((s.first_bit + nbits) - (s.nwords - 1) * 64);  // 1..64
      if (endbit < 64) v &= (1ull << endbit) - 1;
    }
    acc += (uint64_t)__popcll(v);
  }
  acc = ah_wave_sum(acc);
  __shared__ uint64_t sm[kBlock / 64];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t tot = 0;
    for (int k = 0; k < kBlock / 64; k++) tot += sm[k];
    partials[blockIdx.x] = tot;
  }
}

// mb != nullptr: the call's last launch also posts {*extra, total} to the host's mailbox (ah_popcount_post)
__global__ __launch_bounds__(kBlock) void popcount_final_kernel(const unsigned long long* __restrict__ partials, int n,
                                                                 unsigned long long* __restrict__ total, const unsigned long long* __restrict__ extra = nullptr,
                                                                 unsigned long long* mb = nullptr, unsigned long long seq = 0) {
  uint64_t acc = 0;
  for (int i = threadIdx.x; i < n; i += kBlock) acc += partials[i];
  acc = ah_wave_sum(acc);
  __shared__ uint64_t sm[kBlock / 64];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint64_t tot = 0;
    for (int k = 0; k < kBlock / 64; k++) tot += sm[k];
    *total = tot;
    if (mb) { const unsigned long long w[2] = {*extra, tot}; ah_mailbox_post(mb, seq, w, 2); }
  }
}

// Kleene and/or/and_not (scalar_boolean.go:93-104,163-174,289-302)
template <int OP>
__global__ __launch_bounds__(kBlock) void kleene_kernel(const uint8_t* __restrict__ lvalid, const uint8_t* __restrict__ ldata, int64_t loff,
                                                         const uint8_t* __restrict__ rvalid, const uint8_t* __restrict__ rdata, int64_t roff,
                                                         uint8_t* ovalid, uint8_t* odata, int64_t ooff, int64_t nbits) {
  OutSpan sv = make_span(ovalid, ooff, nbits);
  OutSpan sd = make_span(odata, ooff, nbits);
  // the two outputs share ooff but may differ in byte alignment: iterate logical
  // 64-bit groups of the VALIDITY span and write data bits with the generic store
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < sv.nwords; t += stride) {
    int64_t lo; int cnt, sh;
    word_range(sv, t, nbits, &lo, &cnt, &sh);
    if (cnt <= 0) continue;
    uint64_t lv = ah_load_bits64(lvalid, loff + lo, cnt), ld = ah_load_bits64(ldata, loff + lo, cnt);
    uint64_t rv = ah_load_bits64(rvalid, roff + lo, cnt), rd = ah_load_bits64(rdata, roff + lo, cnt);
    uint64_t lT = lv & ld, lF = lv & ~ld, rT = rv & rd, rF = rv & ~rd, v, d;
    if (OP == AH_KLEENE_AND) { v = lF | rF | (lT & rT); d = lT & rT; }
    else if (OP == AH_KLEENE_OR) { v = lT | rT | (lF & rF); d = lT | rT; }
    else { v = lF | rT | (lT & rF); d = lT & rF; }
    uint64_t cm = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1);
    uint64_t mask = cm << sh;
    if (mask == ~0ull) sv.base[t] = v << sh;
    else sv.base[t] = (sv.base[t] & ~mask) | ((v << sh) & mask);
    // data bitmap: same logical range [lo, lo+cnt); map onto its own span
    int64_t pos = sd.first_bit + lo;   // bit position relative to sd.base
    int64_t w0 = pos >> 6; int s0 = (int)(pos & 63);
    uint64_t dv = d & cm;
    // words of the data span are owned by the lane whose validity range covers
    // them; two lanes may share a boundary word → atomics for the partial pieces
    uint64_t m0 = cm << s0;
    if (m0 == ~0ull) sd.base[w0] = dv;
    else {
      atomicAnd((unsigned long long*)&sd.base[w0], (unsigned long long)~m0);
      atomicOr((unsigned long long*)&sd.base[w0], (unsigned long long)((dv << s0) & m0));
    }
    if (s0 != 0 && s0 + cnt > 64) {
      uint64_t m1 = cm >> (64 - s0);
      atomicAnd((unsigned long long*)&sd.base[w0 + 1], (unsigned long long)~m1);
      atomicOr((unsigned long long*)&sd.base[w0 + 1], (unsigned long long)((dv >> (64 - s0)) & m1));
    }
  }
}

template <int OP>
int launch_bitmap(ah_ctx* c, const uint8_t* l, int64_t loff, const uint8_t* r, int64_t roff, uint8_t* out, int64_t ooff,
                  int64_t nbits, uint64_t fill) {
  OutSpan s = make_span(out, ooff, nbits);
  unsigned grid = ah_stream_grid(c, ah_ceil_div(s.nwords, kBlock));
  bitmap_op_kernel<OP><<<grid, kBlock, 0, c->stream>>>(l, loff, r, roff, out, ooff, nbits, fill);
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

}  // namespace

int ah_popcount_async(ah_ctx* c, const uint8_t* bits, int64_t off, int64_t nbits, unsigned long long* total_dev) {
  OutSpan s = make_span((uint8_t*)bits, off, nbits);
  int64_t g = ah_ceil_div(s.nwords, (int64_t)kBlock * 4);  // ≥ 4 words per lane
  unsigned grid = (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
  unsigned long long* partials = (unsigned long long*)&c->dscalars[64];
  popcount_kernel<<<grid, kBlock, 0, c->stream>>>(bits, off, nbits, partials);
  AH_LAUNCH_CHECK(c);
  popcount_final_kernel<<<1, kBlock, 0, c->stream>>>(partials, (int)grid, total_dev);
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}

// the popcount of [off, off + nbits) into *total_dev AND {*extra_dev, that count} to the host in the same two launches: a call that ends
// with "count the output's valid rows, then tell the host" spares the posting launch (out_host[0] = *extra_dev, out_host[1] = the count)
int ah_popcount_post(ah_ctx* c, const uint8_t* bits, int64_t off, int64_t nbits, unsigned long long* total_dev, const unsigned long long* extra_dev,
                     unsigned long long* out_host) {
  OutSpan s = make_span((uint8_t*)bits, off, nbits);
  int64_t g = ah_ceil_div(s.nwords, (int64_t)kBlock * 4);
  unsigned grid = (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
  unsigned long long* partials = (unsigned long long*)&c->dscalars[64];
  unsigned long long* mb;
  unsigned long long seq;
  int rc = ah_mailbox_begin(c, &mb, &seq);
  if (rc != AH_OK) return rc;
  popcount_kernel<<<grid, kBlock, 0, c->stream>>>(bits, off, nbits, partials);
  AH_LAUNCH_CHECK(c);
  popcount_final_kernel<<<1, kBlock, 0, c->stream>>>(partials, (int)grid, total_dev, extra_dev, mb, seq);
  AH_LAUNCH_CHECK(c);
  return ah_mailbox_wait(c, seq, 2, out_host);
}

AH_EXPORT int ah_bitmap_op(ah_ctx* c, int op, const uint8_t* l, int64_t loff, const uint8_t* r, int64_t roff,
                           uint8_t* out, int64_t ooff, int64_t nbits) {
  AH_ENTER(c);
  if (nbits < 0 || loff < 0 || roff < 0 || ooff < 0) return ah_fail(c, AH_EINVALID, "bitmap_op: negative offset/length");
  if (nbits == 0) return AH_OK;
  if (!l || !r || !out) return ah_fail(c, AH_EINVALID, "bitmap_op: null bitmap");
  switch (op) {
    case AH_BIT_AND: return launch_bitmap<AH_BIT_AND>(c, l, loff, r, roff, out, ooff, nbits, 0);
    case AH_BIT_OR: return launch_bitmap<AH_BIT_OR>(c, l, loff, r, roff, out, ooff, nbits, 0);
    case AH_BIT_XOR: return launch_bitmap<AH_BIT_XOR>(c, l, loff, r, roff, out, ooff, nbits, 0);
    case AH_BIT_AND_NOT: return launch_bitmap<AH_BIT_AND_NOT>(c, l, loff, r, roff, out, ooff, nbits, 0);
    case AH_BIT_XNOR: return launch_bitmap<AH_BIT_XNOR>(c, l, loff, r, roff, out, ooff, nbits, 0);
  }
  return ah_fail(c, AH_EINVALID, "bitmap_op: bad op %d", op);
}

AH_EXPORT int ah_copy_bitmap(ah_ctx* c, const uint8_t* src, int64_t soff, int64_t nbits, uint8_t* dst, int64_t doff, int invert) {
  AH_ENTER(c);
  if (nbits < 0 || soff < 0 || doff < 0) return ah_fail(c, AH_EINVALID, "copy_bitmap: negative offset/length");
  if (nbits == 0) return AH_OK;
  if (!src || !dst) return ah_fail(c, AH_EINVALID, "copy_bitmap: null bitmap");
  return invert ? launch_bitmap<101>(c, src, soff, src, soff, dst, doff, nbits, 0)
                : launch_bitmap<100>(c, src, soff, src, soff, dst, doff, nbits, 0);
}

AH_EXPORT int ah_set_bits_to(ah_ctx* c, uint8_t* bits, int64_t off, int64_t nbits, int value) {
  AH_ENTER(c);
  if (nbits < 0 || off < 0) return ah_fail(c, AH_EINVALID, "set_bits_to: negative offset/length");
  if (nbits == 0) return AH_OK;
  if (!bits) return ah_fail(c, AH_EINVALID, "set_bits_to: null bitmap");
  return launch_bitmap<102>(c, nullptr, 0, nullptr, 0, bits, off, nbits, value ? ~0ull : 0ull);
}

AH_EXPORT int ah_count_set_bits(ah_ctx* c, const uint8_t* bits, int64_t off, int64_t nbits, int64_t* out_host) {
  AH_ENTER(c);
  if (!out_host) return ah_fail(c, AH_EINVALID, "count_set_bits: null result pointer");
  if (nbits < 0 || off < 0) return ah_fail(c, AH_EINVALID, "count_set_bits: negative offset/length");
  *out_host = 0;
  if (nbits == 0) return AH_OK;
  if (!bits) { *out_host = nbits; return AH_OK; }
  unsigned long long* total = (unsigned long long*)c->dscalars;
  int rc = ah_popcount_async(c, bits, off, nbits, total);
  if (rc != AH_OK) return rc;
  { int mrc = ah_mailbox_read(c, total, 1, (unsigned long long*)c->pinned); if (mrc != AH_OK) return mrc; }
  *out_host = (int64_t) * (volatile uint64_t*)c->pinned;
  return AH_OK;
}

AH_EXPORT int ah_kleene(ah_ctx* c, int op, const uint8_t* lvalid, const uint8_t* ldata, int64_t loff,
                        const uint8_t* rvalid, const uint8_t* rdata, int64_t roff,
                        uint8_t* ovalid, uint8_t* odata, int64_t ooff, int64_t nbits) {
  AH_ENTER(c);
  if (nbits < 0 || loff < 0 || roff < 0 || ooff < 0) return ah_fail(c, AH_EINVALID, "kleene: negative offset/length");
  if (nbits == 0) return AH_OK;
  if (!ldata || !rdata || !ovalid || !odata) return ah_fail(c, AH_EINVALID, "kleene: null bitmap");
  OutSpan s = make_span(ovalid, ooff, nbits);
  unsigned grid = ah_stream_grid(c, ah_ceil_div(s.nwords, kBlock));
  switch (op) {
    case AH_KLEENE_AND: kleene_kernel<AH_KLEENE_AND><<<grid, kBlock, 0, c->stream>>>(lvalid, ldata, loff, rvalid, rdata, roff, ovalid, odata, ooff, nbits); break;
    case AH_KLEENE_OR: kleene_kernel<AH_KLEENE_OR><<<grid, kBlock, 0, c->stream>>>(lvalid, ldata, loff, rvalid, rdata, roff, ovalid, odata, ooff, nbits); break;
    case AH_KLEENE_AND_NOT: kleene_kernel<AH_KLEENE_AND_NOT><<<grid, kBlock, 0, c->stream>>>(lvalid, ldata, loff, rvalid, rdata, roff, ovalid, odata, ooff, nbits); break;
    default: return ah_fail(c, AH_EINVALID, "kleene: bad op %d", op);
  }
  AH_LAUNCH_CHECK(c);
  return AH_OK;
}
