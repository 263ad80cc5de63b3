/*
 * orc_bits.c — comparisons → packed bitmaps, and the null-bitmap utilities
 * (TEST INFRASTRUCTURE, see oracle.h).
 *
 * Reference:
 *   compare: arrow/compute/internal/kernels/_lib/scalar_comparison.cc:63-208
 *            (== Go kernels/scalar_comparisons.go:51-176): bit i of the output,
 *            LSB-first, starting `out_bit_offset % 8` bits into out[0]; bits
 *            outside [offset, offset+length) are preserved (set_bit_to for the
 *            ≤7-bit prefix and the tail, whole bytes only for 32-batches).
 *   popcount: arrow/bitutil/bitutil.go:89-130 (CountSetBits).
 *   bitmap ops: arrow/bitutil/bitmaps.go:527-590 (aligned/unaligned BitmapOp),
 *            :418-493 (CopyBitmap / InvertBitmap), bitutil.go:158-204 (SetBitsTo).
 *            Intended contract restated: exactly the bits [ooff, ooff+n) of `out`
 *            are written, all others preserved.  (alignedBitmapOp's
 *            `endMask := (lOffset + length%8)`, bitmaps.go:536, leaves the last
 *            byte unwritten when lOffset is a non-zero multiple of 8 and the
 *            range ends on a byte boundary — an upstream defect outside any
 *            reference test; NOT replicated, see DESIGN.md "quirks".)
 *   Kleene:  kernels/scalar_boolean.go:29-65 (computeKleene) and :120-330.
 */
#include "oracle.h"
#include <math.h>

static inline int bget(const uint8_t* b, int64_t i) { return (b[i >> 3] >> (i & 7)) & 1; }
static inline int bget_opt(const uint8_t* b, int64_t i) { return b == 0 ? 1 : bget(b, i); }
static inline void bset_to(uint8_t* b, int64_t i, int v) {
  uint8_t m = (uint8_t)(1u << (i & 7));
  b[i >> 3] = (uint8_t)((b[i >> 3] & ~m) | (v ? m : 0));
}

#define CMP_LOOP(T)                                                                 \
  {                                                                                 \
    const T* l = (const T*)lv; const T* r = (const T*)rv;                           \
    for (int64_t i = 0; i < length; i++) {                                          \
      T a = l[i * ls], b = r[i * rs];                                               \
      int res = cmpop == ORC_CMP_EQ ? a == b : cmpop == ORC_CMP_NE ? a != b         \
              : cmpop == ORC_CMP_GT ? a > b : a >= b;                               \
      bset_to(out_bits, prefix + i, res);                                           \
    }                                                                               \
    return ORC_OK;                                                                  \
  }

int orc_comparison(int cmpop, int shape, int type, const void* lv, const void* rv,
                   uint8_t* out_bits, int64_t length, int out_bit_offset) {
  int ls = shape == ORC_SHAPE_SA ? 0 : 1, rs = shape == ORC_SHAPE_AS ? 0 : 1;
  int64_t prefix = out_bit_offset % 8; /* scalar_comparison.cc:71: `offset % 8` */
  if (cmpop < ORC_CMP_EQ || cmpop > ORC_CMP_GE) return ORC_EINVALID;
  switch (type) {
    case ORC_UINT8: CMP_LOOP(uint8_t)
    case ORC_INT8: CMP_LOOP(int8_t)
    case ORC_UINT16: CMP_LOOP(uint16_t)
    case ORC_INT16: CMP_LOOP(int16_t)
    case ORC_UINT32: CMP_LOOP(uint32_t)
    case ORC_INT32: CMP_LOOP(int32_t)
    case ORC_UINT64: CMP_LOOP(uint64_t)
    case ORC_INT64: CMP_LOOP(int64_t)
    case ORC_FLOAT32: CMP_LOOP(float)
    case ORC_FLOAT64: CMP_LOOP(double)
  }
  return ORC_EINVALID;
}

int64_t orc_count_set_bits(const uint8_t* bits, int64_t off, int64_t nbits) {
  int64_t c = 0;
  for (int64_t i = 0; i < nbits; i++) c += bget(bits, off + i);
  return c;
}

void orc_bitmap_op(int op, const uint8_t* l, int64_t loff, const uint8_t* r, int64_t roff,
                   uint8_t* out, int64_t ooff, int64_t nbits) {
  for (int64_t i = 0; i < nbits; i++) {
    int a = bget(l, loff + i), b = bget(r, roff + i), v;
    switch (op) {
      case ORC_BIT_AND: v = a & b; break;
      case ORC_BIT_OR: v = a | b; break;
      case ORC_BIT_XOR: v = a ^ b; break;
      case ORC_BIT_AND_NOT: v = a & !b; break;
      default: v = !(a ^ b); break;
    }
    bset_to(out, ooff + i, v);
  }
}

void orc_copy_bitmap(const uint8_t* src, int64_t soff, int64_t nbits, uint8_t* dst, int64_t doff, int invert) {
  for (int64_t i = 0; i < nbits; i++) bset_to(dst, doff + i, bget(src, soff + i) ^ (invert ? 1 : 0));
}

void orc_set_bits_to(uint8_t* bits, int64_t off, int64_t nbits, int value) {
  for (int64_t i = 0; i < nbits; i++) bset_to(bits, off + i, value);
}

/*
 * Kleene logic (kernels/scalar_boolean.go:93-104 and_kleene, :163-174
 * or_kleene, :289-302 and_not_kleene, through computeKleene :29-65).  With
 *   lT = lvalid & ldata, lF = lvalid & ~ldata, rT = rvalid & rdata, rF = rvalid & ~rdata
 *   and_kleene:     valid = lF | rF | (lT & rT)   data = lT & rT
 *   or_kleene:      valid = lT | rT | (lF & rF)   data = lT | rT
 *   and_not_kleene: valid = lF | rT | (lT & rF)   data = lT & rF
 * (a missing validity bitmap reads as all-ones, which reduces `data` to the
 * plain and/or/and_not of the no-null fast paths at :94-98,164-168,290-294).
 */
void orc_kleene(int op, const uint8_t* lvalid, const uint8_t* ldata, int64_t loff,
                const uint8_t* rvalid, const uint8_t* rdata, int64_t roff,
                uint8_t* ovalid, uint8_t* odata, int64_t ooff, int64_t nbits) {
  for (int64_t i = 0; i < nbits; i++) {
    int lv = bget_opt(lvalid, loff + i), ld = bget(ldata, loff + i);
    int rv = bget_opt(rvalid, roff + i), rd = bget(rdata, roff + i);
    int lT = lv & ld, lF = lv & !ld, rT = rv & rd, rF = rv & !rd;
    int d, v;
    if (op == 0) { v = lF | rF | (lT & rT); d = lT & rT; }
    else if (op == 1) { v = lT | rT | (lF & rF); d = lT | rT; }
    else { v = lF | rT | (lT & rF); d = lT & rF; }
    bset_to(odata, ooff + i, d);
    bset_to(ovalid, ooff + i, v);
  }
}
