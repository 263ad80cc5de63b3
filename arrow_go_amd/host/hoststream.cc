// hoststream.cc — CallFunction over HOST-resident arguments: chunked, overlapped execution.
//
// The reference's executor cuts every call into spans of at most ExecCtx.ChunkSize rows and runs the kernel span by span
// (arrow/compute/executor.go:47-50 ChunkSize, :499 iterateExecSpans, :658-702 the per-span call).  On this device a span loop over
// device-resident columns would be the wrong shape (one launch over the whole column is), but for a column that lives in HOST memory
// the span loop is exactly what hides the PCIe link: span k + 1 uploads on one stream while span k computes and span k − 1
// downloads on a third (csrc/ah_ingest.hip: three slots of ChunkBytes, per-slot events).  An argument imported with ahc_import_host
// stays where its producer put it; the calls below stream it, every other call uploads it whole first (MaterializeOnDevice) — the
// executor's old behaviour, which also needs the whole column plus the output in HBM.
//
// Streamed (results are host-resident arrays in pinned memory of the session's; same bytes as the whole-array path, which the
// tests compare them with):
//   add / sub / subtract / multiply [+ _unchecked]   array ∘ array, array ∘ scalar, scalar ∘ array of one numeric type
//                                                     (checked integer kernels test valid slots only and fail with "overflow")
//   equal … less_equal                                the same shapes → a Boolean array
//   filter / array_filter                             values and selection vector on the host (ah_ingest_filter_*: two phases, the
//                                                     selection vector stays on the device between them)
//   arrow/math Sum (ahc_math_sum)                     ah_ingest_sum_*: every chunk's partials, ONE final reduction
//   cast                                              numeric → numeric (safe or not): one width in, another out
//   cumulative_sum [_checked]                         integer columns, nulls skipped or none: the running sum is carried from span to span
// Validity bitmaps are 1/64 of the values: they are uploaded whole, combined on the device (intersection, as propagateNulls does:
// executor.go:237-349) and the result's bitmap comes back in one copy.
#include <cstring>
#include "arrowhip_compute.h"

namespace arrowhip {
namespace compute {

namespace {

struct HostArg {
  const ArrayData* arr = nullptr;   // host-resident array, or
  const Scalar* scalar = nullptr;   // a scalar
  const uint8_t* values() const { return (const uint8_t*)arr->buffers[1]->hptr + arr->offset * (arr->type->bit_width / 8); }
  bool has_nulls() const { return arr && arr->buffers[0] && arr->null_count != 0; }
};

bool EndsWith(const std::string& s, const char* suf) {
  const size_t n = strlen(suf);
  return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// the whole validity bitmap of a host array on the device (bit `offset` = row 0)
Status UploadValidity(Session* s, const ArrayData& a, BufferPtr* out) {
  const int64_t vbytes = (a.offset + a.length + 7) / 8;
  AHC_RETURN_NOT_OK(s->Allocate(vbytes, out));
  return s->FromStatus(ah_upload_async(s->ctx(), (*out)->dptr, a.buffers[0]->hptr, (size_t)vbytes));
}

// validity of the result = intersection of the arguments' (null scalar: all null) → host bitmap + null count
Status ResultValidity(Session* s, const HostArg& l, const HostArg& r, int64_t n, BufferPtr* dev_l, BufferPtr* dev_r, BufferPtr* out_host, int64_t* nulls) {
  *nulls = 0;
  const bool null_scalar = (l.scalar && !l.scalar->valid) || (r.scalar && !r.scalar->valid);
  if (!l.has_nulls() && !r.has_nulls() && !null_scalar) return Status::OK();
  BufferPtr ov;
  AHC_RETURN_NOT_OK(s->AllocateBitmap(n, &ov));   // zero-filled
  if (!null_scalar) {
    if (l.has_nulls()) AHC_RETURN_NOT_OK(UploadValidity(s, *l.arr, dev_l));
    if (r.has_nulls()) AHC_RETURN_NOT_OK(UploadValidity(s, *r.arr, dev_r));
    if (l.has_nulls() && r.has_nulls())
      AHC_RETURN_NOT_OK(s->FromStatus(ah_bitmap_op(s->ctx(), AH_BIT_AND, (const uint8_t*)(*dev_l)->dptr, l.arr->offset, (const uint8_t*)(*dev_r)->dptr, r.arr->offset,
                                                   (uint8_t*)ov->dptr, 0, n)));
    else {
      const HostArg& one = l.has_nulls() ? l : r;
      const BufferPtr& dv = l.has_nulls() ? *dev_l : *dev_r;
      AHC_RETURN_NOT_OK(s->FromStatus(ah_copy_bitmap(s->ctx(), (const uint8_t*)dv->dptr, one.arr->offset, n, (uint8_t*)ov->dptr, 0, 0)));
    }
    int64_t set = 0;
    AHC_RETURN_NOT_OK(s->FromStatus(ah_count_set_bits(s->ctx(), (const uint8_t*)ov->dptr, 0, n, &set)));
    *nulls = n - set;
  } else {
    *nulls = n;
  }
  AHC_RETURN_NOT_OK(s->AllocatePinned((n + 7) / 8, out_host));
  AHC_RETURN_NOT_OK(s->FromStatus(ah_download_async(s->ctx(), (*out_host)->hptr, ov->dptr, (size_t)((n + 7) / 8))));
  return s->FromStatus(ah_sync(s->ctx()));
}

// the span loop: upload span k + 1, run `kernel` on span k, download span k — every step on its own stream (ah_ingest slots).
//   in_w[i]   bytes per row of input i (0: that input is a scalar and is not uploaded)
//   out_num / out_den: output bytes per row = out_num / out_den (8 / 1 for Int64 values, 1 / 8 for a bitmap)
//   kernel(rows, row0, in0_dev, in1_dev, out_dev)
template <typename K>
Status SpanLoop(ExecCtx* ctx, int64_t n, const uint8_t* in0, int w0, const uint8_t* in1, int w1, uint8_t* out_host, int out_num, int out_den, K kernel) {
  Session* s = ctx->session;
  ah_ingest* g;
  AHC_RETURN_NOT_OK(s->Ingest((size_t)ctx->ChunkBytes, &g));
  const size_t chunk_bytes = ah_ingest_chunk_bytes(g);
  const int depth = ah_ingest_depth(g);
  int wmax = w0 > w1 ? w0 : w1;
  if (out_den == 1 && out_num > wmax) wmax = out_num;
  const int64_t span = ((int64_t)(chunk_bytes / (size_t)wmax)) & ~(int64_t)63;   // rows per span: whole bitmap words
  const int64_t nspans = (n + span - 1) / span;
  auto upload = [&](int64_t k) -> Status {
    const int slot = (int)(k % depth);
    const int64_t r0 = k * span, rows = n - r0 < span ? n - r0 : span;
    bool first = true;
    if (w0) { AHC_RETURN_NOT_OK(s->FromStatus(ah_ingest_slot_upload(g, slot, 0, 0, in0 + r0 * w0, (size_t)(rows * w0), 1))); first = false; }
    if (w1) AHC_RETURN_NOT_OK(s->FromStatus(ah_ingest_slot_upload(g, slot, 1, 0, in1 + r0 * w1, (size_t)(rows * w1), first ? 1 : 0)));
    return Status::OK();
  };
  Status st = nspans > 0 ? upload(0) : Status::OK();
  for (int64_t k = 0; st.ok() && k < nspans; k++) {
    const int slot = (int)(k % depth);
    const int64_t r0 = k * span, rows = n - r0 < span ? n - r0 : span;
    // the NEXT span's uploads are enqueued before this span's kernel: a kernel that synchronises (the checked ones return their
    // verdict) must not hold the link idle
    if (k + 1 < nspans) st = upload(k + 1);
    if (st.ok()) st = s->FromStatus(ah_ingest_slot_ready(g, slot, 1));
    if (st.ok()) st = kernel(rows, r0, ah_ingest_slot_buffer(g, slot, 0), ah_ingest_slot_buffer(g, slot, 1), ah_ingest_slot_buffer(g, slot, 2));
    if (st.ok()) st = s->FromStatus(ah_ingest_slot_release(g, slot));
    if (st.ok()) st = s->FromStatus(ah_ingest_slot_download(g, slot, 2, 0, out_host + (r0 * out_num) / out_den, (size_t)((rows * out_num + out_den - 1) / out_den)));
  }
  Status w = s->FromStatus(ah_ingest_wait(g));   // also after a failure: no copy may still be in flight into buffers about to be dropped
  return st.ok() ? w : st;
}

ArrayDataPtr MakeHostArray(const DataType* type, int64_t n, const BufferPtr& values, const BufferPtr& validity, int64_t nulls, const std::string& logical) {
  auto d = std::make_shared<ArrayData>();
  d->type = type;
  d->length = n;
  d->null_count = validity ? nulls : 0;
  d->offset = 0;
  d->buffers[0] = validity;
  d->buffers[1] = values;
  d->logical = logical;
  d->on_host = true;
  return d;
}

bool SplitArgs(const std::vector<Datum>& args, HostArg* l, HostArg* r) {
  if (args.size() != 2) return false;
  HostArg* dst[2] = {l, r};
  for (int i = 0; i < 2; i++) {
    if (args[i].kind == DatumKind::Array) {
      if (!args[i].array->on_host) return false;   // mixed residency: upload the host side whole
      if (args[i].array->device_twin) return false;   // already uploaded once: the copy in HBM is nearer than the link
      dst[i]->arr = args[i].array.get();
    } else if (args[i].kind == DatumKind::Scalar) {
      dst[i]->scalar = args[i].scalar.get();
    } else {
      return false;
    }
  }
  if (!l->arr && !r->arr) return false;
  const DataType* t = l->arr ? l->arr->type : r->arr->type;
  if (l->arr && r->arr && (l->arr->type != r->arr->type || l->arr->length != r->arr->length)) return false;   // implicit casts / the length error: the usual path
  if ((l->scalar && l->scalar->type != t) || (r->scalar && r->scalar->type != t)) return false;
  if (!(IsInteger(t->id) || IsFloating(t->id))) return false;
  for (const HostArg* a : {l, r})
    if (a->arr && !a->arr->logical.empty()) return false;   // temporal columns: the type rules live in the usual path
  return true;
}

Status StreamArithmetic(ExecCtx* ctx, int op, bool checked, const HostArg& l, const HostArg& r, Datum* out) {
  Session* s = ctx->session;
  const ArrayData& a = *(l.arr ? l.arr : r.arr);
  const DataType* t = a.type;
  const int64_t n = a.length;
  const int w = t->bit_width / 8;
  const int shape = l.arr && r.arr ? AH_SHAPE_AA : (l.arr ? AH_SHAPE_AS : AH_SHAPE_SA);
  BufferPtr dvl, dvr, hvalid, hvalues;
  int64_t nulls = 0;
  AHC_RETURN_NOT_OK(ResultValidity(s, l, r, n, &dvl, &dvr, &hvalid, &nulls));
  AHC_RETURN_NOT_OK(s->AllocatePinned(n * w, &hvalues));
  const bool null_scalar = (l.scalar && !l.scalar->valid) || (r.scalar && !r.scalar->valid);
  // A null scalar makes every row null, but only ScalarBinaryNotNull — the checked INTEGER add / sub — leaves the payload as
  // allocated (helpers.go:311-314,340-343: "fast path if one side is entirely null").  Every unchecked op, every float op and
  // checked multiply run ScalarBinary, which unboxes the scalar's stored value and computes l[i] ∘ value under the all-null
  // validity (helpers.go:204-222; base_arithmetic.go:273-280 for Mul): those stream like any other call, so the payload is the
  // whole-array path's (kernels.cc ExecArithUnchecked / ah_arithmetic_checked) byte for byte.
  if (null_scalar && checked && op != AH_OP_MUL_CHECKED) {
    memset(hvalues->hptr, 0, (size_t)(n * w));
    *out = Datum::Of(MakeHostArray(t, n, hvalues, hvalid, nulls, ""));
    return Status::OK();
  }
  const int scalar_valid = null_scalar ? 0 : 1;
  auto kernel = [&](int64_t rows, int64_t r0, void* b0, void* b1, void* b2) -> Status {
    const void* lp = l.arr ? b0 : (const void*)l.scalar->value;
    const void* rp = r.arr ? b1 : (const void*)r.scalar->value;
    if (!checked) {
      switch (shape) {
        case AH_SHAPE_AA: return s->FromStatus(ah_arithmetic_binary(s->ctx(), (int)t->id, (int8_t)op, lp, rp, b2, rows));
        case AH_SHAPE_AS: return s->FromStatus(ah_arithmetic_arr_scalar(s->ctx(), (int)t->id, (int8_t)op, lp, rp, b2, rows));
        default: return s->FromStatus(ah_arithmetic_scalar_arr(s->ctx(), (int)t->id, (int8_t)op, lp, rp, b2, rows));
      }
    }
    // the span's rows of the whole-column validity bitmaps (on the device since ResultValidity)
    // (not uploaded under a null scalar — only checked multiply gets here with one, and it never reads validity: ah_arith.hip)
    const uint8_t* lv = l.has_nulls() && dvl ? (const uint8_t*)dvl->dptr : nullptr;
    const uint8_t* rv = r.has_nulls() && dvr ? (const uint8_t*)dvr->dptr : nullptr;
    return s->FromStatus(ah_arithmetic_checked(s->ctx(), (int)t->id, (int8_t)op, shape, lp, lv, l.arr ? l.arr->offset + r0 : 0, rp, rv,
                                               r.arr ? r.arr->offset + r0 : 0, scalar_valid, b2, rows));
  };
  AHC_RETURN_NOT_OK(SpanLoop(ctx, n, l.arr ? l.values() : nullptr, l.arr ? w : 0, r.arr ? r.values() : nullptr, r.arr ? w : 0, (uint8_t*)hvalues->hptr, w, 1, kernel));
  *out = Datum::Of(MakeHostArray(t, n, hvalues, hvalid, nulls, ""));
  return Status::OK();
}

Status StreamCompare(ExecCtx* ctx, int cmpop, const HostArg& l, const HostArg& r, Datum* out) {
  Session* s = ctx->session;
  const ArrayData& a = *(l.arr ? l.arr : r.arr);
  const DataType* t = a.type;
  const int64_t n = a.length;
  const int w = t->bit_width / 8;
  const int shape = l.arr && r.arr ? AH_SHAPE_AA : (l.arr ? AH_SHAPE_AS : AH_SHAPE_SA);
  BufferPtr dvl, dvr, hvalid, hbits;
  int64_t nulls = 0;
  AHC_RETURN_NOT_OK(ResultValidity(s, l, r, n, &dvl, &dvr, &hvalid, &nulls));
  AHC_RETURN_NOT_OK(s->AllocatePinned((n + 7) / 8, &hbits));
  memset(hbits->hptr, 0, (size_t)((n + 7) / 8));
  // (a null scalar: compareKernel still compares every row with the scalar's stored value — ScalarBinary unboxes it whatever its
  // validity, helpers.go:204-222 — so the data bitmap under the all-null validity is the whole-array path's, not zeros)
  auto kernel = [&](int64_t rows, int64_t, void* b0, void* b1, void* b2) -> Status {
    const void* lp = l.arr ? b0 : (const void*)l.scalar->value;
    const void* rp = r.arr ? b1 : (const void*)r.scalar->value;
    // the slot's bytes behind the last row keep whatever an earlier span left: clear the last byte's tail through a whole-byte memset
    AHC_RETURN_NOT_OK(s->FromStatus(ah_memset_async(s->ctx(), (uint8_t*)b2 + rows / 8, 0, 8)));
    return s->FromStatus(ah_comparison(s->ctx(), cmpop, shape, (int)t->id, lp, rp, (uint8_t*)b2, rows, 0));
  };
  AHC_RETURN_NOT_OK(SpanLoop(ctx, n, l.arr ? l.values() : nullptr, l.arr ? w : 0, r.arr ? r.values() : nullptr, r.arr ? w : 0, (uint8_t*)hbits->hptr, 1, 8, kernel));
  *out = Datum::Of(MakeHostArray(GetDataType(Type::BOOL), n, hbits, hvalid, nulls, ""));
  return Status::OK();
}

Status StreamFilter(ExecCtx* ctx, const FunctionOptions* opts, const std::vector<Datum>& args, Datum* out, bool* handled) {
  Session* s = ctx->session;
  if (args.size() != 2 || args[0].kind != DatumKind::Array || args[1].kind != DatumKind::Array) return Status::OK();
  const ArrayData &v = *args[0].array, &f = *args[1].array;
  if (!v.on_host || !f.on_host || v.device_twin || f.device_twin || f.type->id != Type::BOOL || v.type->bit_width < 8 || !(IsInteger(v.type->id) || IsFloating(v.type->id))) return Status::OK();
  if (v.length != f.length) return Status::OK();   // the reference's error comes from the usual path
  const FilterOptions* fo = static_cast<const FilterOptions*>(opts);
  const int null_sel = fo ? (int)fo->NullSelection : 0;
  const int w = v.type->bit_width / 8;
  ah_ingest* g;
  AHC_RETURN_NOT_OK(s->Ingest((size_t)ctx->ChunkBytes, &g));
  const bool fnulls = f.buffers[0] && f.null_count != 0, vnulls = v.buffers[0] && v.null_count != 0;
  int64_t n_out = 0;
  // (the ingest's filter takes one bit offset for the selection vector and its validity: a sliced mask whose two bitmaps start at
  // different bits does not exist in Arrow — both are indexed by the array's offset)
  AHC_RETURN_NOT_OK(s->FromStatus(ah_ingest_filter_count(g, (const uint8_t*)f.buffers[1]->hptr, fnulls ? (const uint8_t*)f.buffers[0]->hptr : nullptr, f.offset, f.length,
                                                         null_sel, &n_out)));
  const bool allocate_validity = vnulls || fnulls;   // vector_selection.go:486-488
  BufferPtr hvalues, hvalid;
  AHC_RETURN_NOT_OK(s->AllocatePinned(n_out * w, &hvalues));
  if (allocate_validity) AHC_RETURN_NOT_OK(s->AllocatePinned((n_out + 7) / 8, &hvalid));
  int64_t nulls = 0;
  AHC_RETURN_NOT_OK(s->FromStatus(ah_ingest_filter_primitive(g, w, (const uint8_t*)v.buffers[1]->hptr + v.offset * w, vnulls ? (const uint8_t*)v.buffers[0]->hptr : nullptr,
                                                             v.offset, v.length, n_out, hvalues->hptr, allocate_validity ? (uint8_t*)hvalid->hptr : nullptr,
                                                             allocate_validity ? &nulls : nullptr)));
  *out = Datum::Of(MakeHostArray(v.type, n_out, hvalues, hvalid, nulls, v.logical));
  *handled = true;
  return Status::OK();
}

// numeric → numeric cast (CastIntToInt / CastFloatingToInteger / CastIntegerToFloating / CastFloatingToFloating, kernels/numeric_cast.go:37-71):
// one input of one width, one output of another; the safe-cast checks look at valid rows only — the span's rows of the whole-column
// validity bitmap, on the device since ResultValidity — and the first span with an offending row fails the call with the whole-array
// path's text (spans run in row order: that row is the column's first offender)
Status StreamCast(ExecCtx* ctx, const CastOptions& co, const ArrayData& a, Datum* out) {
  Session* s = ctx->session;
  const DataType *from = a.type, *to = co.ToType;
  const int64_t n = a.length;
  const int wi = from->bit_width / 8, wo = to->bit_width / 8;
  HostArg l, none;
  l.arr = &a;
  BufferPtr dvl, dvr, hvalid, hvalues;
  int64_t nulls = 0;
  AHC_RETURN_NOT_OK(ResultValidity(s, l, none, n, &dvl, &dvr, &hvalid, &nulls));
  AHC_RETURN_NOT_OK(s->AllocatePinned(n * wo, &hvalues));
  auto kernel = [&](int64_t rows, int64_t r0, void* b0, void*, void* b2) -> Status {
    return s->FromStatus(ah_cast_numeric(s->ctx(), (int)from->id, (int)to->id, b0, l.has_nulls() ? (const uint8_t*)dvl->dptr : nullptr, a.offset + r0, rows,
                                         co.AllowIntOverflow, co.AllowFloatTruncate, b2));
  };
  AHC_RETURN_NOT_OK(SpanLoop(ctx, n, l.values(), wi, nullptr, 0, (uint8_t*)hvalues->hptr, wo, 1, kernel));
  *out = Datum::Of(MakeHostArray(to, n, hvalues, hvalid, nulls, ""));
  return Status::OK();
}

// cumulative_sum / cumulative_sum_checked of an INTEGER column (cumulativeSumExec, kernels/vector_cumulative.go:228-360): span k + 1 starts
// from the running sum span k ended with — the output of the span's last valid row, read back behind the span's kernel (8 bytes and a
// wait on the compute stream; the uploads of the spans ahead go on).  Integer sums wrap, so the bytes do not depend on where the spans
// are cut: they are the whole-array path's.  (Float columns are uploaded whole: their sums depend on the grouping, and the whole-array
// path's tree is a function of the column's length.)  Nulls: skipped (SkipNulls) or none; with SkipNulls = false everything behind
// the first null is null and the call takes the whole-array path.
Status StreamCumulativeSum(ExecCtx* ctx, bool checked, const CumulativeOptions* opts, const ArrayData& a, Datum* out) {
  Session* s = ctx->session;
  const DataType* t = a.type;
  const int64_t n = a.length;
  const int w = t->bit_width / 8;
  alignas(8) uint8_t carry[8] = {0};
  bool have_start = false;
  if (opts && opts->Start) {
    if (!opts->Start->valid) return Status::Make(StatusCode::Invalid, "cumulative sum start value must be valid");
    AHC_RETURN_NOT_OK(SafeCastCumulativeStart(*opts->Start, t, carry));
    have_start = true;
  }
  HostArg l, none;
  l.arr = &a;
  const bool nulls_in = l.has_nulls();
  BufferPtr dvl, hvalid, hvalues, dvalid_out;
  AHC_RETURN_NOT_OK(s->AllocatePinned(n * w, &hvalues));
  int64_t nulls = 0;
  if (nulls_in) {
    // prepareCumulativeOutput (:211-226): the output's bitmap is pre-set to ones, the kernel writes the rows' bits — the bits behind row
    // n − 1 in the last byte stay ones, as in the whole-array path
    AHC_RETURN_NOT_OK(UploadValidity(s, a, &dvl));
    AHC_RETURN_NOT_OK(s->AllocateBitmap(n, &dvalid_out));
    AHC_RETURN_NOT_OK(s->FromStatus(ah_memset_async(s->ctx(), dvalid_out->dptr, 0xFF, (size_t)((n + 7) / 8))));
    AHC_RETURN_NOT_OK(s->FromStatus(ah_copy_bitmap(s->ctx(), (const uint8_t*)dvl->dptr, a.offset, n, (uint8_t*)dvalid_out->dptr, 0, 0)));
    int64_t set = 0;
    AHC_RETURN_NOT_OK(s->FromStatus(ah_count_set_bits(s->ctx(), (const uint8_t*)dvalid_out->dptr, 0, n, &set)));
    nulls = n - set;
    AHC_RETURN_NOT_OK(s->AllocatePinned((n + 7) / 8, &hvalid));
    AHC_RETURN_NOT_OK(s->FromStatus(ah_download_async(s->ctx(), hvalid->hptr, dvalid_out->dptr, (size_t)((n + 7) / 8))));
    AHC_RETURN_NOT_OK(s->FromStatus(ah_sync(s->ctx())));
  }
  const uint8_t* hv = nulls_in ? (const uint8_t*)a.buffers[0]->hptr : nullptr;
  BufferPtr span_valid;   // the kernel wants somewhere to write a span's validity: scratch, one span's worth (the output's bitmap is made above)
  auto kernel = [&](int64_t rows, int64_t r0, void* b0, void*, void* b2) -> Status {
    if (nulls_in && !span_valid) AHC_RETURN_NOT_OK(s->AllocateBitmap(rows + 64, &span_valid));   // (the first span is the longest)
    AHC_RETURN_NOT_OK(s->FromStatus(ah_cumulative_sum(s->ctx(), (int)t->id, b0, nulls_in ? (const uint8_t*)dvl->dptr : nullptr, a.offset + r0, rows,
                                                      have_start ? carry : nullptr, 1, checked ? 1 : 0, b2, nulls_in ? (uint8_t*)span_valid->dptr : nullptr, nullptr)));
    // the running sum behind this span = the output of its last valid row (none: unchanged)
    int64_t last = rows - 1;
    if (hv) while (last >= 0 && !((hv[(a.offset + r0 + last) >> 3] >> ((a.offset + r0 + last) & 7)) & 1)) last--;
    if (last >= 0) {
      AHC_RETURN_NOT_OK(s->FromStatus(ah_download_async(s->ctx(), carry, (const uint8_t*)b2 + last * w, (size_t)w)));
      AHC_RETURN_NOT_OK(s->FromStatus(ah_sync(s->ctx())));
      have_start = true;
    }
    return Status::OK();
  };
  AHC_RETURN_NOT_OK(SpanLoop(ctx, n, l.values(), w, nullptr, 0, (uint8_t*)hvalues->hptr, w, 1, kernel));
  *out = Datum::Of(MakeHostArray(t, n, hvalues, hvalid, nulls, ""));
  return Status::OK();
}

}  // namespace

Status MaterializeOnDevice(Session* s, const ArrayDataPtr& host, ArrayDataPtr* out) {
  if (!host->on_host) { *out = host; return Status::OK(); }
  if (host->device_twin) { *out = host->device_twin; return Status::OK(); }
  auto d = std::make_shared<ArrayData>(*host);
  d->on_host = false;
  d->device_twin = nullptr;
  const int64_t nbits = host->offset + host->length;
  const int64_t vbytes = (nbits + 7) / 8;
  const int64_t dbytes = host->type->bit_width == 1 ? vbytes : nbits * (host->type->bit_width / 8);
  d->buffers[0] = nullptr;
  if (host->buffers[0] && host->null_count != 0) {
    AHC_RETURN_NOT_OK(s->Allocate(vbytes, &d->buffers[0]));
    AHC_RETURN_NOT_OK(s->FromStatus(ah_upload_async(s->ctx(), d->buffers[0]->dptr, host->buffers[0]->hptr, (size_t)vbytes)));
  } else {
    d->null_count = 0;
  }
  AHC_RETURN_NOT_OK(s->Allocate(dbytes, &d->buffers[1], /*zero_all=*/false));
  if (dbytes > 0) AHC_RETURN_NOT_OK(s->FromStatus(ah_upload_async(s->ctx(), d->buffers[1]->dptr, host->buffers[1]->hptr, (size_t)dbytes)));
  AHC_RETURN_NOT_OK(s->FromStatus(ah_sync(s->ctx())));
  host->device_twin = d;
  *out = d;
  return Status::OK();
}

Status MaterializeAllOnDevice(Session* s, std::vector<Datum>* values) {
  for (Datum& d : *values)
    if (d.kind == DatumKind::Array && d.array->on_host) {
      ArrayDataPtr dev;
      AHC_RETURN_NOT_OK(MaterializeOnDevice(s, d.array, &dev));
      d.array = dev;
    }
  return Status::OK();
}

Status CallHostResident(ExecCtx* ctx, const std::string& name, const FunctionOptions* opts, const std::vector<Datum>& args, Datum* out, bool* handled) {
  *handled = false;
  if (name == "filter" || name == "array_filter") return StreamFilter(ctx, opts, args, out, handled);
  // arithmetic: the function names compute.Add / Subtract / Multiply call (arithmetic.go:679-682, 1105-1125) and the expression layer's
  struct A { const char* name; int op; };
  static const A arith[] = {{"add", AH_OP_ADD}, {"sub", AH_OP_SUB}, {"subtract", AH_OP_SUB}, {"multiply", AH_OP_MUL}};
  struct C { const char* name; int op; bool flip; };
  static const C cmps[] = {{"equal", AH_CMP_EQ, false}, {"not_equal", AH_CMP_NE, false}, {"greater", AH_CMP_GT, false}, {"greater_equal", AH_CMP_GE, false},
                           {"less", AH_CMP_GT, true}, {"less_equal", AH_CMP_GE, true}};
  if (name == "cast" && args.size() == 1 && args[0].kind == DatumKind::Array) {
    const ArrayData& a = *args[0].array;
    const CastOptions* co = dynamic_cast<const CastOptions*>(opts);
    auto numeric = [](const DataType* t) { return t && (IsInteger(t->id) || IsFloating(t->id)); };
    if (!co || !co->ToLogical.empty() || !a.on_host || a.device_twin || !a.logical.empty() || a.length == 0 || !numeric(a.type) || !numeric(co->ToType) || a.type->id == co->ToType->id)
      return Status::OK();   // bool / temporal / identity casts, a missing ToType (the reference's error): the usual path
    *handled = true;
    return StreamCast(ctx, *co, a, out);
  }
  if ((name == "cumulative_sum" || name == "cumulative_sum_checked") && args.size() == 1 && args[0].kind == DatumKind::Array) {
    const ArrayData& a = *args[0].array;
    const CumulativeOptions* co = dynamic_cast<const CumulativeOptions*>(opts);
    const bool has_nulls = a.buffers[0] && a.null_count != 0;
    if (!a.on_host || a.device_twin || !a.logical.empty() || !IsInteger(a.type->id) || a.length == 0 || (has_nulls && !(co && co->SkipNulls))) return Status::OK();
    *handled = true;
    return StreamCumulativeSum(ctx, name == "cumulative_sum_checked", co, a, out);
  }
  HostArg l, r;
  for (const A& a : arith) {
    const bool unchecked = name == std::string(a.name) + "_unchecked";
    if (name != a.name && !unchecked) continue;
    if (!SplitArgs(args, &l, &r)) return Status::OK();
    const DataType* t = (l.arr ? l.arr : r.arr)->type;
    // a NoCheckOverflow option on the checked name selects the unchecked kernels (arithmetic.go:1115-1117)
    const ArithmeticOptions* ao = dynamic_cast<const ArithmeticOptions*>(opts);
    const bool checked = !unchecked && !(ao && ao->NoCheckOverflow) && IsInteger(t->id);
    *handled = true;
    return StreamArithmetic(ctx, checked ? a.op + (AH_OP_ADD_CHECKED - AH_OP_ADD) : a.op, checked, l, r, out);   // the checked entry takes the *_CHECKED ids
  }
  for (const C& c : cmps) {
    if (name != c.name) continue;
    if (!SplitArgs(args, &l, &r)) return Status::OK();
    *handled = true;
    return c.flip ? StreamCompare(ctx, c.op, r, l, out) : StreamCompare(ctx, c.op, l, r, out);   // less(a, b) = greater(b, a): makeFlippedCompare
  }
  return Status::OK();
}

Status SumHostResident(ExecCtx* ctx, const ArrayData& a, double* f64, int64_t* i64, uint64_t* u64) {
  Session* s = ctx->session;
  ah_ingest* g;
  AHC_RETURN_NOT_OK(s->Ingest((size_t)ctx->ChunkBytes, &g));
  const uint8_t* p = (const uint8_t*)a.buffers[1]->hptr + a.offset * 8;
  switch (a.type->id) {
    case Type::FLOAT64: return s->FromStatus(ah_ingest_sum_float64(g, (const double*)p, (size_t)a.length, f64));
    case Type::INT64: return s->FromStatus(ah_ingest_sum_int64(g, (const int64_t*)p, (size_t)a.length, i64));
    case Type::UINT64: return s->FromStatus(ah_ingest_sum_int64(g, (const int64_t*)p, (size_t)a.length, (int64_t*)u64));
    default: return Status::Make(StatusCode::TypeError, "arrow/math has Float64, Int64 and Uint64 Sum only");
  }
}

}  // namespace compute
}  // namespace arrowhip
