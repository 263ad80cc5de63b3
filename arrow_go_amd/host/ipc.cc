// Arrow IPC stream reader that lands record-batch bodies in HBM — see ipc.h.
#include "ipc.h"

#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <new>

#include "../../include/arrowhip.h"

namespace arrowhip {
namespace ipc {
namespace {

// ---- a bounds-checked FlatBuffer accessor ------------------------------------------------------
// Layout (flatbuffers internals): a table starts with an int32 pointing BACK to its vtable; the vtable
// is [u16 vtable bytes, u16 table bytes, u16 offset of field 0, …] (0 = field absent → default).
// Strings / vectors / sub-tables are reached through a u32 forward offset stored in the field.
// Every read goes through `at`, which flags the buffer as bad instead of reading outside it.
struct Fb {
  const uint8_t* p;
  int64_t n;
  bool bad = false;

  const uint8_t* at(int64_t off, int64_t len) {
    if (off < 0 || len < 0 || off > n || len > n - off) { bad = true; return nullptr; }
    return p + off;
  }
  template <typename T>
  T rd(int64_t off) {
    const uint8_t* q = at(off, (int64_t)sizeof(T));
    T v{};
    if (q) std::memcpy(&v, q, sizeof(T));
    return v;
  }
  int64_t root() { return (int64_t)rd<uint32_t>(0); }
  // position of field `id` inside table `t`, 0 if absent
  int64_t field(int64_t t, int id) {
    const int64_t vt = t - (int64_t)rd<int32_t>(t);
    const int vt_len = rd<uint16_t>(vt);
    const int slot = 4 + 2 * id;
    if (bad || slot + 2 > vt_len) return 0;
    const int off = rd<uint16_t>(vt + slot);
    return off ? t + off : 0;
  }
  template <typename T>
  T scalar(int64_t t, int id, T dflt) {
    const int64_t f = field(t, id);
    return f ? rd<T>(f) : dflt;
  }
  // target of an offset field (table / string / vector), 0 if absent
  int64_t indirect(int64_t t, int id) {
    const int64_t f = field(t, id);
    return f ? f + (int64_t)rd<uint32_t>(f) : 0;
  }
  int64_t vec_len(int64_t v) { return v ? (int64_t)rd<uint32_t>(v) : 0; }
  std::string str(int64_t s) {
    if (!s) return std::string();
    const int64_t len = (int64_t)rd<uint32_t>(s);
    const uint8_t* q = at(s + 4, len);
    return q ? std::string((const char*)q, (size_t)len) : std::string();
  }
};

// format/Message.fbs
enum { kHeaderSchema = 1, kHeaderDictionaryBatch = 2, kHeaderRecordBatch = 3 };
// format/Schema.fbs: union Type
enum { kTypeInt = 2, kTypeFloatingPoint = 3, kTypeBinary = 4, kTypeUtf8 = 5, kTypeBool = 6, kTypeDate = 8, kTypeTime = 9, kTypeTimestamp = 10,
       kTypeDuration = 18, kTypeLargeBinary = 19, kTypeLargeUtf8 = 20 };

Status Invalid(const std::string& m) { return Status::Make(StatusCode::Invalid, "arrow/ipc: " + m); }
Status NotImpl(const std::string& m) { return Status::Make(StatusCode::NotImplemented, "arrow/ipc: " + m); }

// ---- compressed bodies (format/Message.fbs: BodyCompression { codec: LZ4_FRAME | ZSTD; method: BUFFER }) -----------
// ipc/compression.go:25-40 + file_reader.go:585-612: every buffer of a compressed body is [int64 uncompressed length |
// frame]; −1 means "stored as it is".  The reference decompresses with pierrec/lz4 and klauspost/zstd on the host; here
// the system's liblz4.so.1 / libzstd.so.1 do it (bound at first use — the library has no link-time dependency on them),
// into ONE host buffer laid out like an uncompressed body, which then takes the usual single transfer.
struct Codecs {
  // lz4frame.h
  size_t (*lz4f_create)(void**, unsigned) = nullptr;
  size_t (*lz4f_free)(void*) = nullptr;
  size_t (*lz4f_decompress)(void*, void*, size_t*, const void*, size_t*, const void*) = nullptr;
  unsigned (*lz4f_is_error)(size_t) = nullptr;
  // zstd.h
  size_t (*zstd_decompress)(void*, size_t, const void*, size_t) = nullptr;
  unsigned (*zstd_is_error)(size_t) = nullptr;
};
const Codecs& GetCodecs() {
  static Codecs c;
  static std::once_flag once;
  std::call_once(once, [] {
    if (void* h = dlopen("liblz4.so.1", RTLD_NOW | RTLD_LOCAL)) {
      c.lz4f_create = (decltype(c.lz4f_create))dlsym(h, "LZ4F_createDecompressionContext");
      c.lz4f_free = (decltype(c.lz4f_free))dlsym(h, "LZ4F_freeDecompressionContext");
      c.lz4f_decompress = (decltype(c.lz4f_decompress))dlsym(h, "LZ4F_decompress");
      c.lz4f_is_error = (decltype(c.lz4f_is_error))dlsym(h, "LZ4F_isError");
    }
    if (void* h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL)) {
      c.zstd_decompress = (decltype(c.zstd_decompress))dlsym(h, "ZSTD_decompress");
      c.zstd_is_error = (decltype(c.zstd_is_error))dlsym(h, "ZSTD_isError");
    }
  });
  return c;
}
enum { kCodecLz4Frame = 0, kCodecZstd = 1 };

Status DecompressBuffer(int codec, const uint8_t* src, int64_t srclen, uint8_t* dst, int64_t dstlen) {
  const Codecs& c = GetCodecs();
  if (codec == kCodecZstd) {
    if (!c.zstd_decompress || !c.zstd_is_error) return NotImpl("ZSTD-compressed body: libzstd.so.1 not available");
    const size_t r = c.zstd_decompress(dst, (size_t)dstlen, src, (size_t)srclen);
    if (c.zstd_is_error(r) || (int64_t)r != dstlen) return Invalid("ZSTD buffer does not decompress to the " + std::to_string(dstlen) + " bytes it announces");
    return Status::OK();
  }
  if (!c.lz4f_create || !c.lz4f_free || !c.lz4f_decompress || !c.lz4f_is_error) return NotImpl("LZ4-compressed body: liblz4.so.1 not available");
  void* dctx = nullptr;
  if (c.lz4f_is_error(c.lz4f_create(&dctx, 100 /* LZ4F_VERSION */))) return Invalid("LZ4 decompression context");
  int64_t in = 0, out = 0;
  Status st = Status::OK();
  for (;;) {
    size_t dn = (size_t)(dstlen - out), sn = (size_t)(srclen - in);
    const size_t r = c.lz4f_decompress(dctx, dst + out, &dn, src + in, &sn, nullptr);
    if (c.lz4f_is_error(r)) { st = Invalid("corrupt LZ4 frame"); break; }
    in += (int64_t)sn; out += (int64_t)dn;
    if (r == 0) break;                                   // frame complete
    if (sn == 0 && dn == 0) { st = Invalid("LZ4 frame is truncated or larger than the " + std::to_string(dstlen) + " bytes it announces"); break; }
  }
  c.lz4f_free(dctx);
  if (st.ok() && out != dstlen) st = Invalid("LZ4 buffer decompresses to " + std::to_string(out) + " bytes, announced " + std::to_string(dstlen));
  return st;
}

// one Field table → DataType (metadata.go: typeFromFB / intFromFB / floatFromFB)
Status DecodeField(Fb& fb, int64_t f, FieldInfo* out) {
  out->name = fb.str(fb.indirect(f, 0));
  out->nullable = fb.scalar<uint8_t>(f, 1, 0) != 0;
  const int type_type = fb.scalar<uint8_t>(f, 2, 0);
  const int64_t tt = fb.indirect(f, 3);
  if (const int64_t de = fb.indirect(f, 4)) {  // DictionaryEncoding { id: long; indexType: Int; isOrdered: bool; dictionaryKind }
    out->dict_id = fb.scalar<int64_t>(de, 0, 0);
    const int64_t it = fb.indirect(de, 1);
    const int bits = it ? fb.scalar<int32_t>(it, 0, 0) : 32;   // absent → signed 32-bit (Schema.fbs)
    const bool sgn = it ? fb.scalar<uint8_t>(it, 1, 0) != 0 : true;
    Type iid = Type::NA;
    switch (bits) {
      case 8: iid = sgn ? Type::INT8 : Type::UINT8; break;
      case 16: iid = sgn ? Type::INT16 : Type::UINT16; break;
      case 32: iid = sgn ? Type::INT32 : Type::UINT32; break;
      case 64: iid = sgn ? Type::INT64 : Type::UINT64; break;
      default: return Invalid("field '" + out->name + "': dictionary index of " + std::to_string(bits) + " bits");
    }
    out->index_type = GetDataType(iid);
    if (out->dict_id < 0) return Invalid("field '" + out->name + "': negative dictionary id");
  }
  if (fb.vec_len(fb.indirect(f, 5)) != 0) return NotImpl("field '" + out->name + "' is nested");
  Type id = Type::NA;
  switch (type_type) {
    case kTypeInt: {
      const int bits = tt ? fb.scalar<int32_t>(tt, 0, 0) : 0;
      const bool sgn = tt && fb.scalar<uint8_t>(tt, 1, 0) != 0;
      switch (bits) {
        case 8: id = sgn ? Type::INT8 : Type::UINT8; break;
        case 16: id = sgn ? Type::INT16 : Type::UINT16; break;
        case 32: id = sgn ? Type::INT32 : Type::UINT32; break;
        case 64: id = sgn ? Type::INT64 : Type::UINT64; break;
        default: return Invalid("field '" + out->name + "': integer of " + std::to_string(bits) + " bits");
      }
      break;
    }
    case kTypeFloatingPoint: {
      const int prec = tt ? fb.scalar<int16_t>(tt, 0, 0) : 0;  // HALF 0, SINGLE 1, DOUBLE 2
      if (prec == 1) id = Type::FLOAT32;
      else if (prec == 2) id = Type::FLOAT64;
      else return NotImpl("field '" + out->name + "': float16");
      break;
    }
    case kTypeBool: id = Type::BOOL; break;
    case kTypeUtf8: id = Type::STRING; break;
    case kTypeBinary: id = Type::BINARY; break;
    case kTypeLargeUtf8: id = Type::LARGE_STRING; break;
    case kTypeLargeBinary: id = Type::LARGE_BINARY; break;
    // temporal columns (metadata.go: dateFromFB / timeFromFB / timestampFromFB / durationFromFB): integers with a label,
    // written as the C Data format of the type (TimeUnit: SECOND 0, MILLISECOND 1, MICROSECOND 2, NANOSECOND 3)
    case kTypeDate: {
      const int unit = tt ? fb.scalar<int16_t>(tt, 0, 1) : 1;  // DateUnit: DAY 0, MILLISECOND 1 (default)
      if (unit != 0 && unit != 1) return Invalid("field '" + out->name + "': date unit " + std::to_string(unit));
      out->logical = unit == 0 ? "tdD" : "tdm";
      break;
    }
    case kTypeTime: {
      const int unit = tt ? fb.scalar<int16_t>(tt, 0, 1) : 1;
      const int bits = tt ? fb.scalar<int32_t>(tt, 1, 32) : 32;
      if (unit < 0 || unit > 3 || bits != (unit < 2 ? 32 : 64))
        return Invalid("field '" + out->name + "': time of unit " + std::to_string(unit) + " in " + std::to_string(bits) + " bits");
      out->logical = std::string("tt") + "smun"[unit];
      break;
    }
    case kTypeTimestamp: {
      const int unit = tt ? fb.scalar<int16_t>(tt, 0, 0) : 0;
      if (unit < 0 || unit > 3) return Invalid("field '" + out->name + "': timestamp unit " + std::to_string(unit));
      out->logical = std::string("ts") + "smun"[unit] + ":" + (tt ? fb.str(fb.indirect(tt, 1)) : std::string());
      break;
    }
    case kTypeDuration: {
      const int unit = tt ? fb.scalar<int16_t>(tt, 0, 1) : 1;
      if (unit < 0 || unit > 3) return Invalid("field '" + out->name + "': duration unit " + std::to_string(unit));
      out->logical = std::string("tD") + "smun"[unit];
      break;
    }
    default: return NotImpl("field '" + out->name + "' has flatbuf type " + std::to_string(type_type));
  }
  out->type = out->logical.empty() ? GetDataType(id) : TemporalStorage(out->logical);
  if (!out->type) return NotImpl("field '" + out->name + "': type " + out->logical);
  return Status::OK();
}

}  // namespace

// message.go:207-287: [continuation 0xFFFFFFFF] [int32 metadata bytes] [metadata, padded] [body];
// a length of 0 (with or without the continuation word) or the end of the bytes ends the stream;
// a first word that is not the continuation token is the pre-0.15 framing (the length itself).
Status StreamReader::NextMessage(bool* have, const uint8_t** meta, int64_t* meta_len, const uint8_t** body, int64_t* body_len) {
  *have = false;
  if (pos_ == n_) return Status::OK();
  if (n_ - pos_ < 4) return Invalid("could not read continuation indicator");
  uint32_t word;
  std::memcpy(&word, p_ + pos_, 4);
  pos_ += 4;
  int32_t mlen;
  if (word == 0) return Status::OK();
  if (word == 0xFFFFFFFFu) {
    if (n_ - pos_ < 4) return Invalid("could not read message length");
    std::memcpy(&mlen, p_ + pos_, 4);
    pos_ += 4;
    if (mlen == 0) return Status::OK();
  } else {
    mlen = (int32_t)word;
  }
  if (mlen < 4) return Invalid("invalid message metadata length " + std::to_string(mlen));
  if (mlen > n_ - pos_) return Invalid("could not read message metadata");
  *meta = p_ + pos_;
  *meta_len = mlen;
  pos_ += mlen;
  Fb fb{*meta, *meta_len};
  const int64_t msg = fb.root();
  const int64_t blen = fb.scalar<int64_t>(msg, 3, 0);
  if (fb.bad) return Invalid("invalid message metadata");
  if (blen < 0) return Invalid("invalid message body length " + std::to_string(blen));
  if (blen > n_ - pos_) return Invalid("could not read message body");
  *body = p_ + pos_;
  *body_len = blen;
  pos_ += blen;
  *have = true;
  return Status::OK();
}

Status StreamReader::Open(Session* s, const uint8_t* bytes, int64_t len, std::unique_ptr<StreamReader>* out) {
  std::unique_ptr<StreamReader> r(new StreamReader());
  r->s_ = s;
  r->p_ = bytes;
  r->n_ = len;
  // the FILE format (ipc.NewFileReader, arrow/ipc/file_reader.go:60-160) is the same message sequence between an
  // "ARROW1\0\0" magic and a footer: after the magic it reads as a stream and ends at the EOS marker in front of the footer
  if (len >= 8 && std::memcmp(bytes, "ARROW1\0\0", 8) == 0) r->pos_ = 8;
  bool have;
  const uint8_t *meta, *body;
  int64_t mlen, blen;
  AHC_RETURN_NOT_OK(r->NextMessage(&have, &meta, &mlen, &body, &blen));
  if (!have) return Invalid("stream holds no schema message");  // reader.go:148-160 getSchema
  Fb fb{meta, mlen};
  const int64_t msg = fb.root();
  if (fb.scalar<uint8_t>(msg, 1, 0) != kHeaderSchema) return Invalid("invalid message type (got=" + std::to_string(fb.scalar<uint8_t>(msg, 1, 0)) + ", want=Schema)");
  const int64_t sch = fb.indirect(msg, 2);
  if (!sch || fb.bad) return Invalid("invalid message metadata");
  if (fb.scalar<int16_t>(sch, 0, 0) != 0) return NotImpl("big-endian stream");
  const int64_t fv = fb.indirect(sch, 1);
  const int64_t nf = fb.vec_len(fv);
  for (int64_t i = 0; i < nf && !fb.bad; i++) {
    const int64_t slot = fv + 4 + 4 * i;
    const int64_t f = slot + (int64_t)fb.rd<uint32_t>(slot);
    FieldInfo fi;
    AHC_RETURN_NOT_OK(DecodeField(fb, f, &fi));
    r->fields_.push_back(std::move(fi));
  }
  if (fb.bad) return Invalid("invalid message metadata");
  *out = std::move(r);
  return Status::OK();
}

// reader.go:249-300 (next) + file_reader.go:523-575 (newRecordBatch) + arrayLoaderContext (:660-…):
// nodes and buffers are consumed in field order — validity, then data (fixed width / bool) or
// offsets + data (binary).
Status StreamReader::Next(bool* have, std::vector<ArrayDataPtr>* columns, int64_t* rows) {
  if (columns) columns->clear();
  *rows = 0;
  const uint8_t *meta, *body;
  int64_t mlen, blen;
  for (;;) {
    AHC_RETURN_NOT_OK(NextMessage(have, &meta, &mlen, &body, &blen));
    if (!*have) return Status::OK();
    Fb fb{meta, mlen};
    const int64_t msg = fb.root();
    const int ht = fb.scalar<uint8_t>(msg, 1, 0);
    const int64_t hdr = fb.indirect(msg, 2);
    if (fb.bad || !hdr) return Invalid("invalid message metadata");
    if (ht == kHeaderRecordBatch) return LoadColumns(meta, mlen, hdr, body, blen, fields_, false, columns, rows);
    if (ht != kHeaderDictionaryBatch) return Invalid("invalid message type (got=" + std::to_string(ht) + ", want=RecordBatch)");
    // DictionaryBatch { id: long; data: RecordBatch; isDelta: bool } (reader.go:167-200 readDictionary): one column of the
    // field's value type, kept by id for the record batches that follow
    const int64_t id = fb.scalar<int64_t>(hdr, 0, 0);
    const int64_t data = fb.indirect(hdr, 1);
    if (fb.bad || !data) return Invalid("invalid message metadata");
    const bool is_delta = fb.scalar<uint8_t>(hdr, 2, 0) != 0;
    if (is_delta && !seen_dict_.count(id)) return Invalid("delta dictionary batch for id " + std::to_string(id) + " without a dictionary to extend");
    const FieldInfo* owner = nullptr;
    for (auto& f : fields_) if (f.dict_id == id) owner = &f;
    if (!owner) return Invalid("dictionary batch for unknown dictionary id " + std::to_string(id));
    std::vector<FieldInfo> one{*owner};
    std::vector<ArrayDataPtr> vals;
    int64_t nvals = 0;
    AHC_RETURN_NOT_OK(LoadColumns(meta, mlen, data, body, blen, one, true, columns ? &vals : nullptr, &nvals));
    if (columns) {
      if (is_delta) {  // readDictionary (reader.go:186-196): the new values are appended to the existing dictionary
        ArrayDataPtr merged;
        AHC_RETURN_NOT_OK(compute::Concatenate(s_, {dicts_[id], vals[0]}, owner->type, &merged));
        dicts_[id] = merged;
      } else {
        dicts_[id] = vals[0];
      }
    }
    seen_dict_[id] = true;
  }
}

Status StreamReader::LoadColumns(const uint8_t* meta, int64_t mlen, int64_t rb, const uint8_t* body, int64_t blen,
                                 const std::vector<FieldInfo>& fields_, bool as_values, std::vector<ArrayDataPtr>* columns, int64_t* rows) {
  const bool dry = columns == nullptr;  // validate the metadata against the body, move nothing
  Fb fb{meta, mlen};
  const int64_t nrows = fb.scalar<int64_t>(rb, 0, 0);
  const int64_t nodes = fb.indirect(rb, 1), bufs = fb.indirect(rb, 2);
  const int64_t n_nodes = fb.vec_len(nodes), n_bufs = fb.vec_len(bufs);
  if (nrows < 0 || fb.bad) return Invalid("invalid message metadata");
  if (n_nodes != (int64_t)fields_.size()) return Invalid("record batch has " + std::to_string(n_nodes) + " field nodes, the schema " + std::to_string(fields_.size()) + " fields");
  // a compressed body is inflated on the host into the layout an uncompressed body has; (offset, length) pairs follow it
  std::vector<uint8_t> plain;
  std::vector<std::pair<int64_t, int64_t>> plain_bufs;
  const int64_t comp = fb.indirect(rb, 3);
  if (comp) {
    const int codec = fb.scalar<int8_t>(comp, 0, 0), method = fb.scalar<int8_t>(comp, 1, 0);
    if (fb.bad) return Invalid("invalid message metadata");
    if (codec != kCodecLz4Frame && codec != kCodecZstd) return NotImpl("body compression codec " + std::to_string(codec));
    if (method != 0) return NotImpl("body compression method " + std::to_string(method));
    struct Piece { int64_t src, srclen, dst, dstlen; bool stored; };
    std::vector<Piece> pieces;
    int64_t total = 0;
    for (int64_t i = 0; i < n_bufs; i++) {
      const int64_t e = bufs + 4 + 16 * i;
      const int64_t off = fb.rd<int64_t>(e), len = fb.rd<int64_t>(e + 8);
      if (fb.bad) return Invalid("invalid message metadata");
      if (off < 0 || len < 0 || off > blen || len > blen - off) return Invalid("buffer [" + std::to_string(off) + ", +" + std::to_string(len) + ") lies outside the " + std::to_string(blen) + "-byte body");
      if (len == 0) { pieces.push_back({0, 0, total, 0, true}); continue; }
      if (len < 8) return Invalid("compressed buffer of " + std::to_string(len) + " bytes has no length prefix");
      int64_t ulen;
      std::memcpy(&ulen, body + off, 8);
      const bool stored = ulen == -1;
      if (stored) ulen = len - 8;
      // a frame cannot announce more than the formats' best ratios allow from its own size: (zeros compress ≈ 30 000 : 1 under ZSTD): no allocation bombs from 20 bytes
      if (ulen < 0 || (!stored && ulen / 131072 > len + 1024)) return Invalid("compressed buffer announces " + std::to_string(ulen) + " bytes from " + std::to_string(len));
      pieces.push_back({off + 8, len - 8, total, ulen, stored});
      total += (ulen + 63) & ~(int64_t)63;
      // the sum is bounded as well — an ≈ 8 MB message of maximal ratios could otherwise ask for hundreds of GiB, which an
      // overcommitting host grants and then kills the process for: ARROWHIP_IPC_MAX_INFLATED_BYTES (default 16 GiB per message)
      static const int64_t kMaxInflated = [] { const char* e = getenv("ARROWHIP_IPC_MAX_INFLATED_BYTES"); const long long v = e ? atoll(e) : 0; return v > 0 ? (int64_t)v : ((int64_t)16 << 30); }();
      if (total > kMaxInflated) return Invalid("decompressed body beyond " + std::to_string(kMaxInflated) + " bytes (ARROWHIP_IPC_MAX_INFLATED_BYTES)");
    }
    try { plain.resize((size_t)total); } catch (const std::bad_alloc&) { return Invalid("cannot hold a decompressed body of " + std::to_string(total) + " bytes"); }
    for (const Piece& p : pieces) {
      if (p.dstlen == 0) { /* nothing */ }
      else if (p.stored) std::memcpy(plain.data() + p.dst, body + p.src, (size_t)p.dstlen);
      else AHC_RETURN_NOT_OK(DecompressBuffer(codec, body + p.src, p.srclen, plain.data() + p.dst, p.dstlen));
      plain_bufs.push_back({p.dst, p.dstlen});
    }
    body = plain.data();
    blen = total;
  }

  // the whole body in one transfer; columns are slices of it
  BufferPtr dev;
  if (!dry) AHC_RETURN_NOT_OK(s_->Allocate(blen, &dev));
  if (!dry && blen > 0) {
    AHC_RETURN_NOT_OK(s_->FromStatus(ah_upload_async(s_->ctx(), dev->dptr, body, (size_t)blen)));
    AHC_RETURN_NOT_OK(s_->FromStatus(ah_sync(s_->ctx())));
    uploaded_ += blen;
  }
  int64_t ib = 0;
  auto next_buffer = [&](int64_t* off, int64_t* len) -> Status {
    if (ib >= n_bufs) return Invalid("buffer index out of bound");
    if (comp) {
      *off = plain_bufs[(size_t)ib].first;
      *len = plain_bufs[(size_t)ib].second;
      ib++;
    } else {
      const int64_t e = bufs + 4 + 16 * ib++;  // struct Buffer { offset: long; length: long }
      *off = fb.rd<int64_t>(e);
      *len = fb.rd<int64_t>(e + 8);
    }
    if (fb.bad) return Invalid("invalid message metadata");
    if (*off < 0 || *len < 0 || *off > blen || *len > blen - *off) return Invalid("buffer [" + std::to_string(*off) + ", +" + std::to_string(*len) + ") lies outside the " + std::to_string(blen) + "-byte body");
    if (*len > 0 && (*off & 7)) return Invalid("buffer offset " + std::to_string(*off) + " is not 8-byte aligned");
    return Status::OK();
  };
  auto slice = [&](int64_t off, int64_t len) -> BufferPtr {
    if (dry) return nullptr;
    auto b = std::make_shared<Buffer>();
    s_->Keep(b.get());
    b->dptr = (uint8_t*)dev->dptr + off;
    b->size = len;
    b->owned = false;
    b->owner = dev;
    return b;
  };
  for (size_t i = 0; i < fields_.size(); i++) {
    const FieldInfo& fi = fields_[i];
    const int64_t ne = nodes + 4 + 16 * (int64_t)i;  // struct FieldNode { length: long; null_count: long }
    const int64_t flen = fb.rd<int64_t>(ne), fnulls = fb.rd<int64_t>(ne + 8);
    if (fb.bad) return Invalid("invalid message metadata");
    if (flen != nrows) return Invalid("field '" + fi.name + "' has " + std::to_string(flen) + " rows, the batch " + std::to_string(nrows));
    // every size below is flen × (≤ 8 bytes) or (flen + 1) × 8: a crafted length ≥ 2^60 would wrap those products to something
    // small and pass the "buffer holds enough bytes" tests.  A row needs at least one bit of body, so bound it by that first.
    if (flen < 0 || flen > blen * 8) return Invalid("field '" + fi.name + "': " + std::to_string(flen) + " rows cannot fit a " + std::to_string(blen) + "-byte body");
    if (fnulls < 0 || fnulls > flen) return Invalid("field '" + fi.name + "': null count " + std::to_string(fnulls));
    const bool encoded = fi.dict_id >= 0 && !as_values;   // the column holds indices into dictionary fi.dict_id
    if (encoded && !seen_dict_.count(fi.dict_id)) return Invalid("field '" + fi.name + "': no dictionary batch with id " + std::to_string(fi.dict_id) + " before the record batch");
    const DataType* storage = encoded ? fi.index_type : fi.type;
    auto d = std::make_shared<ArrayData>();
    d->type = storage;
    if (!encoded) d->logical = fi.logical;   // a dictionary-encoded temporal column: the label sits on the dictionary's values
    d->length = flen;
    d->offset = 0;
    int64_t off, len;
    AHC_RETURN_NOT_OK(next_buffer(&off, &len));
    if (fnulls > 0) {
      if (len < (flen + 7) / 8) return Invalid("field '" + fi.name + "': validity bitmap of " + std::to_string(len) + " bytes for " + std::to_string(flen) + " rows");
      d->buffers[0] = slice(off, len);
      d->null_count = fnulls;
    } else {
      d->null_count = 0;  // a bitmap that is present but all set is dropped, like loadCommon (file_reader.go:700-712)
    }
    const int w = storage->bit_width / 8;
    if (IsBaseBinary(storage->id)) {
      AHC_RETURN_NOT_OK(next_buffer(&off, &len));
      if (flen > 0 && len < (flen + 1) * w) return Invalid("field '" + fi.name + "': offsets buffer of " + std::to_string(len) + " bytes for " + std::to_string(flen) + " rows");
      if (len == 0 && !dry) {  // an empty array may carry no offsets at all: give it the single zero Arrow asks for
        BufferPtr z;
        AHC_RETURN_NOT_OK(s_->Allocate(w, &z));
        AHC_RETURN_NOT_OK(s_->FromStatus(ah_memset_async(s_->ctx(), z->dptr, 0, (size_t)w)));
        d->buffers[1] = z;
      } else {
        d->buffers[1] = slice(off, len);
      }
      const int64_t ooff = off, olen = len;
      AHC_RETURN_NOT_OK(next_buffer(&off, &len));
      d->buffers[2] = slice(off, len);
      if (olen > 0) {  // the body is still on the host: the first and last offset bound every device read of the data
        int64_t first, last;
        if (w == 4) { int32_t a, b; std::memcpy(&a, body + ooff, 4); std::memcpy(&b, body + ooff + flen * 4, 4); first = a; last = b; }
        else { std::memcpy(&first, body + ooff, 8); std::memcpy(&last, body + ooff + flen * 8, 8); }
        if (first < 0 || last < first || last > len) return Invalid("field '" + fi.name + "': offsets [" + std::to_string(first) + ", " + std::to_string(last) + "] do not fit the " + std::to_string(len) + "-byte data buffer");
      }
    } else {
      AHC_RETURN_NOT_OK(next_buffer(&off, &len));
      const int64_t need = storage->bit_width == 1 ? (flen + 7) / 8 : flen * w;
      if (len < need) return Invalid("field '" + fi.name + "': data buffer of " + std::to_string(len) + " bytes for " + std::to_string(flen) + " rows");
      d->buffers[1] = slice(off, len);
    }
    if (encoded && !dry) {
      // on the device dictionary indices are int32 (arrowhip_compute.h); the wire type is kept for export
      if (storage->id != Type::INT32 && flen > 0) {
        BufferPtr idx32;
        AHC_RETURN_NOT_OK(s_->Allocate(flen * 4, &idx32, /*zero_all=*/false));
        AHC_RETURN_NOT_OK(s_->FromStatus(ah_cast_numeric(s_->ctx(), (int)storage->id, AH_INT32, d->buffers[1]->dptr, nullptr, 0, flen, 1, 1, idx32->dptr)));
        d->buffers[1] = idx32;
      }
      d->type = GetDataType(Type::DICTIONARY);
      d->dict_index_type = storage;
      d->dict_value_type = fi.type;
      d->dictionary = dicts_[fi.dict_id];
    }
    if (!dry) columns->push_back(d);
  }
  *rows = nrows;
  return Status::OK();
}

}  // namespace ipc
}  // namespace arrowhip
