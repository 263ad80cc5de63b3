// Arrow IPC *stream* → record batches resident in HBM.
//
// What ipc.NewReader / Reader.Next do on the host (arrow/ipc/reader.go:97-120,202-300;
// message framing message.go:207-287; batch assembly file_reader.go:523-575, buffers :583-616,
// type decoding metadata.go) with one difference in where the bytes land: the body of a RecordBatch
// message is sent to the device with ONE copy and every column buffer becomes a slice of that one
// allocation — the IPC body is already laid out as Arrow buffers (8-byte aligned, validity / offsets /
// data in field order), so nothing is unpacked on the host.
//
// Both the stream format and the file format ("ARROW1" magic … footer) are read; the footer's block index is not
// needed for a sequential pass.
// Scope: flat columns of the types the kernels take — Int8..Uint64, Float32/64, Bool, Date32/64, Time32/64, Timestamp,
// Duration, Utf8 / Binary and their Large variants, plain or dictionary-encoded (DictionaryBatch messages, replacement and delta);
// little-endian; bodies uncompressed or compressed per buffer with LZ4_FRAME / ZSTD (ipc/compression.go: inflated on the host
// by the system's liblz4 / libzstd into the layout of an uncompressed body, then the same single transfer).  Anything else is
// ErrNotImplemented with the field named.  The metadata is a FlatBuffer (format/Message.fbs,
// Schema.fbs); it is read with the small bounds-checked accessor in ipc.cc — the bytes come from a
// file or a socket and are not trusted.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "arrowhip_compute.h"

namespace arrowhip {
namespace ipc {

struct FieldInfo {
  std::string name;
  const DataType* type = nullptr;        // the column's type; for a dictionary-encoded field the VALUE type
  std::string logical;                   // temporal columns: C Data format of the type, `type` is then the storage integer
  bool nullable = true;
  int64_t dict_id = -1;                  // ≥ 0: dictionary-encoded (Schema.fbs DictionaryEncoding.id)
  const DataType* index_type = nullptr;  // … with these indices on the wire
};

class StreamReader {
 public:
  // Parses the Schema message.  `bytes` must stay alive and unchanged while the reader is used.
  static Status Open(Session* s, const uint8_t* bytes, int64_t len, std::unique_ptr<StreamReader>* out);
  const std::vector<FieldInfo>& fields() const { return fields_; }
  // Next record batch: *have = false at the end of the stream (EOS marker or end of bytes).
  // columns == nullptr: only check the batch's metadata against its body (no session needed).
  Status Next(bool* have, std::vector<ArrayDataPtr>* columns, int64_t* rows);
  int64_t body_bytes_uploaded() const { return uploaded_; }

 private:
  Status NextMessage(bool* have, const uint8_t** meta, int64_t* meta_len, const uint8_t** body, int64_t* body_len);
  // one RecordBatch table (a record batch message, or the `data` of a dictionary batch) against its body
  Status LoadColumns(const uint8_t* meta, int64_t meta_len, int64_t rb, const uint8_t* body, int64_t body_len,
                     const std::vector<FieldInfo>& fields, bool as_values, std::vector<ArrayDataPtr>* columns, int64_t* rows);
  std::map<int64_t, ArrayDataPtr> dicts_;   // dictionary id → values (dictutils.Memo)
  std::map<int64_t, bool> seen_dict_;
  Session* s_ = nullptr;
  const uint8_t* p_ = nullptr;
  int64_t n_ = 0, pos_ = 0, uploaded_ = 0;
  std::vector<FieldInfo> fields_;
};

}  // namespace ipc
}  // namespace arrowhip
