// ah_cast_wide.hip — the 32/64-bit integer inputs of ah_cast_numeric (see ah_cast.hip).
#include "ah_cast_impl.h"

using namespace ah_cast_impl;

int ah_cast_from_wide(ah_ctx* c, int in_type, int out_type, const void* in, const uint8_t* valid, int64_t off, int64_t n, void* out, int aio,
                      int aft) {
  switch (in_type) {
    case AH_UINT32: return cast_from<uint32_t>(c, out_type, in, valid, off, n, out, aio, aft);
    case AH_INT32: return cast_from<int32_t>(c, out_type, in, valid, off, n, out, aio, aft);
    case AH_UINT64: return cast_from<uint64_t>(c, out_type, in, valid, off, n, out, aio, aft);
    case AH_INT64: return cast_from<int64_t>(c, out_type, in, valid, off, n, out, aio, aft);
  }
  return ah_fail(c, AH_ENOTIMPL, "cast: unsupported input type %d", in_type);
}
