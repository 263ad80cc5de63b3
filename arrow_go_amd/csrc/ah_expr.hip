// ah_expr.hip — fused evaluation of a scalar expression tree in ONE kernel (hiprtc JIT).
//
// Row §8(f)-1 of SURVEY.md: the natural caller of the element-wise kernels is the
// expression executor — compute.Expression / NewCall / NewFieldRef / NewLiteral
// (arrow/compute/expression.go:596-620) evaluated by executeScalarBatch
// (arrow/compute/exprs/exec.go:542-700), which runs ONE scalar kernel per call node and
// materialises every intermediate array.  For (a + b) * c > t that is 3 kernels and
// 24+24+8.125 bytes/row of HBM traffic.  Here the tree (as a postfix program over input
// columns and literals) is turned into HIP source, compiled for gfx950 with hiprtc, cached
// by signature, and evaluated in a single pass: every input column is read once, only the
// final result is written.
//
// Semantics are those of the per-call execution, bit for bit: every call on this path has
// NullHandling = NullIntersection (validity = AND of the operands' validity), unchecked
// kernels compute every slot (null payloads included), checked integer kernels write 0 under
// nulls and fail with "overflow" by the reference's carry test (see ah_arith.hip), compares
// produce bits, and/or/xor/and_not/invert are the plain (non-Kleene) bitmap ops.  Float
// contraction is disabled (-ffp-contract=off): a*b+c rounds twice, like two kernels.
// No implicit casts: the operands of a call must have the same type (→ AH_ENOTIMPL).
#include <hip/hiprtc.h>

#include <map>
#include <string>
#include <vector>

#include "ah_common.h"

struct ah_expr {
  std::string source, signature;
  hipModule_t module = nullptr;
  hipFunction_t fn = nullptr;
  int n_cols = 0, n_lits = 0;
  int out_type = 0;  // arrow.Type id; 1 = BOOL
  bool has_checked = false;
  std::vector<int> col_types, lit_types;
};

namespace {

constexpr int kMaxCols = 16, kMaxLits = 16;
constexpr int AH_BOOL_T = 1;

struct KernelParams {  // must match the struct in the generated source
  const void* col[kMaxCols];
  const uint8_t* valid[kMaxCols];
  long long off[kMaxCols];
  unsigned long long lit[kMaxLits];
  int lit_valid[kMaxLits];
  long long n;
  void* out;
  uint8_t* out_valid;
  unsigned* flag;
};

const char* CType(int t) {
  switch (t) {
    case AH_BOOL_T: return "bool";
    case AH_UINT8: return "unsigned char";
    case AH_INT8: return "signed char";
    case AH_UINT16: return "unsigned short";
    case AH_INT16: return "short";
    case AH_UINT32: return "unsigned int";
    case AH_INT32: return "int";
    case AH_UINT64: return "unsigned long long";
    case AH_INT64: return "long long";
    case AH_FLOAT32: return "float";
    case AH_FLOAT64: return "double";
  }
  return nullptr;
}
const char* UType(int t) {
  switch (t) {
    case AH_UINT8: case AH_INT8: return "unsigned char";
    case AH_UINT16: case AH_INT16: return "unsigned short";
    case AH_UINT32: case AH_INT32: return "unsigned int";
    case AH_UINT64: case AH_INT64: return "unsigned long long";
  }
  return nullptr;
}
bool IsInt(int t) { return t >= AH_UINT8 && t <= AH_INT64; }
bool IsSigned(int t) { return t == AH_INT8 || t == AH_INT16 || t == AH_INT32 || t == AH_INT64; }
bool IsFloat(int t) { return t == AH_FLOAT32 || t == AH_FLOAT64; }
bool IsNum(int t) { return IsInt(t) || IsFloat(t); }
// conversions that keep every value (so the safe cast DispatchBest would insert cannot fail)
bool ValuePreserving(int from, int to) {
  if (!IsNum(from) || !IsNum(to)) return false;
  if (from == to) return true;
  const int fb = ah_type_width(from) * 8, tb = ah_type_width(to) * 8;
  if (IsInt(from) && IsInt(to)) return IsSigned(from) == IsSigned(to) ? tb >= fb : (!IsSigned(from) && IsSigned(to) && tb > fb);
  if (IsInt(from)) return to == AH_FLOAT64 ? fb <= 32 : fb <= 16;
  return from == AH_FLOAT32 && to == AH_FLOAT64;
}

const char* kPrelude = R"SRC(
typedef unsigned long long u64;
struct Params {
  const void* col[16]; const unsigned char* valid[16]; long long off[16];
  u64 lit[16]; int lit_valid[16];
  long long n; void* out; unsigned char* out_valid; unsigned* flag;
};
// 64 bits of a bitmap starting at bit `pos` (cnt valid), whole aligned words only; NULL = all ones
__device__ __forceinline__ u64 load_bits64(const unsigned char* bm, long long pos, int cnt) {
  u64 mask = cnt >= 64 ? ~0ull : ((1ull << cnt) - 1);
  if (cnt <= 0) return 0;
  if (bm == nullptr) return mask;
  unsigned long long addr = (unsigned long long)bm + (unsigned long long)(pos >> 3);
  unsigned long long base = addr & ~7ull;
  int shift = (int)((addr - base) * 8 + (pos & 7));
  // constant address space + wave-uniform address → scalar loads (s_load_dwordx2): the bitmap
  // words cost nothing on the vector memory pipe.  Inputs are never written by this kernel.
  typedef const __attribute__((address_space(4))) u64 cu64;
  const cu64* w = (const cu64*)base;
  u64 lo = w[0] >> shift;
  if (shift != 0 && shift + cnt > 64) lo |= w[1] << (64 - shift);
  return lo & mask;
}
__device__ __forceinline__ void store_word(unsigned char* out, long long row0, int cnt, u64 w) {
  unsigned char* p = out + (row0 >> 3);
  if (cnt >= 64) { *(u64*)p = w; return; }
  int nb = (cnt + 7) >> 3;
  for (int b = 0; b < nb; b++) p[b] = (unsigned char)(w >> (8 * b));
}
// checked integer ops: the reference's carry test (kernels/base_arithmetic.go:249-286,84-106)
template <typename T, typename U, bool SIGNED, int OP>
__device__ __forceinline__ T chk(T a, T b, bool valid, bool& ovf) {
  const int bits = sizeof(T) * 8;
  if (OP == 2) {
    const T tmin = SIGNED ? (T)((U)1 << (bits - 1)) : (T)0;
    const T tmax = SIGNED ? (T)(~((U)1 << (bits - 1))) : (T)~(U)0;
    bool o = false;
    if (a > 0) { if (b > 0) { if (a > (T)(tmax / b)) o = true; } else { if (b < (T)(tmin / a)) o = true; } }
    else if (b > 0) { if (a < (T)(tmin / b)) o = true; }
    else { if (a != 0 && b < (T)(tmax / a)) o = true; }
    ovf |= o;
    return o ? (T)0 : (T)((U)a * (U)b);
  }
  if (!valid) return (T)0;
  U ua = (U)a, ub = (U)b, o, cy;
  if (OP == 0) { o = (U)(ua + ub); cy = (U)((ua & ub) | ((ua | ub) & (U)~o)); }
  else { o = (U)(ua - ub); cy = (U)(((U)~ua & ub) | ((U) ~(ua ^ ub) & o)); }
  bool top = (cy >> (bits - 1)) & 1, next = (cy >> (bits - 2)) & 1;
  ovf |= SIGNED ? (!top && next) : top;
  return (T)o;
}
)SRC";

struct NodeVal { int type; std::string v, ok; };

int Generate(ah_ctx* c, const ah_expr_node* nodes, int n_nodes, const int* col_types, int n_cols, const int* lit_types, int n_lits,
             ah_expr* e) {
  if (n_cols > kMaxCols || n_lits > kMaxLits) return ah_fail(c, AH_ENOTIMPL, "expr: at most %d columns and %d literals", kMaxCols, kMaxLits);
  std::string body;
  std::vector<NodeVal> stack;
  std::vector<bool> col_used(n_cols, false);
  int tmp = 0;
  char buf[512];
  auto push = [&](int type, const std::string& expr, const std::string& ok) {
    std::string v = "v" + std::to_string(tmp), k = "k" + std::to_string(tmp);
    tmp++;
    body += std::string("        const ") + CType(type) + " " + v + " = " + expr + ";\n";
    body += "        const bool " + k + " = " + ok + ";\n";
    stack.push_back({type, v, k});
  };
  for (int i = 0; i < n_nodes; i++) {
    int op = nodes[i].op, arg = nodes[i].arg;
    if (op == AH_X_FIELD) {
      if (arg < 0 || arg >= n_cols || !CType(col_types[arg])) return ah_fail(c, AH_EINVALID, "expr: bad field reference %d", arg);
      col_used[arg] = true;
      std::string ci = std::to_string(arg);
      push(col_types[arg], "c" + ci, "kc" + ci);
    } else if (op == AH_X_LITERAL) {
      if (arg < 0 || arg >= n_lits || !CType(lit_types[arg])) return ah_fail(c, AH_EINVALID, "expr: bad literal reference %d", arg);
      push(lit_types[arg], "l" + std::to_string(arg), "kl" + std::to_string(arg));
    } else if (op >= AH_X_NEGATE && op <= AH_X_SIGN) {
      if (stack.empty()) return ah_fail(c, AH_EINVALID, "expr: stack underflow");
      NodeVal a = stack.back(); stack.pop_back();
      if (!IsNum(a.type)) return ah_fail(c, AH_ENOTIMPL, "expr: unary arithmetic needs a numeric operand");
      std::string T = CType(a.type), ex;
      if (op == AH_X_NEGATE) ex = IsInt(a.type) ? "(" + T + ")((" + UType(a.type) + ")0 - (" + UType(a.type) + ")" + a.v + ")" : "-" + a.v;
      else if (op == AH_X_ABS) {
        if (IsFloat(a.type)) ex = "__builtin_fabs" + std::string(a.type == AH_FLOAT32 ? "f" : "") + "(" + a.v + ")";
        else if (!IsSigned(a.type)) ex = a.v;
        else ex = "(" + T + ")(((" + UType(a.type) + ")" + a.v + " + (" + a.v + " < 0 ? (" + UType(a.type) + ")~(" + UType(a.type) + ")0 : (" + UType(a.type) + ")0)) ^ (" + a.v + " < 0 ? (" + UType(a.type) + ")~(" + UType(a.type) + ")0 : (" + UType(a.type) + ")0))";
      } else {
        if (IsFloat(a.type)) ex = "(" + a.v + " != " + a.v + ") ? " + a.v + " : (" + a.v + " == 0 ? (" + T + ")0 : (__builtin_signbit(" + a.v + ") ? (" + T + ")-1 : (" + T + ")1))";
        else if (!IsSigned(a.type)) ex = "(" + T + ")(" + a.v + " > 0 ? 1 : 0)";
        else ex = "(" + T + ")(" + a.v + " > 0 ? 1 : (" + a.v + " ? -1 : 0))";
      }
      push(a.type, ex, a.ok);
    } else if (op == AH_X_CAST) {
      if (stack.empty()) return ah_fail(c, AH_EINVALID, "expr: stack underflow");
      NodeVal a = stack.back(); stack.pop_back();
      if (!ValuePreserving(a.type, arg))
        return ah_fail(c, AH_ENOTIMPL, "expr: cast %s → type %d is not value-preserving: it needs the checked cast kernel", CType(a.type) ? CType(a.type) : "?", arg);
      push(arg, "(" + std::string(CType(arg)) + ")" + a.v, a.ok);
    } else if (op == AH_X_INVERT) {
      if (stack.empty()) return ah_fail(c, AH_EINVALID, "expr: stack underflow");
      NodeVal a = stack.back(); stack.pop_back();
      if (a.type != AH_BOOL_T) return ah_fail(c, AH_ENOTIMPL, "expr: invert needs a boolean operand");
      push(AH_BOOL_T, "!" + a.v, a.ok);
    } else {
      if (stack.size() < 2) return ah_fail(c, AH_EINVALID, "expr: stack underflow");
      NodeVal b = stack.back(); stack.pop_back();
      NodeVal a = stack.back(); stack.pop_back();
      if (a.type != b.type)
        return ah_fail(c, AH_ENOTIMPL, "expr: operand types differ (%s, %s): the caller inserts AH_X_CAST where DispatchBest would cast", CType(a.type), CType(b.type));
      std::string ok = a.ok + " && " + b.ok, T = CType(a.type);
      if (op >= AH_X_ADD && op <= AH_X_MUL_CHECKED) {
        if (!IsNum(a.type)) return ah_fail(c, AH_ENOTIMPL, "expr: arithmetic needs numeric operands");
        int base = (op - AH_X_ADD) % 3;  // 0 add, 1 sub, 2 mul
        bool checked = op >= AH_X_ADD_CHECKED && IsInt(a.type);  // checked float == unchecked (base_arithmetic_amd64.go:109-117)
        const char* sym = base == 0 ? "+" : base == 1 ? "-" : "*";
        std::string ex;
        if (checked) {
          e->has_checked = true;
          snprintf(buf, sizeof buf, "chk<%s, %s, %s, %d>(%s, %s, %s, ovf)", T.c_str(), UType(a.type), IsSigned(a.type) ? "true" : "false", base,
                   a.v.c_str(), b.v.c_str(), ("(" + ok + ")").c_str());
          ex = buf;
        } else if (IsInt(a.type)) {
          ex = "(" + T + ")((" + UType(a.type) + ")" + a.v + " " + sym + " (" + UType(a.type) + ")" + b.v + ")";
        } else {
          ex = a.v + " " + sym + " " + b.v;
        }
        push(a.type, ex, ok);
      } else if (op >= AH_X_EQ && op <= AH_X_LE) {
        if (!IsNum(a.type)) return ah_fail(c, AH_ENOTIMPL, "expr: comparison needs numeric operands");
        const char* sym[] = {"==", "!=", ">", ">=", "<", "<="};
        push(AH_BOOL_T, a.v + " " + sym[op - AH_X_EQ] + " " + b.v, ok);
      } else if (op >= AH_X_AND && op <= AH_X_AND_NOT) {
        if (a.type != AH_BOOL_T) return ah_fail(c, AH_ENOTIMPL, "expr: boolean op needs boolean operands");
        std::string ex = op == AH_X_AND ? a.v + " && " + b.v : op == AH_X_OR ? a.v + " || " + b.v : op == AH_X_XOR ? a.v + " != " + b.v : a.v + " && !" + b.v;
        push(AH_BOOL_T, ex, ok);
      } else {
        return ah_fail(c, AH_ENOTIMPL, "expr: unknown opcode %d", op);
      }
    }
  }
  if (stack.size() != 1) return ah_fail(c, AH_EINVALID, "expr: program leaves %zu values on the stack", stack.size());
  NodeVal r = stack.back();
  e->out_type = r.type;

  std::string src = kPrelude;
  src += "extern \"C\" __global__ void __launch_bounds__(256) ah_expr_kernel(Params p) {\n"
         "  const int lane = threadIdx.x & 63;\n"
         "  const long long nchunks = (p.n + 63) >> 6;\n"
         "  const long long nwaves = (long long)gridDim.x * 4;\n"
         "  // wave-uniform on purpose (readfirstlane): row0 / cnt / bitmap addresses then live in SGPRs and\n"
         "  // the validity words are fetched once per wave instead of once per lane\n"
         "  const long long wave = (long long)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));\n"
         "  bool ovf = false;\n";
  for (int l = 0; l < n_lits; l++) {
    std::string li = std::to_string(l);
    if (lit_types[l] == AH_BOOL_T) src += "  const bool l" + li + " = (p.lit[" + li + "] & 1) != 0;\n";
    else src += std::string("  const ") + CType(lit_types[l]) + " l" + li + " = *(const " + CType(lit_types[l]) + "*)&p.lit[" + li + "];\n";
    src += "  const bool kl" + li + " = p.lit_valid[" + li + "] != 0;\n";
  }
  src += "  for (long long chunk = wave; chunk < nchunks; chunk += nwaves * 4) {\n"
         "#pragma unroll\n"
         "    for (int u = 0; u < 4; u++) {\n"
         "      const long long ck = chunk + u * nwaves;\n"
         "      const bool act = ck < nchunks;\n"
         "      const long long row0 = ck << 6, row = row0 + lane;\n"
         "      const int cnt = !act ? 0 : (p.n - row0 >= 64 ? 64 : (int)(p.n - row0));\n"
         "      const bool inb = lane < cnt;\n"
         "      {\n";
  for (int i = 0; i < n_cols; i++) {
    if (!col_used[i]) continue;
    std::string ci = std::to_string(i);
    if (col_types[i] == AH_BOOL_T)
      src += "        const bool c" + ci + " = (load_bits64((const unsigned char*)p.col[" + ci + "], p.off[" + ci + "] + row0, cnt) >> lane) & 1;\n";
    else
      src += std::string("        const ") + CType(col_types[i]) + " c" + ci + " = inb ? ((const " + CType(col_types[i]) + "*)p.col[" + ci + "])[row] : (" +
             CType(col_types[i]) + ")0;\n";
    src += "        const bool kc" + ci + " = (load_bits64(p.valid[" + ci + "], p.off[" + ci + "] + row0, cnt) >> lane) & 1;\n";
  }
  src += body;
  if (r.type == AH_BOOL_T) {
    src += "        { u64 w = __ballot(inb && " + r.v + "); if (lane == 0 && cnt > 0) store_word((unsigned char*)p.out, row0, cnt, w); }\n";
  } else {
    src += std::string("        if (inb) ((") + CType(r.type) + "*)p.out)[row] = " + r.v + ";\n";
  }
  src += "        if (p.out_valid) { u64 w = __ballot(inb && " + r.ok + "); if (lane == 0 && cnt > 0) store_word(p.out_valid, row0, cnt, w); }\n";
  src += "      }\n    }\n  }\n";
  src += "  if (__any(ovf) && lane == 0) atomicOr(p.flag, 1u);\n}\n";
  e->source = std::move(src);
  return AH_OK;
}

int Compile(ah_ctx* c, ah_expr* e) {
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, e->source.c_str(), "ah_expr.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
    return ah_fail(c, AH_EHIP, "expr: hiprtcCreateProgram failed");
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17"};
  hiprtcResult r = hiprtcCompileProgram(prog, 4, opts);
  if (r != HIPRTC_SUCCESS) {
    size_t ls = 0;
    hiprtcGetProgramLogSize(prog, &ls);
    std::string log(ls, ' ');
    if (ls) hiprtcGetProgramLog(prog, &log[0]);
    hiprtcDestroyProgram(&prog);
    return ah_fail(c, AH_EHIP, "expr: hiprtc compile failed: %.400s", log.c_str());
  }
  size_t cs = 0;
  hiprtcGetCodeSize(prog, &cs);
  std::vector<char> code(cs);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  AH_HIP(c, hipModuleLoadData(&e->module, code.data()));
  AH_HIP(c, hipModuleGetFunction(&e->fn, e->module, "ah_expr_kernel"));
  return AH_OK;
}

using Cache = std::map<std::string, ah_expr*>;

}  // namespace

AH_EXPORT int ah_expr_compile(ah_ctx* c, const ah_expr_node* nodes, int n_nodes, const int* col_types, int n_cols,
                              const int* lit_types, int n_lits, ah_expr** out, int* out_type_host) {
  AH_ENTER(c);
  if (!out || !nodes || n_nodes <= 0) return ah_fail(c, AH_EINVALID, "expr: empty program");
  *out = nullptr;
  // signature = structure + types (literal VALUES are kernel arguments, not part of it)
  std::string sig;
  for (int i = 0; i < n_nodes; i++) sig += std::to_string(nodes[i].op) + ":" + std::to_string(nodes[i].arg) + ",";
  sig += "|";
  for (int i = 0; i < n_cols; i++) sig += std::to_string(col_types[i]) + ",";
  sig += "|";
  for (int i = 0; i < n_lits; i++) sig += std::to_string(lit_types[i]) + ",";
  if (!c->expr_cache) c->expr_cache = new Cache();
  Cache* cache = (Cache*)c->expr_cache;
  auto it = cache->find(sig);
  if (it != cache->end()) {
    *out = it->second;
    if (out_type_host) *out_type_host = it->second->out_type;
    return AH_OK;
  }
  ah_expr* e = new ah_expr();
  e->signature = sig;
  e->n_cols = n_cols;
  e->n_lits = n_lits;
  e->col_types.assign(col_types, col_types + n_cols);
  e->lit_types.assign(lit_types, lit_types + n_lits);
  int rc = Generate(c, nodes, n_nodes, col_types, n_cols, lit_types, n_lits, e);
  if (rc == AH_OK) rc = Compile(c, e);
  if (rc != AH_OK) { delete e; return rc; }
  (*cache)[sig] = e;
  *out = e;
  if (out_type_host) *out_type_host = e->out_type;
  return AH_OK;
}

AH_EXPORT const char* ah_expr_source(ah_expr* e) { return e ? e->source.c_str() : ""; }

AH_EXPORT int ah_expr_execute(ah_ctx* c, ah_expr* e, const void* const* col_values, const uint8_t* const* col_valid,
                              const int64_t* col_offsets, const void* lit_values_host, const int* lit_valid_host, int64_t len,
                              void* out_values, uint8_t* out_valid) {
  AH_ENTER(c);
  if (!e || !e->fn) return ah_fail(c, AH_EINVALID, "expr: not compiled");
  if (len < 0) return ah_fail(c, AH_EINVALID, "expr: negative length");
  if (len == 0) return AH_OK;
  if (!out_values) return ah_fail(c, AH_EINVALID, "expr: null output");
  KernelParams p;
  memset(&p, 0, sizeof(p));
  for (int i = 0; i < e->n_cols; i++) {
    p.col[i] = col_values[i];
    p.valid[i] = col_valid ? col_valid[i] : nullptr;
    p.off[i] = col_offsets ? col_offsets[i] : 0;
  }
  for (int i = 0; i < e->n_lits; i++) {
    memcpy(&p.lit[i], (const uint8_t*)lit_values_host + 8 * i, 8);
    p.lit_valid[i] = lit_valid_host ? lit_valid_host[i] : 1;
  }
  p.n = len;
  p.out = out_values;
  p.out_valid = out_valid;
  p.flag = (unsigned*)&c->dscalars[10];
  if (e->has_checked) AH_HIP(c, hipMemsetAsync(p.flag, 0, sizeof(unsigned), c->stream));
  int64_t nchunks = (len + 63) / 64;
  int64_t blocks = ah_ceil_div(nchunks, 4 * 4);  // 4 waves per block, 4 chunks per wave
  if (blocks > ((int64_t)1 << 30)) blocks = (int64_t)1 << 30;
  void* args[] = {&p};
  AH_HIP(c, hipModuleLaunchKernel(e->fn, (unsigned)blocks, 1, 1, 256, 1, 1, 0, c->stream, args, nullptr));
  if (e->has_checked) {
    AH_HIP(c, hipMemcpyAsync(c->pinned, p.flag, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    AH_HIP(c, hipStreamSynchronize(c->stream));
    if (*(volatile unsigned*)c->pinned & 1u) return ah_fail(c, AH_EOVERFLOW, "overflow");
  }
  return AH_OK;
}

AH_EXPORT int ah_expr_codegen(const ah_expr_node* nodes, int n_nodes, const int* col_types, int n_cols, const int* lit_types,
                              int n_lits, int do_compile, char* src_buf, size_t src_cap, char* err_buf, size_t err_cap,
                              int* out_type_host) {
  ah_ctx tmp;
  memset(&tmp, 0, sizeof(tmp));
  ah_expr e;
  int rc = (nodes && n_nodes > 0) ? Generate(&tmp, nodes, n_nodes, col_types, n_cols, lit_types, n_lits, &e)
                                  : ah_fail(&tmp, AH_EINVALID, "expr: empty program");
  if (rc == AH_OK && do_compile) {
    hiprtcProgram prog;
    const char* opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17"};
    if (hiprtcCreateProgram(&prog, e.source.c_str(), "ah_expr.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS ||
        hiprtcCompileProgram(prog, 4, opts) != HIPRTC_SUCCESS) {
      size_t ls = 0;
      hiprtcGetProgramLogSize(prog, &ls);
      std::string log(ls, ' ');
      if (ls) hiprtcGetProgramLog(prog, &log[0]);
      rc = ah_fail(&tmp, AH_EHIP, "expr: hiprtc compile failed: %.400s", log.c_str());
    }
    hiprtcDestroyProgram(&prog);
  }
  if (src_buf && src_cap) snprintf(src_buf, src_cap, "%s", e.source.c_str());
  if (err_buf && err_cap) snprintf(err_buf, err_cap, "%s", tmp.err);
  if (out_type_host) *out_type_host = e.out_type;
  return rc;
}

// called from ah_ctx_destroy
void ah_expr_cache_free(ah_ctx* c) {
  if (!c->expr_cache) return;
  Cache* cache = (Cache*)c->expr_cache;
  for (auto& kv : *cache) {
    if (kv.second->module) (void)hipModuleUnload(kv.second->module);
    delete kv.second;
  }
  delete cache;
  c->expr_cache = nullptr;
}
