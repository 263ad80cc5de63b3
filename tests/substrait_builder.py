"""A tiny protobuf WRITER for substrait.ExtendedExpression messages, shaped like the builders the reference's tests use (substrait-go's
ExprBuilder, arrow/compute/exprs/exec_test.go).  Test infrastructure: the reader under test is arrow_go_amd/host/substrait.cc; this
writer shares no code with it.  Used by tests/test_expressions.py (GPU: evaluation) and tests/test_substrait_reader.py (CPU: what the
reader understood, truncated and mutated inputs)."""
# Field numbers: substrait-io/substrait proto/substrait/{extended_expression,algebra,type}.proto.
def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _vi(field, v):
    return _varint(field << 3) + _varint(v)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + bytes(payload)


def _f64(field, x):
    import struct
    return _varint((field << 3) | 1) + struct.pack("<d", x)


def _f32(field, x):
    import struct
    return _varint((field << 3) | 5) + struct.pack("<f", x)


ARITH, CMP, BOOLF = ("https://github.com/substrait-io/substrait/blob/main/extensions/functions_arithmetic.yaml",
                     "https://github.com/substrait-io/substrait/blob/main/extensions/functions_comparison.yaml",
                     "https://github.com/substrait-io/substrait/blob/main/extensions/functions_boolean.yaml")
_URI_ANCHOR = {ARITH: 1, CMP: 2, BOOLF: 3}
_TYPE_FIELD = {"bool": 1, "i8": 2, "i16": 3, "i32": 5, "i64": 7, "fp32": 10, "fp64": 11, "string": 12, "binary": 13, "date": 16}
_UNSIGNED = {"u8": ("i8", 1), "u16": ("i16", 2), "u32": ("i32", 3), "u64": ("i64", 4)}   # arrow-go: a type VARIATION of the signed type (exprs/types.go:58-78)


class SB:
    """builds one substrait.ExtendedExpression the way the reference's tests do with substrait-go's builders"""

    def __init__(self, schema):
        self.schema = schema            # [(name, type name)]
        self.funcs = {}                 # (uri, name) → anchor

    def typ(self, t):
        if isinstance(t, tuple):        # ("struct", [(name, type) …]) / ("list", type) / ("map", key type, value type): nested types
            if t[0] == "struct":
                return _ld(25, b"".join(_ld(1, self.typ(ft)) for _, ft in t[1]) + _vi(3, 1))
            if t[0] == "list":
                return _ld(27, _ld(1, self.typ(t[1])) + _vi(3, 1))
            return _ld(28, _ld(1, self.typ(t[1])) + _ld(2, self.typ(t[2])) + _vi(4, 1))
        if t in _UNSIGNED:
            base, var = _UNSIGNED[t]
            return _ld(_TYPE_FIELD[base], _vi(1, var) + _vi(2, 1))
        return _ld(_TYPE_FIELD[t], _vi(2, 1))

    def field(self, i, child=None):
        seg = _vi(1, i) if i else b""
        if child is not None:
            seg += _ld(2, _ld(2, _vi(1, child) if child else b""))
        return _ld(2, _ld(1, _ld(2, seg)) + _ld(4, b""))            # selection {direct_reference {struct_field}, root_reference}

    def lit(self, t, v):
        if v is None:
            return _ld(1, _ld(29, self.typ(t)) + _vi(50, 1))
        if t in _UNSIGNED:
            base, var = _UNSIGNED[t]
            return _ld(1, _vi(_TYPE_FIELD[base], v) + _vi(51, var))
        body = {"bool": lambda: _vi(1, int(v)), "fp32": lambda: _f32(10, v), "fp64": lambda: _f64(11, v)}.get(t, lambda: _vi(_TYPE_FIELD[t], v))()
        return _ld(1, body)

    def call(self, uri, name, *args, options=None, out_type="bool"):
        anchor = self.funcs.setdefault((uri, name), len(self.funcs) + 1)
        body = _vi(1, anchor) + _ld(3, self.typ(out_type))
        for a in args:
            body += _ld(4, _ld(3, a))
        for k, prefs in (options or {}).items():
            body += _ld(5, _ld(1, k.encode()) + b"".join(_ld(2, p.encode()) for p in prefs))
        return _ld(3, body)

    def cast(self, t, inp, behavior=2):
        return _ld(11, _ld(1, self.typ(t)) + _ld(2, inp) + (_vi(3, behavior) if behavior else b""))

    def build(self, *exprs, measure=False):
        msg = b""
        for uri, anchor in _URI_ANCHOR.items():
            msg += _ld(1, _vi(1, anchor) + _ld(2, uri.encode()))
        for name, (base, var) in _UNSIGNED.items():
            msg += _ld(2, _ld(2, _vi(1, 1) + _vi(2, var) + _ld(3, name.encode())))            # extension_type_variation
        for (uri, name), anchor in self.funcs.items():
            msg += _ld(2, _ld(3, _vi(1, _URI_ANCHOR[uri]) + _vi(2, anchor) + _ld(3, name.encode())))
        for e in exprs:
            msg += _ld(3, (_ld(2, b"\x08\x01") if measure else _ld(1, e)) + _ld(3, b"out"))
        def dfs(name, t):               # NamedStruct.names: depth-first, the fields of nested structs included
            out = [name]
            if isinstance(t, tuple):
                if t[0] == "struct":
                    for fn, ft in t[1]:
                        out += dfs(fn, ft)
                else:
                    for sub in t[1:]:
                        out += dfs(None, sub)[1:]
            return out
        names = b"".join(_ld(1, n.encode()) for col, t in self.schema for n in dfs(col, t))
        types = b"".join(_ld(1, self.typ(t)) for _, t in self.schema)
        return msg + _ld(4, names + _ld(2, types + _vi(3, 2)))


