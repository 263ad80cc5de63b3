"""The Substrait ExtendedExpression reader of the host layer (arrow_go_amd/host/substrait.cc) WITHOUT a device: what it understood of
a message, rendered by ahc_substrait_inspect.  Evaluation over device columns is tests/test_expressions.py (GPU); here: the function
mapping of the reference's default extension set (arrow/compute/exprs/extension_types.go, builders.go), literals of every primitive
type in both unsigned conventions, casts, the reference's refusals with their error class (arrow/compute/exprs/exec.go:465-700), plans
serialized by Arrow C++ (pyarrow.substrait — a producer that shares no code with this reader), and — because the reader takes bytes
from outside the process — every truncation and thousands of mutations of those messages under AddressSanitizer (tests/wire_fuzz.cc).
"""
import os
import struct
import subprocess

import numpy as np
import pyarrow as pa
import pyarrow.compute as pc
import pytest

from arrow_go_amd import compute as ac
from tests.substrait_builder import SB, ARITH, CMP, BOOLF, _ld, _vi

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def expr_of(text):
    """the single expression of an inspect line"""
    schema, *exprs = text.split("|")
    assert len(exprs) == 1, text
    name, _, body = exprs[0].partition("=")
    assert name == "out"
    return body


def test_function_mapping_of_the_default_extension_set():
    """substrait name (+ options) → the registry's function, as exprs/builders.go + exec.go:630-700 map them"""
    b = SB([("a", "i32"), ("b", "i32"), ("p", "bool"), ("q", "bool")])
    f0, f1, p, q = b.field(0), b.field(1), b.field(2), b.field(3)
    for uri, name, args, want in [
        (CMP, "lt", (f0, f1), "less($0, $1)"), (CMP, "gt", (f0, f1), "greater($0, $1)"),
        (CMP, "lte", (f0, f1), "less_equal($0, $1)"), (CMP, "gte", (f0, f1), "greater_equal($0, $1)"),
        (CMP, "equal", (f0, f1), "equal($0, $1)"), (CMP, "not_equal", (f0, f1), "not_equal($0, $1)"),
        (CMP, "is_null", (f0,), "is_null($0)"), (CMP, "is_not_null", (f0,), "is_not_null($0)"),
        (BOOLF, "and", (p, q), "and_kleene($2, $3)"), (BOOLF, "or", (p, q), "or_kleene($2, $3)"), (BOOLF, "not", (p,), "invert($2)"),
        (ARITH, "add", (f0, f1), "add_unchecked($0, $1)"), (ARITH, "subtract", (f0, f1), "subtract_unchecked($0, $1)"),
        (ARITH, "multiply", (f0, f1), "multiply_unchecked($0, $1)"),
    ]:
        assert expr_of(ac.inspect_substrait(b.build(b.call(uri, name, *args)))) == want, name
    # the overflow option (exprs/types.go:172-192): SILENT → _unchecked, ERROR → the checked kernel, the first IMPLEMENTED preference wins
    for prefs, want in ((["SILENT"], "add_unchecked"), (["ERROR"], "add"), (["SATURATE", "ERROR"], "add"), (["SATURATE", "SILENT"], "add_unchecked")):
        e = b.call(ARITH, "add", f0, f1, options={"overflow": prefs}, out_type="i32")
        assert expr_of(ac.inspect_substrait(b.build(e))) == f"{want}($0, $1)", prefs
    e = b.call(ARITH, "add", f0, f1, options={"overflow": ["SATURATE"]}, out_type="i32")
    assert "!not implemented" in ac.inspect_substrait(b.build(e)) and "SATURATE" in ac.inspect_substrait(b.build(e))


def test_base_schema_and_literals():
    """every primitive literal (exec.go:118-330 literalToDatum), little-endian payload rendered most significant byte first; unsigned
    integers as arrow-go writes them (a type VARIATION of the signed type, exprs/types.go:58-78)"""
    b = SB([("a", "i8"), ("b", "i16"), ("c", "i32"), ("d", "i64"), ("e", "u8"), ("f", "u16"), ("g", "u32"), ("h", "u64"), ("x", "fp32"), ("y", "fp64"),
            ("p", "bool"), ("s", "string")])
    text = ac.inspect_substrait(b.build(b.call(CMP, "equal", b.field(0), b.lit("i8", -1))))
    assert text.split("|")[0] == "a:int8,b:int16,c:int32,d:int64,e:uint8,f:uint16,g:uint32,h:uint64,x:float32,y:float64,p:bool,s:?string"
    cases = [("i8", -1, "int8(ff)"), ("i8", 127, "int8(7f)"), ("i16", -2, "int16(fffe)"), ("i32", -(2**31), "int32(80000000)"),
             ("i64", -5, "int64(fffffffffffffffb)"), ("i64", 2**62, "int64(4000000000000000)"),
             ("u8", 255, "uint8(ff)"), ("u16", 65535, "uint16(ffff)"), ("u32", 4_000_000_000, "uint32(ee6b2800)"), ("u64", 2**63 + 1, "uint64(8000000000000001)"),
             ("bool", True, "bool(01)"), ("bool", False, "bool(00)"),
             ("fp32", 1.5, "float32(" + struct.pack(">f", 1.5).hex() + ")"), ("fp64", -0.25, "float64(" + struct.pack(">d", -0.25).hex() + ")"),
             ("i32", None, "int32(null)"), ("u64", None, "uint64(null)"), ("fp64", None, "float64(null)"), ("bool", None, "bool(null)")]
    for t, v, want in cases:
        got = expr_of(ac.inspect_substrait(b.build(b.call(CMP, "equal", b.field(2), b.lit(t, v)))))
        assert got == f"equal($2, {want})", (t, v)
    # a field of a type this layer does not carry is an error only where it is referenced (test_expressions: at execution)
    assert "s:?string" in ac.inspect_substrait(b.build(b.call(CMP, "is_null", b.field(11))))


def test_nested_schema_names_and_the_last_oneof_member():
    """NamedStruct.names is depth-first with the fields of nested structs included (substrait/type.proto): a schema with a struct
    column (also inside a list / a map) keeps its other columns addressable — the reference executes such a schema as long as the
    expression does not touch the struct.  And rex_type is a oneof: of two members on the wire the last one counts."""
    inner = ("struct", [("z", "i64"), ("w", ("list", ("struct", [("q", "i8")])))])
    b = SB([("a", "i32"), ("s", ("struct", [("x", "i32"), ("y", inner)])), ("m", ("map", "string", ("struct", [("k", "fp64")]))), ("b", "i64")])
    text = ac.inspect_substrait(b.build(b.call(CMP, "gt", b.field(3), b.lit("i64", 7))))
    assert text.split("|")[0] == "a:int32,s:?struct,m:?type #28,b:int64", text
    assert expr_of(text) == "greater($3, int64(0000000000000007))"
    # a name too few / too many is still refused, with the count of nested fields in the message
    good = b.build(b.call(CMP, "is_null", b.field(0)))
    with pytest.raises(ac.ErrInvalid, match="names for .* nested struct fields"):
        ac.inspect_substrait(good + _ld(4, _ld(1, b"stray")))          # a name that no column accounts for
    # two members of the oneof: literal, then field reference → the field reference
    flat = SB([("a", "i32"), ("b", "i32")])
    both = flat.lit("i32", 5) + flat.field(1)
    assert expr_of(ac.inspect_substrait(flat.build(flat.call(CMP, "equal", flat.field(0), both)))) == "equal($0, $1)"
    both = flat.field(1) + flat.lit("i32", 5)
    assert expr_of(ac.inspect_substrait(flat.build(flat.call(CMP, "equal", flat.field(0), both)))) == "equal($0, int32(00000005))"


def test_casts():
    b = SB([("a", "i32"), ("x", "fp64")])
    assert expr_of(ac.inspect_substrait(b.build(b.cast("i64", b.field(0))))) == "cast($0 -> int64 unsafe)"        # THROW_EXCEPTION → UnsafeCastOptions (exec.go:571)
    assert expr_of(ac.inspect_substrait(b.build(b.call(ARITH, "multiply", b.cast("fp64", b.field(0)), b.field(1), out_type="fp64")))) == \
        "multiply_unchecked(cast($0 -> float64 unsafe), $1)"
    assert "!invalid: cast behavior unspecified" in ac.inspect_substrait(b.build(b.cast("i64", b.field(0), behavior=0)))      # :573
    assert "!not implemented: cast behavior return nil" in ac.inspect_substrait(b.build(b.cast("i64", b.field(0), behavior=1)))   # :575
    assert "!not implemented" in ac.inspect_substrait(b.build(b.cast("string", b.field(0))))


def test_what_the_reference_refuses_is_refused_with_its_error_class():
    b = SB([("a", "i32")])
    ok = b.call(CMP, "equal", b.field(0), b.lit("i32", 2))
    assert "!not implemented: measures not implemented" in ac.inspect_substrait(b.build(ok, measure=True))             # exec.go:477
    assert "!not implemented" in ac.inspect_substrait(b.build(b.call(ARITH, "modulus", b.field(0), b.field(0), out_type="i32")))   # :606-609
    assert "!not implemented" in ac.inspect_substrait(b.build(b.field(0, child=1)))                                     # nested reference
    for number, what in ((6, "if-then"), (7, "switch"), (8, "singular-or-list"), (9, "multi-or-list"), (12, "subqueries"), (13, "nested")):
        assert f"!not implemented: substrait: {what}" in ac.inspect_substrait(b.build(_ld(number, b"")))                # :699-705
    assert "!invalid" in ac.inspect_substrait(b.build(_ld(5, b"")))                                                     # a window function: "non-scalar", :545-548
    # an unknown function anchor (declared nowhere)
    bad = _ld(3, _vi(1, 99) + _ld(3, b.typ("bool")) + _ld(4, _ld(3, b.field(0))))
    assert "!" in ac.inspect_substrait(b.build(bad))
    # nesting beyond any plan a producer writes: refused, not recursed into
    deep = b.field(0)
    for _ in range(400):
        deep = b.call(BOOLF, "not", deep)
    assert "nested too deeply" in ac.inspect_substrait(b.build(deep))
    # the message as a whole
    with pytest.raises(ac.ErrInvalid, match="malformed"):
        ac.inspect_substrait(b.build(ok)[:-3])
    with pytest.raises(ac.ErrInvalid, match="names for"):
        ac.inspect_substrait(b.build(ok) + _ld(4, _ld(1, b"extra")))      # a second NamedStruct with a name and no type
    assert ac.inspect_substrait(b"") == ""                                # an empty message is an empty plan: "no referred expression" at execution


def _arrow_cpp_plans():
    import pyarrow.substrait as ps
    schema = pa.schema([("i32", pa.int32()), ("j32", pa.int32()), ("i64", pa.int64()), ("u32", pa.uint32()), ("f64", pa.float64()), ("g64", pa.float64()),
                        ("b", pa.bool_()), ("c", pa.bool_())])
    F = pc.field
    exprs = [
        (pc.add(F("i32"), F("j32")), "add_unchecked($0, $1)"),
        (pc.add_checked(F("i32"), F("j32")), "add($0, $1)"),
        (pc.greater(pc.multiply(pc.add(F("f64"), F("g64")), F("g64")), pc.scalar(0.25)),
         "greater(multiply_unchecked(add_unchecked($4, $5), $5), float64(3fd0000000000000))"),
        (pc.and_kleene(pc.less(F("i64"), pc.scalar(pa.scalar(0, pa.int64()))), pc.invert(F("b"))), "and_kleene(less($2, int64(0000000000000000)), invert($6))"),
        (pc.or_kleene(F("b"), pc.greater_equal(F("f64"), F("g64"))), "or_kleene($6, greater_equal($4, $5))"),
        (pc.less_equal(F("u32"), pc.scalar(pa.scalar(3_000_000_000, pa.uint32()))), "less_equal($3, uint32(b2d05e00))"),   # Arrow C++: a user-defined literal
        (pc.subtract(F("i64"), pc.scalar(pa.scalar(-5, pa.int64()))), "subtract_unchecked($2, int64(fffffffffffffffb))"),
        (pc.is_null(F("f64")), "is_null($4)"),
        (pc.not_equal(F("c"), F("b")), "not_equal($7, $6)"),
        (pc.multiply(F("i32").cast(pa.float64(), safe=False), F("f64")), "multiply_unchecked(cast($0 -> float64 unsafe), $4)"),
        (pc.equal(F("i32"), pc.scalar(pa.scalar(None, pa.int32()))), "equal($0, int32(null))"),
    ]
    return schema, [(bytes(ps.serialize_expressions([e], ["out"], schema)), want) for e, want in exprs]


def test_plans_written_by_arrow_cpp():
    """Arrow C++ writes unsigned types and literals as USER-DEFINED types of its own extension (arrow-go: type variations), function
    options as enum arguments or options, 64-bit negatives as ten-byte varints: all read to the same trees"""
    schema, plans = _arrow_cpp_plans()
    for msg, want in plans:
        text = ac.inspect_substrait(msg)
        assert text.split("|")[0] == "i32:int32,j32:int32,i64:int64,u32:uint32,f64:float64,g64:float64,b:bool,c:bool"
        assert expr_of(text) == want


def _harness():
    """tests/wire_fuzz.cc built with AddressSanitizer + UBSan from the host layer's sources (rebuilt when any of them is newer)"""
    build = os.path.join(HERE, "build")
    os.makedirs(build, exist_ok=True)
    exe = os.path.join(build, "wire_fuzz")
    host = os.path.join(ROOT, "arrow_go_amd", "host")
    srcs = [os.path.join(HERE, "wire_fuzz.cc")] + [os.path.join(host, f) for f in ("core.cc", "kernels.cc", "expression.cc", "substrait.cc", "ipc.cc", "hoststream.cc", "capi.cc")]
    deps = srcs + [os.path.join(host, "arrowhip_compute.h"), os.path.join(host, "ipc.h"), os.path.join(ROOT, "include", "arrowhip_compute.h")]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        lib = os.path.join(ROOT, "arrow_go_amd")
        cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-I" + host, "-o", exe] + srcs + \
              ["-L" + lib, "-larrowhip", "-ldl", "-Wl,-rpath," + lib]
        subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def _run_harness(tmp_path, kind, msgs, mutations, seed):
    corpus = tmp_path / f"{kind}.corpus"
    with open(corpus, "wb") as f:
        for m in msgs:
            f.write(struct.pack("<I", len(m)) + m)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([_harness(), str(corpus), kind, str(mutations), str(seed)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    nmsg, ntrunc, nmut, accepted = map(int, r.stdout.split())
    assert nmsg == len(msgs) and accepted >= nmsg      # every untouched message is among the truncations
    return ntrunc, nmut, accepted


def test_truncated_and_mutated_messages_under_asan(tmp_path):
    """every prefix of every message and 3000 seeded mutations each (bit flips, random bytes, 0xFF runs that keep a varint going, a byte
    removed or inserted), plus plain noise: an error or a plan, never a read outside the buffer (the copy the reader sees is an exact-size
    heap block: ASan's red zone starts at its last byte), never undefined behaviour, never unbounded recursion"""
    b = SB([("a", "i32"), ("u", "u32"), ("x", "fp64"), ("p", "bool")])
    ours = [b.build(b.call(CMP, "lt", b.field(0), b.lit("i32", -2))),
            b.build(b.call(BOOLF, "and", b.call(CMP, "gte", b.field(2), b.lit("fp64", 0.5)), b.call(BOOLF, "not", b.call(CMP, "equal", b.field(1), b.lit("u32", 4_000_000_000))))),
            b.build(b.call(ARITH, "add", b.field(0), b.lit("i32", None), options={"overflow": ["SATURATE", "ERROR"]}, out_type="i32")),
            b.build(b.cast("i64", b.field(0))), b.build(b.field(0, child=1)), b.build(), b""]
    _, plans = _arrow_cpp_plans()
    ntrunc, nmut, accepted = _run_harness(tmp_path, "substrait", ours + [m for m, _ in plans], 3000, 20260926)
    assert ntrunc > 3000 and nmut > 50_000
    assert accepted < ntrunc + nmut        # (and most damaged messages are refused)


def test_ipc_streams_truncated_and_mutated_under_asan(tmp_path):
    """the same for the other reader of outside bytes: Arrow IPC streams (host/ipc.cc through ahc_ipc_inspect) written by Arrow C++ —
    plain, dictionary-encoded and temporal columns, several batches, an empty batch"""
    rng = np.random.default_rng(5)
    n = 300
    t1 = pa.table({"i": pa.array(rng.integers(-9, 9, n), pa.int32()), "f": pa.array(rng.random(n), mask=rng.random(n) < 0.2),
                   "b": pa.array(rng.random(n) < 0.5), "s": pa.array([f"k{v}" for v in rng.integers(0, 5, n)]),
                   "t": pa.array(rng.integers(0, 10**12, n), pa.timestamp("us"))})
    t2 = pa.table({"d": pa.array([f"v{v}" for v in rng.integers(0, 7, n)]).dictionary_encode(), "u": pa.array(rng.integers(0, 2**60, n), pa.uint64())})
    msgs = []
    for t, chunk in ((t1, 100), (t2, 150), (t1.slice(0, 0), 10)):
        sink = pa.BufferOutputStream()
        with pa.ipc.new_stream(sink, t.schema) as w:
            for batch in t.to_batches(max_chunksize=chunk):
                w.write_batch(batch)
        msgs.append(sink.getvalue().to_pybytes())
    ntrunc, nmut, accepted = _run_harness(tmp_path, "ipc", msgs, 1500, 7)
    assert nmut > 4000
