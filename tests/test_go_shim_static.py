"""go/arrowhip/*.go checked without a Go toolchain (tests/go_static.py): duplicate declarations, brackets, unused imports, and every
C.ah_* call's argument count and argument types against include/arrowhip.h.  The checker itself is tested by mutation: each class of
error is planted in a copy of the package and must be reported."""
import os
import re
import shutil

import pytest

from tests import go_static as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# entry points the Go package deliberately leaves unbound, each with its reason (also stated in go/arrowhip/comm.go)
DELIBERATELY_UNBOUND = {
    "ah_comm_init_transport": "takes a table of C callbacks; only test rigs that put several ranks on one GPU use it",
}


@pytest.fixture(scope="module")
def chk():
    return G.check()


def test_package_is_clean(chk):
    assert not chk.errors, "\n".join(chk.errors)


def test_every_c_call_argument_is_type_checked(chk):
    unverified = [f"{c.file}:{c.line} {c.cname}: {u}" for c in chk.calls for u in c.unverified]
    assert not unverified, "\n".join(unverified)
    assert len(chk.calls) >= 100 and sum(c.nargs for c in chk.calls) >= 600


def test_every_header_entry_is_bound_or_listed(chk):
    assert set(chk.unbound()) == set(DELIBERATELY_UNBOUND), chk.unbound()
    assert len(chk.protos) >= 104


def test_checked_arithmetic_is_wired(chk):
    """compute.Add / Subtract / Multiply default to the CHECKED kernels (compute/arithmetic.go:635-636, 1095-1105): the shim must call
    ah_arithmetic_checked and register it under the stock names in SwapInPlace and under *_hip in Register"""
    callers = {c.func for c in chk.calls if c.cname == "ah_arithmetic_checked"}
    assert callers == {"Context.ArithmeticChecked"}
    src = open(os.path.join(G.GO_DIR, "register.go")).read()
    # … "sub" too: compute.Subtract itself calls impl(ctx, "sub", …) (+ "_unchecked" with NoCheckOverflow), arithmetic.go:679-682, 1115-1117
    assert re.search(r'"add": opAdd, "subtract": opSub, "sub": opSub, "multiply": opMul', src)
    assert re.search(r'"add_unchecked": opAdd, "subtract_unchecked": opSub, "sub_unchecked": opSub, "multiply_unchecked": opMul', src)
    assert re.search(r'"add_hip": opAdd, "subtract_hip": opSub, "multiply_hip": opMul', src)
    assert "checkedExec(x, dt.ID(), width(dt), op)" in src
    # AH_EOVERFLOW → arrow.ErrInvalid (the reference's errOverflow is fmt.Errorf("%w: overflow", arrow.ErrInvalid))
    assert re.search(r"case C\.AH_EINVALID, C\.AH_EOVERFLOW:\s*return fmt\.Errorf\(\"%w: %s\", arrow\.ErrInvalid", open(os.path.join(G.GO_DIR, "arrowhip.go")).read())


def test_arrow_go_symbols_and_members_exist(chk):
    """every `pkg.Name` the shim writes is exported by that arrow-go package, every field / method it selects on a value of an arrow-go
    type exists on that type (embedded types and aliases followed) — the round-4 package called arrow.IsNumeric, which arrow-go does not
    have.  The list of exported names is a committed fixture (scripts/gen_go_exports.py reads the reference's sources); where the
    reference is present the fixture must be current."""
    assert chk.ext, "tests/golden/go_reference_exports.json is missing"
    assert chk.ext_checked >= 250 and chk.local_calls_checked >= 150 and chk.local_args_typed >= 150   # (calls of the package's own functions / methods: counts and types)
    if os.path.isdir("/root/reference/arrow"):
        import subprocess, sys
        assert subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_go_exports.py"), "--check"]).returncode == 0, \
            "regenerate: python scripts/gen_go_exports.py"


def test_vector_swap_dispatches_on_the_runtime_width():
    """array_take / array_filter hold ONE kernel for all fixed-width primitives (exec.Primitive(), vector_selection.go:2340-2360): the
    wrapper must pick the HIP kernel by the width of the column it is handed, not by the first type that matched the signature"""
    src = open(os.path.join(G.GO_DIR, "register.go")).read()
    assert "hip := map[int]exec.ArrayKernelExec{1: mk(1), 2: mk(2), 4: mk(4), 8: mk(8)}" in src
    assert "hip[width(t)]" in src


def test_flipped_comparisons_are_swapped_themselves():
    """makeFlippedCompare (scalar_compare.go:84-99) copies greater's kernels and captures their ExecFn VALUE (flippedData.unflippedExec)
    when the registry is built: swapping "greater" in place does not reach "less" / "less_equal" — SwapInPlace must replace their ExecFn
    too (the same comparison, operands exchanged)"""
    src = open(os.path.join(G.GO_DIR, "register.go")).read()
    assert re.search(r'map\[string\]int\{"less": cmpGT, "less_equal": cmpGE\}', src)
    assert "compareExec(x, dt.ID(), width(dt), c, true)" in src
    assert "swapped for free" not in src


def test_math_shaped_like_arrow_math(chk):
    """arrow/math/float64.go:25-39: Float64Funcs.Sum(*array.Float64) float64 — same shape on the GPU context"""
    sigs = {(f.recv_type, f.name): (f.params, f.results) for f in chk.pkg.funcs}
    for recv, arr, res in (("Float64Funcs", "*array.Float64", "float64"), ("Int64Funcs", "*array.Int64", "int64"), ("Uint64Funcs", "*array.Uint64", "uint64")):
        params, results = sigs[(recv, "Sum")]
        assert list(params.values()) == [arr] and results == [res]
    assert {"Float64", "Int64", "Uint64"} <= set(chk.pkg.structs["Math"])


def test_binding_table_in_integration_md_is_current(chk):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- go-binding-table:begin -->\n(.*?)<!-- go-binding-table:end -->", text, flags=re.S)
    assert m, "INTEGRATION.md lost its generated binding table (python tests/go_static.py --table)"
    assert m.group(1) == G.binding_table(chk), "regenerate: python scripts/update_binding_table.py"


# ---- the checker must catch what it claims to catch ----------------------------------------------------------------------------
MUTATIONS = [
    ("duplicate type", "comm.go", "func UniqueID()", "type Comm struct{ a int }\n\nfunc UniqueID()", r"Comm redeclared"),
    ("duplicate method", "comm.go", "func UniqueID()", "func (c *Comm) Close() {}\n\nfunc UniqueID()", r"Comm\.Close redeclared"),
    ("duplicate func across files", "math.go", "func NewMath(", "func boolInt(b bool) C.int { return 0 }\n\nfunc NewMath(", r"boolInt redeclared"),
    ("missing brace", "graph.go", "func (g *Graph) Close() {", "func (g *Graph) Close() {{", r"never closed|unbalanced"),
    ("argument dropped", "arrowhip.go", "C.ah_sum_int64(x.c, (*C.int64_t)(values), C.size_t(n), &r)", "C.ah_sum_int64(x.c, (*C.int64_t)(values), &r)", r"ah_sum_int64 called with 3 arguments, the header declares 4"),
    ("wrong pointee", "arrowhip.go", "C.ah_sum_int64(x.c, (*C.int64_t)(values)", "C.ah_sum_int64(x.c, (*C.uint64_t)(values)", r"ah_sum_int64 argument 2 .* wants int64_t\*"),
    ("scalar where a pointer goes", "arrowhip.go", "C.size_t(n), &r))\n\treturn int64(r), err", "C.size_t(n), r))\n\treturn int64(r), err", r"argument 4 `r` has type C\.int64_t, the header wants int64_t\*"),
    ("Go int passed raw", "extra.go", "C.ah_event_record(x.c, C.int(slot))", "C.ah_event_record(x.c, slot)", r"ah_event_record argument 2 `slot` has type int"),
    ("wrong scalar width", "extra.go", "dst, C.int(byteValue), C.size_t(nbytes)", "dst, C.int64_t(byteValue), C.size_t(nbytes)", r"ah_memset_async argument 3 .* wants int\b"),
    ("unsafe.Pointer for a typed pointer", "math.go", "(*C.uint64_t)(values), C.size_t(n), &r", "values, C.size_t(n), &r", r"ah_sum_uint64 argument 2 `values` has type unsafe\.Pointer"),
    ("unknown entry point", "extra.go", "C.ah_ingest_wait(i.g)", "C.ah_ingest_waitall(i.g)", r"C\.ah_ingest_waitall is not declared"),
    ("unknown constant", "arrowhip.go", "C.AH_SHIFT_DIVIDE", "C.AH_SHIFT_DIVIDED", r"C\.AH_SHIFT_DIVIDED is not declared"),
    ("unknown constant in a package-level table", "expr_lower.go", '"invert": C.AH_X_INVERT', '"invert": C.AH_X_NOT', r"C\.AH_X_NOT is not declared"),
    ("unused import", "math.go", '"unsafe"\n', '"unsafe"\n\t"fmt"\n', r'"fmt" imported and not used'),
    ("unknown field", "comm.go", "C.ah_comm_destroy(c.m)", "C.ah_comm_destroy(c.comm)", r"type Comm has no field or method comm"),
    ("unknown method", "register.go", "g, err := x.NewIngest(0, 0)", "g, err := x.MakeIngest(0, 0)", r"type Context has no field or method MakeIngest"),
    ("build tag lost", "graph.go", "//go:build hip\n", "", r"no '//go:build hip' line"),
    ("own method called with an argument missing", "register.go", "return finishVector(ctx, out, values.Type, n, nulls, w, do, dvo)", "return finishVector(ctx, out, values.Type, n, nulls, w, do)",
     r"finishVector called with 7 arguments, declared with 8"),
    ("two variables for a call that returns three", "register.go", "ndict, nullID, err := x.HashU64Encode(", "ndict, err := x.HashU64Encode(", r"assignment mismatch: 2 variables but the call returns 3 values"),
    ("int64 passed where the package's own function takes an int", "register.go", "return finishVector(ctx, out, values.Type, n, nulls, w, do, dvo)", "return finishVector(ctx, out, values.Type, n, nulls, nulls, do, dvo)",
     r"finishVector argument 6 `nulls` has type int64, the parameter is int"),
    ("a return with a value missing", "arrowhip.go", "\treturn int64(r), err\n}", "\treturn int64(r)\n}", r"returns 1 values here, its signature has 2"),
    ("package used but not imported", "comm.go", '\t"fmt"\n', "", r"undefined: fmt"),
    ("local declared and not used", "math.go", "func NewMath(", "func unusedLocal() int {\n\tleft, right := 1, 2\n\treturn left\n}\n\nfunc NewMath(", r"right declared and not used"),
    # against the arrow-go packages the shim imports (tests/golden/go_reference_exports.json)
    ("symbol arrow-go does not export", "register.go", "arrow.IsInteger(id)", "arrow.IsNumeric(id)", r"arrow\.IsNumeric is not exported by github.com/apache/arrow-go/v18/arrow"),
    ("member an arrow-go type does not have", "register.go", "out.Buffers[1].WrapBuffer(data)", "out.Buffers[1].Wrap(data)", r"type exec\.BufferSpan has no field or method Wrap"),
    ("field of an arrow-go struct misspelt", "register.go", "values, indices := &batch.Values[0].Array, &batch.Values[1].Array", "values, indices := &batch.Values[0].Arr, &batch.Values[1].Array", r"type exec\.ExecValue has no field or method Arr"),
    ("arrow-go function called with the wrong number of arguments", "register.go", "compute.NewChildRegistry(compute.GetFunctionRegistry())", "compute.NewChildRegistry(compute.GetFunctionRegistry(), nil)",
     r"compute\.NewChildRegistry called with 2 arguments, arrow-go declares 1"),
    ("arrow-go method called with the wrong number of arguments", "register.go", "out.Buffers[1].WrapBuffer(data)", "out.Buffers[1].WrapBuffer(data, true)",
     r"WrapBuffer called with 2 arguments, arrow-go declares 1"),
    ("method of an embedded arrow-go type misspelt", "math.go", "a.Len()", "a.Length()", r"type array\.Float64 has no field or method Length"),
]


@pytest.mark.parametrize("label,fname,old,new,expect", MUTATIONS, ids=[m[0] for m in MUTATIONS])
def test_checker_catches(tmp_path, label, fname, old, new, expect):
    d = tmp_path / "arrowhip"
    shutil.copytree(G.GO_DIR, d)
    src = (d / fname).read_text()
    assert src.count(old) >= 1, f"mutation anchor for '{label}' is gone from {fname}"
    (d / fname).write_text(src.replace(old, new, 1))
    c = G.check(str(d))
    assert any(re.search(expect, e) for e in c.errors), (label, c.errors[:5])
