"""Float64 Sum's accumulator (csrc/ah_ddsum.h) on the CPU: the header compiled for the host, its Python restatement
(tests/ddx_model.py) and the oracle's fixed-point superaccumulator (orc_sum_float64_xreal) must agree on every class of input —
ordinary data, ±inf, NaN, finite overflow, overflow in an intermediate sum only, rows around the 2^960 class boundary, subnormals —
and the oracle must agree with BOTH reference orders (oracle/_ref: the AVX2 machine code and the strict-sequential C) wherever
those agree with each other (arrow/math/float64.go:41-47, _lib/float64.c:20-26)."""
import ctypes as C
import math
import os
import subprocess
import tempfile

import numpy as np
import pytest

from tests import ddx_model as M
from tests import oracle_lib as OL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
inf, nan = math.inf, math.nan


@pytest.fixture(scope="module")
def host():
    d = tempfile.mkdtemp(prefix="ddx_")
    so = os.path.join(d, "libddx.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", os.path.join(ROOT, "tests", "ddx_harness.cc"), "-o", so])
    lib = C.CDLL(so)
    lib.ddx_sum.restype = C.c_double
    lib.ddx_sum.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    lib.ddx_merge_result.restype = C.c_double
    lib.ddx_merge_result.argtypes = [C.c_void_p, C.c_size_t]
    return lib


def host_sum(lib, a, lanes):
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = np.zeros(4)
    r = lib.ddx_sum(a.ctypes.data, a.size, lanes, out.ctypes.data)
    return r, out


def same(a, b):
    return (math.isnan(a) and math.isnan(b)) or a == b


CASES = {
    "inf among finite": [1.0, inf, 2.0],
    "both infinities": [inf, -inf],
    "finite overflow": [1e308] * 3,
    "negative finite overflow": [-1e308] * 3,
    "-inf and nan": [-inf, nan],
    "nan alone": [nan],
    "-inf alone": [3.0, -inf, 5.0],
    "exactly DBL_MAX stays finite": [1.7976931348623157e308, 9.9e291],
    "half an ulp over DBL_MAX": [1.7976931348623157e308, 9.98e291],
    "class boundary below": [2.0 ** 959, 2.0 ** 959, 1.0],
    "class boundary above": [2.0 ** 960, 2.0 ** 960, 1.0, -(2.0 ** 960)],
    "big rows cancel exactly, small rows remain": [1e308, 3.5, -1e308, 1e-300],
    "big rows cancel to a remainder": [2.0 ** 1000, 2.0 ** 960, -(2.0 ** 1000), 7.0],
    "subnormals": [5e-324, 5e-324, -1e-320],
    "negative zero": [-0.0],
    "empty": [],
}
# here the reference's own two orders part ways: the strict left-to-right sum overflows on the way, the 32 strided partials do not
ORDER_DEPENDENT = {
    # −inf met AFTER the finite rows have overflowed to +inf: NaN in the reference, −inf with the −inf row first; the rule says −inf
    "inf after overflowing finite rows": [1e308, 1e308, 1e308, -inf] + [0.0] * 60,
    "intermediate overflow, exact sum 0": [1e308, 1e308, -1e308, -1e308] + [0.0] * 60,
    "intermediate overflow, exact sum finite": [1e308, 1e308, -1e308, 4.0] + [0.0] * 60,
}


# finite cancellation is where the reference's plain additions round (its two orders lose the small rows); every other case is exact in
# plain double arithmetic, so both reference orders must return the rule's value
REF_INEXACT = {"big rows cancel exactly, small rows remain", "big rows cancel to a remainder", "class boundary above"}


@pytest.mark.parametrize("name", [k for k in CASES if k not in REF_INEXACT])
def test_oracle_agrees_with_both_reference_orders(orc, name):
    ref = OL.load_reference()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    a = np.array(CASES[name], dtype=np.float64)
    want = float(orc.sum_float64_xreal(a))
    assert same(want, float(ref.sum("seq", a))), name
    assert same(want, float(ref.sum("avx2", a))), name
    # ... also wherever the special rows sit in a longer column (32-row groups, scalar tail)
    rng = np.random.default_rng(len(name))
    if a.size and not np.all(np.isfinite(a)):
        for n in (31, 64, 100, 1000):
            col = rng.uniform(-1, 1, n)
            col[rng.choice(n, a.size, replace=False)] = a
            want = float(orc.sum_float64_xreal(col))
            assert same(want, float(ref.sum("seq", col))) and same(want, float(ref.sum("avx2", col))), (name, n)


def test_reference_orders_disagree_on_intermediate_overflow(orc):
    ref = OL.load_reference()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    for name, want in (("intermediate overflow, exact sum 0", 0.0), ("intermediate overflow, exact sum finite", 1e308 + 4.0)):
        a = np.array(ORDER_DEPENDENT[name], dtype=np.float64)
        seq, avx = float(ref.sum("seq", a)), float(ref.sum("avx2", a))
        assert seq == inf and avx == want                    # the two reference paths return different numbers
        assert float(orc.sum_float64_xreal(a)) == want         # the rule: the order-free one
    a = np.array(ORDER_DEPENDENT["inf after overflowing finite rows"], dtype=np.float64)
    assert math.isnan(float(ref.sum("seq", a))) and float(ref.sum("seq", a[::-1].copy())) == -inf   # the same rows, two answers
    assert float(orc.sum_float64_xreal(a)) == -inf


@pytest.mark.parametrize("name", list(CASES) + list(ORDER_DEPENDENT))
@pytest.mark.parametrize("lanes", [1, 2, 64, 1000])
def test_header_model_oracle_agree(orc, host, name, lanes):
    a = np.array({**CASES, **ORDER_DEPENDENT}[name], dtype=np.float64)
    want = float(orc.sum_float64_xreal(a))
    got, parts = host_sum(host, a, lanes)
    mod = M.accumulate(a.tolist(), lanes)
    assert same(got, want), (name, lanes, got, want)
    assert same(M.result(mod), want)
    # the header and the model hold the same four words (NaN payloads aside)
    for x, y in zip(parts.tolist(), mod):
        assert same(x, y), (name, lanes, parts, mod)


def test_xreal_is_fsum_on_finite_data(orc):
    rng = np.random.default_rng(5)
    for n in (0, 1, 2, 33, 1000, 100003):
        for a in (rng.uniform(-1, 1, n), rng.uniform(-1, 1, n) * 1e-310, rng.standard_normal(n) * np.exp(rng.uniform(-700, 700, n)),
                  rng.integers(-2**52, 2**52, n).astype(np.float64)):
            assert float(orc.sum_float64_xreal(a)) == math.fsum(a.tolist())
            assert float(orc.sum_float64_xreal(a)) == float(orc.sum_float64_exact(a))


@pytest.mark.parametrize("seed", range(6))
def test_random_mixtures(orc, host, seed):
    """wide-range finite data with big rows that nearly cancel, optionally salted with non-finite rows, any lane count: within the
    double-double bound of the exact sum, class (finite / ±inf / NaN) exact"""
    rng = np.random.default_rng(100 + seed)
    n = 20000
    a = rng.standard_normal(n) * np.exp(rng.uniform(-50, 50, n))
    big = rng.standard_normal(200) * 2.0 ** rng.integers(955, 1022, 200)
    pos = rng.choice(n, 400, replace=False)
    a[pos[:200]] = big
    a[pos[200:]] = -big * (1.0 if seed % 2 else (1 + 2.0 ** -30))
    if seed >= 4:
        a[rng.integers(0, n)] = [inf, -inf][seed % 2]
    want = float(orc.sum_float64_xreal(a))
    for lanes in (1, 7, 256):
        got, _ = host_sum(host, a, lanes)
        if math.isfinite(want):
            # the double-double bound under cancellation (tests/test_gpu_parity.py::test_sum_float64 (c)): ulp(exact) + n·2^-104·Σ|x|,
            # taken at 2^-128 scale because Σ|x| itself is beyond DBL_MAX here
            bound = math.ulp(want) * M.DOWN + n * 2.0 ** -104 * float(np.abs(a * M.DOWN).sum())
            assert math.isfinite(got) and abs(got * M.DOWN - want * M.DOWN) <= bound, (lanes, got, want)
        else:
            assert same(got, want)


def test_merge_of_rank_partials(orc, host):
    """the cross-rank combine of ah_comm_cmp_filter_sum_f64: per-rank accumulators merged in rank order, rounded once — the result does
    not depend on how the rows fall over the ranks, overflow across ranks included"""
    rows = np.array([1e308, 2.5, 1e308, -1e308, inf][:4] + [1e-3] * 5)
    want = float(orc.sum_float64_xreal(rows))
    assert want == 1e308 + 2.5 + 5e-3 or abs(want - 1e308) <= math.ulp(1e308)
    for world in (1, 2, 3, 9):
        parts = []
        for r in range(world):
            lo, hi = rows.size * r // world, rows.size * (r + 1) // world
            _, p = host_sum(host, rows[lo:hi], 3)
            parts.append(p)
        flat = np.ascontiguousarray(np.concatenate(parts))
        assert host.ddx_merge_result(flat.ctypes.data, world) == want, world
