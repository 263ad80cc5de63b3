"""HIP vs oracle, byte for byte, AT THE SIZES BASELINE.json's configs name — with nulls.

  C2  2^27-row Int64 / Float64 columns: checked Add with both validities, compare at a bit offset
  C3  2^27-row Int64 column with 10 % nulls: Filter (Drop / Emit, mask nulls, s ∈ {0.01, 0.5, 0.9}),
      Take by 2^27 uniformly random int32 indices with 10 % nulls on both sides
  C5  2^26-row Int64 keys / Float64 values: dictionary_encode + hash_sum at 2^10 … 2^24 keys, one Zipf(1.1) column

The oracle (C, one core) does 2^27 rows in about a second per kernel, so these are plain equality tests of the downloaded
bytes — values, validity, null counts, ids, dictionaries — not property checks.  Reference behaviour:
vector_selection_test.go:554-613 (random compare-then-filter), :213-253 (take), vector_hash_test.go:768 (table growth).
"""
import numpy as np
import pytest

from tests import oracle_lib as OL
from tests.backends import HipBackend, OracleBackend, STATUS_OK

pytestmark = pytest.mark.gpu

N27 = 1 << 27
N26 = 1 << 26


def bits(rng, n, p):
    """n random bits, P(set) = p, packed LSB-first (+ 8 bytes of slack so a bit offset can be applied)"""
    out = np.empty(n // 8 + 9, np.uint8)
    step = 1 << 24
    for i in range(0, n, step):
        m = min(step, n - i)
        out[i // 8:i // 8 + (m + 7) // 8] = np.packbits(rng.random(m) < p, bitorder="little")
    out[(n + 7) // 8:] = 0xA5   # slack is not zero: nothing may depend on it
    return out


@pytest.fixture(scope="module")
def hip(ctx):
    return HipBackend(ctx)


@pytest.fixture(scope="module")
def orc_be():
    return OracleBackend()


@pytest.fixture(scope="module")
def column():
    """the C3 column: 2^27 Int64 values, Bernoulli(0.9) validity"""
    rng = np.random.default_rng(1)
    values = rng.integers(-2**62, 2**62, N27, dtype=np.int64)
    return values, bits(rng, N27, 0.9)


def same(a, b, what):
    assert a.shape == b.shape, what
    if a.tobytes() != b.tobytes():
        bad = np.flatnonzero(a != b)
        raise AssertionError(f"{what}: {bad.size} of {a.size} differ, first at {bad[0]}: {a[bad[0]]!r} vs {b[bad[0]]!r}")


@pytest.mark.parametrize("sel,null_sel,mask_nulls", [(0.5, 0, False), (0.01, 0, False), (0.9, 1, True), (0.5, 0, True), (0.5, 1, True)],
                         ids=["s0.5-drop", "s0.01-drop", "s0.9-emit-masknulls", "s0.5-drop-masknulls", "s0.5-emit-masknulls"])
def test_c3_filter_2_27(hip, orc_be, column, sel, null_sel, mask_nulls):
    values, vvalid = column
    rng = np.random.default_rng(int(sel * 100) + 7 * null_sel + mask_nulls)
    fdata = bits(rng, N27, sel)
    fvalid = bits(rng, N27, 0.9) if mask_nulls else None
    g = hip.filter(values, vvalid, 0, fdata, fvalid, 0, N27, null_sel, True)
    e = orc_be.filter(values, vvalid, 0, fdata, fvalid, 0, N27, null_sel, True)
    assert g[0].size == e[0].size and g[2] == e[2]
    same(g[0], e[0], "filter payload")      # incl. payload under null outputs (copied for null values, 0 for emitted nulls)
    same(g[1], e[1], "filter validity")


def test_c3_filter_2_27_runs_and_offsets(hip, orc_be, column):
    """a mask made of long runs (geometric, mean 256) at non-zero bit offsets on every bitmap"""
    values, vvalid = column
    rng = np.random.default_rng(11)
    n = N27 - 77
    runs = rng.geometric(1 / 256, n // 200)
    on = np.repeat(np.arange(runs.size) % 2 == 0, runs)[:n + 13]
    on = np.concatenate([on, np.zeros(max(0, n + 13 - on.size), bool)])
    fdata = np.concatenate([np.packbits(on, bitorder="little"), np.full(9, 0xA5, np.uint8)])
    g = hip.filter(values[5:5 + n], vvalid, 5, fdata, None, 13, n, 0, True, misalign=1)
    e = orc_be.filter(values[5:5 + n], vvalid, 5, fdata, None, 13, n, 0, True)
    assert g[2] == e[2]
    same(g[0], e[0], "filter payload (runs)")
    same(g[1], e[1], "filter validity (runs)")


@pytest.mark.parametrize("binned", [1, 0], ids=["auto", "direct"])
def test_c3_take_2_27_random_with_nulls(ctx, hip, orc_be, column, binned):
    """2^27 uniformly random int32 indices over [0, 2^27), 10 % nulls in values AND in indices — through the binned path
    (what `auto` picks here) and through the direct kernel"""
    values, vvalid = column
    rng = np.random.default_rng(21)
    idx = rng.integers(0, N27, N27, dtype=np.int32)
    ivalid = bits(rng, N27, 0.9)
    ctx.set_option("take_binned", binned)
    try:
        g = hip.take(values, vvalid, 0, idx, ivalid, 0, True, True)
    finally:
        ctx.set_option("take_binned", 1)
    e = orc_be.take(values, vvalid, 0, idx, ivalid, 0, True, True)
    assert g[0] == e[0] == STATUS_OK and g[3] == e[3]
    same(g[1], e[1], "take payload")
    same(g[2], e[2], "take validity")


def test_c3_take_2_27_no_nulls_and_bad_index(ctx, hip, orc_be, column):
    values, _ = column
    rng = np.random.default_rng(22)
    idx = rng.integers(0, N27, N27, dtype=np.int32)
    g = hip.take(values, None, 0, idx, None, 0, True, False)
    e = orc_be.take(values, None, 0, idx, None, 0, True, False)
    assert g[0] == e[0] == STATUS_OK
    same(g[1], e[1], "take payload (no nulls)")
    idx[100_000_007] = -5
    idx[100_000_001] = N27            # the FIRST offender in index order is reported
    g = hip.take(values, None, 0, idx, None, 0, True, False)
    e = orc_be.take(values, None, 0, idx, None, 0, True, False)
    assert g[0] == e[0] != STATUS_OK and g[4] == e[4] == N27


def test_c2_checked_add_and_compare_2_27(hip, orc_be, column):
    a, av = column
    rng = np.random.default_rng(31)
    b = rng.integers(-2**62, 2**62, N27, dtype=np.int64)
    bv = bits(rng, N27, 0.9)
    n = N27 - 19
    g = hip.arithmetic_checked(21, 0, a[:n], av, 3, b[:n], bv, 11)      # OpAddChecked, array ∘ array
    e = orc_be.arithmetic_checked(21, 0, a[:n], av, 3, b[:n], bv, 11)
    assert g[0] == e[0] == STATUS_OK       # |a|, |b| < 2^62: no overflow; null slots hold 0
    same(g[1], e[1], "checked add")
    init = np.full(N27 // 8 + 16, 0xA5, np.uint8)
    thr = np.array([12345], np.int64)
    gc = hip.comparison(2, 1, a[:n], thr, init, 5)                      # greater(array, scalar) at out.Offset = 5
    ec = orc_be.comparison(2, 1, a[:n], thr, init, 5)
    same(gc, ec, "compare bitmap incl. the bits around the range")
    f = a.view(np.float64)                                              # arbitrary bit patterns incl. NaNs / denormals
    gf = hip.comparison(3, 0, f[:n], b.view(np.float64)[:n], init, 2)   # greater_equal(array, array), float64
    ef = orc_be.comparison(3, 0, f[:n], b.view(np.float64)[:n], init, 2)
    same(gf, ef, "float compare bitmap")


def zipf_keys(rng, n, card, s=1.1):
    """Zipf(s) over `card` distinct 64-bit keys (inverse-CDF on the truncated distribution)"""
    w = 1.0 / np.arange(1, card + 1) ** s
    cdf = np.cumsum(w); cdf /= cdf[-1]
    ranks = np.searchsorted(cdf, rng.random(n))
    pool = rng.integers(-2**63, 2**63 - 1, card, dtype=np.int64)
    return pool[ranks]


@pytest.mark.parametrize("lg,dist", [(10, "uniform"), (16, "uniform"), (20, "uniform"), (22, "uniform"), (24, "uniform"), (20, "zipf")])
def test_c5_hash_2_26(hip, orc_be, lg, dist):
    rng = np.random.default_rng(40 + lg + (dist == "zipf"))
    if dist == "zipf":
        keys = zipf_keys(rng, N26, 1 << lg)
    else:
        pool = rng.integers(-2**63, 2**63 - 1, 1 << lg, dtype=np.int64)
        keys = pool[rng.integers(0, 1 << lg, N26)]
    kvalid = bits(rng, N26, 0.98)
    for enc in ((False, True) if lg < 22 else (False,)):   # (2^24 keys: ≈ 10 s of oracle per call)
        g = hip.hash_encode(keys, kvalid, 0, enc)
        e = orc_be.hash_encode(keys, kvalid, 0, enc)
        same(g[0], e[0], f"ids enc={enc}")
        same(g[1], e[1], "id validity")
        same(g[2], e[2], "dictionary (first-seen order)")
        assert g[3] == e[3]
    # hash + sum: Int64 values are exact (wrapping); Float64 values chosen integer-valued (< 2^31 per row, < 2^53 per group)
    # so that every order of additions gives the same bits — the general-float tolerance is tested at small sizes
    vvalid = bits(rng, N26, 0.9)
    if lg in (16, 20):
        vals_i = rng.integers(-2**62, 2**62, N26, dtype=np.int64)
        g = hip.hash_sum("i64", keys, kvalid, 0, vals_i, vvalid, 0)
        e = orc_be.hash_sum("i64", keys, kvalid, 0, vals_i, vvalid, 0)
        for k, what in enumerate(("group keys", "sums", "counts")):
            same(g[k], e[k], f"hash_sum i64 {what}")
        assert g[3] == e[3]
        same(g[4], e[4], "first rows")
    vals_f = rng.integers(-2**20, 2**20, N26).astype(np.float64)
    g = hip.hash_sum("f64", keys, kvalid, 0, vals_f, vvalid, 0)
    e = orc_be.hash_sum("f64", keys, kvalid, 0, vals_f, vvalid, 0)
    for k, what in enumerate(("group keys", "sums", "counts")):
        same(g[k], e[k], f"hash_sum f64 {what}")


def test_c2_float64_sum_nonfinite_2_27(ctx):
    """Float64 Sum and the fused C4 sum at 2^27 rows with ±inf / NaN / overflowing rows in the head, the body, the odd tail and the last
    workgroup: the extended-real rule of csrc/ah_ddsum.h — the value both reference orders return (arrow/math/float64.go:41-47) — and
    the ordinary column still within 1 ULP of the exact sum"""
    import math
    from tests import oracle_lib as OL
    o, ref = OL.load_oracle(), OL.load_reference()
    rng = np.random.default_rng(2027)
    n = N27 - 1                                   # odd tail row
    base = rng.standard_normal(n)
    d = ctx.alloc(base.nbytes + 64)
    inf, nan = math.inf, math.nan
    exact = float(o.sum_float64_xreal(base))
    for misalign in (0, 1):
        d.upload(base, misalign * 8)
        got = ctx.sum_float64(d.ptr + misalign * 8, n)
        assert abs(got - exact) <= math.ulp(exact), (got, exact)
    for label, edits, want in (("+inf head", {0: inf}, inf), ("-inf tail", {n - 1: -inf}, -inf), ("nan body", {n // 2 + 1: nan}, nan),
                               ("both infinities, far apart", {1: inf, n - 2: -inf}, nan), ("+inf in the last workgroup", {n - 4097: inf}, inf),
                               ("finite overflow", {7: 1e308, n // 3: 1e308, n - 9: 1e308}, inf),
                               ("negative finite overflow", {8: -1e308, n // 3: -1e308, n - 10: -1.5e308}, -inf),
                               ("intermediate overflow only", {7: 1e308, n // 3: 1e308, n // 3 + 2: -1e308, n - 9: -1e308}, None)):
        col = base.copy()
        for i, v in edits.items():
            col[i] = v
        if want is None:
            want = float(o.sum_float64_xreal(col))
            assert math.isfinite(want)
        elif ref is not None:
            for which in ("seq", "avx2"):
                r = float(ref.sum(which, col))
                assert (math.isnan(r) and math.isnan(want)) or r == want, (label, which, r)
        d.upload(col, 8)
        got = ctx.sum_float64(d.ptr + 8, n)
        ok = (math.isnan(got) and math.isnan(want)) or got == want or (math.isfinite(want) and abs(got - want) <= math.ulp(want))
        assert ok, (label, got, want)
        s, c = ctx.cmp_filter_sum_f64(3, d.ptr + 8, None, 0, n, -inf)       # x >= -inf keeps every row but NaN
        keep_nan = any(isinstance(v, float) and math.isnan(v) for v in edits.values())
        assert c == n - (1 if keep_nan else 0), (label, c)
        want_f = float(o.sum_float64_xreal(col[~np.isnan(col)])) if keep_nan else want
        ok = (math.isnan(s) and math.isnan(want_f)) or s == want_f or (math.isfinite(want_f) and abs(s - want_f) <= math.ulp(want_f))
        assert ok, (label, "fused", s, want_f)


def test_c2_sum_and_cumulative_sum_2_27(hip, orc_be, column):
    a, av = column
    assert hip.sum(a) == orc_be.sum(a)
    g = hip.cumulative_sum(a, None, 0)
    e = orc_be.cumulative_sum(a, None, 0)
    assert g[0] == e[0] == STATUS_OK
    same(g[1], e[1], "cumulative_sum int64 (wrapping)")
    g = hip.cumulative_sum(a[:N27 - 5], av, 5, skip_nulls=True)
    e = orc_be.cumulative_sum(a[:N27 - 5], av, 5, skip_nulls=True)
    assert g[0] == e[0] == STATUS_OK and g[3] == e[3]
    same(g[1], e[1], "cumulative_sum skip_nulls payload")
    same(g[2], e[2], "cumulative_sum skip_nulls validity")
    # Float64, one pass with the fixed-grouping look-back: 2^27 rows = 8192 tiles = two super blocks (the super-block chain is
    # crossed once); integer-valued addends — every order of additions is exact, so the bytes are the sequential oracle's
    f = (a % 7).astype(np.float64)
    g = hip.cumulative_sum(f, None, 0)
    e = orc_be.cumulative_sum(f, None, 0)
    assert g[0] == e[0] == STATUS_OK
    same(g[1], e[1], "cumulative_sum float64 (one pass)")


@pytest.mark.parametrize("kind", ["int64", "float64"])
def test_sort_indices_2_27(hip, kind):
    """sort_indices at the size of the C2 / C3 columns through the MSD path: the permutation must be THE stable one
    (numpy's stable argsort of the same keys; a stable sort has exactly one answer), nulls last"""
    rng = np.random.default_rng(90)
    a = rng.integers(-2**63, 2**63 - 1, N27, dtype=np.int64) if kind == "int64" else rng.standard_normal(N27)
    a[rng.integers(0, N27, 1000)] = a[0]          # some ties
    g = hip.sort_indices(a, None, 0, False, False)
    e = np.argsort(a, kind="stable").astype(np.uint64)
    same(g, e, f"sort_indices {kind}")
    valid = bits(rng, N27, 0.9)
    g = hip.sort_indices(a, valid, 0, True, True)   # descending, nulls first: the partition pass in front of the MSD path
    ok = np.unpackbits(valid, bitorder="little")[:N27].astype(bool)
    nulls = np.flatnonzero(~ok).astype(np.uint64)
    rest = np.flatnonzero(ok)
    key = a[rest]
    order = rest[np.argsort(-key if kind == "float64" else ~key, kind="stable")].astype(np.uint64)
    same(g, np.concatenate([nulls, order]), f"sort_indices {kind} descending, nulls first")


@pytest.mark.parametrize("pct", [10, 50, 90])
def test_c4_fused_2_27_vs_oracle(hip, orc_be, column, pct):
    """Config C4 per GPU — Compare(>) → Filter(DropNulls) → Sum over a 2^27-row shard with 10 % nulls — against orc_fused.c (the
    unfused reference chain collapsed per element), thresholds at the 10th / 50th / 90th percentile: Int64 sum and
    count bit-exact; Float64 count exact and the sum within 1 ULP of the exact sum of the survivors (the Sum rule of DESIGN §4)."""
    a, av = column
    thr = int(np.percentile(a[:1 << 20], pct))
    g, e = hip.cmp_filter_sum_i64(2, a, av, 0, thr), orc_be.cmp_filter_sum_i64(2, a, av, 0, thr)
    assert g == e, (pct, g, e)
    g, e = hip.cmp_filter_sum_i64(2, a[:N27 - 11], av, 11, thr), orc_be.cmp_filter_sum_i64(2, a[:N27 - 11], av, 11, thr)   # validity at a bit offset
    assert g == e, (pct, g, e)
    rng = np.random.default_rng(300 + pct)
    x = rng.standard_normal(N27) * 1e3
    tf = float(np.percentile(x[:1 << 20], pct))
    gs, gc = hip.cmp_filter_sum_f64(2, x, av, 0, tf)
    es, ec = orc_be.cmp_filter_sum_f64(2, x, av, 0, tf)     # (exact sum of the survivors, count)
    assert gc == ec
    assert abs(gs - es) <= np.spacing(abs(es)), (pct, gs, es)
    gs2, gc2 = hip.cmp_filter_sum_f64(2, x, av, 0, tf)
    assert gs2 == gs and gc2 == gc                            # the double-double tree does not depend on timing


@pytest.mark.parametrize("lg", [10, 16, 20])
def test_c5_hash_sum_f64_general_doubles_2_26(hip, orc_be, lg):
    """hash + sum of 2^26 rows with GENERAL doubles (round-2 review: only integer-valued addends had been checked at this size).
    The column spans < 42 binades, so the 128-bit fixed point holds every addend whole: each group's sum must be the correctly
    rounded exact sum (math.fsum) — checked on the groups of 4096 sampled keys — and two runs must give the same bytes."""
    import math
    rng = np.random.default_rng(500 + lg)
    pool = rng.integers(-2**63, 2**63 - 1, 1 << lg, dtype=np.int64)
    keys = pool[rng.integers(0, 1 << lg, N26)]
    vals = (1.0 + rng.random(N26)) * np.exp(rng.uniform(-10, 10, N26)) * rng.choice([-1.0, 1.0], N26)
    vvalid = bits(rng, N26, 0.9)
    g = hip.hash_sum("f64", keys, None, 0, vals, vvalid, 0)
    g2 = hip.hash_sum("f64", keys, None, 0, vals, vvalid, 0)
    same(g[1].view(np.uint64), g2[1].view(np.uint64), "hash_sum f64 run to run")
    ok = np.unpackbits(vvalid, bitorder="little")[:N26].astype(bool)
    first = {int(k): i for i, k in enumerate(g[0].view(np.int64))}
    assert len(first) == g[0].size
    sample = pool[rng.integers(0, 1 << lg, 4096 if lg > 10 else 64)]
    # one pass: rows of the sampled keys only
    want = np.isin(keys, sample)
    ks, vs = keys[want & ok], vals[want & ok]
    order = np.argsort(ks, kind="stable")
    ks, vs = ks[order], vs[order]
    bounds = np.flatnonzero(np.diff(ks)) + 1
    for lo, hi in zip(np.concatenate([[0], bounds]), np.concatenate([bounds, [ks.size]])):
        exact = math.fsum(vs[lo:hi].tolist())             # correctly rounded exact sum
        got = float(g[1][first[int(ks[lo])]])
        assert got == exact, (lg, int(ks[lo]), got, exact, hi - lo)
        assert int(g[2][first[int(ks[lo])]]) == hi - lo


def test_c3_take_2_27_clustered_vec_path(hip, orc_be, column):
    """the 16 / W rows-per-lane Take (ah_take.hip take_vec_kernel) at the size of config C3, chosen by the neighbour sample:
    identity, reversed and a shifted slice with 10 % nulls on both sides, byte for byte"""
    a, av = column
    rng = np.random.default_rng(61)
    iv = bits(rng, N27, 0.9)
    for name, idx in (("identity", np.arange(N27, dtype=np.int32)), ("reverse", np.arange(N27 - 1, -1, -1, dtype=np.int32)),
                      ("slice", np.minimum(np.arange(N27, dtype=np.int64) + 777, N27 - 1).astype(np.int32))):
        g = hip.take(a, av, 0, idx, iv, 0, True, True)
        e = orc_be.take(a, av, 0, idx, iv, 0, True, True)
        assert g[0] == e[0] == STATUS_OK
        same(g[1], e[1], f"take {name} values")
        same(g[2], e[2], f"take {name} validity")
        assert g[3] == e[3]
