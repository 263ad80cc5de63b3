"""Float64 Sum on ±inf, NaN and overflowing inputs, through every entry point that shares the accumulator of csrc/ah_ddsum.h:
ah_sum_float64, ah_sum_float64_dev, ah_cmp_filter_sum_f64, ah_cmp_filter_sum_f64_dev, ah_ingest_sum_float64 (and, with two and
three ranks on one GPU, ah_comm_cmp_filter_sum_f64: scripts/dist_gpu_ranks.py; at 2^27 rows: tests/test_full_size.py).

Columns of at most 31 rows through the three Sum entry points are the exception: there both reference paths are ONE sequential loop
(arrow/math/_lib/float64_avx2.s:16-17 `cmp rsi, 31 ; jbe`, arrow/math/float64.go:41-47) and the device returns that loop's bytes,
intermediate overflow included (tests/test_sum_short.py); the fused chain has no reference counterpart and stays on the rule.

The rule, from 32 rows on: the sum over the extended reals, rounded once.  Expected values come from the oracle's fixed-point
superaccumulator (orc_sum_float64_xreal) and — for every case in which the reference's two summation orders agree with each other —
from BOTH reference orders run here: oracle/_ref's AVX2 machine code (ref.sum("avx2")) and its strict-sequential C (ref.sum("seq")),
arrow/math/float64.go:41-47, _lib/float64.c:20-26.  The special rows are placed where the kernel treats rows differently: the
unaligned head row, the 16-byte body, the odd tail row, the ragged last iteration, the last workgroup."""
import math

import numpy as np
import pytest

from tests import oracle_lib as OL

pytestmark = pytest.mark.gpu
inf, nan = math.inf, math.nan
GT, GE = 2, 3

# name → (special rows, both reference orders agree with the rule wherever the rows sit)
CASES = {
    "one +inf": ([inf], True),
    "one -inf": ([-inf], True),
    "+inf and -inf": ([inf, -inf], True),
    "nan": ([nan], True),
    "-inf and nan": ([-inf, nan], True),
    "+inf twice": ([inf, inf], True),
    "three 1e308": ([1e308] * 3, True),
    "three -1e308": ([-1e308] * 3, True),
    # from 32 rows on the running sum of a reference order may or may not pass through ±inf here, depending on where the rows sit:
    # only the rule is checked (up to 31 rows there is one reference order, and seq_sum below is it)
    "1e308 twice and back": ([1e308, 1e308, -1e308, -1e308], False),
    "overflow then the other infinity": ([1e308, 1e308, 1e308, -inf], False),
}
PLACEMENTS = ("head", "body", "tail", "spread", "last_workgroup")


def same(a, b):
    return (math.isnan(a) and math.isnan(b)) or a == b


def place(rng, n, special, where):
    """a column of n ordinary rows with `special` written over it at the named place"""
    col = rng.uniform(-1, 1, n)
    k = len(special)
    if k > n:
        special, k = special[:n], n
    if where == "head":
        idx = np.arange(k)
    elif where == "tail":
        idx = np.arange(n - k, n)
    elif where == "body":
        idx = n // 2 + np.arange(k) if n // 2 + k <= n else np.arange(k)
    elif where == "last_workgroup":
        idx = np.maximum(n - 1 - 3 * np.arange(k) - 5, 0)
        if len(set(idx.tolist())) < k:
            idx = np.arange(k)
    else:
        idx = np.sort(rng.choice(n, k, replace=False))
    col[idx] = special
    return col


def seq_sum(col):
    """acc = +0.0; acc += x left to right in IEEE doubles: arrow/math/float64.go:41-47, and _lib/float64_avx2.s below 32 rows"""
    acc = 0.0
    for v in col.tolist():
        acc += v
    return acc


def check_short(got, col, ref, label):
    want = seq_sum(col)
    assert same(got, want) and (math.isnan(want) or np.float64(got).tobytes() == np.float64(want).tobytes()), (label, got, want)
    if ref is not None:
        assert same(want, float(ref.sum("seq", col))) and same(want, float(ref.sum("avx2", col))), label


def check_value(got, want, label):
    if math.isfinite(want):
        assert math.isfinite(got) and abs(got - want) <= math.ulp(want), (label, got, want)
    else:
        assert same(got, want), (label, got, want)


@pytest.fixture(scope="module")
def refs():
    return OL.load_oracle(), OL.load_reference()


@pytest.mark.parametrize("n", [3, 5, 8192, 8193, (1 << 20) + 7])
@pytest.mark.parametrize("name", list(CASES))
def test_sum_float64_extended_reals(ctx, refs, name, n):
    o, ref = refs
    special, ref_agrees = CASES[name]
    rng = np.random.default_rng(n + len(name))
    res = ctx.alloc(64)
    for where in PLACEMENTS:
        for misalign in (0, 1):
            col = place(rng, n, special, where)
            want = float(o.sum_float64_xreal(col))
            if ref_agrees and len(special) <= n and ref is not None and not math.isfinite(want):
                assert same(want, float(ref.sum("seq", col))) and same(want, float(ref.sum("avx2", col))), (name, where, n)
            buf = ctx.alloc(col.nbytes + 64)
            buf.upload(col, misalign * 8)
            p = buf.ptr + misalign * 8
            label = (name, where, n, misalign)
            got = ctx.sum_float64(p, n)                                          # ah_sum_float64
            ctx.sum_float64_dev(p, n, res)                                       # ah_sum_float64_dev
            got_dev = float(res.download(np.float64, 1)[0])
            if n <= 31:
                check_short(got, col, ref, label)
                check_short(got_dev, col, ref, label)
                continue
            check_value(got, want, label)
            check_value(got_dev, want, label)


@pytest.mark.parametrize("n", [3, 8192, (1 << 20) + 7])
@pytest.mark.parametrize("name", list(CASES))
def test_cmp_filter_sum_f64_extended_reals(ctx, refs, name, n):
    """the fused chain keeps what Compare(>=, −inf) keeps — every valid row but NaN (a NaN row compares false and is dropped, as in the
    reference's greater_equal → Filter) — and sums it by the same rule; null rows never reach the sum, non-finite or not"""
    o, ref = refs
    special, ref_agrees = CASES[name]
    rng = np.random.default_rng(n * 7 + len(name))
    ds, dc = ctx.alloc(64), ctx.alloc(64)
    for where in PLACEMENTS:
        for with_valid in (False, True):
            col = place(rng, n, special, where)
            vb = rng.random(n) < 0.9 if with_valid else np.ones(n, bool)
            valid = np.packbits(vb, bitorder="little") if with_valid else None
            for op, thr in ((GE, -inf), (GT, 0.0)):
                keep = col[vb & ((col >= thr) if op == GE else (col > thr))]
                want = float(o.sum_float64_xreal(keep))
                if ref_agrees and ref is not None and not math.isfinite(want):   # (a finite sum is where the reference's own orders round)
                    assert same(want, float(ref.sum("seq", keep))) and same(want, float(ref.sum("avx2", keep)))
                dx = ctx.to_device(col, pad=64)
                dv = ctx.to_device(valid, pad=64) if with_valid else None
                label = (name, where, n, with_valid, op)
                s, c = ctx.cmp_filter_sum_f64(op, dx, dv, 0, n, thr)              # ah_cmp_filter_sum_f64
                assert c == keep.size, label
                check_value(s, want, label)
                ctx.cmp_filter_sum_f64_dev(op, dx, dv, 0, n, thr, ds, dc)         # ah_cmp_filter_sum_f64_dev
                assert int(dc.download(np.int64, 1)[0]) == keep.size
                check_value(float(ds.download(np.float64, 1)[0]), want, label)


@pytest.mark.parametrize("chunk_kib,depth", [(4, 2), (256, 3)])
def test_ingest_sum_float64_extended_reals(ctx, refs, chunk_kib, depth):
    """the chunked host → HBM sum: the special rows in the first chunk, in the last (ragged) one, split over two chunks"""
    import arrow_go_amd as ah
    o, ref = refs
    ing = ah.Ingest(ctx, chunk_kib << 10, depth)
    rows_per_chunk = (chunk_kib << 10) // 8
    try:
        for n in (3, rows_per_chunk * 3 + 11):
            for name, (special, ref_agrees) in CASES.items():
                for where in PLACEMENTS:
                    rng = np.random.default_rng(n + len(name) + len(where))
                    col = place(rng, n, special, where)
                    if where == "spread" and n > rows_per_chunk and len(special) > 1:      # straddle a chunk boundary
                        col = rng.uniform(-1, 1, n)
                        col[rows_per_chunk - 1:rows_per_chunk - 1 + len(special)] = special
                    want = float(o.sum_float64_xreal(col))
                    if ref_agrees and ref is not None and not math.isfinite(want):
                        assert same(want, float(ref.sum("seq", col))) and same(want, float(ref.sum("avx2", col)))
                    pb = ctx.alloc_pinned(col.nbytes + 64)
                    v = pb.view(np.float64, n)
                    v[...] = col
                    if n <= 31:
                        check_short(ing.sum_float64(v, n), col, ref, (name, where, n))
                    else:
                        check_value(ing.sum_float64(v, n), want, (name, where, n))
                    pb.free()
    finally:
        ing.close()


def test_sum_float64_class_boundary_and_cancellation(ctx, refs):
    """rows around 2^960 (where a row changes accumulators), big rows that cancel, exactly or to a remainder, next to small rows: within the
    double-double bound of the exact sum (tests/test_gpu_parity.py::test_sum_float64 (c)), taken at 2^-128 scale"""
    o, _ = refs
    rng = np.random.default_rng(77)
    n = 200_003
    for variant in range(4):
        a = rng.standard_normal(n) * np.exp(rng.uniform(-40, 40, n))
        big = rng.standard_normal(500) * 2.0 ** rng.integers(957, 1022, 500)
        pos = rng.choice(n, 1000, replace=False)
        a[pos[:500]] = big
        a[pos[500:]] = -big if variant % 2 == 0 else -big * (1 + 2.0 ** -40)
        if variant >= 2:
            a[pos[:500]] = np.abs(big) * 2.0 ** -8        # no cancellation: the sum is far above 2^960
            a[pos[500:]] = 2.0 ** 959.5
        want = float(o.sum_float64_xreal(a))
        got = ctx.sum_float64(ctx.to_device(a), n)
        if not math.isfinite(want):
            assert same(got, want)
            continue
        bound = math.ulp(want) * 2.0 ** -128 + n * 2.0 ** -104 * float(np.abs(a * 2.0 ** -128).sum())
        assert math.isfinite(got) and abs(got * 2.0 ** -128 - want * 2.0 ** -128) <= bound, (variant, got, want)
