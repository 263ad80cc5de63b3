"""Float64 Sum of 0 … 31 rows: bit-exact with BOTH reference paths.

Up to 31 rows the reference's AVX2 kernel is its scalar loop (arrow/math/_lib/float64_avx2.s:16-17: `cmp rsi, 31 ; jbe .LBB0_2` →
the `vaddsd` loop at .LBB0_4) and the pure-Go path is the same loop (arrow/math/float64.go:41-47): acc = +0.0, acc += x left to
right.  There is one reference answer — also where a partial sum passes through ±inf ([1e308, 1e308, -1e308] → +inf) or where the
exact sum rounds differently ([0.1]·10 → 0.9999999999999999) — and ah_sum_float64, ah_sum_float64_dev and ah_ingest_sum_float64
(the whole column in one call) return its bytes.  A NaN result is compared as "is NaN" (Go does not define NaN payloads; x86's
inf − inf is the negative default NaN, gfx950's the positive one).

From 32 rows on the two reference orders differ from each other and the order-free rule of DESIGN.md §4 applies
(tests/test_sum_nonfinite.py); the last test here pins the hand-over at 32 rows."""
import math

import numpy as np
import pytest

from tests import oracle_lib as OL
from tests.test_sum_nonfinite import seq_sum, same

inf, nan = math.inf, math.nan


def columns(n, rng):
    """the columns of the review: random, [0.1]·n, overflow through an intermediate sum, ±inf / NaN at every kind of place"""
    out = [("random", rng.uniform(-1e3, 1e3, n)),
           ("tenths", np.full(n, 0.1)),
           ("wide", rng.standard_normal(n) * np.exp(rng.uniform(-300, 300, n))),
           ("signed zeros", np.where(rng.random(n) < 0.5, -0.0, 0.0))]
    for name, special in (("1e308 twice and back", [1e308, 1e308, -1e308, -1e308]),
                          ("1e308 twice, once back", [1e308, 1e308, -1e308]),
                          ("overflow then -inf", [1e308, 1e308, -inf]),
                          ("-1e308 twice and back", [-1e308, -1e308, 1e308, 1e308]),
                          ("+inf", [inf]), ("-inf", [-inf]), ("+inf -inf", [inf, -inf]), ("nan", [nan]), ("-inf nan", [-inf, nan])):
        k = len(special)
        if k > n:
            continue
        for where, idx in (("head", np.arange(k)), ("tail", np.arange(n - k, n)), ("spread", np.sort(rng.choice(n, k, replace=False)))):
            col = rng.uniform(-1, 1, n)
            col[idx] = special
            out.append((f"{name} @ {where}", col))
            z = np.zeros(n)
            z[idx] = special
            out.append((f"{name} @ {where} in zeros", z))
    return out


def expect(col, ref):
    want = seq_sum(col)
    if ref is not None:   # the reference's own machine code and its C loop, run here: they are the same loop below 32 rows
        for order in ("seq", "avx2"):
            r = float(ref.sum(order, col))
            assert same(r, want) and (math.isnan(want) or np.float64(r).tobytes() == np.float64(want).tobytes()), (order, col)
    return want


def check(got, want, label):
    assert same(got, want), (label, got, want)
    if not math.isnan(want):
        assert np.float64(got).tobytes() == np.float64(want).tobytes(), (label, got, want)   # −0.0 vs +0.0 included


def test_seq_sum_is_the_reference_below_32_rows():
    """CPU: the Python model of the loop == oracle/_ref's two orders for every column the GPU test uses (and the two orders really
    part ways from 32 rows on, which is why the hand-over sits there)"""
    ref = OL.load_reference()
    if ref is None:
        pytest.skip("oracle/_ref not built")
    rng = np.random.default_rng(5)
    for n in range(0, 32):
        for _, col in columns(n, rng):
            expect(col, ref)
    col = np.zeros(64)
    col[[0, 1, 2, 3]] = [1e308, 1e308, -1e308, -1e308]
    assert float(ref.sum("seq", col)) != float(ref.sum("avx2", col)) or math.isinf(float(ref.sum("seq", col)))


@pytest.mark.gpu
@pytest.mark.parametrize("n", list(range(0, 32)))
def test_sum_float64_short_columns_bit_exact(ctx, n):
    import arrow_go_amd as ah
    ref = OL.load_reference()
    rng = np.random.default_rng(1000 + n)
    res = ctx.alloc(64)
    ing = ah.Ingest(ctx, 4096, 2)
    pb = ctx.alloc_pinned(4096)
    try:
        for name, col in columns(n, rng):
            want = expect(col, ref) if n else 0.0
            for misalign in (0, 1):
                buf = ctx.alloc(col.nbytes + 64)
                buf.upload(col, misalign * 8)
                p = buf.ptr + misalign * 8
                check(ctx.sum_float64(p, n), want, (name, n, misalign, "ah_sum_float64"))
                ctx.sum_float64_dev(p, n, res)
                check(float(res.download(np.float64, 1)[0]), want, (name, n, misalign, "ah_sum_float64_dev"))
                v = pb.view(np.float64, n + 1)[misalign:misalign + n]
                v[...] = col
                check(ing.sum_float64(v, n), want, (name, n, misalign, "ah_ingest_sum_float64"))
    finally:
        ing.close()
        pb.free()


@pytest.mark.gpu
def test_sum_float64_hand_over_at_32_rows(ctx):
    """31 rows: the sequential loop's +inf; 32 rows of the same shape: the order-free rule's finite answer (the reference's own two
    orders disagree there: DESIGN.md §4)"""
    o = OL.load_oracle()
    for n, sequential in ((31, True), (32, False), (33, False)):
        col = np.zeros(n)
        col[:4] = [1e308, 1e308, -1e308, -1e308]
        got = ctx.sum_float64(ctx.to_device(col), n)
        if sequential:
            assert got == inf == seq_sum(col)
        else:
            assert got == float(o.sum_float64_xreal(col)) == 0.0
            # the KNOWN DIVERGENCE from 32 rows on ("Known divergences"): the reference's sequential order passes through
            # +inf on this vector and stays there; the order-free rule returns the exact sum
            ref = OL.load_reference()
            if ref is not None:
                assert float(ref.sum("seq", col)) == inf
    tenths = np.full(10, 0.1)
    assert ctx.sum_float64(ctx.to_device(tenths), 10) == 0.9999999999999999 == seq_sum(tenths)
