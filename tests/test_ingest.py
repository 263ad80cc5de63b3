"""Overlapped host → HBM ingest (csrc/ah_ingest.hip): chunked Sum / Add / Filter straight from pinned host memory must give the
bytes of the resident entry points — which are the oracle's — whatever the chunk size: chunks smaller than a filter tile, ragged
last chunks, more chunks than slots (every slot reused many times), bit offsets, empty inputs."""
import math

import numpy as np
import pytest

from tests.backends import OracleBackend

pytestmark = pytest.mark.gpu
DROP, EMIT = 0, 1


@pytest.fixture(scope="module")
def orc_be():
    return OracleBackend()


def pinned_copy(ctx, arr):
    a = np.ascontiguousarray(arr)
    pb = ctx.alloc_pinned(a.nbytes + 64)
    v = pb.view(a.dtype, a.size)
    v[...] = a
    return pb, v


@pytest.mark.parametrize("chunk_kib,depth", [(4, 2), (64, 3), (1024, 4)])
def test_ingest_sum(ctx, orc_be, chunk_kib, depth):
    import arrow_go_amd as ah
    rng = np.random.default_rng(chunk_kib)
    ing = ah.Ingest(ctx, chunk_kib << 10, depth)
    try:
        for n in (0, 1, 511, 512, 513, 70001, 300007):
            xi = rng.integers(-2**62, 2**62, n, dtype=np.int64)
            pb, v = pinned_copy(ctx, xi)
            assert ing.sum_int64(v, n) == int(np.sum(xi, dtype=np.uint64).view(np.int64) if n else 0)
            pb.free()
            xf = rng.integers(-10**6, 10**6, n).astype(np.float64)           # integer-valued: exact in any order → bit-exact
            pb, v = pinned_copy(ctx, xf)
            assert ing.sum_float64(v, n) == float(orc_be.sum(xf)) if n else ing.sum_float64(v, n) == 0.0
            pb.free()
            xg = rng.standard_normal(n) * np.exp(rng.uniform(-20, 20, n))     # general: within 1 ULP of the exact sum, and the
            pb, v = pinned_copy(ctx, xg)                                      # SAME bytes as the resident kernel's one reduction
            got = ing.sum_float64(v, n)
            exact = math.fsum(xg.tolist())
            assert abs(got - exact) <= math.ulp(exact), (n, got, exact)
            if n:
                d = ctx.to_device(xg)
                assert got == ctx.sum_float64(d, n)
                d.free()
            pb.free()
        # pageable (unpinned) host memory: same results, just no overlap
        x = rng.standard_normal(100003)
        assert abs(ing.sum_float64(x, x.size) - math.fsum(x.tolist())) <= math.ulp(math.fsum(x.tolist()))
    finally:
        ing.close()


@pytest.mark.parametrize("dtype,tid", [(np.int64, 9), (np.float64, 12), (np.int32, 7), (np.uint8, 2)])
def test_ingest_arithmetic(ctx, orc_be, dtype, tid):
    import arrow_go_amd as ah
    rng = np.random.default_rng(tid)
    for chunk_kib, depth in ((4, 2), (256, 3)):
        ing = ah.Ingest(ctx, chunk_kib << 10, depth)
        try:
            for n in (0, 1, 1000, 70001, 262144 + 17):
                if np.dtype(dtype).kind == "f":
                    a, b = rng.standard_normal(n).astype(dtype), rng.standard_normal(n).astype(dtype)
                else:
                    info = np.iinfo(dtype)
                    a, b = rng.integers(info.min, info.max, n, dtype=dtype, endpoint=True), rng.integers(info.min, info.max, n, dtype=dtype, endpoint=True)
                pa, va = pinned_copy(ctx, a); pb, vb = pinned_copy(ctx, b)
                po = ctx.alloc_pinned(a.nbytes + 64); vo = po.view(dtype, n); vo[...] = 0x55 if n else 0
                for op in (0, 1, 2):
                    ing.arithmetic_binary(tid, op, va, vb, vo, n)
                    assert vo.tobytes() == orc_be.arithmetic(op, 0, a, b).tobytes(), (dtype, n, op, chunk_kib)
                for p in (pa, pb, po):
                    p.free()
        finally:
            ing.close()


@pytest.mark.parametrize("dtype", [np.int64, np.float32, np.uint8], ids=str)
def test_ingest_filter(ctx, orc_be, dtype):
    import arrow_go_amd as ah
    rng = np.random.default_rng(17)
    w = np.dtype(dtype).itemsize
    for chunk_kib, depth in ((8, 2), (128, 3)):
        ing = ah.Ingest(ctx, chunk_kib << 10, depth)
        try:
            for n in (0, 1, 63, 1025, 16384 + 5, 70001):
                vals = rng.integers(0, 255, n).astype(dtype) if np.dtype(dtype).kind != "f" else rng.standard_normal(n).astype(dtype)
                if dtype == np.int64:
                    vals = rng.integers(-2**62, 2**62, n, dtype=np.int64)
                for sel_p in (0.0, 0.03, 0.5, 1.0):
                    for voff, foff in ((0, 0), (3, 13)):
                        fdata = np.packbits(rng.random(foff + n + 64) < sel_p, bitorder="little")
                        for vvalid, fvalid in ((None, None), (np.packbits(rng.random(voff + n + 64) < 0.9, bitorder="little"), np.packbits(rng.random(foff + n + 64) < 0.9, bitorder="little"))):
                            want_valid = vvalid is not None
                            for null_sel in (DROP, EMIT):
                                e = orc_be.filter(vals, vvalid, voff, fdata, fvalid, foff, n, null_sel, want_valid)
                                k = ing.filter_count(fdata, fvalid, foff, n, null_sel)
                                assert k == len(e[0]), (n, sel_p, voff, foff, null_sel)
                                out = np.full(k + 8, 0x33, dtype=dtype)
                                ov = np.full((k + 7) // 8 + 8, 0xCC, np.uint8) if want_valid else None
                                nulls = ing.filter_primitive(w, vals, vvalid, voff, n, k, out, ov)
                                assert out[:k].tobytes() == e[0].tobytes(), (dtype, n, sel_p, voff, foff, null_sel, chunk_kib)
                                if want_valid:
                                    assert ov[:(k + 7) // 8].tobytes() == e[1].tobytes() and nulls == e[2], (dtype, n, sel_p, voff, foff, null_sel)
        finally:
            ing.close()


def test_ingest_filter_needs_its_count(ctx):
    import arrow_go_amd as ah
    ing = ah.Ingest(ctx, 1 << 16, 2)
    try:
        with pytest.raises(ah.ErrInvalid):
            ing.filter_primitive(8, np.zeros(100, np.int64), None, 0, 100, 5, np.zeros(5, np.int64), None)
        k = ing.filter_count(np.full(20, 0xFF, np.uint8), None, 0, 100, DROP)
        assert k == 100
        with pytest.raises(ah.ErrInvalid):
            ing.filter_primitive(8, np.zeros(100, np.int64), None, 0, 100, 7, np.zeros(100, np.int64), None)   # n_out ≠ the count
    finally:
        ing.close()
    with pytest.raises(ah.ErrInvalid):
        ah.Ingest(ctx, 1000, 3)       # not a multiple of 4096
    with pytest.raises(ah.ErrInvalid):
        ah.Ingest(ctx, 1 << 16, 9)
