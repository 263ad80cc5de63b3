"""power / power_unchecked through the function registry (compute.Power, arithmetic.go:929-930, 1181-1190).  The kernel itself is
pinned in tests/test_golden.py (TestPower's vectors, oracle and device) and tests/test_gpu_parity.py (random, bit-exact)."""
import pyarrow as pa
import pytest


@pytest.fixture(scope="module")
def sess():
    from arrow_go_amd import compute as ac
    s = ac.Session(0)
    yield s
    s.close()


# ---- power / power_unchecked (arithmetic_test.go:482-510) -----------------------------------------------------------
@pytest.mark.gpu
def test_power_through_the_registry(sess):
    import pyarrow.compute as pc
    from arrow_go_amd import compute as ac
    for t in [pa.int8(), pa.uint8(), pa.int16(), pa.uint16(), pa.int32(), pa.uint32(), pa.int64(), pa.uint64()]:
        a, b = pa.array([None, 2, 3, None, 5, 0, 1], t), pa.array([1, 6, 2, 5, 1, 0, 0], t)   # every result (and every square) fits int8
        for fn, ref in (("power", pc.power_checked), ("power_unchecked", pc.power)):
            assert sess.call_function(fn, [a, b]).equals(ref(a, b)), (fn, t)
            assert sess.call_function(fn, [a, pa.scalar(2, t)]).equals(ref(a, pa.scalar(2, t)))
            assert sess.call_function(fn, [pa.scalar(3, t), pa.array([None, 3, 4, None, 2], t)]).to_pylist() == [None, 27, 81, None, 9]
        mx = pa.array([2 ** (t.bit_width - (1 if pa.types.is_signed_integer(t) else 0)) - 1], t)
        with pytest.raises(ac.ErrInvalid, match="overflow"):
            sess.call_function("power", [mx, pa.array([10], t)])
        assert sess.call_function("power_unchecked", [mx, pa.array([10], t)]).to_pylist() == [1]
        if pa.types.is_signed_integer(t):
            for fn in ("power", "power_unchecked"):
                with pytest.raises(ac.ErrInvalid, match="integers to negative integer powers are not allowed"):
                    sess.call_function(fn, [pa.array([2, 3], t), pa.array([1, -1], t)])
