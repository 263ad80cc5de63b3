"""Python restatement of csrc/ah_ddsum.h (TEST INFRASTRUCTURE): the four-word Float64 accumulator {s, e, bs, be} of Sum, the fused
Compare→Filter→Sum and the cross-rank combine.  Python floats are IEEE doubles and CPython never contracts a*b+c, so these are the
same operations in the same order as the C header; tests compare the header (compiled for the host by tests/ddx_harness.cc, and on the
GPU through the C ABI) with this model and with the oracle's fixed-point superaccumulator (orc_sum_float64_xreal)."""
import math
import struct

BIG_HI = 0x7BF00000       # high word (sign cleared) of 2^960
DOWN, UP = 2.0 ** -128, 2.0 ** 128


def hi_abs(x: float) -> int:
    return (struct.unpack("<Q", struct.pack("<d", x))[0] >> 32) & 0x7FFFFFFF


def _sub(a, b):
    return a - b


def dd_add(s, e, x):
    t = s + x
    bp = t - s
    e = e + ((s - (t - bp)) + (x - bp))
    return t, e


def dd_merge(s, e, os_, oe):
    t = s + os_
    bp = t - s
    e = e + (((s - (t - bp)) + (os_ - bp)) + oe)
    return t, e


def zero():
    return [0.0, 0.0, 0.0, 0.0]


def add(a, x):
    if hi_abs(x) >= BIG_HI:
        a[2], a[3] = dd_add(a[2], a[3], x * DOWN)
    else:
        a[0], a[1] = dd_add(a[0], a[1], x)
    return a


def merge(a, o):
    a[0], a[1] = dd_merge(a[0], a[1], o[0], o[1])
    a[2], a[3] = dd_merge(a[2], a[3], o[2], o[3])
    return a


def result(a) -> float:
    s, e, bs, be = a
    if bs == 0.0 and be == 0.0:
        return s + e
    if not math.isfinite(bs):
        return bs
    hs = bs + be
    bp = hs - bs
    he = (bs - (hs - bp)) + (be - bp)
    if hs == 0.0 and he == 0.0:
        return s + e
    if abs(hs) < 2.0 ** 850:
        U, V = hs * UP, he * UP
        t = U + s
        q = t - U
        err = (U - (t - q)) + (s - q)
        return t + ((err + V) + e)
    s2, e2 = s * DOWN, e * DOWN
    t = hs + s2
    q = t - hs
    err = (hs - (t - q)) + (s2 - q)
    r = t + ((err + he) + e2)
    try:
        return r * UP
    except OverflowError:        # Python raises where IEEE returns ±inf
        return math.copysign(math.inf, r)


def accumulate(values, lanes: int = 1):
    """`lanes` interleaved accumulators merged in order — the shape of the kernel's per-lane / per-workgroup tree"""
    accs = [zero() for _ in range(lanes)]
    for i, x in enumerate(values):
        add(accs[i % lanes], float(x))
    tot = zero()
    for a in accs:
        merge(tot, a)
    return tot
