"""divide / bit-wise / shift / sqrt kernels at 2^27 rows (algorithmic GB/s)"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
N = ah._native
ctx = ah.Context(0)
rows = 1 << 27
a = ctx.alloc(rows * 8); b = ctx.alloc(rows * 8); c = ctx.alloc(rows * 8)
rng = np.random.default_rng(0)
chunk = rng.integers(1, 1 << 40, 1 << 22, dtype=np.int64) | np.int64(0x0000000100000001)  # no zero divisor in the int32 view either
for off in range(0, rows, 1 << 22):
    a.upload(chunk, off * 8); b.upload(chunk[::-1].copy(), off * 8)
res = {}
def timed(name, nbytes, fn, reps=10):
    fn(); ctx.sync(); ctx.event_record(1)
    for _ in range(reps): fn()
    ctx.event_record(2)
    ms = ctx.event_elapsed_ms(1, 2) / reps
    res[name] = {"ms": round(ms, 4), "GB/s": round(nbytes / ms / 1e6, 1)}
for name, t, w in (("int64", N.INT64, 8), ("int32", N.INT32, 4), ("float64", N.FLOAT64, 8), ("float32", N.FLOAT32, 4)):
    n = rows * 8 // w
    timed(f"divide_{name}", 3 * w * n, lambda: ctx.arithmetic_ext(t, N.OP_DIV, N.SHAPE_AA, a, None, 0, b, None, 0, True, c, n))
    if name.startswith("int"):
        timed(f"bit_wise_xor_{name}", 3 * w * n, lambda: ctx.arithmetic_ext(t, N.OP_BIT_XOR, N.SHAPE_AA, a, None, 0, b, None, 0, True, c, n))
        timed(f"shift_right_{name}_scalar", 2 * w * n, lambda: ctx.arithmetic_ext(t, N.OP_SHIFT_RIGHT_CHECKED, N.SHAPE_AS, a, None, 0, np.array([3], f"i{w}"), None, 0, True, c, n))
        timed(f"abs_checked_{name}", 2 * w * n, lambda: ctx.arithmetic_ext(t, N.OP_ABS_CHECKED, N.SHAPE_AS, a, None, 0, None, None, 0, True, c, n))
    else:
        timed(f"sqrt_{name}", 2 * w * n, lambda: ctx.arithmetic_ext(t, N.OP_SQRT, N.SHAPE_AS, a, None, 0, None, None, 0, True, c, n))
print(json.dumps(res))
