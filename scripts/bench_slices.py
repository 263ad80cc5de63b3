"""Element-aligned Arrow slices (value buffers 8 bytes off a 16-byte boundary): Int64 / Int32 Add, compare, Filter on 2^26-row slices
next to the 16-byte-aligned columns."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
import bench
N = ah._native
ctx = ah.Context(0); rows = 1 << 26
rng = np.random.default_rng(1)
a = ctx.alloc(rows * 8 + 64); b = ctx.alloc(rows * 8 + 64); c = ctx.alloc(rows * 8 + 64); m = ctx.alloc(rows // 8 + 64)
chunk = rng.integers(-2**40, 2**40, 1 << 22, dtype=np.int64)
for off in range(0, rows, 1 << 22): a.upload(chunk, off * 8); b.upload(chunk[::-1].copy(), off * 8)
mask = ctx.to_device(bench.random_bits(rng, rows, 0.5))
def timed(fn, reps=10):
    fn(); ctx.sync(); ctx.event_record(10)
    for _ in range(reps): fn()
    ctx.event_record(11); return round(ctx.event_elapsed_ms(10, 11) / reps, 4)
def Off(buf, o): return buf.ptr + o
res = {}
for name, o8, o4 in (("aligned", 0, 0), ("slice", 8, 4)):
    A, B, Cc = Off(a, o8), Off(b, o8), Off(c, o8)
    res[f"add_int64_{name}_ms"] = timed(lambda: ctx.arithmetic(N.INT64, N.OP_ADD, N.SHAPE_AA, A, B, Cc, rows - 8))
    res[f"greater_int64_{name}_ms"] = timed(lambda: ctx.comparison(N.CMP_GT, N.SHAPE_AS, N.INT64, A, np.array([0], np.int64), m, rows - 8, 0))
    A4, B4, C4 = Off(a, o4), Off(b, o4), Off(c, o4)
    res[f"add_int32_{name}_ms"] = timed(lambda: ctx.arithmetic(N.INT32, N.OP_ADD, N.SHAPE_AA, A4, B4, C4, 2 * rows - 8))
    k = ctx.filter_count(mask, None, 0, rows - 8, 0)
    res[f"filter_int64_{name}_ms"] = timed(lambda: ctx.filter_primitive(8, A, None, 0, mask, None, 0, rows - 8, 0, k, Cc, None))
print(json.dumps(res))
