"""cumulative_sum over 2^27 rows vs segment size (bytes of input per reduce-then-scan segment; 0 = one segment), and what a
re-read of a just-streamed range costs (Sum over 32 MiB … 1 GiB working sets)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
N = ah._native
ctx = ah.Context(0); rows = 1 << 27
rng = np.random.default_rng(3)
a = ctx.to_device(rng.integers(-2**40, 2**40, rows, dtype=np.int64), 64)
x = ctx.to_device(rng.uniform(-1e6, 1e6, rows), 64)
c = ctx.alloc(rows * 8 + 64); vvalid = ctx.to_device(np.packbits(rng.random(rows) < 0.9, bitorder="little"), 64); ov = ctx.alloc(rows // 8 + 64)
res = ctx.alloc(64)
def timed(fn, reps=10):
    fn(); ctx.sync(); ctx.event_record(10)
    for _ in range(reps): fn()
    ctx.event_record(11); return round(ctx.event_elapsed_ms(10, 11) / reps, 4)
out = {}
for lg in (0, 24, 25, 26, 27, 28):
    ctx.set_option("scan_segment_log2", lg)
    out[f"seg2^{lg}"] = {"int64": timed(lambda: ctx.cumulative_sum(N.INT64, a, None, 0, rows, None, False, False, c, None)),
                         "float64": timed(lambda: ctx.cumulative_sum(N.FLOAT64, x, None, 0, rows, None, False, False, c, None)),
                         "int32": timed(lambda: ctx.cumulative_sum(N.INT32, a, None, 0, rows, None, False, False, c, None)),
                         "int64_nulls10_skip": timed(lambda: ctx.cumulative_sum(N.INT64, a, vvalid, 0, rows, None, True, False, c, ov))}
ctx.set_option("scan_segment_log2", 0)
mall = {}
for lg in (22, 23, 24, 25, 26, 27):
    n = 1 << lg
    ms = timed(lambda: ctx.sum_int64_dev(a, n, res), 20)
    mall[f"{n*8>>20}MiB"] = {"us": round(ms * 1e3, 1), "GBps": round(n * 8 / ms / 1e6)}
out["reread_sum"] = mall
print(json.dumps(out))
