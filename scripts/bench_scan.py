"""cumulative_sum throughput at 2^27 rows (row §8(f)-2).  Algorithmic bytes: w read + w written per row."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
N = ah._native
ctx = ah.Context(0); rows = 1 << 27
rng = np.random.default_rng(1)
a = ctx.alloc(rows * 8); c = ctx.alloc(rows * 8); v = ctx.alloc(rows // 8 + 64); ov = ctx.alloc(rows // 8 + 64)
chunk = rng.integers(-2**40, 2**40, 1 << 22, dtype=np.int64)
for off in range(0, rows, 1 << 22): a.upload(chunk, off * 8)
vb = np.packbits(rng.random(1 << 22) >= 0.1, bitorder="little")
for off in range(0, rows // 8, len(vb)): v.upload(vb, off)
def timed(fn, reps=5):
    fn(); ctx.sync(); ctx.event_record(10)
    for _ in range(reps): fn()
    ctx.event_record(11); return ctx.event_elapsed_ms(10, 11) / reps
res = {}
T = {"int64": (N.INT64, 8), "int32": (N.INT32, 4), "int16": (N.INT16, 2), "int8": (N.INT8, 1), "float64": (N.FLOAT64, 8), "float32": (N.FLOAT32, 4)}
for name, (tid, w) in T.items():
    ms = timed(lambda: ctx.cumulative_sum(tid, a, None, 0, rows, None, False, False, c, None))
    res[name] = {"ms": round(ms, 4), "GBps": round(2 * w * rows / ms / 1e6, 1)}
ms = timed(lambda: ctx.cumulative_sum(N.INT64, a, None, 0, rows, None, False, True, c, None))
res["int64_checked"] = {"ms": round(ms, 4), "GBps": round(16 * rows / ms / 1e6, 1)}
z = ctx.alloc(rows * 4); z.memset(0)
ms = timed(lambda: ctx.cumulative_sum(N.INT32, z, None, 0, rows, None, False, True, c, None))
res["int32_checked"] = {"ms": round(ms, 4), "GBps": round(8 * rows / ms / 1e6, 1)}
ms = timed(lambda: ctx.cumulative_sum(N.INT64, a, v, 0, rows, None, True, False, c, ov))
res["int64_nulls_skip"] = {"ms": round(ms, 4), "GBps": round((16 + 0.25) * rows / ms / 1e6, 1)}
ms = timed(lambda: ctx.cumulative_sum(N.INT64, a, v, 0, rows, None, False, False, c, ov))
res["int64_nulls_propagate"] = {"ms": round(ms, 4)}
ms = timed(lambda: ctx.cumulative_sum(N.INT64, a, v, 0, rows, None, True, True, c, ov))
res["int64_checked_nulls_skip"] = {"ms": round(ms, 4), "GBps": round((16 + 0.25) * rows / ms / 1e6, 1)}
# … and the reduce-then-scan path on the same columns (option scan_onepass 0): what rounds 1-4 ran for checked / null-carrying columns
ctx.set_option("scan_onepass", 0)
for key, fn in (("int64", lambda: ctx.cumulative_sum(N.INT64, a, None, 0, rows, None, False, False, c, None)),
                ("int64_checked", lambda: ctx.cumulative_sum(N.INT64, a, None, 0, rows, None, False, True, c, None)),
                ("int64_nulls_skip", lambda: ctx.cumulative_sum(N.INT64, a, v, 0, rows, None, True, False, c, ov)),
                ("int64_checked_nulls_skip", lambda: ctx.cumulative_sum(N.INT64, a, v, 0, rows, None, True, True, c, ov))):
    res[key + "_two_pass"] = {"ms": round(timed(fn), 4)}
# … and one pass with the validity copy + popcount as separate launches (option 2: round 5's first version)
ctx.set_option("scan_onepass", 2)
for key, fn in (("int64_nulls_skip", lambda: ctx.cumulative_sum(N.INT64, a, v, 0, rows, None, True, False, c, ov)),
                ("int64_nulls_propagate", lambda: ctx.cumulative_sum(N.INT64, a, v, 0, rows, None, False, False, c, ov)),
                ("int64_checked_nulls_skip", lambda: ctx.cumulative_sum(N.INT64, a, v, 0, rows, None, True, True, c, ov))):
    res[key + "_separate_validity"] = {"ms": round(timed(fn), 4)}
ctx.set_option("scan_onepass", 1)
# the same column through Sum for scale (8 B/row read only)
ms = timed(lambda: ctx.sum_int64(a, rows))
res["sum_int64_for_scale"] = {"ms": round(ms, 4), "GBps": round(8 * rows / ms / 1e6, 1)}
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/bench_scan.json", "w"), indent=1)
