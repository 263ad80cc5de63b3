"""cumulative_sum Float64 (and Int64 for scale) over 2^27 rows: ms per call."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
N = ah._native
ctx = ah.Context(0)
rows = 1 << 27
rng = np.random.default_rng(2)
x = ctx.alloc(rows * 8); out = ctx.alloc(rows * 8)
for off in range(0, rows, 1 << 22):
    x.upload(rng.uniform(-1e6, 1e6, 1 << 22), off * 8)
res = {}
for name, typ, opt in (("float64", N.FLOAT64, 1), ("float64_two_pass", N.FLOAT64, 5), ("int64", N.INT64, 1), ("float64_again", N.FLOAT64, 1)):
    ctx.set_option("scan_onepass", opt)
    for _ in range(5): ctx.cumulative_sum(typ, x, None, 0, rows, None, False, False, out, None)
    ctx.sync(); ctx.event_record(1)
    for _ in range(10): ctx.cumulative_sum(typ, x, None, 0, rows, None, False, False, out, None)
    ctx.event_record(2)
    res[name] = round(ctx.event_elapsed_ms(1, 2) / 10, 4)
ctx.set_option("scan_onepass", 1)
print(json.dumps(res))
