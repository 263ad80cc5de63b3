"""hash_sum Float64 at 2^26 rows for the given cardinalities (log2): ms per call, 6 calls after 3 warm-ups."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0)
hrows = 1 << 26
rng = np.random.default_rng(1)
keys = ctx.alloc(hrows * 8); vals = ctx.alloc(hrows * 8)
dic = ctx.alloc((hrows + 1) * 8); sums = ctx.alloc((hrows + 1) * 8); cnts = ctx.alloc((hrows + 1) * 8)
for off in range(0, hrows, 1 << 22):
    vals.upload(rng.uniform(-1, 1, 1 << 22), off * 8)
res = {}
for lg in [int(a) for a in sys.argv[1:]] or [16]:
    for off in range(0, hrows, 1 << 22):
        keys.upload((rng.integers(0, 1 << lg, 1 << 22, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64), off * 8)
    for _ in range(3): ctx.hash_sum("f64", keys, None, 0, vals, None, 0, hrows, dic, sums, cnts)
    ctx.sync(); ctx.event_record(1)
    for _ in range(6): ctx.hash_sum("f64", keys, None, 0, vals, None, 0, hrows, dic, sums, cnts)
    ctx.event_record(2)
    res[f"sum_f64 2^{lg}"] = round(ctx.event_elapsed_ms(1, 2) / 6, 3)
print(json.dumps(res))
