"""hash_sum (Float64) of 2^26 rows with few groups — the direct path: quick look → prep → aggregate → finish (four launches)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0)
n = 1 << 26
rng = np.random.default_rng(3)
keys = ctx.alloc(n * 8); vals = ctx.alloc(n * 8)
outs = [ctx.alloc((n + 1) * 8 + 64) for _ in range(4)]
vchunk = rng.uniform(-1, 1, 1 << 22)
for off in range(0, n, 1 << 22): vals.upload(vchunk, off * 8)
def timed(fn, reps=10):
    fn(); ctx.sync(); ctx.event_record(1)
    for _ in range(reps): fn()
    ctx.event_record(2); return round(ctx.event_elapsed_ms(1, 2) / reps, 4)
res = {}
for lg in (0, 4, 8, 10, 11):
    kchunk = (rng.integers(0, 1 << lg, 1 << 22, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    for off in range(0, n, 1 << 22): keys.upload(kchunk, off * 8)
    res[f"2^{lg}_groups_ms"] = timed(lambda: ctx.hash_sum("f64", keys, None, 0, vals, None, 0, n, *[o.ptr for o in outs]))
print(json.dumps(res))
