"""hash_sum (Float64 values, Int64 keys) of 2^26 rows at 2^12 … 2^24 groups (+ Zipf): the partition-first path, auto choice."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0)
n = 1 << 26
rng = np.random.default_rng(1)
keys = ctx.alloc(n * 8); vals = ctx.alloc(n * 8)
outs = [ctx.alloc((n + 1) * 8 + 64) for _ in range(4)]
vchunk = rng.standard_normal(1 << 22)
for off in range(0, n, 1 << 22): vals.upload(vchunk, off * 8)
def timed(fn, reps=5):
    fn(); ctx.sync(); ctx.event_record(1)
    for _ in range(reps): fn()
    ctx.event_record(2); return round(ctx.event_elapsed_ms(1, 2) / reps, 4)
res = {}
for lg, zipf in ((12, False), (16, False), (18, False), (20, False), (20, True), (24, False)):
    for off in range(0, n, 1 << 22):
        k = (rng.zipf(1.1, 1 << 22) % (1 << lg)) if zipf else rng.integers(0, 1 << lg, 1 << 22)
        keys.upload((k.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64), off * 8)
    r = {}
    for reserve in (0, 1):   # histogram → offsets → scatter against the reserving scatter (ah_partition.h 1b)
        ctx.set_option("groupby_reserve", reserve)
        r[f"f64_reserve{reserve}_ms"] = timed(lambda: ctx.hash_sum("f64", keys, None, 0, vals, None, 0, n, *[o.ptr for o in outs]), reps=10)
        r[f"i64_reserve{reserve}_ms"] = timed(lambda: ctx.hash_sum("i64", keys, None, 0, vals, None, 0, n, *[o.ptr for o in outs]))
    res[f"2^{lg}{'_zipf' if zipf else ''}"] = r
print(json.dumps(res))
