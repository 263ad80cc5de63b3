"""hash + sum at 2^26 rows: expected keys per partition (groupby_keys option) swept.   python scripts/bench_gb_keys.py"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0)
hrows = 1 << 26
rng = np.random.default_rng(1)
keys = ctx.alloc(hrows * 8); vals = ctx.alloc(hrows * 8)
for off in range(0, hrows, 1 << 22):
    vals.upload(rng.standard_normal(1 << 22), off * 8)
dic = ctx.alloc((hrows + 1) * 8); sums = ctx.alloc((hrows + 1) * 8); cnts = ctx.alloc((hrows + 1) * 8)
def timed(fn, reps=3):
    fn(); ctx.sync(); ctx.event_record(1)
    for _ in range(reps): fn()
    ctx.event_record(2)
    return round(ctx.event_elapsed_ms(1, 2) / reps, 3)
res = {}
for lg, zipf in [(14, False), (16, False), (18, False), (19, False), (20, False), (21, False), (16, True), (20, True)]:
    for off in range(0, hrows, 1 << 22):
        k = (rng.zipf(1.1, 1 << 22) % (1 << lg)) if zipf else rng.integers(0, 1 << lg, 1 << 22)
        keys.upload((k.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64), off * 8)
    r = {}
    for kpp in (1280, 1700, 2100, 2500):
        ctx.set_option("groupby_keys", kpp)
        r[str(kpp)] = timed(lambda: ctx.hash_sum("f64", keys, None, 0, vals, None, 0, hrows, dic, sums, cnts))
    res[f"2^{lg}" + ("z" if zipf else "")] = r
    sys.stderr.write(f"2^{lg}{'z' if zipf else ''} {r}\n")
print(json.dumps(res))
