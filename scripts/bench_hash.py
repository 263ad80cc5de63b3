"""dictionary_encode / unique / hash_sum at 2^26 Int64 rows across cardinalities (G rows/s).
   python scripts/bench_hash.py [lg ...]   (default 10 13 16 20 22 24)"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0)
hrows = 1 << 26
rng = np.random.default_rng(1)
keys = ctx.alloc(hrows * 8); vals = ctx.alloc(hrows * 8); vals.memset(0)
ids = ctx.alloc(hrows * 4)
dic = ctx.alloc((hrows + 1) * 8); sums = ctx.alloc((hrows + 1) * 8); cnts = ctx.alloc((hrows + 1) * 8)
lgs = [int(a) for a in sys.argv[1:]] or [10, 13, 16, 20, 22, 24]
res = {}
def timed(fn, reps=3):
    fn(); ctx.sync(); ctx.event_record(1)
    for _ in range(reps): fn()
    ctx.event_record(2)
    return ctx.event_elapsed_ms(1, 2) / reps
for lg in lgs:
    card = 1 << lg
    for off in range(0, hrows, 1 << 22):
        keys.upload((rng.integers(0, card, 1 << 22, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64), off * 8)
    r = {}
    r["encode_ms"] = round(timed(lambda: ctx.hash_u64_encode(keys, None, 0, hrows, False, ids, None, dic)), 3)
    r["unique_ms"] = round(timed(lambda: ctx.hash_u64_encode(keys, None, 0, hrows, False, None, None, dic)), 3)
    r["hash_sum_f64_ms"] = round(timed(lambda: ctx.hash_sum("f64", keys, None, 0, vals, None, 0, hrows, dic, sums, cnts)), 3)
    r["encode_Grows/s"] = round(hrows / r["encode_ms"] / 1e6, 2)
    r["hash_sum_Grows/s"] = round(hrows / r["hash_sum_f64_ms"] / 1e6, 2)
    res[f"2^{lg}"] = r
print(json.dumps({"rows": hrows, "results": res}))
