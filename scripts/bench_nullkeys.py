"""dictionary_encode / hash + sum at 2^26 rows with and without 10 % null keys (and null values): what the validity bitmaps cost.
   python scripts/bench_nullkeys.py"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0)
hrows = 1 << 26
rng = np.random.default_rng(1)
keys = ctx.alloc(hrows * 8); vals = ctx.alloc(hrows * 8)
kv = ctx.alloc(hrows // 8 + 64); vv = ctx.alloc(hrows // 8 + 64)
for off in range(0, hrows, 1 << 22):
    vals.upload(rng.standard_normal(1 << 22), off * 8)
for b in (kv, vv):
    for off in range(0, hrows // 8, 1 << 19):
        b.upload(np.packbits(rng.random(1 << 22) >= 0.1, bitorder="little"), off)
ids = ctx.alloc(hrows * 4); idv = ctx.alloc(hrows // 8 + 64)
dic = ctx.alloc((hrows + 1) * 8); sums = ctx.alloc((hrows + 1) * 8); cnts = ctx.alloc((hrows + 1) * 8)
def timed(fn, reps=3):
    fn(); ctx.sync(); ctx.event_record(1)
    for _ in range(reps): fn()
    ctx.event_record(2)
    return round(ctx.event_elapsed_ms(1, 2) / reps, 3)
res = {}
only = int(sys.argv[sys.argv.index("--only") + 1]) if "--only" in sys.argv else None
for lg in ((only,) if only else (10, 16, 20)):
    for off in range(0, hrows, 1 << 22):
        k = rng.integers(0, 1 << lg, 1 << 22)
        keys.upload((k.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64), off * 8)
    r = {}
    if only:
        r["hash_sum_nullkeys_nullvals_ms"] = timed(lambda: ctx.hash_sum("f64", keys, kv, 0, vals, vv, 0, hrows, dic, sums, cnts))
        r["encode_nullkeys_ms"] = timed(lambda: ctx.hash_u64_encode(keys, kv, 0, hrows, False, ids, idv, dic))
        sys.stderr.write(f"2^{lg} {r}\n")
        continue
    r["encode_ms"] = timed(lambda: ctx.hash_u64_encode(keys, None, 0, hrows, False, ids, None, dic))
    r["encode_nullkeys_ms"] = timed(lambda: ctx.hash_u64_encode(keys, kv, 0, hrows, False, ids, idv, dic))
    r["unique_nullkeys_ms"] = timed(lambda: ctx.hash_u64_encode(keys, kv, 0, hrows, True, None, None, dic))
    r["hash_sum_ms"] = timed(lambda: ctx.hash_sum("f64", keys, None, 0, vals, None, 0, hrows, dic, sums, cnts))
    r["hash_sum_nullkeys_nullvals_ms"] = timed(lambda: ctx.hash_sum("f64", keys, kv, 0, vals, vv, 0, hrows, dic, sums, cnts))
    res[f"2^{lg}"] = r
    sys.stderr.write(f"2^{lg} {r}\n")
print(json.dumps({"rows": hrows, "results": res}))
