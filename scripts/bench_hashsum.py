"""hash_sum (group-by sum) at 2^26 rows across cardinalities; ARROWHIP_HASH_XCD=0 disables the per-XCD copies."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0)
hrows = 1 << 26
rng = np.random.default_rng(1)
keys = ctx.alloc(hrows * 8); vals = ctx.alloc(hrows * 8); vals.memset(0)
dic = ctx.alloc((hrows + 1) * 8); sums = ctx.alloc((hrows + 1) * 8); cnts = ctx.alloc((hrows + 1) * 8)
res = {}
for lg in (10, 13, 16, 18, 20, 22, 24):
    card = 1 << lg
    for off in range(0, hrows, 1 << 22):
        keys.upload((rng.integers(0, card, 1 << 22, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64), off * 8)
    for kind in ("f64", "i64"):
        ctx.hash_sum(kind, keys, None, 0, vals, None, 0, hrows, dic, sums, cnts)
        ctx.sync(); ctx.event_record(1)
        for _ in range(2): ng, _ = ctx.hash_sum(kind, keys, None, 0, vals, None, 0, hrows, dic, sums, cnts)
        ctx.event_record(2)
        res[f"{kind} 2^{lg}"] = round(ctx.event_elapsed_ms(1, 2) / 2, 2)
print(os.environ.get("ARROWHIP_HASH_XCD"), json.dumps(res))
