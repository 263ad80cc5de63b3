"""hash_sum / dictionary_encode at 2^26 rows, many groups (2^20 … 2^24): per-call ms (events, 5 calls after 3 warm-ups)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0)
hrows = 1 << 26
rng = np.random.default_rng(1)
keys = ctx.alloc(hrows * 8); vals = ctx.alloc(hrows * 8)
dic = ctx.alloc((hrows + 1) * 8); sums = ctx.alloc((hrows + 1) * 8); cnts = ctx.alloc((hrows + 1) * 8); ids = ctx.alloc(hrows * 4)
for off in range(0, hrows, 1 << 22):
    vals.upload(rng.uniform(-1, 1, 1 << 22), off * 8)
res = {}
lgs = [int(a) for a in sys.argv[1:]] or [16, 20, 22, 24]
for lg in lgs:
    card = 1 << lg
    for off in range(0, hrows, 1 << 22):
        keys.upload((rng.integers(0, card, 1 << 22, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64), off * 8)
    for kind in ("f64", "i64"):
        for _ in range(3): ctx.hash_sum(kind, keys, None, 0, vals, None, 0, hrows, dic, sums, cnts)
        ctx.sync(); ctx.event_record(1)
        for _ in range(5): ng, _ = ctx.hash_sum(kind, keys, None, 0, vals, None, 0, hrows, dic, sums, cnts)
        ctx.event_record(2)
        res[f"sum_{kind} 2^{lg}"] = round(ctx.event_elapsed_ms(1, 2) / 5, 3)
    if os.environ.get("ENC", "1") == "1":
        for _ in range(3): ctx.hash_u64_encode(keys, None, 0, hrows, False, ids, None, dic)
        ctx.sync(); ctx.event_record(1)
        for _ in range(5): ctx.hash_u64_encode(keys, None, 0, hrows, False, ids, None, dic)
        ctx.event_record(2)
        res[f"encode 2^{lg}"] = round(ctx.event_elapsed_ms(1, 2) / 5, 3)
print(json.dumps(res))
