"""dictionary_encode of 2^26 rows, 2^20 / 2^22 keys: workgroups of the resolve pass (option encode_resolve_wgs)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0)
hrows = 1 << 26
rng = np.random.default_rng(1)
keys = ctx.alloc(hrows * 8); ids = ctx.alloc(hrows * 4); dic = ctx.alloc((hrows + 1) * 8)
def timed(fn, reps=5):
    fn(); ctx.sync(); ctx.event_record(1)
    for _ in range(reps): fn()
    ctx.event_record(2)
    return round(ctx.event_elapsed_ms(1, 2) / reps, 3)
res = {}
for lg in (20, 22):
    for off in range(0, hrows, 1 << 22):
        keys.upload((rng.integers(0, 1 << lg, 1 << 22, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64), off * 8)
    r = {}
    for rnd in (0, 1):
        for w in (256, 1024, 2048, 4096, 8192):
            ctx.set_option("encode_resolve_wgs", w)
            r[f"wgs{w}_round{rnd}_ms"] = timed(lambda: ctx.hash_u64_encode(keys, None, 0, hrows, False, ids, None, dic))
    res[f"2^{lg}"] = r
ctx.set_option("encode_resolve_wgs", 1024)
print(json.dumps(res))
