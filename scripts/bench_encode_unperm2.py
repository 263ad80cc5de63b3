"""dictionary_encode of 2^26 rows with 2^23 / 2^24 keys (two cuts): level-2 un-permute per virtual tile (0) or per group of four (4)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0)
hrows = 1 << 26
rng = np.random.default_rng(1)
keys = ctx.alloc(hrows * 8); ids = ctx.alloc(hrows * 4); dic = ctx.alloc((hrows + 1) * 8)
def timed(fn, reps=4):
    fn(); ctx.sync(); ctx.event_record(1)
    for _ in range(reps): fn()
    ctx.event_record(2)
    return round(ctx.event_elapsed_ms(1, 2) / reps, 3)
res = {}
for lg in (24,):
    for off in range(0, hrows, 1 << 22):
        keys.upload((rng.integers(0, 1 << lg, 1 << 22, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64), off * 8)
    r = {}
    for rnd in (0, 1):
        for g in (0, 4):
            ctx.set_option("encode_unperm2_group", g)
            r[f"group{g}_round{rnd}_ms"] = timed(lambda: ctx.hash_u64_encode(keys, None, 0, hrows, False, ids, None, dic))
    res[f"2^{lg}"] = r
ctx.set_option("encode_unperm2_group", 4)
print(json.dumps(res))
