"""dictionary_encode of 2^26 Int64 rows on the partition-first path: the final un-permute one tile per workgroup (1) against four
consecutive tiles per workgroup (4), with 8192- and 4096-slot tables.   python scripts/bench_encode_unperm.py [lg ...]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0)
hrows = 1 << 26
rng = np.random.default_rng(1)
keys = ctx.alloc(hrows * 8); ids = ctx.alloc(hrows * 4); dic = ctx.alloc((hrows + 1) * 8)
lgs = [int(a) for a in sys.argv[1:]] or [19, 20, 21, 22, 24]
res = {}
def timed(fn, reps=5):
    fn(); ctx.sync(); ctx.event_record(1)
    for _ in range(reps): fn()
    ctx.event_record(2)
    return round(ctx.event_elapsed_ms(1, 2) / reps, 3)
for lg in lgs:
    card = 1 << lg
    for off in range(0, hrows, 1 << 22):
        keys.upload((rng.integers(0, card, 1 << 22, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64), off * 8)
    r = {}
    for rnd in (0, 1):
        for batch in (0, 1):
            ctx.set_option("encode_table_batch", batch)
            r[f"batch{batch}_round{rnd}_ms"] = timed(lambda: ctx.hash_u64_encode(keys, None, 0, hrows, False, ids, None, dic))
    res[f"2^{lg}"] = r
ctx.set_option("encode_table_batch", 1)
print(json.dumps({"rows": hrows, "results": res}))
