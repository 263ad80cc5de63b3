"""dictionary_encode / unique of 2^26 Int64 rows: the global-table path against the partition-first path (ah_hash_part.hip)
   across cardinalities, uniform keys + one Zipf(1.1) column.   python scripts/bench_encode_part.py [lg ...]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
import bench
ctx = ah.Context(0)
hrows = 1 << 26
rng = np.random.default_rng(1)
keys = ctx.alloc(hrows * 8); ids = ctx.alloc(hrows * 4); dic = ctx.alloc((hrows + 1) * 8)
lgs = [int(a) for a in sys.argv[1:]] or [17, 18, 19, 20, 21, 22]
res = {}
def timed(fn, reps=3):
    fn(); ctx.sync(); ctx.event_record(1)
    for _ in range(reps): fn()
    ctx.event_record(2)
    return round(ctx.event_elapsed_ms(1, 2) / reps, 3)
def run(tag):
    r = {}
    for name, mode in (("global_table", 0), ("auto", 1), ("forced_256", 8), ("forced_1024", 10), ("forced_2048", 11), ("forced_4096", 12), ("forced_8192", 13)):
        ctx.set_option("encode_partition", mode)
        r[name + "_encode_ms"] = timed(lambda: ctx.hash_u64_encode(keys, None, 0, hrows, False, ids, None, dic))
        if mode in (0, 1):
            r[name + "_unique_ms"] = timed(lambda: ctx.hash_u64_encode(keys, None, 0, hrows, False, None, None, dic))
    ctx.set_option("encode_partition", 1)
    ctx.set_option("encode_part_slots", 4096)
    r["auto_slots4096_encode_ms"] = timed(lambda: ctx.hash_u64_encode(keys, None, 0, hrows, False, ids, None, dic))
    ctx.set_option("encode_part_slots", 8192)
    res[tag] = r
for lg in lgs:
    card = 1 << lg
    for off in range(0, hrows, 1 << 22):
        keys.upload((rng.integers(0, card, 1 << 22, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64), off * 8)
    run(f"2^{lg}")
for off in range(0, hrows, 1 << 22):
    keys.upload((bench.zipf_ranks(rng, 1 << 22, 1 << 20) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64), off * 8)
run("2^20_zipf1.1")
print(json.dumps({"rows": hrows, "results": res}))
