"""hash + sum group-by (C5) at 2^26 rows: partition-first path (ah_groupby.hip) against the id-based path, and
dictionary_encode with / without the re-packed table.   python scripts/bench_groupby.py [--quick]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
quick = "--quick" in sys.argv
ctx = ah.Context(0)
hrows = 1 << 26
rng = np.random.default_rng(1)
keys = ctx.alloc(hrows * 8); vals = ctx.alloc(hrows * 8)
for off in range(0, hrows, 1 << 22):
    vals.upload(rng.standard_normal(1 << 22), off * 8)
ids = ctx.alloc(hrows * 4)
dic = ctx.alloc((hrows + 1) * 8); sums = ctx.alloc((hrows + 1) * 8); cnts = ctx.alloc((hrows + 1) * 8)
def timed(fn, reps=3):
    fn(); ctx.sync(); ctx.event_record(1)
    for _ in range(reps): fn()
    ctx.event_record(2)
    return round(ctx.event_elapsed_ms(1, 2) / reps, 3)
def fill(card, zipf):
    for off in range(0, hrows, 1 << 22):
        k = (rng.zipf(1.1, 1 << 22) % card) if zipf else rng.integers(0, card, 1 << 22)
        keys.upload((k.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64), off * 8)
res = {}
only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
quick = quick or only is not None
cases = [(int(only.rstrip("z")), only.endswith("z"))] if only else [(10, False), (12, False), (16, False), (20, False), (20, True)] if quick else [(10, False), (12, False), (13, False), (14, False), (16, False), (18, False), (20, False), (21, False), (22, False), (24, False), (16, True), (20, True)]
for lg, zipf in cases:
    fill(1 << lg, zipf)
    r = {}
    for kind in ("f64", "i64"):
        hs = lambda: ctx.hash_sum(kind, keys, None, 0, vals, None, 0, hrows, dic, sums, cnts)
        ctx.set_option("groupby_partition", 0); r[f"{kind}_idbased_ms"] = timed(hs)
        ctx.set_option("groupby_partition", 1); r[f"{kind}_auto_ms"] = timed(hs)
        r[f"{kind}_groups"] = int(hs()[0])
    if not zipf and lg <= 12:
        ctx.set_option("groupby_partition", -2); r["f64_direct_ms"] = timed(lambda: ctx.hash_sum("f64", keys, None, 0, vals, None, 0, hrows, dic, sums, cnts))
        ctx.set_option("groupby_partition", 1)
    if not quick and not zipf and lg >= 21:
        ctx.set_option("groupby_partition", 2); r["f64_sorted_ms"] = timed(lambda: ctx.hash_sum("f64", keys, None, 0, vals, None, 0, hrows, dic, sums, cnts))
        ctx.set_option("groupby_partition", 1)
    if not quick and not zipf and lg in (16, 20):
        for lp in ((6, 8, 9) if lg == 16 else (9, 10)):
            ctx.set_option("groupby_partition", lp + 2); r[f"f64_P2^{lp}_ms"] = timed(lambda: ctx.hash_sum("f64", keys, None, 0, vals, None, 0, hrows, dic, sums, cnts))
        ctx.set_option("groupby_partition", 1)
    if only:
        res[f"2^{lg}" + ("_zipf1.1" if zipf else "")] = r
        continue
    enc = lambda: ctx.hash_u64_encode(keys, None, 0, hrows, False, ids, None, dic)
    ctx.set_option("hash_direct", 3); r["encode_sparse_table_ms"] = timed(enc)
    ctx.set_option("hash_direct", 2); r["encode_ms"] = timed(enc)
    res[f"2^{lg}" + ("_zipf1.1" if zipf else "")] = r
print(json.dumps({"rows": hrows, "results": res}))
