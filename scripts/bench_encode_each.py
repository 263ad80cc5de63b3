"""dictionary_encode of 2^26 Int64 rows over 2^lg keys: every call timed on its own (HIP events), so a call that takes another path or
waits for an allocation shows up instead of being averaged away.  python scripts/bench_encode_each.py [lg …]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0)
hrows = 1 << 26
rng = np.random.default_rng(7)
c = ctx.alloc(hrows * 8); hids = ctx.alloc(hrows * 4 + 64); hdic = ctx.alloc(hrows * 8 + 64)
res = {}
for lg in [int(a) for a in sys.argv[1:]] or [20, 24]:
    ranks = rng.integers(0, 1 << lg, hrows, dtype=np.uint64)
    c.upload(ranks * np.uint64(0x9E3779B97F4A7C15))
    ms = []
    for i in range(12):
        ctx.sync(); ctx.event_record(30)
        nd = ctx.hash_u64_encode(c, None, 0, hrows, False, hids, None, hdic)
        ctx.event_record(31); ms.append(round(ctx.event_elapsed_ms(30, 31), 3))
    res[f"2^{lg}"] = {"keys": int(nd[0]) if isinstance(nd, (tuple, list)) else int(nd), "ms_each": ms}
print(json.dumps(res))
