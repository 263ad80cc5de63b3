"""dictionary_encode at 2^26 rows for the given cardinalities (log2): ms per call, 5 calls after 3 warm-ups."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0)
hrows = 1 << 26
rng = np.random.default_rng(1)
keys = ctx.alloc(hrows * 8); dic = ctx.alloc((hrows + 1) * 8); ids = ctx.alloc(hrows * 4)
res = {}
for lg in [int(a) for a in sys.argv[1:]] or [20]:
    for off in range(0, hrows, 1 << 22):
        keys.upload((rng.integers(0, 1 << lg, 1 << 22, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64), off * 8)
    for _ in range(3): ctx.hash_u64_encode(keys, None, 0, hrows, False, ids, None, dic)
    ctx.sync(); ctx.event_record(1)
    for _ in range(5): ctx.hash_u64_encode(keys, None, 0, hrows, False, ids, None, dic)
    ctx.event_record(2)
    res[f"encode 2^{lg}"] = round(ctx.event_elapsed_ms(1, 2) / 5, 3)
print(json.dumps(res))
