"""is_in throughput at 2^27 Int64 rows (row §8(f)-2).  Algorithmic bytes: 8 + 2/8 per row."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0); rows = 1 << 27
rng = np.random.default_rng(3)
a = ctx.alloc(rows * 8); od = ctx.alloc(rows // 8 + 64); ov = ctx.alloc(rows // 8 + 64)
def timed(fn, reps=5):
    fn(); ctx.sync(); ctx.event_record(10)
    for _ in range(reps): fn()
    ctx.event_record(11); return ctx.event_elapsed_ms(10, 11) / reps
res = {}
for set_n in (16, 1024, 2048, 4096, 1 << 16, 1 << 20, 1 << 24):
    universe = 4 * set_n
    chunk = rng.integers(0, universe, 1 << 22, dtype=np.int64)
    for off in range(0, rows, 1 << 22): a.upload(chunk, off * 8)
    vs = ctx.to_device(rng.choice(universe, set_n, replace=False).astype(np.int64))
    ms = timed(lambda: ctx.is_in(8, a, None, 0, rows, vs, None, 0, set_n, 0, od, ov, 0))
    hit = ctx.count_set_bits(od, 0, rows) / rows
    res[f"int64 set={set_n}"] = {"ms": round(ms, 4), "GBps": round(8.25 * rows / ms / 1e6), "hit_rate": round(hit, 3)}
for w, name in ((1, "int8"), (2, "int16"), (4, "int32")):
    vs = ctx.to_device(np.arange(0, 100, 3).astype({1: np.int8, 2: np.int16, 4: np.int32}[w]))
    ms = timed(lambda: ctx.is_in(w, a, None, 0, rows, vs, None, 0, 34, 0, od, ov, 0))
    res[f"{name} set=34"] = {"ms": round(ms, 4), "GBps": round((w + 0.25) * rows / ms / 1e6)}
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/bench_isin.json", "w"), indent=1)
