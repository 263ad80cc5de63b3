"""sort_indices throughput (row §8(f)-2)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
N = ah._native
ctx = ah.Context(0)
rng = np.random.default_rng(9)
res = {}
def timed(fn, reps=3):
    fn(); ctx.sync(); ctx.event_record(10)
    for _ in range(reps): fn()
    ctx.event_record(11); return ctx.event_elapsed_ms(10, 11) / reps
for lg in (20, 24, 27):
    rows = 1 << lg
    a = ctx.alloc(rows * 8); out = ctx.alloc(rows * 8)
    chunk = rng.integers(-2**62, 2**62, min(rows, 1 << 22), dtype=np.int64)
    for off in range(0, rows, chunk.size): a.upload(chunk + off, off * 8)
    ms = timed(lambda: ctx.sort_indices(N.INT64, a, None, 0, rows, False, False, out))
    res[f"int64 random 2^{lg}"] = {"ms": round(ms, 3), "Mrows_per_s": round(rows / ms / 1e3)}
    ctx.set_option("sort_msd", 0)
    res[f"int64 random 2^{lg}"]["ms_lsd_only"] = round(timed(lambda: ctx.sort_indices(N.INT64, a, None, 0, rows, False, False, out)), 3)
    ctx.set_option("sort_msd", 1)
    if lg >= 24:
        f = rng.standard_normal(min(rows, 1 << 22))
        for off in range(0, rows, f.size): a.upload(f + off * 1e-9, off * 8)
        ms = timed(lambda: ctx.sort_indices(N.FLOAT64, a, None, 0, rows, False, False, out))
        res[f"float64 normal 2^{lg}"] = {"ms": round(ms, 3), "Mrows_per_s": round(rows / ms / 1e3)}
        ctx.set_option("sort_msd", 0)
        res[f"float64 normal 2^{lg}"]["ms_lsd_only"] = round(timed(lambda: ctx.sort_indices(N.FLOAT64, a, None, 0, rows, False, False, out)), 3)
        ctx.set_option("sort_msd", 1)
        for off in range(0, rows, chunk.size): a.upload(chunk + off, off * 8)
    if lg == 27:
        ms = timed(lambda: ctx.sort_indices(N.FLOAT64, a, None, 0, rows, True, False, out))
        res["float64(bits) desc 2^27"] = {"ms": round(ms, 3), "Mrows_per_s": round(rows / ms / 1e3)}
        small = rng.integers(0, 100000, 1 << 22, dtype=np.int64)
        for off in range(0, rows, small.size): a.upload(small, off * 8)
        ms = timed(lambda: ctx.sort_indices(N.INT64, a, None, 0, rows, False, False, out))
        res["int64 in [0,1e5) 2^27 (3 varying bytes)"] = {"ms": round(ms, 3), "Mrows_per_s": round(rows / ms / 1e3)}
        ms = timed(lambda: ctx.sort_indices(N.INT32, a, None, 0, rows, False, False, out))
        res["int32 2^27"] = {"ms": round(ms, 3), "Mrows_per_s": round(rows / ms / 1e3)}
    a.free(); out.free()
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/bench_sort.json", "w"), indent=1)
