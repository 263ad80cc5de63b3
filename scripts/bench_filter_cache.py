"""Filter of 2^27 Int64 rows (10 % nulls): the two-phase call with the count's tile prefixes kept for the fill (option filter_cache 1,
the default of ah_ctx_create) against the same call with the fill recounting (0), and the fill alone — HIP events and wall clock.
The masks are bench.py's (same generator, same order), so the lines can be set beside its filter_* lines."""
import os, sys, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
import bench
ctx = ah.Context(0)
rows = 1 << 27
rng = np.random.default_rng(5)
a = ctx.alloc(rows * 8 + 64); c = ctx.alloc(rows * 8 + 64)
bench.fill_random(ctx, a, rows, np.int64, 1)
vvalid = ctx.to_device(bench.random_bits(rng, rows, 0.9)); ovalid = ctx.alloc(rows // 8 + 64); fmask = ctx.alloc(rows // 8 + 64)
res = {}
def ev(fn, reps=20):
    fn(); ctx.sync(); ctx.event_record(20)
    for _ in range(reps): fn()
    ctx.event_record(21); return round(ctx.event_elapsed_ms(20, 21) / reps, 4)
def wall(fn, reps=20):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync()
    return round((time.perf_counter() - t0) / reps * 1e3, 4)
only = sys.argv[1:]   # e.g. "0.5": one selectivity (for a kernel trace)
for sel in (0.01, 0.1, 0.5, 0.9):
    fmask.upload(bench.random_bits(rng, rows, sel))
    if only and str(sel) not in only: continue
    k = ctx.filter_count(fmask, None, 0, rows, 0)
    def two():
        kk = ctx.filter_count(fmask, None, 0, rows, 0)
        ctx.filter_primitive(8, a, vvalid, 0, fmask, None, 0, rows, 0, kk, c, ovalid, want_null_count=False)
    def fill():
        ctx.filter_primitive(8, a, vvalid, 0, fmask, None, 0, rows, 0, k, c, ovalid, want_null_count=False)
    r = {"selected": round(k / rows, 4)}
    for cache in (1, 0, 1, 0):
        ctx.set_option("filter_cache", cache)
        r.setdefault(f"two_phase_cache{cache}_events_ms", []).append(ev(two))
        r.setdefault(f"two_phase_cache{cache}_wall_ms", []).append(wall(two))
    ctx.set_option("filter_cache", 1)
    r["fill_only_events_ms"] = ev(fill)
    r["count_only_events_ms"] = ev(lambda: ctx.filter_count(fmask, None, 0, rows, 0))
    res[f"sel{sel}"] = r
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/bench_filter_cache.json", "w"), indent=1)
