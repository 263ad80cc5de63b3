"""Filter of 2^27 Int64 rows (10 % nulls), s = 0.5: the two-phase call (ah_filter_count + ah_filter_primitive) against the one-call entry
(ah_filter_primitive_once: outputs sized for n rows), both as a host language calls them (wall clock through ctypes) and by HIP events."""
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
def wall(fn, reps=20):
    fn(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    ctx.sync()
    return round((time.perf_counter() - t0) / reps * 1e3, 4)
for sel in (0.1, 0.5, 0.9):
    fmask.upload(bench.random_bits(rng, rows, sel))
    def two():
        k = ctx.filter_count(fmask, None, 0, rows, 0)
        ctx.filter_primitive(8, a, vvalid, 0, fmask, None, 0, rows, 0, k, c, ovalid, want_null_count=False)
    def once():
        ctx.filter_primitive_once(8, a, vvalid, 0, fmask, None, 0, rows, 0, c, ovalid)
    k = ctx.filter_count(fmask, None, 0, rows, 0)
    traffic = (8 + 0.25) * rows + (8 + 0.125) * k
    t2, t1 = wall(two), wall(once)
    res[f"sel{sel}"] = {"two_phase_ms": t2, "one_call_ms": t1, "two_phase_GBps": round(traffic / t2 / 1e6, 1), "one_call_GBps": round(traffic / t1 / 1e6, 1),
                        "one_call_frac_of_8TBps": round(traffic / t1 / 1e6 / 8000, 3)}
print(json.dumps(res))
