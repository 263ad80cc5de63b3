"""hash_sum at 2^16 groups: does the VALUE column matter (zeros vs random doubles)?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import arrow_go_amd as ah
ctx = ah.Context(0)
hrows = 1 << 26
rng = np.random.default_rng(1)
keys = ctx.alloc(hrows * 8); vals = ctx.alloc(hrows * 8)
dic = ctx.alloc((hrows + 1) * 8); sums = ctx.alloc((hrows + 1) * 8); cnts = ctx.alloc((hrows + 1) * 8)
def timed(fn, reps=3):
    fn(); ctx.sync(); ctx.event_record(1)
    for _ in range(reps): fn()
    ctx.event_record(2)
    return ctx.event_elapsed_ms(1, 2) / reps
for lg in (16, 20):
    kc = (rng.integers(0, 1 << lg, 1 << 22, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    for off in range(0, hrows, 1 << 22): keys.upload(kc, off * 8)
    for name, gen in (("zeros", lambda: np.zeros(1 << 22)), ("uniform(-1,1)", lambda: rng.uniform(-1, 1, 1 << 22)), ("normal*1e6", lambda: rng.standard_normal(1 << 22) * 1e6)):
        vc = gen()
        for off in range(0, hrows, 1 << 22): vals.upload(vc, off * 8)
        print(f"2^{lg} {name}: f64 {timed(lambda: ctx.hash_sum('f64', keys, None, 0, vals, None, 0, hrows, dic, sums, cnts)):.3f} ms   i64 {timed(lambda: ctx.hash_sum('i64', keys, None, 0, vals, None, 0, hrows, dic, sums, cnts)):.3f} ms", flush=True)
