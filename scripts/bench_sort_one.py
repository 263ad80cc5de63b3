"""one sort_indices case for profiling: python scripts/bench_sort_one.py <lg> <int|normal> <msd 0|1> [unique]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
N = ah._native
lg, kind, msd = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
ctx = ah.Context(0)
ctx.set_option("sort_msd", msd)
rng = np.random.default_rng(9)
rows = 1 << lg
a = ctx.alloc(rows * 8); out = ctx.alloc(rows * 8)
step = 1 << 22
for off in range(0, rows, step):
    a.upload(rng.integers(-2**62, 2**62, step, dtype=np.int64) if kind == "int" else rng.standard_normal(step), off * 8)
t = N.INT64 if kind == "int" else N.FLOAT64
fn = lambda: ctx.sort_indices(t, a, None, 0, rows, False, False, out)
fn(); ctx.sync(); ctx.event_record(10)
for _ in range(3): fn()
ctx.event_record(11)
print(json.dumps({"lg": lg, "kind": kind, "msd": msd, "ms": round(ctx.event_elapsed_ms(10, 11) / 3, 3)}))
