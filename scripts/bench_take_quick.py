"""random Take of 2^27 Int64 values by 2^27 int32 indices, with and without 10 % nulls: the binned path in a few seconds.
   python scripts/bench_take_quick.py"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0); rows = 1 << 27
rng = np.random.default_rng(1)
a = ctx.alloc(rows * 8); c = ctx.alloc(rows * 8); idx = ctx.alloc(rows * 4)
vv = ctx.alloc(rows // 8 + 64); iv = ctx.alloc(rows // 8 + 64); ov = ctx.alloc(rows // 8 + 64)
chunk = rng.integers(-2**62, 2**62, 1 << 22, dtype=np.int64)
for off in range(0, rows, 1 << 22):
    a.upload(chunk, off * 8)
    idx.upload(rng.integers(0, rows, 1 << 22, dtype=np.int64).astype(np.int32), off * 4)
for b in (vv, iv):
    for off in range(0, rows // 8, 1 << 19):
        b.upload(np.packbits(rng.random(1 << 22) >= 0.1, bitorder="little"), off)
def timed(fn, reps=3):
    fn(); ctx.sync(); ctx.event_record(10)
    for _ in range(reps): fn()
    ctx.event_record(11); return round(ctx.event_elapsed_ms(10, 11) / reps, 3)
res = {"random_ms": timed(lambda: ctx.take_primitive(8, a, None, 0, rows, 4, True, idx, None, 0, rows, True, c, None)),
       "random_nulls10_ms": timed(lambda: ctx.take_primitive(8, a, vv, 0, rows, 4, True, idx, iv, 0, rows, True, c, ov))}
print(json.dumps(res))
