"""numeric cast throughput at 2^27 rows (row §8(f)-2).  Algorithmic bytes: w_in + w_out per row."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
N = ah._native
ctx = ah.Context(0); rows = 1 << 27
a = ctx.alloc(rows * 8); c = ctx.alloc(rows * 8); v = ctx.alloc(rows // 8 + 64)
a.memset(0)  # zeros: in range for every target, finite as floats
v.memset(0xEE)
def timed(fn, reps=5):
    fn(); ctx.sync(); ctx.event_record(10)
    for _ in range(reps): fn()
    ctx.event_record(11); return ctx.event_elapsed_ms(10, 11) / reps
W = {N.INT8: 1, N.INT16: 2, N.INT32: 4, N.INT64: 8, N.UINT8: 1, N.UINT32: 4, N.UINT64: 8, N.FLOAT32: 4, N.FLOAT64: 8}
NAME = {N.INT8: "i8", N.INT16: "i16", N.INT32: "i32", N.INT64: "i64", N.UINT8: "u8", N.UINT32: "u32", N.UINT64: "u64", N.FLOAT32: "f32", N.FLOAT64: "f64"}
res = {}
pairs = [(N.INT32, N.INT64), (N.INT64, N.INT32), (N.INT64, N.FLOAT64), (N.FLOAT64, N.INT64), (N.FLOAT64, N.FLOAT32), (N.FLOAT32, N.FLOAT64),
         (N.INT8, N.INT64), (N.INT64, N.INT8), (N.INT32, N.FLOAT32), (N.UINT8, N.INT16), (N.INT64, N.UINT64)]
for fi, to in pairs:
    for safe in (False, True):
        ms = timed(lambda: ctx.cast_numeric(fi, to, a, None, 0, rows, not safe, not safe, c))
        res[f"{NAME[fi]}->{NAME[to]}{' safe' if safe else ''}"] = {"ms": round(ms, 4), "GBps": round((W[fi] + W[to]) * rows / ms / 1e6)}
ms = timed(lambda: ctx.cast_numeric(N.INT64, N.INT32, a, v, 3, rows, False, False, c))
res["i64->i32 safe, validity"] = {"ms": round(ms, 4), "GBps": round(12.125 * rows / ms / 1e6)}
ms = timed(lambda: ctx.cast_bool_to_numeric(N.INT64, v, 0, rows, c))
res["bool->i64"] = {"ms": round(ms, 4), "GBps": round(8.125 * rows / ms / 1e6)}
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/bench_cast.json", "w"), indent=1)
