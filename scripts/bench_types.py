#!/usr/bin/env python3
"""Bandwidth of the element-wise family across all 10 numeric types (1 GiB per operand):
unchecked add (AA / AS), checked add with 10 % nulls, compare → bitmap, abs."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
N = ah._native
ctx = ah.Context(0)
nbytes = 1 << 30
a = ctx.alloc(nbytes); b = ctx.alloc(nbytes); c = ctx.alloc(nbytes); m = ctx.alloc(nbytes // 8 + 64)
rng = np.random.default_rng(0)
chunk = rng.integers(0, 50, 1 << 24, dtype=np.uint8)  # small values: no checked overflow in any width
for off in range(0, nbytes, 1 << 24): a.upload(chunk, off); b.upload(chunk, off)
vb = np.packbits(rng.random(1 << 22) < 0.9, bitorder="little")
lv = ctx.alloc((1 << 30) // 8 + 64)
for off in range(0, (1 << 30) // 8, vb.size): lv.upload(vb, off)
def timed(fn, reps=10):
    fn(); ctx.event_record(1)
    for _ in range(reps): fn()
    ctx.event_record(2); return ctx.event_elapsed_ms(1, 2) / reps
out = {}
types = {"uint8": (N.UINT8, 1), "int8": (N.INT8, 1), "uint16": (N.UINT16, 2), "int16": (N.INT16, 2), "uint32": (N.UINT32, 4),
         "int32": (N.INT32, 4), "uint64": (N.UINT64, 8), "int64": (N.INT64, 8), "float32": (N.FLOAT32, 4), "float64": (N.FLOAT64, 8)}
for name, (tid, w) in types.items():
    n = nbytes // w
    r = {}
    ms = timed(lambda: ctx.arithmetic(tid, N.OP_ADD, N.SHAPE_AA, a, b, c, n)); r["add_AA_GB/s"] = round(3 * nbytes / ms / 1e6)
    ms = timed(lambda: ctx.arithmetic(tid, N.OP_MUL, N.SHAPE_AS, a, np.array([3]).astype(name), c, n)); r["mul_AS_GB/s"] = round(2 * nbytes / ms / 1e6)
    ms = timed(lambda: ctx.arithmetic_unary(tid, N.OP_ABS, a, c, n)); r["abs_GB/s"] = round(2 * nbytes / ms / 1e6)
    ms = timed(lambda: ctx.comparison(N.CMP_GT, N.SHAPE_AS, tid, a, np.array([50]).astype(name), m, n, 0)); r["gt_AS_GB/s"] = round((nbytes + n / 8) / ms / 1e6)
    ms = timed(lambda: ctx.comparison(N.CMP_EQ, N.SHAPE_AA, tid, a, b, m, n, 0)); r["eq_AA_GB/s"] = round((2 * nbytes + n / 8) / ms / 1e6)
    if "float" not in name:
        ms = timed(lambda: ctx.arithmetic_checked(tid, N.OP_ADD_CHECKED, N.SHAPE_AA, a, lv, 0, b, None, 0, True, c, n), reps=3)
        r["add_checked_nulls10_GB/s"] = round((3 * nbytes + n / 8) / ms / 1e6)
    out[name] = r
    print(name, r, flush=True)
json.dump(out, open("gpurun_out/bench_types.json", "w"), indent=1)
