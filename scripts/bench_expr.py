#!/usr/bin/env python3
"""(a + b) * c > t over 2^27 int64 rows: one fused JIT kernel vs the three-kernel chain."""
import json, os, struct, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
N = ah._native
ctx = ah.Context(0); rows = 1 << 27
rng = np.random.default_rng(0)
cols = [ctx.alloc(rows * 8) for _ in range(3)]
chunk = rng.integers(-1000, 1000, 1 << 22, dtype=np.int64)
for b in cols:
    for off in range(0, rows, 1 << 22): b.upload(chunk, off * 8)
t1 = ctx.alloc(rows * 8); t2 = ctx.alloc(rows * 8); mask = ctx.alloc(rows // 8 + 64); mask2 = ctx.alloc(rows // 8 + 64)
vb = np.packbits(rng.random(1 << 22) < 0.9, bitorder="little")
valid = ctx.alloc(rows // 8 + 64); ovalid = ctx.alloc(rows // 8 + 64)
for off in range(0, rows // 8, vb.size): valid.upload(vb, off)
def timed(fn, reps=10):
    fn(); ctx.sync(); ctx.event_record(1)
    for _ in range(reps): fn()
    ctx.event_record(2); return ctx.event_elapsed_ms(1, 2) / reps
thr = np.array([5], np.int64)
prog = [(N.X_FIELD, 0), (N.X_FIELD, 1), (N.X_ADD, 0), (N.X_FIELD, 2), (N.X_MUL, 0), (N.X_LITERAL, 0), (N.X_GT, 0)]
h, ot = ctx.expr_compile(prog, [N.INT64] * 3, [N.INT64])
def chain():
    ctx.arithmetic(N.INT64, N.OP_ADD, N.SHAPE_AA, cols[0], cols[1], t1, rows)
    ctx.arithmetic(N.INT64, N.OP_MUL, N.SHAPE_AA, t1, cols[2], t2, rows)
    ctx.comparison(N.CMP_GT, N.SHAPE_AS, N.INT64, t2, thr, mask, rows, 0)
def fused(): ctx.expr_execute(h, cols, [None] * 3, [0] * 3, [struct.pack("<q", 5)], [1], rows, mask2, None)
def fused_nulls(): ctx.expr_execute(h, cols, [valid, None, valid], [0, 0, 3], [struct.pack("<q", 5)], [1], rows, mask2, ovalid)
out = {}
ms = timed(chain); out["chain_add_mul_greater"] = {"ms": round(ms, 4), "traffic_GB/s": round((24 + 24 + 8.125) * rows / ms / 1e6, 1)}
ms = timed(fused); out["fused_one_kernel"] = {"ms": round(ms, 4), "traffic_GB/s": round((24 + 0.125) * rows / ms / 1e6, 1)}
ms = timed(fused_nulls); out["fused_one_kernel_2_validity_bitmaps"] = {"ms": round(ms, 4), "traffic_GB/s": round((24 + 0.5) * rows / ms / 1e6, 1)}
assert mask.download(np.uint8, 1 << 20).tobytes() == mask2.download(np.uint8, 1 << 20).tobytes() or True
chain(); fused(); ctx.sync()
assert mask.download(np.uint8, rows // 8).tobytes() == mask2.download(np.uint8, rows // 8).tobytes(), "fused != chain"
out["speedup"] = round(out["chain_add_mul_greater"]["ms"] / out["fused_one_kernel"]["ms"], 2)
print(json.dumps(out, indent=1))
