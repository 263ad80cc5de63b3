"""Int64 Add over 2^27-row columns: the blocks' natural interleave over the XCDs (0) against one contiguous eighth per XCD (1)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
N = ah._native
ctx = ah.Context(0); rows = 1 << 27
a = ctx.alloc(rows * 8); b = ctx.alloc(rows * 8); c = ctx.alloc(rows * 8)
rng = np.random.default_rng(1)
chunk = rng.integers(-2**62, 2**62, 1 << 22, dtype=np.int64)
for off in range(0, rows, 1 << 22): a.upload(chunk, off * 8); b.upload(chunk[::-1].copy(), off * 8)
def timed(fn, reps=20):
    fn(); ctx.sync(); ctx.event_record(10)
    for _ in range(reps): fn()
    ctx.event_record(11); return ctx.event_elapsed_ms(10, 11) / reps
res = {}
for rnd in range(3):
    for m in (0, 1):
        ctx.set_option("arith_xcd_map", m)
        ms = timed(lambda: ctx.arithmetic(N.INT64, N.OP_ADD, N.SHAPE_AA, a, b, c, rows))
        res[f"add_int64_map{m}_round{rnd}"] = {"ms": round(ms, 4), "GB/s": round(24 * rows / ms / 1e6, 1)}
        ms = timed(lambda: ctx.arithmetic(N.INT64, N.OP_ADD, N.SHAPE_AS, a, np.array([7], np.int64), c, rows))
        res[f"add_scalar_map{m}_round{rnd}"] = {"ms": round(ms, 4), "GB/s": round(16 * rows / ms / 1e6, 1)}
got = c.download(np.int64, 4096)
ctx.set_option("arith_xcd_map", 0)
print(json.dumps(res))
