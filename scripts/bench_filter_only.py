import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0); rows = 1 << 27
rng = np.random.default_rng(1)
a = ctx.alloc(rows * 8); c = ctx.alloc(rows * 8)
chunk = rng.integers(-2**62, 2**62, 1 << 22, dtype=np.int64)
for off in range(0, rows, 1 << 22): a.upload(chunk, off * 8)
mask = ctx.alloc(rows // 8 + 64); vvalid = ctx.alloc(rows // 8 + 64); ovalid = ctx.alloc(rows // 8 + 64)
vb = np.packbits(rng.random(1 << 22) < 0.9, bitorder="little")
for off in range(0, rows // 8, vb.size): vvalid.upload(vb, off)
def timed(fn, reps=10):
    fn(); ctx.sync(); ctx.event_record(10)
    for _ in range(reps): fn()
    ctx.event_record(11); return ctx.event_elapsed_ms(10, 11) / reps
res = {}
for p in (0.1, 0.5, 0.9):
    mb = np.packbits(rng.random(1 << 22) < p, bitorder="little")
    for off in range(0, rows // 8, mb.size): mask.upload(mb, off)
    n_out = ctx.filter_count(mask, None, 0, rows, 0)
    for name, vv, ov in (("nonulls", None, None), ("nulls10", vvalid, ovalid)):
        ms = timed(lambda: ctx.filter_primitive(8, a, vv, 0, mask, None, 0, rows, 0, n_out, c, ov, want_null_count=False))
        res[f"{name}_sel{p}"] = round(ms, 4)
print(os.environ.get("ARROWHIP_LIB", "default"), json.dumps(res))
