import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0); rows = 1 << 27
rng = np.random.default_rng(1)
a = ctx.alloc(rows * 8); c = ctx.alloc(rows * 8); idx = ctx.alloc(rows * 4)
chunk = rng.integers(-2**62, 2**62, 1 << 22, dtype=np.int64)
for off in range(0, rows, 1 << 22): a.upload(chunk, off * 8)
for off in range(0, rows, 1 << 22): idx.upload(rng.integers(0, rows, 1 << 22, dtype=np.int64).astype(np.int32), off * 4)
def timed(fn, reps=3):
    fn(); ctx.sync(); ctx.event_record(10)
    for _ in range(reps): fn()
    ctx.event_record(11); return ctx.event_elapsed_ms(10, 11) / reps
res = {}
for bpc in (0, 8, 16, 32):
    os.environ["ARROWHIP_BLOCKS_PER_CU"] = str(bpc)
    c2 = ah.Context(0)
    c2.take_primitive(8, a, None, 0, rows, 4, True, idx, None, 0, rows, True, c, None)
    c2.sync(); c2.event_record(1)
    for _ in range(3): c2.take_primitive(8, a, None, 0, rows, 4, True, idx, None, 0, rows, True, c, None)
    c2.event_record(2); res[f"bpc{bpc}"] = round(c2.event_elapsed_ms(1, 2) / 3, 3)
print(os.environ.get("ARROWHIP_LIB", "default"), json.dumps(res))
