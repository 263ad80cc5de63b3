import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
N = ah._native
ctx = ah.Context(0); rows = 1 << 27
rng = np.random.default_rng(9)
a = ctx.alloc(rows * 8); out = ctx.alloc(rows * 8)
chunk = rng.integers(-2**62, 2**62, 1 << 22, dtype=np.int64)
for off in range(0, rows, chunk.size): a.upload(chunk + off, off * 8)
for _ in range(3):
    ctx.sort_indices(N.INT64, a, None, 0, rows, False, False, out)
ctx.sync(); ctx.event_record(1)
ctx.sort_indices(N.INT64, a, None, 0, rows, False, False, out)
ctx.event_record(2); print("sort 2^27 int64:", round(ctx.event_elapsed_ms(1, 2), 2), "ms")
