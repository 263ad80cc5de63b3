"""Take of 2^27 Int64 values by an identity Int32 index vector with 10 % nulls on both sides, twelve calls in a row (for a kernel trace)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
import bench
ctx = ah.Context(0)
rows = 1 << 27
rng = np.random.default_rng(5)
a = ctx.alloc(rows * 8 + 64); c = ctx.alloc(rows * 8 + 64); idx = ctx.alloc(rows * 4 + 64)
bench.fill_random(ctx, a, rows, np.int64, 1)
vvalid = ctx.to_device(bench.random_bits(rng, rows, 0.9)); ivalid = ctx.to_device(bench.random_bits(rng, rows, 0.9)); ovalid = ctx.alloc(rows // 8 + 64)
idx.upload(np.arange(rows, dtype=np.int32))
if len(sys.argv) > 1: ctx.set_option("take_hint_cache", int(sys.argv[1]))
for _ in range(12):
    ctx.take_primitive(8, a, vvalid, 0, rows, 4, True, idx, ivalid, 0, rows, True, c, ovalid)
ctx.sync()
