"""Int64 Add of 2^27 rows (the headline kernel) with the three columns placed at different relative offsets inside one allocation:
does the time depend on how the streams' addresses line up (HBM channel interleave)?  python scripts/bench_add_placement.py"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
N = ah._native
ctx = ah.Context(0)
rows = 1 << 27
col = rows * 8
pool = ctx.alloc(3 * col + (1 << 28))
pool.memset(1)
def timed(l, r, o, reps=20):
    for _ in range(3): ctx.arithmetic(N.INT64, N.OP_ADD, N.SHAPE_AA, l, r, o, rows)
    ctx.sync(); ctx.event_record(40)
    for _ in range(reps): ctx.arithmetic(N.INT64, N.OP_ADD, N.SHAPE_AA, l, r, o, rows)
    ctx.event_record(41); return round(ctx.event_elapsed_ms(40, 41) / reps, 4)
res = {}
base = (pool.ptr + (1 << 21) - 1) & ~((1 << 21) - 1)   # 2 MiB aligned
for name, d in (("0", 0), ("256B", 256), ("4KiB", 4096), ("64KiB", 1 << 16), ("1MiB", 1 << 20), ("1MiB+4KiB", (1 << 20) + 4096), ("3MiB+17*256B", (3 << 20) + 17 * 256), ("32MiB", 1 << 25)):
    l, r, o = base, base + col + d, base + 2 * col + 2 * d
    res[name] = [timed(l, r, o), timed(l, r, o)]
# three separate allocations, as bench.py has them
a, b, c = ctx.alloc(col), ctx.alloc(col), ctx.alloc(col)
a.memset(1); b.memset(1)
res["separate_allocations"] = [timed(a.ptr, b.ptr, c.ptr), timed(a.ptr, b.ptr, c.ptr)]
res["separate_ptr_mod_1GiB"] = [hex(p.ptr & ((1 << 30) - 1)) for p in (a, b, c)]
print(json.dumps(res))
