#!/usr/bin/env python3
"""A/B sweep of the streaming-kernel tunables on the GPU box (ARROWHIP_NT,
ARROWHIP_BLOCKS_PER_CU are read at context creation).  Prints one line per setting."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
N = ah._native
rows = 1 << 27

def bench(ctx, fn, nbytes, reps=20):
    fn(); fn()
    ctx.event_record(1); 
    for _ in range(reps): fn()
    ctx.event_record(2)
    ms = ctx.event_elapsed_ms(1, 2) / reps
    return nbytes / ms / 1e6

for nt in (1, 0):
    for bpc in (2, 4, 8, 16, 32, 64):
        os.environ["ARROWHIP_NT"] = str(nt); os.environ["ARROWHIP_BLOCKS_PER_CU"] = str(bpc)
        ctx = ah.Context(0)
        a = ctx.alloc(rows * 8); b = ctx.alloc(rows * 8); c = ctx.alloc(rows * 8); r = ctx.alloc(64)
        rng = np.random.default_rng(0)
        chunk = rng.integers(-2**62, 2**62, 1 << 22, dtype=np.int64)
        for off in range(0, rows, 1 << 22):
            a.upload(chunk, off * 8); b.upload(chunk[::-1].copy(), off * 8)
        add = bench(ctx, lambda: ctx.arithmetic(N.INT64, N.OP_ADD, N.SHAPE_AA, a, b, c, rows), 24 * rows)
        sm = bench(ctx, lambda: ctx.sum_int64_dev(a, rows, r), 8 * rows)
        smf = bench(ctx, lambda: ctx.sum_float64_dev(a, rows, r), 8 * rows)
        print(f"nt={nt} blocks_per_cu={bpc:3d}  add_i64 {add:7.1f} GB/s   sum_i64 {sm:7.1f}   sum_f64 {smf:7.1f}", flush=True)
        for x in (a, b, c, r): x.free()
        ctx.close()
