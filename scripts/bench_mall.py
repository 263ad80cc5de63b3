"""Does a re-read of a recently streamed range come from the memory-side cache (MALL)?  Sum over working sets of 32 MiB … 1 GiB."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0); rows = 1 << 27
a = ctx.alloc(rows * 8); a.memset(1)
res_dev = ctx.alloc(64)
res = {}
for lg in (22, 23, 24, 25, 26, 27):
    n = 1 << lg
    for _ in range(3): ctx.sum_int64_dev(a, n, res_dev)
    ctx.sync(); ctx.event_record(10)
    reps = 20
    for _ in range(reps): ctx.sum_int64_dev(a, n, res_dev)
    ctx.event_record(11); ms = ctx.event_elapsed_ms(10, 11) / reps
    res[f"{n*8>>20}MiB"] = {"us": round(ms * 1e3, 1), "GBps": round(n * 8 / ms / 1e6)}
print(json.dumps(res))
