"""ah_shift_time (temporal unit change) at 2^27 rows: the four width pairs × multiply / divide, checked and not (algorithmic GB/s)"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0)
rows = 1 << 27
a = ctx.alloc(rows * 8); c = ctx.alloc(rows * 8)
rng = np.random.default_rng(0)
chunk = rng.integers(-(1 << 20), 1 << 20, 1 << 22, dtype=np.int64) * 1000   # multiples of 1000 small enough for every check, in both views
for off in range(0, rows, 1 << 22):
    a.upload(chunk, off * 8)
res = {}
def timed(name, nbytes, fn, reps=10):
    fn(); ctx.sync(); ctx.event_record(1)
    for _ in range(reps): fn()
    ctx.event_record(2)
    ms = ctx.event_elapsed_ms(1, 2) / reps
    res[name] = {"ms": round(ms, 4), "GB/s": round(nbytes / ms / 1e6, 1)}
for ib, ob in ((64, 64), (32, 64), (64, 32), (32, 32)):
    n = rows if ib == 64 or ob == 64 else rows * 2
    per_row = ib // 8 + ob // 8
    if ib == 32 and ob == 64:
        n = rows  # the output buffer holds 2^27 int64
    chk = ib == 64   # the int32 view of the data is not built to pass the checks
    timed(f"mul1000_{ib}to{ob}" + ("_checked" if chk else ""), per_row * n, lambda: ctx.shift_time(ib, ob, 0, 1000, chk, a, None, 0, n, c))
    timed(f"div1000_{ib}to{ob}" + ("_checked" if chk else ""), per_row * n, lambda: ctx.shift_time(ib, ob, 1, 1000, chk, a, None, 0, n, c))
timed("convert_32to64", 12 * rows, lambda: ctx.shift_time(32, 64, 0, 1, True, a, None, 0, rows, c))
print(json.dumps(res))
