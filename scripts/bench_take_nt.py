"""Clustered take (2^27 int32 indices, 1 GiB Int64 column): nontemporal hints of take_vec_kernel — option take_vec_nt 0 / 4 / 5 / 7."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
import bench
ctx = ah.Context(0); rows = 1 << 27
rng = np.random.default_rng(1)
a = ctx.alloc(rows * 8); c = ctx.alloc(rows * 8); idx = ctx.alloc(rows * 4 + 64)
chunk = rng.integers(-2**62, 2**62, 1 << 22, dtype=np.int64)
for off in range(0, rows, 1 << 22): a.upload(chunk, off * 8)
vvalid = ctx.to_device(bench.random_bits(rng, rows, 0.9)); ivalid = ctx.to_device(bench.random_bits(rng, rows, 0.9)); ovalid = ctx.alloc(rows // 8 + 64)
def timed(fn, reps=10):
    fn(); ctx.sync(); ctx.event_record(10)
    for _ in range(reps): fn()
    ctx.event_record(11); return ctx.event_elapsed_ms(10, 11) / reps
res = {}
pats = {"identity": lambda: np.arange(rows, dtype=np.int32), "reverse": lambda: np.arange(rows - 1, -1, -1, dtype=np.int32),
        "sorted_random": lambda: np.sort(rng.integers(0, rows, rows, dtype=np.int32))}
for name, mk in pats.items():
    idx.upload(mk())
    for nulls in (False, True):
        for nt in (0, 4, 5, 7):
            ctx.set_option("take_vec", 2); ctx.set_option("take_vec_nt", nt)
            ms = timed(lambda: ctx.take_primitive(8, a, vvalid if nulls else None, 0, rows, 4, True, idx, ivalid if nulls else None, 0, rows, True, c, ovalid if nulls else None))
            res[f"{name}{'_nulls10' if nulls else ''}_nt{nt}"] = {"ms": round(ms, 4), "GB/s": round((20 + (0.375 if nulls else 0)) * rows / ms / 1e6, 1)}
ctx.set_option("take_vec", 1); ctx.set_option("take_vec_nt", 7)
print(json.dumps(res))
