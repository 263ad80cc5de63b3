"""Take with clustered indices (identity / reverse / sorted-random / slices), 2^27 int32 indices into a 1 GiB Int64 column, with and
without 10 % nulls on both sides: one row per lane (take_vec 0) against 16 / W rows per lane (take_vec 1 = by the sample, 2 = always)."""
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
def timed(fn, reps=5):
    fn(); ctx.sync(); ctx.event_record(10)
    for _ in range(reps): fn()
    ctx.event_record(11); return ctx.event_elapsed_ms(10, 11) / reps
res = {}
pats = {"identity": lambda: np.arange(rows, dtype=np.int32), "reverse": lambda: np.arange(rows - 1, -1, -1, dtype=np.int32),
        "sorted_random": lambda: np.sort(rng.integers(0, rows, rows, dtype=np.int32)), "slice_shifted": lambda: (np.arange(rows, dtype=np.int64) + 12345).clip(0, rows - 1).astype(np.int32)}
for name, mk in pats.items():
    idx.upload(mk())
    for nulls in (False, True):
        for mode in (0, 1, 2):
            ctx.set_option("take_vec", mode)
            ms = timed(lambda: ctx.take_primitive(8, a, vvalid if nulls else None, 0, rows, 4, True, idx, ivalid if nulls else None, 0, rows, True, c, ovalid if nulls else None))
            res[f"{name}{'_nulls10' if nulls else ''}_vec{mode}"] = {"ms": round(ms, 4), "GB/s": round((20 + (0.375 if nulls else 0)) * rows / ms / 1e6, 1)}
ctx.set_option("take_vec", 1)
print(json.dumps(res))
