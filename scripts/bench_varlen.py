"""var-length take / filter throughput (row §8(f)-4): 2^24 strings."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0)
rng = np.random.default_rng(4)
res = {}
def timed(fn, reps=3):
    fn(); ctx.sync(); ctx.event_record(10)
    for _ in range(reps): fn()
    ctx.event_record(11); return ctx.event_elapsed_ms(10, 11) / reps
for mean_len in (8, 32, 256):
    n = 1 << (24 if mean_len < 100 else 22)
    lens = rng.integers(0, 2 * mean_len + 1, n).astype(np.int64)
    offs = np.zeros(n + 1, np.int32); offs[1:] = np.cumsum(lens)
    total = int(offs[-1])
    ob = ctx.to_device(offs); db = ctx.alloc(total + 64); db.memset(7)
    for name, idx in (("sorted", np.sort(rng.integers(0, n, n)).astype(np.int32)), ("random", rng.integers(0, n, n).astype(np.int32))):
        ib = ctx.to_device(idx)
        oo = ctx.alloc((n + 1) * 4 + 64)
        nulls, tot = ctx.take_binary_offsets(4, ob, None, 0, n, 4, True, ib, None, 0, n, oo, None)
        od = ctx.alloc(tot + 64)
        ms1 = timed(lambda: ctx.take_binary_offsets(4, ob, None, 0, n, 4, True, ib, None, 0, n, oo, None))
        ms2 = timed(lambda: ctx.take_binary_data(4, ob, db, 0, 4, ib, n, oo, od))
        res[f"take 2^{n.bit_length()-1} rows mean_len={mean_len} {name}"] = {"offsets_ms": round(ms1, 3), "data_ms": round(ms2, 3), "out_MiB": tot >> 20,
                                                   "data_GBps": round(2 * tot / ms2 / 1e6), "Mrows_per_s": round(n / (ms1 + ms2) / 1e3)}
        for b in (ib, oo, od): b.free()
    ob.free(); db.free()
print(json.dumps(res))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/bench_varlen.json", "w"), indent=1)
