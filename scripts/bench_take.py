"""C3 Take, the stated config: 2^27 int32 indices into a 2^27-row Int64 column with 10 % nulls (values and indices).
Index patterns of SURVEY §8d: uniform random over [0, 2^27) (2^27 INDEPENDENT draws — not a tiled chunk), sorted-random,
identity, reverse; direct kernel vs the binned path (ah_take_binned.hip) over its two knobs.  One JSON line.
    python scripts/bench_take.py [--quick] [--rows-log2 27]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah

p = argparse.ArgumentParser(); p.add_argument("--quick", action="store_true"); p.add_argument("--rows-log2", type=int, default=27)
args = p.parse_args()
N = ah._native
ctx = ah.Context(0)
rows = 1 << args.rows_log2
rng = np.random.default_rng(5)


def bits(n, p):
    out = np.empty(n // 8 + 64, np.uint8)
    for i in range(0, n, 1 << 24):
        m = min(1 << 24, n - i)
        out[i // 8:i // 8 + (m + 7) // 8] = np.packbits(rng.random(m) < p, bitorder="little")
    return out


vals = rng.integers(-2**62, 2**62, rows, dtype=np.int64)
a = ctx.to_device(vals, 64)
vvalid = ctx.to_device(bits(rows, 0.9))
ivalid = ctx.to_device(bits(rows, 0.9))
out_d = ctx.alloc(rows * 8 + 64); out_b = ctx.alloc(rows * 8 + 64)
ov_d = ctx.alloc(rows // 8 + 64); ov_b = ctx.alloc(rows // 8 + 64)
idx = ctx.alloc(rows * 4 + 64)
mask = ctx.alloc(rows // 8 + 64)


def timed(fn, reps=3):
    fn(); ctx.sync(); ctx.event_record(10)
    for _ in range(reps):
        fn()
    ctx.event_record(11)
    return ctx.event_elapsed_ms(10, 11) / reps


def take(out, ov, nulls):
    ctx.take_primitive(8, a, vvalid if nulls else None, 0, rows, 4, True, idx, ivalid if nulls else None, 0, rows, True, out, ov if nulls else None)


def line(ms):
    return {"ms": round(ms, 3), "GB/s": round(20 * rows / ms / 1e6, 1)}


res = {"rows": rows}
patterns = {"random": lambda: rng.integers(0, rows, rows, dtype=np.int32)}
if not args.quick:
    patterns["sorted_random"] = lambda: np.sort(rng.integers(0, rows, rows, dtype=np.int32))
    patterns["identity"] = lambda: np.arange(rows, dtype=np.int32)
    patterns["reverse"] = lambda: np.arange(rows - 1, -1, -1, dtype=np.int32)
for pname, gen in patterns.items():
    idx.upload(gen())
    for nulls in (False, True):
        tag = pname + ("_nulls10" if nulls else "")
        ctx.set_option("take_binned", 0)
        res[tag + "/direct"] = line(timed(lambda: take(out_d, ov_d, nulls)))
        ctx.set_option("take_binned", 1)
        res[tag + "/auto"] = line(timed(lambda: take(out_b, ov_b, nulls)))
        # same bytes from both paths (device-side compare + validity xor popcount)
        ctx.comparison(N.CMP_NE, N.SHAPE_AA, N.INT64, out_d, out_b, mask, rows, 0)
        diff = ctx.count_set_bits(mask, 0, rows)
        if nulls:
            ctx.bitmap_op(N.BIT_XOR, ov_d, 0, ov_b, 0, mask, 0, rows)
            diff += ctx.count_set_bits(mask, 0, rows)
        res[tag + "/mismatches"] = int(diff)
        if pname == "random" and not args.quick:
            ctx.set_option("take_binned", 2)
            for wl in (20, 21, 22):
                for wg in (2, 4, 8):
                    ctx.set_option("take_window_log2", wl); ctx.set_option("take_gather_wg_per_cu", wg)
                    res[f"{tag}/binned_w{wl}_g{wg}"] = line(timed(lambda: take(out_b, ov_b, nulls)))
            ctx.set_option("take_window_log2", 22); ctx.set_option("take_gather_wg_per_cu", 8)
            for ld in (1, 2):
                ctx.set_option("take_gather_load", ld)
                res[f"{tag}/binned_w22_g8_ld{ld}"] = line(timed(lambda: take(out_b, ov_b, nulls)))
            ctx.set_option("take_gather_load", 0); ctx.set_option("take_binned", 1)
print(json.dumps(res))
