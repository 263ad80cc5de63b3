"""One process, many workloads, each delimited by a MARKER launch (a 64-bit kleene_kernel, which no workload uses) so that a
rocprofv3 database can be cut into (workload, kernel) rows afterwards (scripts/pmc_by_workload.py).  Per workload: one
untimed run (warm-up), then marker, REPS runs, marker: only the dispatches between a pair of markers are the workload's.  The order of WORKLOADS is the order of the
segments.  Run it bare for HIP-event timings, or under
    rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE --kernel-trace | --pmc WRITE_SIZE --kernel-trace   (separate passes)
    python scripts/prof_workloads.py [--rows-log2 27] [--only name,name]"""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
from bench import random_bits, zipf_ranks

p = argparse.ArgumentParser(); p.add_argument("--rows-log2", type=int, default=27); p.add_argument("--only", default="")
p.add_argument("--list", action="store_true")
args = p.parse_args()
REPS = 3
N = ah._native
rows = 1 << args.rows_log2
hrows = min(rows, 1 << 26)
WORKLOADS = ["add_int64", "sum_float64", "copy_kernel", "filter_sel0.01", "filter_sel0.50", "filter_sel0.90", "filter_sel0.50_masknulls_emit", "filter_sel0.50_count_and_fill",
             "take_random_nulls10", "take_random", "take_random_nulls10_direct", "take_identity_nulls10", "take_identity", "take_reverse_nulls10", "take_stride8_calib", "take_stride16_calib",
             "cumulative_sum_int64", "cumulative_sum_float64", "dict_encode_2^10", "dict_encode_2^16", "dict_encode_2^20", "dict_encode_2^22", "dict_encode_2^24", "dict_encode_2^20_zipf", "unique_2^20",
             "hash_sum_2^10", "hash_sum_2^16", "hash_sum_2^20", "hash_sum_2^24", "hash_sum_2^20_zipf", "sort_indices_int64_2^27"]
if args.list:
    print("\n".join(WORKLOADS)); sys.exit(0)
only = [w for w in args.only.split(",") if w]
ctx = ah.Context(0)
rng = np.random.default_rng(5)
a = ctx.to_device(rng.integers(-2**62, 2**62, rows, dtype=np.int64), 64)
b = ctx.to_device(rng.integers(-2**62, 2**62, rows, dtype=np.int64), 64)
x = ctx.to_device(rng.uniform(-1e6, 1e6, rows), 64)
c = ctx.alloc(rows * 8 + 64)
vvalid = ctx.to_device(random_bits(rng, rows, 0.9)); ivalid = ctx.to_device(random_bits(rng, rows, 0.9)); fvalid = ctx.to_device(random_bits(rng, rows, 0.9))
ovalid = ctx.alloc(rows // 8 + 64); fmask = ctx.alloc(rows // 8 + 64); idx = ctx.alloc(rows * 4 + 64)
res = ctx.alloc(64)
hids = ctx.alloc(hrows * 4 + 64); hdic, hsum, hcnt = (ctx.alloc((hrows + 1) * 8 + 64) for _ in range(3))
keys = ctx.alloc(hrows * 8 + 64)
sort_out = None
mk = [ctx.to_device(np.zeros(64, np.uint8)) for _ in range(6)]


def marker():
    ctx.kleene(0, mk[0], mk[1], 0, mk[2], mk[3], 0, mk[4], mk[5], 0, 64)


def setup_filter(sel, fv, null_sel):
    fmask.upload(random_bits(rng, rows, sel))
    n_out = ctx.filter_count(fmask, fv, 0, rows, null_sel)
    return lambda: ctx.filter_primitive(8, a, vvalid, 0, fmask, fv, 0, rows, null_sel, n_out, c, ovalid, want_null_count=False)


def setup_take(gen, nulls, binned=1):
    idx.upload(gen())
    def run():
        ctx.set_option("take_binned", binned)
        ctx.take_primitive(8, a, vvalid if nulls else None, 0, rows, 4, True, idx, ivalid if nulls else None, 0, rows, True, c, ovalid if nulls else None)
        ctx.set_option("take_binned", 1)
    return run


def setup_keys(lg, zipf):
    r = zipf_ranks(rng, hrows, 1 << lg) if zipf else rng.integers(0, 1 << lg, hrows, dtype=np.uint64)
    keys.upload((r * np.uint64(0x9E3779B97F4A7C15)).view(np.int64))


def build(name):
    global sort_out
    if name == "add_int64": return lambda: ctx.arithmetic(N.INT64, N.OP_ADD, N.SHAPE_AA, a, b, c, rows)
    if name == "sum_float64": return lambda: ctx.sum_float64_dev(x, rows, res)
    if name == "copy_kernel":
        return lambda: N.check(ctx.handle, N.lib.ah_copy_async(ctx.handle, c.ptr, a.ptr, rows * 8))
    if name == "filter_sel0.50_count_and_fill":   # the two-phase call as a host makes it: count (tile prefixes stay in the context), then fill
        fmask.upload(random_bits(rng, rows, 0.5))
        def both():
            k = ctx.filter_count(fmask, None, 0, rows, 0)
            ctx.filter_primitive(8, a, vvalid, 0, fmask, None, 0, rows, 0, k, c, ovalid, want_null_count=False)
        return both
    if name.startswith("filter_sel"):
        sel = float(name[10:14])
        return setup_filter(sel, fvalid if "masknulls" in name else None, 1 if name.endswith("emit") else 0)
    if name.startswith("take_random"): return setup_take(lambda: rng.integers(0, rows, rows, dtype=np.int32), "nulls" in name, 0 if name.endswith("direct") else 1)
    if name == "take_identity_nulls10": return setup_take(lambda: np.arange(rows, dtype=np.int32), True)
    if name == "take_identity": return setup_take(lambda: np.arange(rows, dtype=np.int32), False)
    if name == "take_reverse_nulls10": return setup_take(lambda: np.arange(rows - 1, -1, -1, dtype=np.int32), True)
    if name.startswith("take_stride"):  # calibration: every gather its own 64-byte (stride 8) / 128-byte (stride 16) piece of the column, direct kernel
        st = int(name[11:].split("_")[0])
        return setup_take(lambda: ((np.arange(rows, dtype=np.int64) * st) % rows).astype(np.int32), False, 0)
    if name == "cumulative_sum_int64": return lambda: ctx.cumulative_sum(N.INT64, a, None, 0, rows, None, False, False, c, None)
    if name == "cumulative_sum_float64": return lambda: ctx.cumulative_sum(N.FLOAT64, x, None, 0, rows, None, False, False, c, None)
    if name.startswith("unique"):
        setup_keys(int(name.split("^")[1]), False)
        return lambda: ctx.hash_u64_encode(keys, None, 0, hrows, False, None, None, hdic)
    if name.startswith("dict_encode") or name.startswith("hash_sum"):
        lg = int(name.split("^")[1].split("_")[0])
        setup_keys(lg, name.endswith("zipf"))
        if name.startswith("dict"): return lambda: ctx.hash_u64_encode(keys, None, 0, hrows, False, hids, None, hdic)
        return lambda: ctx.hash_sum("f64", keys, None, 0, x, None, 0, hrows, hdic, hsum, hcnt)
    if name == "sort_indices_int64_2^27":
        sort_out = sort_out or ctx.alloc(rows * 8 + 64)
        return lambda: ctx.sort_indices(N.INT64, a, None, 0, rows, False, False, sort_out)
    raise KeyError(name)


out = {"rows": rows, "hash_rows": hrows, "reps": REPS, "workloads": []}
for name in WORKLOADS:
    if only and name not in only:
        continue
    fn = build(name)
    fn(); ctx.sync()
    marker()
    ctx.event_record(1)
    for _ in range(REPS):
        fn()
    ctx.event_record(2)
    marker()
    out["workloads"].append({"name": name, "ms": round(ctx.event_elapsed_ms(1, 2) / REPS, 4)})
print(json.dumps(out))
