#!/usr/bin/env python3
"""Secondary measurements quoted in DESIGN.md / BASELINE.md (not the driver's bench line):
PCIe H2D/D2H rate and the PCIe-inclusive Sum rate, Filter across selectivities, Take
access patterns, unique / dictionary_encode / group-by across cardinalities."""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
N = ah._native
lib = N.lib
rows = 1 << 27
ctx = ah.Context(0)
out = {}

def timed(fn, reps=5, warm=1):
    for _ in range(warm): fn()
    ctx.sync(); ctx.event_record(10)
    for _ in range(reps): fn()
    ctx.event_record(11)
    return ctx.event_elapsed_ms(10, 11) / reps

# ---- PCIe -------------------------------------------------------------------------------
hp = C.c_void_p()
N.check(ctx.handle, lib.ah_host_alloc_pinned(ctx.handle, rows * 8, C.byref(hp)))
host = np.ctypeslib.as_array((C.c_double * rows).from_address(hp.value))
host[:] = np.random.default_rng(0).uniform(-1, 1, rows)
d = ctx.alloc(rows * 8)
def h2d():
    N.check(ctx.handle, lib.ah_upload_async(ctx.handle, d.ptr, hp.value, rows * 8)); ctx.sync()
t0 = time.perf_counter(); h2d(); h2d(); dt = (time.perf_counter() - t0) / 2
out["pcie_h2d_pinned_GB/s"] = round(rows * 8 / dt / 1e9, 2)
def d2h():
    N.check(ctx.handle, lib.ah_download_async(ctx.handle, hp.value, d.ptr, rows * 8)); ctx.sync()
t0 = time.perf_counter(); d2h(); d2h(); dt = (time.perf_counter() - t0) / 2
out["pcie_d2h_pinned_GB/s"] = round(rows * 8 / dt / 1e9, 2)
t0 = time.perf_counter(); h2d(); s = ctx.sum_float64(d, rows); dt = time.perf_counter() - t0
out["sum_float64_including_h2d_GB/s"] = round(rows * 8 / dt / 1e9, 2)
pageable = np.array(host[: 1 << 24])
t0 = time.perf_counter(); d.upload(pageable); dt = time.perf_counter() - t0
out["pcie_h2d_pageable_GB/s"] = round(pageable.nbytes / dt / 1e9, 2)

# ---- data -------------------------------------------------------------------------------
rng = np.random.default_rng(1)
a = ctx.alloc(rows * 8); c = ctx.alloc(rows * 8)
chunk = rng.integers(-2**62, 2**62, 1 << 22, dtype=np.int64)
for off in range(0, rows, 1 << 22): a.upload(chunk, off * 8)
mask = ctx.alloc(rows // 8 + 64); vvalid = ctx.alloc(rows // 8 + 64); ovalid = ctx.alloc(rows // 8 + 64)
vb = np.packbits(rng.random(1 << 22) < 0.9, bitorder="little")
for off in range(0, rows // 8, vb.size): vvalid.upload(vb, off)

# ---- filter across selectivities ------------------------------------------------------------
for p in (0.01, 0.1, 0.5, 0.9, 1.0):
    mb = np.packbits(rng.random(1 << 22) < p, bitorder="little")
    for off in range(0, rows // 8, mb.size): mask.upload(mb, off)
    n_out = ctx.filter_count(mask, None, 0, rows, 0)
    for name, vv, ov in (("nonulls", None, None), ("nulls10", vvalid, ovalid)):
        ms = timed(lambda: ctx.filter_primitive(8, a, vv, 0, mask, None, 0, rows, 0, n_out, c, ov, want_null_count=False))
        traffic = (8.125 + (0.125 if vv else 0)) * rows + (8 + (0.125 if vv else 0)) * n_out
        out[f"filter_i64_{name}_sel{p}"] = {"ms": round(ms, 4), "input_GB/s": round(8 * rows / ms / 1e6, 1), "traffic_GB/s": round(traffic / ms / 1e6, 1)}
# runs mask (geometric run lengths, mean 256)
runs = np.repeat(rng.random((1 << 25) // 256) < 0.5, 256)
mb = np.packbits(runs, bitorder="little")
for off in range(0, rows // 8, mb.size): mask.upload(mb[: min(mb.size, rows // 8 - off)], off)
n_out = ctx.filter_count(mask, None, 0, rows, 0)
ms = timed(lambda: ctx.filter_primitive(8, a, None, 0, mask, None, 0, rows, 0, n_out, c, None, want_null_count=False))
out["filter_i64_runs256_sel0.5"] = {"ms": round(ms, 4), "input_GB/s": round(8 * rows / ms / 1e6, 1), "traffic_GB/s": round((8.125 * rows + 8 * n_out) / ms / 1e6, 1)}

# ---- take patterns ----------------------------------------------------------------------------
idx = ctx.alloc(rows * 4)
def fill_idx(gen):
    for off in range(0, rows, 1 << 22):
        idx.upload(gen(off, min(1 << 22, rows - off)), off * 4)
pats = {
  "random": lambda off, m: rng.integers(0, rows, m, dtype=np.int64).astype(np.int32),
  "sorted_random": lambda off, m: np.sort(rng.integers(off, off + m, m, dtype=np.int64)).astype(np.int32),
  "identity": lambda off, m: np.arange(off, off + m, dtype=np.int32),
  "reverse": lambda off, m: np.arange(rows - 1 - off, rows - 1 - off - m, -1, dtype=np.int32),
  "random_within_64MiB": lambda off, m: (off // (1 << 23) * (1 << 23) + rng.integers(0, 1 << 23, m, dtype=np.int64)).astype(np.int32),
}
for name, gen in pats.items():
    fill_idx(gen)
    ms = timed(lambda: ctx.take_primitive(8, a, None, 0, rows, 4, True, idx, None, 0, rows, True, c, None), reps=3)
    out[f"take_i64_i32_{name}"] = {"ms": round(ms, 4), "GB/s": round(20 * rows / ms / 1e6, 1)}
fill_idx(pats["random"])
ms = timed(lambda: ctx.take_primitive(8, a, vvalid, 0, rows, 4, True, idx, vvalid, 0, rows, True, c, ovalid), reps=3)
out["take_i64_i32_random_nulls10"] = {"ms": round(ms, 4), "GB/s": round(20.375 * rows / ms / 1e6, 1)}

# ---- hash ----------------------------------------------------------------------------------------
hrows = 1 << 26
keys = ctx.alloc(hrows * 8); vals = ctx.alloc(hrows * 8)
ids = ctx.alloc(hrows * 4); dic = ctx.alloc((hrows + 1) * 8); sums = ctx.alloc((hrows + 1) * 8); cnts = ctx.alloc((hrows + 1) * 8)
fv = rng.uniform(-1, 1, 1 << 22)
for off in range(0, hrows, 1 << 22): vals.upload(fv, off * 8)
for card in (1 << 10, 1 << 16, 1 << 20, 1 << 24):
    for off in range(0, hrows, 1 << 22):
        keys.upload((rng.integers(0, card, 1 << 22, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64), off * 8)
    t0 = time.perf_counter(); nd, _ = ctx.hash_u64_encode(keys, None, 0, hrows, False, ids, None, dic); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); nd, _ = ctx.hash_u64_encode(keys, None, 0, hrows, False, ids, None, dic); dt = time.perf_counter() - t0
    out[f"dictionary_encode_i64_card2^{card.bit_length()-1}"] = {"ms": round(dt * 1e3, 3), "rows/s": round(hrows / dt / 1e9, 3), "GB/s_keys": round(8 * hrows / dt / 1e9, 1), "ndict": nd}
    t0 = time.perf_counter(); ng, _ = ctx.hash_sum("f64", keys, None, 0, vals, None, 0, hrows, dic, sums, cnts); dt = time.perf_counter() - t0
    out[f"hash_sum_f64_card2^{card.bit_length()-1}"] = {"ms": round(dt * 1e3, 3), "GB/s_16B_per_row": round(16 * hrows / dt / 1e9, 1), "groups": ng}
print(json.dumps(out, indent=1))
