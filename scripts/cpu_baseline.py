#!/usr/bin/env python3
"""CPU baseline of BASELINE.md §3: the reference's own kernels (AVX2 machine code assembled
from its clang output; C sum kernels compiled strict-sequential = noasm order) and the oracle's
restatements of the Go-only kernels, timed on this host at 64 KiB (cache-resident, comparable
with the README table) and 1 GiB (DRAM-resident, comparable with the GPU runs), 1 thread."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import oracle_lib as OL
ref, o = OL.load_reference(), OL.load_oracle()
out = {"host": {"nproc": os.cpu_count(), "model": next((l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?")},
       "note": "C-kernel timings through ctypes (not `go test -bench`); 1 thread; median of repetitions"}

def med(fn, reps):
    fn(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))

rng = np.random.default_rng(0)
for label, n, reps in (("64KiB", 8192, 2000), ("1GiB", 1 << 27, 5)):
    x = rng.uniform(-1, 1, n); xi = rng.integers(-2**62, 2**62, n, dtype=np.int64); yi = rng.integers(-2**62, 2**62, n, dtype=np.int64)
    r = {}
    for name, fn, nbytes in [
        ("sum_float64_avx2 (reference asm)", lambda: ref.sum("avx2", x), 8 * n),
        ("sum_float64_noasm_order (reference C, -O2 -fno-tree-vectorize)", lambda: ref.sum("seq", x), 8 * n),
        ("sum_int64_avx2 (reference asm)", lambda: ref.sum("avx2", xi), 8 * n),
        ("add_int64_avx2 (reference asm, incl. output alloc)", lambda: ref.arithmetic(0, 0, xi, yi), 24 * n),
        ("greater_int64_scalar_avx2 (reference asm)", lambda: ref.comparison(2, 1, xi, np.array([0], np.int64), np.zeros(n // 8 + 8, np.uint8)), 8.125 * n),
    ]:
        t = med(fn, reps)
        r[name] = {"ns_per_op": round(t * 1e9, 1), "GB/s": round(nbytes / t / 1e9, 2)}
    if label == "1GiB":
        m = 1 << 24  # Go-only kernels: oracle restatements, bounded sample (2^24 rows)
        v = xi[:m]; mask = OL.pack_bits(rng.random(m) < 0.5); vv = OL.pack_bits(rng.random(m) < 0.9)
        idx = rng.integers(0, m, m, dtype=np.int64).astype(np.int32)
        t = med(lambda: o.filter_primitive(v, vv, 0, mask, None, 0, m, 0, True), 2); r["filter_int64_sel0.5_nulls10 (oracle port, 2^24 rows)"] = {"ns_per_op": round(t * 1e9), "input_GB/s": round(8 * m / t / 1e9, 2)}
        t = med(lambda: o.take_primitive(v, None, 0, idx, None, 0, True, False), 2); r["take_int64_random_i32 (oracle port, 2^24 rows)"] = {"ns_per_op": round(t * 1e9), "GB/s": round(20 * m / t / 1e9, 2)}
        keys = (rng.integers(0, 1 << 16, m, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
        t = med(lambda: o.hash_u64_encode(keys, None, 0, False), 1); r["dictionary_encode_int64_card2^16 (oracle port of the memo table, 2^24 rows)"] = {"ns_per_op": round(t * 1e9), "Mrows/s": round(m / t / 1e6, 1)}
    out[label] = r
print(json.dumps(out, indent=1))
