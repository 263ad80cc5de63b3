"""dictionary_encode of 2^26 Int64 rows with 2^lg distinct keys, a few calls (a workload to put under rocprofv3).   python scripts/bench_encode_one.py 20"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0)
hrows = 1 << 26
lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(1)
keys = ctx.alloc(hrows * 8); ids = ctx.alloc(hrows * 4); dic = ctx.alloc((hrows + 1) * 8)
for off in range(0, hrows, 1 << 22):
    keys.upload((rng.integers(0, 1 << lg, 1 << 22, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64), off * 8)
for _ in range(4):
    nd = ctx.hash_u64_encode(keys, None, 0, hrows, False, ids, None, dic)
ctx.sync()
print("ndict", nd)
