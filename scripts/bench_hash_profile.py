#!/usr/bin/env python3
"""one dictionary_encode + hash_sum at 2^26 rows, card 2^24 / 2^16 — run under rocprofv3 for the per-kernel split"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0)
hrows = 1 << 26
rng = np.random.default_rng(1)
keys = ctx.alloc(hrows * 8); vals = ctx.alloc(hrows * 8)
ids = ctx.alloc(hrows * 4); dic = ctx.alloc((hrows + 1) * 8); sums = ctx.alloc((hrows + 1) * 8); cnts = ctx.alloc((hrows + 1) * 8)
for card in (1 << 24, 1 << 16):
    for off in range(0, hrows, 1 << 22):
        keys.upload((rng.integers(0, card, 1 << 22, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64), off * 8)
    for _ in range(2):
        nd, _ = ctx.hash_u64_encode(keys, None, 0, hrows, False, ids, None, dic)
    print(card, nd)
