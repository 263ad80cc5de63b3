#!/usr/bin/env python3
"""One dictionary_encode whose automatic partition-first attempt voids itself after the global table already holds 2^16 rows
(tests/test_gpu_parity.py::test_hash_encode_voided_partition_attempt_keeps_prefix_ids) — run under rocprofv3 --kernel-trace to see both
paths in one call: the partition pass (enc_*_kernel) and, after it, the global insert with the restored slot numbers."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import arrow_go_amd as ah  # noqa: E402
from tests.backends import HipBackend, OracleBackend  # noqa: E402

rng = np.random.default_rng(6701)
n, head = (1 << 22) + 4321, 1 << 21
k = np.concatenate([rng.integers(0, 400_000, head), rng.integers(0, 1_800_000, n - head)])
keys = (k.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
with ah.Context(0) as ctx:
    g, e = HipBackend(ctx).hash_encode(keys, None, 0, False), OracleBackend().hash_encode(keys, None, 0, False)
    assert g[0].tobytes() == e[0].tobytes() and g[2].tobytes() == e[2].tobytes() and g[3] == e[3]
print("void_attempt_trace ok:", e[2].size, "keys")
