#!/usr/bin/env python3
"""The filter's count → fill pair on a stream SHARED with another producer (tests/test_gpu_parity.py runs this in a subprocess: torch
has to be imported before libarrowhip.so).  torch rewrites the mask on the shared stream between ah_filter_count and
ah_filter_primitive — with a mask of the same popcount, so the caller's n_out still fits — and the fill must follow the NEW mask:
a context made by ah_ctx_create_on_stream never leaves the count's tile prefixes for the fill (option "filter_cache" = 0 there).
With the cache forced on, the stale prefixes are demonstrably used (which is why it is off)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import arrow_go_amd as ah  # noqa: E402
from tests import oracle_lib as OL  # noqa: E402

o = OL.load_oracle()
torch.cuda.set_device(0)
stream = torch.cuda.current_stream()
ctx = ah.Context(0, stream=stream.cuda_stream)
rng = np.random.default_rng(808)
n = (1 << 20) + 123
values = rng.integers(-2**62, 2**62, n, dtype=np.int64)
m1 = rng.random(n) < 0.5
m2 = rng.permutation(m1)                       # same popcount, other rows
p1, p2 = np.packbits(m1, bitorder="little"), np.packbits(m2, bitorder="little")
vt = torch.from_numpy(values).cuda()
mt = torch.from_numpy(np.concatenate([p1, np.zeros(64, np.uint8)])).cuda()
m2t = torch.from_numpy(np.concatenate([p2, np.zeros(64, np.uint8)])).cuda()
out = torch.zeros(n + 8, dtype=torch.int64, device="cuda")


def run(cache):
    ctx.set_option("filter_cache", cache)
    mt.copy_(torch.from_numpy(np.concatenate([p1, np.zeros(64, np.uint8)])))
    n_out = ctx.filter_count(mt.data_ptr(), None, 0, n, 0)
    assert n_out == int(m1.sum())
    mt.copy_(m2t)                              # a FOREIGN kernel on the shared stream rewrites the mask
    ctx.filter_primitive(8, vt.data_ptr(), None, 0, mt.data_ptr(), None, 0, n, 0, n_out, out.data_ptr(), None)
    torch.cuda.synchronize()
    return out[:n_out].cpu().numpy()


want = values[m2]
got = run(0)                                   # the default of a shared-stream context
assert got.tobytes() == want.tobytes(), "fill on a shared stream did not follow the rewritten mask"
# the default IS 0 on this kind of context: a fresh one, no option set
ctx2 = ah.Context(0, stream=stream.cuda_stream)
mt.copy_(torch.from_numpy(np.concatenate([p1, np.zeros(64, np.uint8)])))
n_out = ctx2.filter_count(mt.data_ptr(), None, 0, n, 0)
mt.copy_(m2t)
ctx2.filter_primitive(8, vt.data_ptr(), None, 0, mt.data_ptr(), None, 0, n, 0, n_out, out.data_ptr(), None)
torch.cuda.synchronize()
assert out[:n_out].cpu().numpy().tobytes() == want.tobytes(), "default of ah_ctx_create_on_stream must be filter_cache = 0"
stale = run(1)                                 # vouching for the mask while a foreign kernel rewrites it: the hazard itself
print("shared_stream_filter_check ok: fill follows the rewritten mask; with the cache forced on the result is", "STALE (as expected)" if stale.tobytes() != want.tobytes() else "still right")
