"""Does the relative placement of the three streams of Int64 Add (a, b → c) matter?  Same kernel, same sizes,
operands at different byte offsets inside one allocation (HBM channel interleave experiment)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
N = ah._native
ctx = ah.Context(0)
rows = 1 << 27
span = rows * 8
pad = 64 << 20
big = ctx.alloc(3 * (span + pad) + (1 << 20))
big.memset(1)
class View(int):
    pass
class _Unused:
    def __init__(self, ptr): self.ptr = ptr
res = {}
def run(name, oa, ob, oc, reps=20):
    a, b, c = big.ptr + oa, big.ptr + span + pad + ob, big.ptr + 2 * (span + pad) + oc
    f = lambda: ctx.arithmetic(N.INT64, N.OP_ADD, N.SHAPE_AA, a, b, c, rows)
    f(); ctx.sync(); ctx.event_record(1)
    for _ in range(reps): f()
    ctx.event_record(2)
    ms = ctx.event_elapsed_ms(1, 2) / reps
    res[name] = {"ms": round(ms, 4), "GB/s": round(24 * rows / ms / 1e6, 1)}
K = 1024
for name, (oa, ob, oc) in {
    "aligned": (0, 0, 0),
    "b+256 c+512": (0, 256, 512),
    "b+1K c+2K": (0, K, 2 * K),
    "b+4K c+8K": (0, 4 * K, 8 * K),
    "b+8K+256 c+16K+512": (0, 8 * K + 256, 16 * K + 512),
    "b+32K c+64K": (0, 32 * K, 64 * K),
    "b+1M c+2M": (0, K * K, 2 * K * K),
    "b+1M+4K c+2M+8K": (0, K * K + 4 * K, 2 * K * K + 8 * K),
    "b+11M c+23M": (0, 11 * K * K, 23 * K * K),
    "aligned again": (0, 0, 0),
}.items():
    run(name, oa, ob, oc)
print(json.dumps(res, indent=1))
