"""Randomised cross-check of the many-groups group-by paths (two-level cut: reserving levels, records binned by first row) against the id-based
path: sizes, cardinalities, hot keys, value nulls, all-ones keys; Int64 and Float64 sums; bytes must be equal.  Prints one line per case."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
from tests.backends import HipBackend
ctx = ah.Context(0)
hip = HipBackend(ctx, dirty_outputs=True)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = np.random.default_rng(seed)
bad = 0
for case in range(ncases):
    lg = int(rng.integers(22, 26))
    n = (1 << lg) + int(rng.integers(-5000, 5000))
    card = int(rng.choice([n // 2, n // 3, n // 6, n, 3 << 20, 1 << 22]))
    hot = float(rng.choice([0.0, 0.0, 0.003, 0.02]))
    mode = int(rng.choice([1, 1, 3, 4]))
    keys = rng.integers(0, max(card, 2), n).astype(np.int64) * 1000003
    if hot:
        keys[rng.random(n) < hot] = 7 * 1000003
    if rng.random() < 0.5:
        keys[rng.integers(0, n, 3)] = -1
    vvalid = np.packbits(rng.random(n + 16) < 0.9, bitorder="little") if rng.random() < 0.6 else None
    voff = int(rng.integers(0, 8)) if vvalid is not None else 0
    iv = rng.integers(-2**62, 2**62, n, dtype=np.int64)
    fv = (1.0 + rng.random(n)) * np.exp(rng.uniform(-10, 10, n)) * rng.choice([-1.0, 1.0], n)
    if rng.random() < 0.3:
        fv[rng.integers(0, n, 2)] = np.inf
    ctx.set_option("groupby_partition", 0)
    base = [hip.hash_sum(k, keys, None, 0, v, vvalid, voff) for k, v in (("i64", iv), ("f64", fv))]
    ctx.set_option("groupby_partition", mode)
    got = [hip.hash_sum(k, keys, None, 0, v, vvalid, voff) for k, v in (("i64", iv), ("f64", fv))]
    ctx.set_option("groupby_partition", 1)
    ok = True
    for g, b in zip(got, base):
        same_sums = g[1].tobytes() == b[1].tobytes() or (np.isnan(g[1]) == np.isnan(b[1])).all() and (g[1][~np.isnan(g[1])].tobytes() == b[1][~np.isnan(b[1])].tobytes())
        ok = ok and g[0].tobytes() == b[0].tobytes() and same_sums and g[2].tobytes() == b[2].tobytes() and g[3] == b[3] and g[4].tobytes() == b[4].tobytes()
    bad += 0 if ok else 1
    print(json.dumps({"case": case, "n": n, "card": card, "hot": hot, "mode": mode, "nulls": vvalid is not None, "groups": int(base[0][0].size), "ok": bool(ok)}), flush=True)
print("FAILED" if bad else "all equal", bad)
