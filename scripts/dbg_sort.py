import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import arrow_go_amd as ah
N = ah._native
ctx = ah.Context(0)
rng = np.random.default_rng(9)
for name, lg, gen in [("int uniform", 22, lambda n: rng.integers(-2**62, 2**62, n, dtype=np.int64)), ("int uniform", 24, lambda n: rng.integers(-2**62, 2**62, n, dtype=np.int64)),
                      ("normal", 22, lambda n: rng.standard_normal(n)), ("small ints", 22, lambda n: rng.integers(0, 2**40, n, dtype=np.int64)),
                      ("normal", 24, lambda n: rng.standard_normal(n)), ("lognormal", 24, lambda n: np.exp(rng.standard_normal(n) * 8)),
                      ("uniform(-1,1)", 24, lambda n: rng.uniform(-1, 1, n)), ("cauchy", 24, lambda n: rng.standard_cauchy(n)),
                      ("int uniform", 26, lambda n: rng.integers(-2**62, 2**62, n, dtype=np.int64)), ("normal", 26, lambda n: rng.standard_normal(n)),
                      ("3000 distinct", 24, lambda n: rng.integers(0, 2**50, 3000, dtype=np.int64)[rng.integers(0, 3000, n)]),
                      ("hot key 20%", 24, lambda n: np.where(rng.random(n) < 0.2, np.int64(123456789012345), rng.integers(-2**62, 2**62, n, dtype=np.int64))),
                      ("each key 4x", 24, lambda n: np.repeat(rng.integers(-2**62, 2**62, n // 4, dtype=np.int64), 4)[rng.permutation(n)]),
                      ("exp", 24, lambda n: rng.exponential(1.0, n)), ("clustered", 24, lambda n: 10**15 + rng.integers(0, 10**11, n)),
                      ("two clusters", 24, lambda n: np.where(rng.random(n) < 0.3, rng.integers(0, 10**6, n), 2**60 + rng.integers(0, 2**40, n)))]:
    n = 1 << lg
    v = gen(n)
    a = ctx.alloc(n * 8); out = ctx.alloc(n * 8)
    a.upload(v)
    print(name, lg, flush=True)
    ctx.sort_indices(N.FLOAT64 if v.dtype.kind == "f" else N.INT64, a, None, 0, n, False, False, out)
    got = out.download(np.uint64, n)
    if lg <= 24:
        exp = np.argsort(v, kind="stable")
        print("  equal:", bool((got == exp.astype(np.uint64)).all()), flush=True)
    else:
        w = v[got.astype(np.int64)]
        print("  sorted:", bool((w[1:] >= w[:-1]).all()), flush=True)
    a.free(); out.free()
