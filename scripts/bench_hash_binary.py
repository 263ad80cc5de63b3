"""dictionary_encode of a String column (2^25 rows, keys "k%07d"-style 8..24 bytes) across cardinalities.
   python scripts/bench_hash_binary.py"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import arrow_go_amd as ah
ctx = ah.Context(0)
n = 1 << 25
rng = np.random.default_rng(3)
res = {}
def timed(fn, reps=3):
    fn(); ctx.sync(); ctx.event_record(1)
    for _ in range(reps): fn()
    ctx.event_record(2)
    return ctx.event_elapsed_ms(1, 2) / reps
for klen in (8, 24):
    for lg in (10, 16, 20, 23):
        card = 1 << lg
        # fixed-length keys: the id in hex, zero-padded — built as a byte matrix
        ids = rng.integers(0, card, n)
        digits = np.frombuffer(b"0123456789abcdef", np.uint8)
        mat = np.empty((n, klen), np.uint8); mat[:] = ord("k")
        for d in range(8):
            mat[:, klen - 1 - d] = digits[(ids >> (4 * d)) & 15]
        offsets = (np.arange(n + 1, dtype=np.int64) * klen).astype(np.int32)
        ofb = ctx.alloc(offsets.nbytes); ofb.upload(offsets)
        db = ctx.alloc(mat.nbytes); db.upload(mat.reshape(-1))
        idb = ctx.alloc(n * 4); frb = ctx.alloc((n + 1) * 8)
        nd = [0]
        def run():
            nd[0], _ = ctx.hash_binary_encode(4, ofb, db, None, 0, n, False, idb, None, frb)
        ms = timed(run)
        res[f"len{klen} 2^{lg}"] = {"ms": round(ms, 3), "Grows/s": round(n / ms / 1e6, 2), "GB/s_in": round((mat.nbytes + offsets.nbytes) / ms / 1e6, 1), "ndict": nd[0]}
        del ofb, db, idb, frb
print(json.dumps({"rows": n, "results": res}))
