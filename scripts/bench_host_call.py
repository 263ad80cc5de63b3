"""What one CallFunction costs on top of its kernel: Int64 add_unchecked / greater / filter over device-resident 2^27-row
arrays through the host layer (allocation of the output included, as in the Go executor)."""
import os, sys, time, json
import numpy as np
import pyarrow as pa
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from arrow_go_amd import compute as ac
s = ac.Session(0)
n = 1 << 27
rng = np.random.default_rng(0)
a = pa.array(rng.integers(-10**9, 10**9, n))
da = s.call_function("add_unchecked", [a, pa.scalar(1, pa.int64())], keep_on_device=True)
db = s.call_function("add_unchecked", [a, pa.scalar(2, pa.int64())], keep_on_device=True)
del a
res = {}
def timed(name, fn, reps=10):
    out = fn(); del out
    lib_sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn(); del out
    lib_sync()
    res[name] = round((time.perf_counter() - t0) * 1e3 / reps, 3)
def lib_sync():
    import ctypes as C
    x = s.call_function("add_unchecked", [pa.array([1]), pa.array([1])])   # round trip = stream sync
timed("add_unchecked 2^27 (ms per call, output allocated)", lambda: s.call_function("add_unchecked", [da, db], keep_on_device=True))
timed("greater 2^27", lambda: s.call_function("greater", [da, db], keep_on_device=True))
m = s.call_function("greater", [da, db], keep_on_device=True)
timed("filter 2^27", lambda: s.call_function("filter", [da, m], keep_on_device=True))
timed("cumulative_sum 2^27", lambda: s.call_function("cumulative_sum", [da], keep_on_device=True))
timed("sort_indices 2^27", lambda: s.call_function("sort_indices", [da], "order=ascending", keep_on_device=True), reps=3)
print(json.dumps(res))
