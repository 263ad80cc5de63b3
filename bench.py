#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on BASELINE.json's config.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (configs[1], "C2"): one STEP = Int64 Add (array + array → array) followed by
Float64 Sum, each over a contiguous 2^27-row (1 GiB) Arrow value buffer that is already
resident in HBM — 24 + 8 = 32 algorithmic bytes per row per step.  With
N > 1 every rank owns its own record-batch shard of the same size (weak scaling) and the
per-shard Float64 partial sums are combined by ONE RCCL all-reduce of 8 bytes per step —
the only exchange step this path has.

`value` = rows·32·N / step time, in GB/s (whole job).  `roofline` is for the dominant
kernel (the Int64 Add: 24 of the 32 bytes), timed with HIP events recorded on the
library's own compute stream around every Add launch INSIDE the timed region.
`cpu_baseline` = the reference's own AVX2 machine code (oracle/_ref/libref_avx2.so,
assembled from the reference's clang output) running the same step on one host core
over a bounded sample.  `kernels` lists achieved algorithmic GB/s for the other
kernels of the metric (Sum / Add / Compare / Filter / Take / fused), measured after the
timed region.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling ≈ 6.3 TB/s


PMC_PROFILE = "r06_pmc_by_workload.json"   # THIS round's counters (scripts/prof_workloads.py + pmc_by_workload.py on the GPU box)


def pmc_traffic(rows):
    """(HBM bytes per launch of the Int64 Add kernel, where the figure comes from): the current round's committed rocprofv3
    PMC passes (profiles/r06_pmc_by_workload.json — FETCH_SIZE with the calibrated gfx950 streaming factor + WRITE_SIZE), or
    (None, why) when that file is absent or was taken at another size — never an older round's file."""
    path = os.path.join(ROOT, "profiles", PMC_PROFILE)
    try:
        d = json.load(open(path))
    except Exception:
        return None, f"profiles/{PMC_PROFILE} not present (counters are collected in a separate rocprofv3 --pmc run)"
    if d.get("rows") != rows:
        return None, f"profiles/{PMC_PROFILE} was taken at {d.get('rows')} rows, this run has {rows}"
    for w in d.get("workloads", {}).values():
        for k, v in w.get("kernels", {}).items():
            if "binary_kernel<unsigned long, 0, 0, true" in k and "fetch_kib_raw" in v:
                # streaming 16-byte lane loads are tallied at half their bytes (128-byte requests counted as 64: the file's own
                # calibration rows show it), writes at face value
                return int(v["fetch_kib_raw"] * 1024 * 2 + v["write_kib"] * 1024), f"profiles/{PMC_PROFILE}: FETCH_SIZE x2 (calibrated) + WRITE_SIZE"
    return None, f"profiles/{PMC_PROFILE} holds no row for the Int64 Add kernel"


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--rows", type=int, default=1 << 27, help="rows per GPU (default 2^27 = 1 GiB columns)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-c5-merge", action="store_true",
                   help="C5 secondary line: skip the merge of the per-GPU aggregates by key-hash owner (all-to-all of group tuples over "
                        "RCCL), which is ON whenever more than one rank runs — the config is 'partitioned by key-hash over xGMI'")
    p.add_argument("--collectives", choices=["auto", "ah", "torch"], default="auto",
                   help="who performs the data-path collectives under torch.distributed.run: 'ah' = ah_comm_* of libarrowhip.so (RCCL through "
                        "the C ABI, on the library's stream — what a Go host would call; a failure to set it up is fatal), 'torch' = "
                        "torch.distributed's nccl group, 'auto' (default) = 'ah', and if EVERY rank agrees it could not be set up, 'torch' with the "
                        "reason written into config.collectives (never silently)")
    p.add_argument("--secondary-timeout", type=float, default=150.0,
                   help="multi-GPU runs: seconds the sections after the headline (C4 / C5 lines) may take before rank 0 prints the headline "
                        "line without them and every rank exits — a rank stuck in a collective must not cost the run its number")
    p.add_argument("--transport", choices=["rccl", "gloo"], default="rccl",
                   help="multi-rank runs: 'rccl' (default) = one GPU per rank, ah_comm over RCCL / xGMI; 'gloo' = the host-transport communicator "
                        "(ah_comm_init_transport) carried by a gloo group, every rank on device LOCAL_RANK mod the visible devices — lets the "
                        "bench's own N > 1 code (communicator vote, watchdog, C4 / C5 with the owner merge) run on a 1-GPU box; not a measurement")
    p.add_argument("--dump", default=None, help="rank 0 writes the last C4 result and the merged C5 groups to this .npz (tests compare them with the oracle)")
    p.add_argument("--no-kernels", action="store_true", help="skip the per-kernel table")
    p.add_argument("--traffic", type=float, default=None, help="HBM bytes/launch of the Add kernel from a rocprofv3 --pmc pass")
    return p.parse_args()


def fill_random(ctx, buf, rows, dtype, seed):
    """tile one random 2^22-row chunk across the buffer (content is irrelevant to a
    bandwidth-bound kernel, but it must not be zeros: DVFS clocks higher on zeros)"""
    rng = np.random.default_rng(seed)
    chunk_rows = min(rows, 1 << 22)
    if np.dtype(dtype).kind == "f":
        chunk = rng.uniform(-1e6, 1e6, chunk_rows).astype(dtype)
    else:
        chunk = rng.integers(-2**62, 2**62, chunk_rows, dtype=dtype)
    w = np.dtype(dtype).itemsize
    total = 0
    exact_chunks = 0
    for off in range(0, rows, chunk_rows):
        m = min(chunk_rows, rows - off)
        buf.upload(chunk[:m], off * w)
        total += 1
    return chunk


def cpu_baseline(sample_rows, reps):
    """Reference AVX2 kernels (or, if oracle/_ref was never built, our C port) on one core.  The Add writes into a
    PREALLOCATED, already-touched output (the rate of the kernel itself); the rate with a fresh zero-filled output per call —
    what the Go executor pays through memory.Allocator, page faults included — is reported beside it in `sample`."""
    import ctypes as C
    from tests import oracle_lib as OL
    ref = OL.load_reference()
    rng = np.random.default_rng(1)
    a = rng.integers(-2**62, 2**62, sample_rows, dtype=np.int64)
    b = rng.integers(-2**62, 2**62, sample_rows, dtype=np.int64)
    x = rng.uniform(-1e6, 1e6, sample_rows)
    out = np.ones(sample_rows, np.int64)
    if ref is not None:
        kind = "reference"
        f = ref.avx2.arithmetic_binary_avx2
        f.restype = None
        f.argtypes = [C.c_int, C.c_int8, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        add = lambda: f(OL.TYPE_IDS[a.dtype], 0, a.ctypes.data, b.ctypes.data, out.ctypes.data, sample_rows)
        add_alloc = lambda: ref.arithmetic(0, 0, a, b)
        ssum = lambda: ref.sum("avx2", x)
    else:
        kind = "port"
        o = OL.load_oracle()
        add = add_alloc = lambda: o.arithmetic(0, 0, a, b)
        ssum = lambda: o.sum_float64_seq(x)

    def rate(add_fn):
        add_fn(); ssum()  # warm
        t0 = time.perf_counter()
        for _ in range(reps):
            add_fn()
            ssum()
        return 32.0 * sample_rows * reps / (time.perf_counter() - t0) / 1e9

    gbs, gbs_alloc = rate(add), rate(add_alloc)
    assert out.tobytes() == (a + b).tobytes(), "cpu baseline: reference Add produced wrong values"
    return {"value": round(gbs, 3), "unit": "GB/s", "cores": 1, "kind": kind,
            "sample": f"{reps} steps of Int64 Add + Float64 Sum over {sample_rows} rows "
                      f"({'reference AVX2 machine code, oracle/_ref/libref_avx2.so' if kind == 'reference' else 'oracle C port'}; "
                      f"output preallocated and touched; {gbs_alloc:.1f} GB/s with a fresh zero-filled output per call = first touch "
                      f"included, as the Go executor pays), host has {os.cpu_count()} logical cores"}


def cpu_baseline_mt():
    """the same reference AVX2 kernels on ALL host cores (oracle/_ref/bench_ref_mt: N threads over contiguous 1/N shards with a
    final combine — what a Go user gets from chunking + ExecCtx.NumParallel), 2^27-row columns"""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "bench_ref_mt")
    if not os.path.exists(exe):
        return {"error": "oracle/_ref/bench_ref_mt not built"}
    nproc = os.cpu_count() or 1
    cand = sorted({nproc, max(1, nproc // 2), max(1, nproc // 4)}, reverse=True)[:3]
    t0 = time.perf_counter()
    r = json.loads(subprocess.run([exe, os.path.join(ROOT, "oracle", "_ref")] + [str(t) for t in cand], capture_output=True, text=True, timeout=240, check=True).stdout)
    best = None
    for t in cand:
        ms = r[f"Int64_Add_1GiB_threads{t}"]["ms"] + r[f"Float64_Sum_1GiB_threads{t}"]["ms"]
        gbs = 32.0 * (1 << 27) / (ms * 1e-3) / 1e9
        if best is None or gbs > best[0]:
            best = (gbs, t, ms)
    return {"value": round(best[0], 1), "unit": "GB/s", "cores": best[1], "kind": "reference",
            "sample": f"best of {cand} threads: Int64 Add + Float64 Sum over 2^27-row columns, each thread a contiguous shard, best of 3 timed "
                      f"repetitions ({best[2]:.1f} ms per step; whole measurement {time.perf_counter() - t0:.0f} s)", "all": r}


def cpu_baseline_c3_c5():
    """CPU baselines beside configs C3 and C5: the oracle's C ports of the reference's Go kernels (the reference has no assembly for
    them) on the host cores — PrimitiveFilter at s = 0.5 with 10 % nulls and PrimitiveTake with uniformly random Int32 indices and
    10 % nulls on both sides (kernels/vector_selection.go:267-395, 878-988), on 1 thread and on all cores (contiguous shards, as a
    chunked column under ExecCtx.NumParallel); dictionary_encode over 2^16 keys and the hash + sum built on it
    (hashing/xxh3_memo_table_types.go:283-294, vector_hash.go:359-385) on ONE core — the reference feeds every chunk through one memo
    table.  2^24 rows; the ports are checked against the reference's golden vectors (tests/test_golden.py) before anything is timed."""
    import subprocess
    import numpy as np
    from tests import test_golden as TG
    from tests.backends import OracleBackend
    be = OracleBackend()
    TG.test_filter_vectors(be, np.int64, False)
    TG.test_filter_null_payload_rules(be)
    TG.test_take_vectors(be, np.int64, np.int32)
    TG.test_unique_vectors(be, np.int64)
    TG.test_dictionary_encode_vectors(be)
    TG.test_dictionary_encode_resizes_memo_table(be)
    exe = os.path.join(ROOT, "oracle", "_ref", "bench_port_mt")
    if not os.path.exists(exe):
        return {"error": "oracle/_ref/bench_port_mt not built (make -C oracle port_bench)"}, {"error": "oracle/_ref/bench_port_mt not built"}
    nproc = os.cpu_count() or 1
    t0 = time.perf_counter()
    r = json.loads(subprocess.run([exe, os.path.join(ROOT, "oracle"), "24", "1", str(nproc)], capture_output=True, text=True, timeout=300, check=True).stdout)
    took = time.perf_counter() - t0
    note = (f"oracle C ports (kind 'port': the reference's kernels for this config are Go, restated in oracle/orc_select.c / orc_hash.c and pinned to its "
            f"golden vectors, checked again before this timing), {r['rows']} rows, best of 3 repetitions; whole measurement {took:.0f} s")
    f1, fn = r["filter_int64_nulls10_sel0.50_threads1"], r[f"filter_int64_nulls10_sel0.50_threads{nproc}"]
    t1, tn = r["take_int64_random_i32_nulls10_threads1"], r[f"take_int64_random_i32_nulls10_threads{nproc}"]
    c3 = {"value": f1["GB/s"], "unit": "GB/s (Filter s = 0.5, 10 % nulls: algorithmic bytes, as the GPU line counts them)", "cores": 1, "kind": "port",
          "sample": "PrimitiveFilter (Drop) " + note,
          "filter": {"threads1": f1, f"threads{nproc}": fn}, "take_random_i32_nulls10": {"threads1": t1, f"threads{nproc}": tn},
          "all_cores": {"cores": nproc, "filter_GB/s": fn["GB/s"], "take_GB/s": tn["GB/s"]}}
    e1, h1 = r["dictionary_encode_int64_2^16_keys_threads1"], r["hash_sum_float64_2^16_groups_threads1"]
    c5 = {"value": h1["Mrows/s"], "unit": "M rows/s (hash + sum, 2^16 groups)", "cores": 1, "kind": "port",
          "sample": "dictionary_encode + row-order accumulation " + note + "; one core only: the reference's memo table is one sequential structure "
                    "shared by all chunks of a column (vector_hash.go), so there is no N-core figure to quote for it",
          "dictionary_encode_2^16_keys": e1, "hash_sum_float64_2^16_groups": h1}
    return c3, c5


def random_bits(rng, n, p, pad=64):
    """n Bernoulli(p) bits packed LSB-first"""
    out = np.zeros(n // 8 + pad, np.uint8)
    for i in range(0, n, 1 << 24):
        m = min(1 << 24, n - i)
        out[i // 8:i // 8 + (m + 7) // 8] = np.packbits(rng.random(m) < p, bitorder="little")
    return out


def zipf_ranks(rng, n, card, s=1.1):
    """ranks of a Zipf(s) distribution truncated to `card` values (inverse CDF)"""
    w = 1.0 / np.arange(1, card + 1) ** s
    cdf = np.cumsum(w)
    cdf /= cdf[-1]
    return np.searchsorted(cdf, rng.random(n)).astype(np.uint64)


def per_kernel_table(ctx, rows, a, b, c, x):
    """achieved algorithmic GB/s per kernel, HIP events on the ctx stream (10 launches each)"""
    import arrow_go_amd as ah
    N = ah._native
    reps = 10
    out = {}

    def timed(name, nbytes, fn, reps=reps, warm=5):
        # warm-up calls: code, TLB — and the clocks.  A line measured right behind a host-side pause (a mask generated with numpy, a
        # column uploaded) read 7 % high with one warm call: the first tens of launches after an idle stretch run below the
        # steady-state clock (scripts/bench_filter_cache.py: the same two-phase Filter 0.2995 ms first, 0.2781 ms a few ms later); five calls, eight where
        # a new cardinality makes the first calls size tables and arenas (dictionary_encode at 2^20 keys: 4.5 ms, then 1.27, 1.26, 1.23, 1.20 …)
        for r in range(reps + 1):
            ctx.event_record(1000 + r)   # (events exist before the timed region)
        for _ in range(warm):
            fn()
        ctx.event_record(1000)
        for r in range(reps):
            fn()
            ctx.event_record(1001 + r)
        ms = ctx.event_elapsed_ms(1000, 1000 + reps) / reps
        out[name] = {"ms": round(ms, 4), "GB/s": round(nbytes / ms / 1e6, 1), "bytes": int(nbytes)}
        each = [ctx.event_elapsed_ms(1000 + r, 1001 + r) for r in range(reps)]
        if max(each) > 1.3 * min(each):   # a call that took another path or waited for an allocation: shown, not averaged away
            out[name]["ms_each"] = [round(v, 4) for v in each]

    res = ctx.alloc(64)
    # the chip's streaming ceiling in this very session: a device-to-device copy of 1 GiB, 2 bytes moved per byte copied
    from arrow_go_amd._native import check as _check, lib as _lib
    # (ah_copy_async = the library's own 16-byte-per-lane copy kernel; hipMemcpyDtoD, which rounds 1-2 reported here, stops at ≈ 5 TB/s)
    timed("ceiling_copy_kernel", 16 * rows, lambda: _check(ctx.handle, _lib.ah_copy_async(ctx.handle, c.ptr, a.ptr, 8 * rows)))
    timed("sum_float64", 8 * rows, lambda: ctx.sum_float64_dev(x, rows, res))
    timed("sum_int64", 8 * rows, lambda: ctx.sum_int64_dev(a, rows, res))
    timed("add_int64", 24 * rows, lambda: ctx.arithmetic(N.INT64, N.OP_ADD, N.SHAPE_AA, a, b, c, rows))
    timed("add_float64", 24 * rows, lambda: ctx.arithmetic(N.FLOAT64, N.OP_ADD, N.SHAPE_AA, x, b, c, rows))
    timed("add_int64_scalar", 16 * rows, lambda: ctx.arithmetic(N.INT64, N.OP_ADD, N.SHAPE_AS, a, np.array([7], np.int64), c, rows))
    mask = ctx.alloc(rows // 8 + 64)
    thr = np.array([0], np.int64)
    timed("greater_int64_scalar", 8.125 * rows, lambda: ctx.comparison(N.CMP_GT, N.SHAPE_AS, N.INT64, a, thr, mask, rows, 0))
    # ---- C3: Filter on a 1 GiB Int64 column with 10 % value nulls: Bernoulli masks at four selectivities, Drop;
    # Drop / Emit with 10 % mask nulls; a mask of geometric runs (mean 256).  Per line: "GB/s" = the full algorithmic traffic
    # 8 + 1/8 (mask) + 1/8 (value validity) [+ 1/8 mask validity] per input row + (8 + 1/8) per output row; "input_GB/s" = 8·n / t,
    # the "GB/s processed" headline of BASELINE.md.
    rng = np.random.default_rng(5)
    vvalid = ctx.to_device(random_bits(rng, rows, 0.9))
    ovalid = ctx.alloc(rows // 8 + 64)
    fmask = ctx.alloc(rows // 8 + 64)
    fvalid = ctx.to_device(random_bits(rng, rows, 0.9))

    def filter_case(name, fv, null_sel):
        # the whole two-phase call the protocol needs (vector_selection.go:459 then :475): ah_filter_count (the host gets n_out and
        # sizes the output; its tile prefixes stay in the context for the fill) + ah_filter_primitive.  The fill alone — what this
        # line timed in rounds 1–2 — is kept beside it as "fill_only_ms".
        n_out = ctx.filter_count(fmask, fv, 0, rows, null_sel)
        traffic = (8 + 0.125 + 0.125 + (0.125 if fv is not None else 0)) * rows + (8 + 0.125) * n_out

        def call():
            k = ctx.filter_count(fmask, fv, 0, rows, null_sel)
            ctx.filter_primitive(8, a, vvalid, 0, fmask, fv, 0, rows, null_sel, k, c, ovalid, want_null_count=False)

        timed(name + "_fill_only", traffic, lambda: ctx.filter_primitive(8, a, vvalid, 0, fmask, fv, 0, rows, null_sel, n_out, c, ovalid, want_null_count=False))
        fill_only = out.pop(name + "_fill_only")["ms"]
        timed(name, traffic, call, warm=12)
        out[name]["fill_only_ms"] = fill_only
        # the ONE-call flavour for a caller that sizes its output for n rows (ah_filter_primitive_once): counts + fill back to back, the
        # selection count through the mailbox at the end — no turnaround between two launches
        timed(name + "_once", traffic, lambda: ctx.filter_primitive_once(8, a, vvalid, 0, fmask, fv, 0, rows, null_sel, c, ovalid))
        out[name]["one_call_ms"] = out.pop(name + "_once")["ms"]
        out[name]["input_GB/s"] = round(8 * rows / out[name]["ms"] / 1e6, 1)
        out[name]["selected"] = round(n_out / rows, 4)

    for sel in (0.01, 0.1, 0.5, 0.9):
        fmask.upload(random_bits(rng, rows, sel))
        filter_case("filter_int64_nulls10_sel%.2f" % sel, None, 0)
        if sel == 0.5:
            filter_case("filter_int64_nulls10_sel0.50_masknulls10_drop", fvalid, 0)
            filter_case("filter_int64_nulls10_sel0.50_masknulls10_emit", fvalid, 1)
    runs = rng.geometric(1 / 256, rows // 200)
    on = np.repeat(np.arange(runs.size) % 2 == 0, runs)[:rows]
    fmask.upload(np.packbits(np.concatenate([on, np.zeros(rows - on.size, bool)]), bitorder="little"))
    del runs, on
    filter_case("filter_int64_nulls10_runs256", None, 0)
    out["filter_input_GB/s"] = out["filter_int64_nulls10_sel0.50"]["input_GB/s"]
    timed("filter_count", 0.125 * rows, lambda: ctx.filter_count(fmask, None, 0, rows, 0))
    # mask for the later lines (bitmap_and, count_set_bits): a > 0
    ctx.comparison(N.CMP_GT, N.SHAPE_AS, N.INT64, a, thr, mask, rows, 0)
    # ---- C3: Take, 2^27 int32 indices into the 1 GiB column.  "random" = 2^27 INDEPENDENT uniform draws over [0, rows) —
    # every line of the column is a target; with 10 % nulls on both sides (the stated config) and without; then sorted-random,
    # identity, reverse.  20 algorithmic bytes per output row (+ 3/8 with validity: two bitmaps read, one written).
    idx = ctx.alloc(rows * 4 + 64)
    ivalid = ctx.to_device(random_bits(rng, rows, 0.9))
    ridx = rng.integers(0, rows, rows, dtype=np.int32)

    def take_case(name, nulls, reps=3, first_take=False):
        timed(name, (20 + (0.375 if nulls else 0)) * rows,
              lambda: ctx.take_primitive(8, a, vvalid if nulls else None, 0, rows, 4, True, idx, ivalid if nulls else None, 0, rows, True, c,
                                         ovalid if nulls else None), reps=reps)
        if first_take:
            # The line above calls Take repeatedly with ONE index vector: from the second call on the context's hint cache (option
            # take_hint_cache, csrc/ah_take.hip) skips the path sample and its host wait — the steady state of "the second column of a
            # record batch".  The same calls with the cache off = every call a first Take of its index vector, what rounds 1–4 timed.
            ctx.set_option("take_hint_cache", 0)
            try:
                take_case(name + "_first_take", nulls, reps)
            finally:
                ctx.set_option("take_hint_cache", 1)
            out[name]["first_take_ms"] = out[name + "_first_take"]["ms"]

    idx.upload(ridx)
    take_case("take_int64_random_i32_nulls10", True)
    take_case("take_int64_random_i32", False)
    ctx.set_option("take_binned", 0)
    take_case("take_int64_random_i32_nulls10_direct_kernel", True)
    ctx.set_option("take_binned", 1)
    ridx.sort()
    idx.upload(ridx)
    take_case("take_int64_sorted_random_i32_nulls10", True, first_take=True)
    del ridx
    idx.upload(np.arange(rows, dtype=np.int32))
    take_case("take_int64_identity_i32", False)
    take_case("take_int64_identity_i32_nulls10", True, first_take=True)
    idx.upload(np.arange(rows - 1, -1, -1, dtype=np.int32))
    take_case("take_int64_reverse_i32_nulls10", True)
    fmask.free(); fvalid.free(); ivalid.free()
    timed("fused_gt_filter_sum_int64", 8 * rows, lambda: ctx.cmp_filter_sum_i64_dev(N.CMP_GT, a, None, 0, rows, 0, res))
    timed("fused_gt_filter_sum_int64_nulls10", 8.125 * rows, lambda: ctx.cmp_filter_sum_i64_dev(N.CMP_GT, a, vvalid, 0, rows, 0, res))
    # (f) rows: cumulative_sum (16 B/row algorithmic; 24 moved: reduce-then-scan) and numeric cast (w_in + w_out)
    timed("cumulative_sum_int64", 16 * rows, lambda: ctx.cumulative_sum(N.INT64, a, None, 0, rows, None, False, False, c, None))
    timed("cumulative_sum_float64", 16 * rows, lambda: ctx.cumulative_sum(N.FLOAT64, x, None, 0, rows, None, False, False, c, None))
    # checked sums need a column whose running sums stay in range: small values (one 2^22-row chunk tiled), 10 % nulls skipped
    sm = ctx.alloc(rows * 8 + 64)
    sm_chunk = np.random.default_rng(11).integers(-1000, 1000, min(rows, 1 << 22), dtype=np.int64)
    for o in range(0, rows, sm_chunk.size):
        sm.upload(sm_chunk[:min(sm_chunk.size, rows - o)], o * 8)
    timed("cumulative_sum_checked_int64", 16 * rows, lambda: ctx.cumulative_sum(N.INT64, sm, None, 0, rows, None, False, True, c, None))
    timed("cumulative_sum_checked_int64_nulls10", 16.25 * rows, lambda: ctx.cumulative_sum(N.INT64, sm, vvalid, 0, rows, None, True, True, c, ovalid))
    timed("cumulative_sum_int64_nulls10", 16.25 * rows, lambda: ctx.cumulative_sum(N.INT64, a, vvalid, 0, rows, None, True, False, c, ovalid))
    sm.free()
    timed("cast_int64_to_int32_unsafe", 12 * rows, lambda: ctx.cast_numeric(N.INT64, N.INT32, a, None, 0, rows, True, True, c))
    timed("cast_int32_to_int64", 12 * rows, lambda: ctx.cast_numeric(N.INT32, N.INT64, a, None, 0, rows, False, False, c))
    timed("cast_float64_to_float32", 12 * rows, lambda: ctx.cast_numeric(N.FLOAT64, N.FLOAT32, x, None, 0, rows, False, False, c))
    # temporal unit change (ShiftTime): ns → us truncating, then us → ns with the overflow check on (16 B/row)
    t_us = ctx.alloc(rows * 8)
    timed("shift_time_div1000_int64", 16 * rows, lambda: ctx.shift_time(64, 64, 1, 1000, False, a, None, 0, rows, t_us))
    timed("shift_time_mul1000_int64_checked", 16 * rows, lambda: ctx.shift_time(64, 64, 0, 1000, True, t_us, None, 0, rows, c))
    timed("shift_time_mul1000_int64_checked_nulls10", 16.125 * rows, lambda: ctx.shift_time(64, 64, 0, 1000, True, t_us, vvalid, 0, rows, c))
    del t_us
    # sort_indices (row (f)-2): stable argsort of the 2^27-row columns; 16 B/row algorithmic (8 read + the uint64 index written).
    # Keys are independent draws (the tiled columns above repeat every key 32 times): full-range Int64, then Float64 from a normal
    # distribution — the MSD path — and the Int64 column again through the LSD passes alone.
    step = min(rows, 1 << 22)
    for off in range(0, rows, step):
        b.upload(rng.integers(-2**63, 2**63 - 1, step, dtype=np.int64), off * 8)
    timed("sort_indices_int64_random", 16 * rows, lambda: ctx.sort_indices(N.INT64, b, None, 0, rows, False, False, c), reps=3)
    ctx.set_option("sort_msd", 0)
    timed("sort_indices_int64_random_lsd_passes_only", 16 * rows, lambda: ctx.sort_indices(N.INT64, b, None, 0, rows, False, False, c), reps=2)
    ctx.set_option("sort_msd", 1)
    for off in range(0, rows, step):
        b.upload(rng.standard_normal(step), off * 8)
    timed("sort_indices_float64_normal", 16 * rows, lambda: ctx.sort_indices(N.FLOAT64, b, None, 0, rows, False, False, c), reps=3)
    timed("min_max_int64", 8 * rows, lambda: ctx.min_max(N.INT64, a, rows, np.int64))
    timed("bitmap_and", 0.375 * rows, lambda: ctx.bitmap_op(N.BIT_AND, mask, 0, vvalid, 0, ovalid, 0, rows))
    timed("count_set_bits", 0.125 * rows, lambda: ctx.count_set_bits(mask, 0, rows))
    # ---- C5 per GPU: 2^26 Int64 keys (512 MiB; 4 GiB over 8 GPUs) + Float64 values — dictionary_encode and hash + sum at the
    # cardinalities of SURVEY §8d, uniform keys, plus one Zipf(1.1) column over 2^20 keys.  Keys are 2^26 independent draws.
    # Algorithmic bytes: 8 (key) + 4 (id) per row for encode, 8 + 8 for the group-by (outputs are per group).
    hrows = min(rows, 1 << 26)
    hids = ctx.alloc(hrows * 4 + 64)
    hdic, hsum, hcnt = (ctx.alloc((hrows + 1) * 8 + 64) for _ in range(3))
    mult = np.uint64(0x9E3779B97F4A7C15)
    for lg, dist in ((10, "uniform"), (16, "uniform"), (20, "uniform"), (24, "uniform"), (20, "zipf1.1")):
        ranks = zipf_ranks(rng, hrows, 1 << lg) if dist != "uniform" else rng.integers(0, 1 << lg, hrows, dtype=np.uint64)
        c.upload((ranks * mult).view(np.int64))
        del ranks
        tag = "2^%d_%s" % (lg, dist) if dist != "uniform" else "2^%d" % lg
        timed("dictionary_encode_int64_%s_keys" % tag, 12 * hrows, lambda: ctx.hash_u64_encode(c, None, 0, hrows, False, hids, None, hdic), reps=3, warm=8)   # (the first calls at a new cardinality still size tables and arenas)
        timed("hash_sum_float64_%s_groups" % tag, 16 * hrows, lambda: ctx.hash_sum("f64", c, None, 0, x, None, 0, hrows, hdic, hsum, hcnt), reps=3, warm=8)
        out["hash_sum_float64_%s_groups" % tag]["Grows/s"] = round(hrows / out["hash_sum_float64_%s_groups" % tag]["ms"] / 1e6, 2)
    for bfr in (res, mask, vvalid, ovalid, idx, hids, hdic, hsum, hcnt):
        bfr.free()
    return out


def host_ingest_legs(ctx, rows):
    """The same kernels when the Arrow buffers start (and end) in PINNED HOST memory: ah_ingest_* cuts them into 32 MiB chunks and
    overlaps upload k + 1 / kernel k / download k − 1 on three streams with per-slot events (csrc/ah_ingest.hip).  Wall-clock of
    the synchronous calls; `h2d_GB/s` = bytes that crossed PCIe towards the device ÷ time, to be read against `h2d_pinned_1GiB`
    (one plain pinned hipMemcpyAsync of the same size, the rate of the link) — an ingest that overlaps runs at the link's rate."""
    import arrow_go_amd as ah
    N = ah._native
    out = {}
    rng = np.random.default_rng(99)
    nbytes = rows * 8
    pa, pb, po = ctx.alloc_pinned(nbytes + 64), ctx.alloc_pinned(nbytes + 64), ctx.alloc_pinned(nbytes + 64)
    va, vb, vo = pa.view(np.int64, rows), pb.view(np.int64, rows), po.view(np.int64, rows)
    chunk = rng.integers(-2**62, 2**62, 1 << 22, dtype=np.int64)
    for off in range(0, rows, 1 << 22):
        m = min(1 << 22, rows - off)
        va[off:off + m] = chunk[:m]
        vb[off:off + m] = chunk[:m][::-1]
    vx = pb.view(np.float64, rows)   # the same bytes read as doubles would hold NaNs: a separate fill for the Sum leg below
    dev = ctx.alloc(nbytes + 64)

    def wall(name, fn, h2d_bytes, d2h_bytes, reps=3):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        t = min(ts)
        out[name] = {"ms": round(t * 1e3, 3), "h2d_GB/s": round(h2d_bytes / t / 1e9, 2), "pcie_both_ways_GB/s": round((h2d_bytes + d2h_bytes) / t / 1e9, 2),
                     "ms_mean": round(float(np.mean(ts)) * 1e3, 3)}

    def plain_upload():
        ah._native.check(ctx.handle, ah._native.lib.ah_upload_async(ctx.handle, dev.ptr, pa.ptr, nbytes))
        ctx.sync()

    def plain_download():
        ah._native.check(ctx.handle, ah._native.lib.ah_download_async(ctx.handle, po.ptr, dev.ptr, nbytes))
        ctx.sync()

    wall("h2d_pinned_1GiB", plain_upload, nbytes, 0)
    wall("d2h_pinned_1GiB", plain_download, 0, nbytes)
    out["d2h_pinned_1GiB"]["d2h_GB/s"] = round(nbytes / (out["d2h_pinned_1GiB"]["ms"] * 1e-3) / 1e9, 2)
    link = out["h2d_pinned_1GiB"]["h2d_GB/s"]
    ing = ah.Ingest(ctx)   # 32 MiB chunks, 3 slots
    try:
        # Int64 Add: 2 GiB up, 1 GiB down
        wall("add_int64_from_pinned_host", lambda: ing.arithmetic_binary(N.INT64, N.OP_ADD, va, vb, vo, rows), 2 * nbytes, nbytes)
        assert vo[:4096].tobytes() == (va[:4096] + vb[:4096]).tobytes() and vo[-4096:].tobytes() == (va[-4096:] + vb[-4096:]).tobytes(), "ingest Add: wrong values"
        # the unpipelined way (what rounds 1-2 had): upload both, compute, download, each behind the other
        b2 = ctx.alloc(nbytes + 64); c2 = ctx.alloc(nbytes + 64)

        def serial_add():
            ah._native.check(ctx.handle, ah._native.lib.ah_upload_async(ctx.handle, dev.ptr, pa.ptr, nbytes))
            ah._native.check(ctx.handle, ah._native.lib.ah_upload_async(ctx.handle, b2.ptr, pb.ptr, nbytes))
            ctx.arithmetic(N.INT64, N.OP_ADD, N.SHAPE_AA, dev, b2, c2, rows)
            ah._native.check(ctx.handle, ah._native.lib.ah_download_async(ctx.handle, po.ptr, c2.ptr, nbytes))
            ctx.sync()

        wall("add_int64_from_pinned_host_unpipelined", serial_add, 2 * nbytes, nbytes)
        b2.free(); c2.free()
        # Filter: s = 0.5, 10 % value nulls; mask + validity up (32 MiB), values up (1 GiB), survivors down
        fmask, vvalid = random_bits(rng, rows, 0.5), random_bits(rng, rows, 0.9)
        pov = ctx.alloc_pinned(rows // 8 + 64)
        kk = [0]

        def filt():
            kk[0] = ing.filter_count(fmask, None, 0, rows, 0)
            ing.filter_primitive(8, va, vvalid, 0, rows, kk[0], vo, pov)

        wall("filter_int64_sel0.50_nulls10_from_pinned_host", filt, nbytes + rows // 4, 0)
        k = kk[0]
        out["filter_int64_sel0.50_nulls10_from_pinned_host"]["pcie_both_ways_GB/s"] = round((nbytes + rows // 4 + k * 8 + k // 8) / (out["filter_int64_sel0.50_nulls10_from_pinned_host"]["ms"] * 1e-3) / 1e9, 2)
        out["filter_int64_sel0.50_nulls10_from_pinned_host"]["selected"] = round(k / rows, 4)
        sel = np.unpackbits(fmask[:512], bitorder="little")[:4096].astype(bool)
        assert vo[:int(sel.sum())].tobytes() == va[:4096][sel].tobytes(), "ingest Filter: wrong values"
        pov.free()
        # Float64 Sum: 1 GiB up, 8 bytes down
        xchunk = rng.uniform(-1e6, 1e6, 1 << 22)
        for off in range(0, rows, 1 << 22):
            vx[off:off + min(1 << 22, rows - off)] = xchunk[:min(1 << 22, rows - off)]
        res = [0.0]

        def ssum():
            res[0] = ing.sum_float64(vx, rows)

        wall("sum_float64_from_pinned_host", ssum, nbytes, 0)
        exact = math.fsum(xchunk.tolist()) * (rows // (1 << 22)) if rows % (1 << 22) == 0 else None
        if exact is not None:
            assert abs(res[0] - exact) <= 2 * abs(np.spacing(exact)) * (rows // (1 << 22)), "ingest Sum: wrong value"
    finally:
        ing.close()
    for name, v in out.items():
        if name.endswith("_from_pinned_host"):
            v["frac_of_h2d_link"] = round(v["h2d_GB/s"] / link, 3)
    for b_ in (pa, pb, po, dev):
        b_.free()
    return out


def graph_replay_legs(ctx):
    """Launch-bound chains (config C1's size class): Add → Compare → bitmap AND → fused compare-filter-sum → Sum over Int64 columns of
    2^16 and 2^20 rows, timed eager (five submissions) and as one captured hipGraph (ah_graph_*)."""
    import arrow_go_amd as ah
    N = ah._native
    out = {}
    for lg in (16, 20):
        n = 1 << lg
        rng = np.random.default_rng(lg)
        a = ctx.to_device(rng.integers(-1000, 1000, n, dtype=np.int64)); b = ctx.to_device(rng.integers(-1000, 1000, n, dtype=np.int64))
        c = ctx.alloc(n * 8); m1, m2, m3 = (ctx.alloc(n // 8 + 64) for _ in range(3))
        m2.memset(0xA5)
        r1, r2 = ctx.alloc(16), ctx.alloc(8)
        thr = np.array([5], np.int64)

        def chain():
            ctx.arithmetic(N.INT64, N.OP_ADD, N.SHAPE_AA, a, b, c, n)
            ctx.comparison(N.CMP_GT, N.SHAPE_AS, N.INT64, c, thr, m1, n, 0)
            ctx.bitmap_op(N.BIT_AND, m1, 0, m2, 0, m3, 0, n)
            ctx.cmp_filter_sum_i64_dev(N.CMP_GT, c, m2, 0, n, 5, r1)
            ctx.sum_int64_dev(c, n, r2)

        chain(); ctx.sync()
        ctx.graph_begin(); chain(); g = ctx.graph_end()
        reps = 200

        def timed(fn):
            fn()
            ctx.event_record(1002)
            for _ in range(reps):
                fn()
            ctx.event_record(1003)
            return ctx.event_elapsed_ms(1002, 1003) / reps

        eager, replay = timed(chain), timed(g.launch)
        out[f"2^{lg}_rows"] = {"calls": 5, "eager_ms": round(eager, 5), "graph_ms": round(replay, 5), "speedup": round(eager / replay, 2)}
        g.close()
        for d in (a, b, c, m1, m2, m3, r1, r2):
            d.free()
    return out


def c1_sum_8192(ctx):
    """Config C1 (BASELINE.json configs[0]): BenchmarkFloat64Funcs_Sum_8192 — arrow/math/float64_test.go:50-75 sums makeArrayFloat64(8192)
    (buf[i] = i) with math.Float64.Sum.  CPU: the reference's own kernels re-timed here by oracle/_ref/bench_ref (oracle/bench_ref.c:
    sum_float64_x86 compiled strict-sequential = the noasm order of float64.go:41-47, and the AVX2 path's machine code), one core.
    GPU: the same 8192 rows through ah_sum_float64 (result to the host: launch + polled read), ah_sum_float64_dev (result stays on the
    device; launches back to back) and the latter replayed from a captured hipGraph.  µs per call — a launch-bound size: no GB/s, no
    speed-up claim (one CPU core finishes before a kernel launch has reached the GPU)."""
    import subprocess
    n = 8192
    out = {"rows": n, "unit": "us per call", "input": "makeArrayFloat64(8192): buf[i] = i (float64_test.go:32-48)"}
    exe = os.path.join(ROOT, "oracle", "_ref", "bench_ref")
    if os.path.exists(exe):
        r = json.loads(subprocess.run([exe, os.path.join(ROOT, "oracle", "_ref")], capture_output=True, text=True, timeout=120, check=True).stdout)
        out["cpu_noasm_order_us"] = round(r["Float64_Sum_noasm_order_8192"]["ns_per_op"] / 1e3, 3)
        out["cpu_avx2_us"] = round(r["Float64_Sum_avx2_8192"]["ns_per_op"] / 1e3, 3)
        out["cpu"] = {"cores": 1, "kind": "reference", "sample": "oracle/_ref/bench_ref: 2·10^8 / 8192 calls per kernel, the reference's sum_float64 C source (noasm order) and AVX2 machine code",
                      "noasm_MB/s": r["Float64_Sum_noasm_order_8192"]["MB/s"], "avx2_MB/s": r["Float64_Sum_avx2_8192"]["MB/s"]}
    else:
        out["cpu"] = {"error": "oracle/_ref/bench_ref not built"}
    x = np.arange(n, dtype=np.float64)
    dx = ctx.to_device(x)
    res = ctx.alloc(8)
    want = float(n * (n - 1) // 2)
    assert ctx.sum_float64(dx, n) == want, "C1: wrong sum"
    reps = 500
    for _ in range(20):
        ctx.sum_float64(dx, n)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.sum_float64(dx, n)
    out["gpu_sum_to_host_us"] = round((time.perf_counter() - t0) / reps * 1e6, 2)   # wall clock of the synchronous call, as a Go caller of math.Float64.Sum sees it

    def dev_timed(fn):
        for _ in range(20):
            fn()
        ctx.event_record(1002)
        for _ in range(reps):
            fn()
        ctx.event_record(1003)
        return round(ctx.event_elapsed_ms(1002, 1003) / reps * 1e3, 2)

    out["gpu_sum_dev_result_us"] = dev_timed(lambda: ctx.sum_float64_dev(dx, n, res))
    ctx.sync()
    ctx.graph_begin(); ctx.sum_float64_dev(dx, n, res); g = ctx.graph_end()
    out["gpu_sum_dev_result_graph_replay_us"] = dev_timed(g.launch)
    g.close()
    assert res.download(np.float64, 1)[0] == want, "C1: wrong sum from the device-result flavour"
    dx.free(); res.free()
    return out


def main():
    args = parse()
    # stdout carries exactly ONE JSON line: libraries that chat on fd 1 (RCCL's version
    # banner, rocm-smi hints) are sent to stderr for the duration of the run
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
    dist = torch = None
    # launched by torch.distributed.run (RANK set) → collective path, also at world size 1
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if use_dist:
        # torch FIRST: its bundled libamdhip64 must be the one HIP runtime in the process
        import torch
        import torch.distributed as dist
        if args.transport == "gloo":
            local_rank = local_rank % max(1, torch.cuda.device_count())   # several ranks share a device
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    gloo = use_dist and args.transport == "gloo"
    tdev = "cpu" if gloo else f"cuda:{local_rank}"   # where the launcher's own little tensors (votes, times) live
    import arrow_go_amd as ah
    N = ah._native
    comm = None
    comm_note = None
    if gloo:
        from arrow_go_amd.distributed import GlooTransport
        ctx = ah.Context(local_rank)
        gloo_transport = GlooTransport(dist)
        comm = ah.Comm.from_transport(ctx, rank, world, gloo_transport)
        comm_note = "ah_comm over the host transport (gloo) — functional run of the multi-rank code, not a measurement"
    elif use_dist:
        ctx = ah.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
        if args.collectives in ("ah", "auto"):
            # rank 0's RCCL unique id reaches the others over the launcher's process group; from here on the data-path
            # collectives are C-ABI calls on the library's own stream.
            # `--collectives ah`: no fallback — it either measures ah_comm_* or fails (a silent switch to torch's collectives would put
            # another library's numbers under this one's name).  `auto`: the ranks vote; only if the communicator could be set up on
            # NONE of them (librccl not loadable, …) the run goes on with torch's group and says so in config.collectives.
            err = None
            try:
                uid = [ah.Comm.unique_id() if rank == 0 else None]
            except Exception as e:
                uid, err = [None], e
            dist.broadcast_object_list(uid, src=0)
            if uid[0] is not None:
                try:
                    comm = ah.Comm(ctx, rank, world, uid[0])
                except Exception as e:
                    err = e
            else:
                err = err or RuntimeError("rank 0 could not create the RCCL unique id")
            ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=f"cuda:{local_rank}")
            dist.all_reduce(ok)
            if int(ok.item()) != world:
                if args.collectives == "ah" or int(ok.item()) != 0:
                    raise RuntimeError(f"bench: ah_comm set up on {int(ok.item())} of {world} ranks: {err!r}")
                comm_note = f"torch.distributed (ah_comm could not be set up on any rank: {err!r})"
                print("bench: " + comm_note, file=sys.stderr)
    else:
        ctx = ah.Context(0)

    rows = args.rows
    a = ctx.alloc(rows * 8); b = ctx.alloc(rows * 8); c = ctx.alloc(rows * 8); x = ctx.alloc(rows * 8)
    ca = fill_random(ctx, a, rows, np.int64, 10 + rank)
    cb = fill_random(ctx, b, rows, np.int64, 20 + rank)
    cx = fill_random(ctx, x, rows, np.float64, 30 + rank)
    if use_dist and not gloo:
        part = torch.zeros(1, dtype=torch.float64, device=f"cuda:{local_rank}")
        part_ptr = part.data_ptr()
    else:
        part_buf = ctx.alloc(64)
        part_ptr = part_buf.ptr

    def barrier():
        if use_dist:
            dist.barrier()
            if not gloo:
                torch.cuda.synchronize()
        ctx.sync()

    def step(i, mark):
        if mark:
            ctx.event_record(2 * i)
        ctx.arithmetic(N.INT64, N.OP_ADD, N.SHAPE_AA, a, b, c, rows)
        if mark:
            ctx.event_record(2 * i + 1)
        ctx.sum_float64_dev(x, rows, part_ptr)
        if comm is not None:
            comm.allreduce_sum(N.FLOAT64, part_ptr, part_ptr, 1)  # 8-byte RCCL all-reduce through the C ABI: the path's only exchange step
        elif use_dist:
            dist.all_reduce(part)

    for i in range(args.warmup):
        step(i, False)
    barrier()
    # correctness spot-check (not timed): first chunk of c == a + b; partial sum == tiled chunk sum
    got = c.download(np.int64, 4096)
    assert got.tobytes() == (ca[:4096] + cb[:4096]).tobytes(), "bench: Add produced wrong values"

    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt * 1e3 / max(args.steps, 1)
    add_ms = [ctx.event_elapsed_ms(2 * i, 2 * i + 1) for i in range(args.steps)]
    add_avg_ms = float(np.mean(add_ms)) if add_ms else float("nan")

    result = None
    if rank == 0:
        bytes_per_step = 32.0 * rows * args.gpus
        value = bytes_per_step / (ms_per_step * 1e-3) / 1e9
        achieved = 24.0 * rows / (add_avg_ms * 1e-3) / 1e9
        result = {
            "metric": "GB/s processed per kernel (Sum/Add/Filter/Take) vs HBM roofline",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "int64+f64", "data": "synthetic",
            "config": {"workload": "C2: Int64 Add (array+array) + Float64 Sum over contiguous Arrow value buffers resident in HBM"
                                   + (" + 8-byte RCCL all-reduce of the partial sums" + (" (ah_comm_allreduce_sum)" if comm is not None else " (torch.distributed)")
                                      if use_dist else ""),
                       "rows_per_gpu": rows, "bytes_per_row_per_step": 32,
                       "parallelism": f"record-batch shards, one per GPU (x{args.gpus}), no data-path collective",
                       "collectives": (comm_note if gloo else "ah_comm (RCCL through the C ABI)" if comm is not None else (comm_note or "torch.distributed")) if use_dist else "none (single process)"},
            "roofline": {"bound": "hbm", "kernel": "binary_kernel<uint64, ADD, array∘array> (Int64 Add)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": args.traffic if args.traffic is not None else pmc_traffic(rows)[0],
                         "traffic_source": "--traffic" if args.traffic is not None else pmc_traffic(rows)[1],
                         "avg_launch_ms": round(add_avg_ms, 5), "algorithmic_bytes_per_launch": int(24 * rows)},
        }
    # One JSON line, whatever happens after this point: with more than one rank the secondary sections contain collectives, and a
    # rank that fails or hangs inside one would leave the others waiting at a barrier.  A watchdog prints the headline without them
    # and ends every rank.
    import threading
    emit_lock, emitted, secondary_done = threading.Lock(), [False], threading.Event()

    def emit(res):
        with emit_lock:
            if not emitted[0]:
                emitted[0] = True
                os.write(real_stdout, (json.dumps(res) + "\n").encode())

    def watchdog():
        if secondary_done.wait(args.secondary_timeout):
            return
        if rank == 0:
            res = dict(result)
            res["c4_filter_aggregate"] = res["c5_group_by"] = {"error": f"not finished after {args.secondary_timeout:.0f} s (--secondary-timeout): headline printed by the watchdog"}
            emit(res)
        os._exit(0)

    if world > 1:
        threading.Thread(target=watchdog, daemon=True).start()

    # Secondary line (config C4, the 1/2/4/8-GPU curve of the record-batch-sharded filter + aggregate): every rank runs the
    # fused Compare(>) → Filter → Sum over its 2^27-row Int64 shard, the 16-byte (sum, count) partial is all-reduced.
    # Same protocol as above — barrier, K steps, barrier, MAX over ranks — reported next to the headline, never as `value`.
    c4 = None
    try:
        if use_dist and not gloo:
            pair = torch.zeros(2, dtype=torch.int64, device=f"cuda:{local_rank}")
            pair_ptr = pair.data_ptr()
        else:
            pair_buf = ctx.alloc(64)
            pair_ptr = pair_buf.ptr

        c4_last = [None]

        def c4_step():
            if comm is not None:
                c4_last[0] = comm.cmp_filter_sum_i64(N.CMP_GT, a, None, 0, rows, 0)   # ONE C-ABI call: fused kernel + 16-byte all-reduce + the result on the host
                return
            ctx.cmp_filter_sum_i64_dev(N.CMP_GT, a, None, 0, rows, 0, pair_ptr)
            if use_dist:
                dist.all_reduce(pair)

        for _ in range(max(args.warmup, 1)):
            c4_step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            c4_step()
        barrier()
        c4_dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([c4_dt], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            c4_dt = float(t.item())
        c4_ms = c4_dt * 1e3 / max(args.steps, 1)
        c4 = {"workload": "C4: fused Compare(>) -> Filter -> Sum over an Int64 record-batch shard per GPU + 16-byte all-reduce",
              "ms_per_step": round(c4_ms, 5), "GB/s": round(8.0 * rows * args.gpus / (c4_ms * 1e-3) / 1e9, 1), "rows_per_gpu": rows, "n_gpus": args.gpus}
    except Exception as e:  # informative only
        c4 = {"error": repr(e)}

    # Secondary line (config C5): hash + sum group-by, 2^26 Int64 keys (2^16 distinct) + Float64 values per GPU; under
    # torch.distributed the local aggregates are merged by key-hash owner (all-to-all of O(groups) tuples, distributed.py).
    c5 = None
    try:
        hrows = min(rows, 1 << 26)
        krng = np.random.default_rng(77 + rank)
        # Keys: 2^26 INDEPENDENT draws over 2^16 values, like the kernel table's rows.  Rounds 1-4 tiled ONE 2^22-row block sixteen times:
        # with every eighth of the column holding the same rows the eight XCDs run the cut in lockstep on identical data (same run lengths,
        # destinations a constant stride apart) and a sampled key is met by sixteen workgroups at once — an artefact of the generator that
        # costs the partitioned path ≈ 0.15 ms (scatter 466 -> 563 us, sample 41 -> 107 us in profiles/r05_bench_kernel_stats.csv).  The
        # tiled column's time is kept beside the line as `ms_per_step_tiled_block_keys` for continuity with the earlier rounds' numbers.
        kchunk = (krng.integers(0, 1 << 16, 1 << 22, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
        for off in range(0, hrows, 1 << 22):
            c.upload(kchunk[:min(1 << 22, hrows - off)], off * 8)
        tiled_ms = None
        # the owner merge is ONE C-ABI call (ah_comm_merge_groups); it runs whenever the ah_comm communicator exists and world > 1
        merge = comm is not None and world > 1 and not args.no_c5_merge
        obufs = [ctx.alloc((hrows + 1) * 8 + 64) for _ in range(4)]
        optr = [b_.ptr for b_ in obufs]
        mcap = (1 << 16) * 2 + 64
        mbufs = [ctx.alloc(mcap * 8 + 64) for _ in range(4)] if merge else []
        ngroups = [0]

        def c5_step():
            ng, _ = ctx.hash_sum("f64", c, None, 0, x, None, 0, hrows, optr[0], optr[1], optr[2], optr[3])
            if merge:
                ng = comm.merge_groups(True, optr[0], optr[1], optr[2], optr[3], ng, rank * hrows, mcap, *mbufs)
            ngroups[0] = int(ng)

        c5_steps = max(1, min(args.steps, 5))
        if world == 1:   # the tiled column of rounds 1-4 first (continuity), then the independent draws the line is quoted on
            c5_step()
            barrier()
            t0 = time.perf_counter()
            for _ in range(c5_steps):
                c5_step()
            barrier()
            tiled_ms = round((time.perf_counter() - t0) * 1e3 / c5_steps, 4)
        for off in range(0, hrows, 1 << 22):
            kc = (krng.integers(0, 1 << 16, min(1 << 22, hrows - off), dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
            c.upload(kc, off * 8)
        c5_step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(c5_steps):
            c5_step()
        barrier()
        c5_dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([c5_dt], dtype=torch.float64, device=tdev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            c5_dt = float(t.item())
        c5_ms = c5_dt * 1e3 / c5_steps
        if args.dump and rank == 0:
            gcount = ngroups[0]
            src = mbufs if merge else obufs
            np.savez(args.dump, c4=np.array(c4_last[0] if c4_last[0] is not None else (0, 0), np.int64), merged=np.array([1 if merge else 0]),
                     keys=src[0].download(np.uint64, gcount), sums=src[1].download(np.float64, gcount), counts=src[2].download(np.int64, gcount),
                     first_rows=src[3].download(np.int64, gcount))
        c5 = {"workload": "C5: hash + sum group-by over Int64 keys / Float64 values per GPU" + (", merged by key-hash owner (ah_comm_merge_groups: all-to-all of group tuples)" if merge else
                                                                                                        " (local aggregate; the owner merge over RCCL runs when world > 1 with --collectives ah)"),
              "ms_per_step": round(c5_ms, 4), "Grows/s": round(hrows * args.gpus / (c5_ms * 1e-3) / 1e9, 2), "rows_per_gpu": hrows,
              "groups": ngroups[0], "n_gpus": args.gpus, "steps": c5_steps, "keys": "2^26 independent draws over 2^16 values per GPU",
              "ms_per_step_tiled_block_keys": tiled_ms}
    except Exception as e:  # informative only
        c5 = {"error": repr(e)}

    if rank == 0:
        result["c4_filter_aggregate"] = c4
        result["c5_group_by"] = c5
        if world == 1 and not args.no_kernels:
            try:
                result["kernels"] = per_kernel_table(ctx, rows, a, b, c, x)
            except Exception as e:  # the table is informative; never lose the headline over it
                result["kernels"] = {"error": repr(e)}
            try:
                result["host_ingest"] = host_ingest_legs(ctx, rows)
            except Exception as e:
                result["host_ingest"] = {"error": repr(e)}
            try:
                result["graph_replay"] = graph_replay_legs(ctx)
            except Exception as e:
                result["graph_replay"] = {"error": repr(e)}
            try:
                result["c1_sum_8192"] = c1_sum_8192(ctx)
            except Exception as e:
                result["c1_sum_8192"] = {"error": repr(e)}
            if isinstance(result["kernels"].get("ceiling_copy_kernel"), dict):
                result["roofline"]["measured_copy_GB/s"] = result["kernels"]["ceiling_copy_kernel"]["GB/s"]
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline(min(rows, 1 << 26), 8)
            except Exception as e:
                result["cpu_baseline"] = {"error": repr(e)}
            try:
                result["cpu_baseline_mt"] = cpu_baseline_mt()
            except Exception as e:
                result["cpu_baseline_mt"] = {"error": repr(e)}
            try:
                result["cpu_baseline_c3"], result["cpu_baseline_c5"] = cpu_baseline_c3_c5()
            except Exception as e:
                result["cpu_baseline_c3"] = result["cpu_baseline_c5"] = {"error": repr(e)}
        emit(result)
    secondary_done.set()
    if comm is not None:
        comm.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
