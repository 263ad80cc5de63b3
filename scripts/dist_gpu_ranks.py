#!/usr/bin/env python3
"""Configs C4 / C5 through the C ABI with SEVERAL ranks on one GPU: every rank is a process with its own ah_ctx on device 0, the
communicator is the host-transport flavour (ah_comm_init_transport) carried by a gloo group — RCCL refuses two ranks on one
device, and a 1-GPU box has no second one.  Launched by tests/test_distributed_gpu.py under torch.distributed.run; every rank
checks the global results of ah_comm_cmp_filter_sum_{i64,f64} and ah_comm_merge_groups against the CPU oracle over the
UNDIVIDED data, byte for byte, and against each other through the protocol model of tests/dist_model.py."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()

import arrow_go_amd as ah  # noqa: E402
from arrow_go_amd.distributed import GlooTransport, ShardedGpu, shard_bounds  # noqa: E402
from tests import oracle_lib as OL  # noqa: E402

o = OL.load_oracle()
ctx = ah.Context(0)
tr = GlooTransport(dist)
comm = ah.Comm.from_transport(ctx, rank, world, tr)
sg = ShardedGpu(comm)
rng = np.random.default_rng(4321)   # the same "table" on every rank; each takes its shard
n = 400_003
x = rng.integers(-10**9, 10**9, n, dtype=np.int64)
valid_bits = rng.random(n) < 0.9
xf = rng.standard_normal(n)
lo, hi = shard_bounds(n, rank, world)
vall = np.packbits(valid_bits, bitorder="little")
dx = ctx.to_device(x[lo:hi]); dxf = ctx.to_device(xf[lo:hi]); dv = ctx.to_device(np.packbits(valid_bits[lo:hi], bitorder="little"), pad=64)
# ---- C4
for thr in (-5 * 10**8, 0, 7 * 10**8):
    assert sg.cmp_filter_sum(2, dx, dv, 0, hi - lo, thr, np.int64) == o.cmp_filter_sum_i64(2, x, vall, 0, thr), thr
gotf = sg.cmp_filter_sum(2, dxf, dv, 0, hi - lo, 0.25, np.float64)
# every rank's UN-ROUNDED double-double accumulator is gathered and merged in rank order, rounded once: within 1 ULP of the exact
# sum over the undivided column whatever the world size (orc_sum_float64_xreal: the fixed-point superaccumulator)
kept = xf[(xf > 0.25) & valid_bits]
exact = float(o.sum_float64_xreal(kept))
assert gotf[1] == kept.size
assert abs(gotf[0] - exact) <= np.spacing(abs(exact)), (gotf[0], exact)
allf = [None] * world
dist.all_gather_object(allf, float(gotf[0]).hex())
assert len(set(allf)) == 1, allf                                                    # the same bytes on every rank
# non-finite rows and overflow follow the extended reals however the rows fall over the ranks (arrow/math/float64.go:41-47 for the
# cases in which the reference's orders agree): the special row in the first / last rank's shard, in both, finite overflow across ranks
inf, nan = np.inf, np.nan
for label, edits in (("+inf on the first rank", {3: inf}), ("+inf on the last rank", {n - 2: inf}), ("+inf and -inf on different ranks", {5: inf, n - 7: -inf}),
                     ("-inf only", {n // 2: -inf}), ("NaN is dropped by the compare", {11: nan, n - 11: nan}),
                     ("finite overflow across ranks", {1: 1e308, n // 2 + 1: 1e308, n - 3: 1e308}),
                     ("intermediate overflow only", {1: 1e308, 2: 1e308, n - 3: -1e308, n - 4: -1e308})):
    y = xf.copy()
    vb = valid_bits.copy()
    for i, v in edits.items():
        y[i] = v
        vb[i] = True
    dy = ctx.to_device(y[lo:hi]); dvb = ctx.to_device(np.packbits(vb[lo:hi], bitorder="little"), pad=64)
    thr = -np.inf
    got = sg.cmp_filter_sum(3, dy, dvb, 0, hi - lo, thr, np.float64)     # x >= -inf: everything but NaN
    keep = y[(y >= thr) & vb]
    want = float(o.sum_float64_xreal(keep))
    assert got[1] == keep.size, label
    assert (np.isnan(want) and np.isnan(got[0])) or got[0] == want or (np.isfinite(want) and abs(got[0] - want) <= np.spacing(abs(want))), (label, got[0], want)
    assert np.isfinite(got[0]) == np.isfinite(want) and np.isnan(got[0]) == np.isnan(want), (label, got[0], want)


def local_aggregate(kind, keys, vals):
    m = keys.size
    dk = ctx.to_device(keys); dvv = ctx.to_device(vals)
    outs = [ctx.alloc((m + 1) * 8 + 64) for _ in range(4)]
    ng, _ = ctx.hash_sum(kind, dk, None, 0, dvv, None, 0, m, outs[0], outs[1], outs[2], outs[3]) if m else (0, -1)
    return ng, outs


def merged(kind, keys, vals, row_offset, capacity):
    ng, outs = local_aggregate(kind, keys, vals)
    res = [ctx.alloc(capacity * 8 + 64) for _ in range(4)]
    G = sg.merge_groups(kind == "f64", outs[0], outs[1], outs[2], outs[3], ng, row_offset, capacity, *res)
    sdt = np.float64 if kind == "f64" else np.int64
    return G, res[0].download(np.uint64, G), res[1].download(sdt, G), res[2].download(np.int64, G), res[3].download(np.int64, G)


# ---- C5: few groups, many groups, every key its own group
for card in (777, 50_000, 10**9):
    keys = rng.integers(0, card, n).astype(np.int64) * 1000003
    vals = rng.integers(-2**40, 2**40, n, dtype=np.int64)
    G, mk, ms, mc, mf = merged("i64", keys[lo:hi], vals[lo:hi], lo, n + 1)
    ek, es, ec, _nid, ef = o.hash_sum("i64", keys, None, 0, vals, None, 0)
    assert G == ek.size and mk.tobytes() == ek.tobytes() and ms.tobytes() == es.tobytes() and mc.tobytes() == ec.tobytes() and mf.tobytes() == ef.tobytes(), card
    fv = rng.integers(-1000, 1000, n).astype(np.float64)          # integer-valued: exact in any merge order
    G, mk, ms, mc, mf = merged("f64", keys[lo:hi], fv[lo:hi], lo, n + 1)
    ek, es, ec, _nid, ef = o.hash_sum("f64", keys, None, 0, fv, None, 0)
    assert G == ek.size and mk.tobytes() == ek.tobytes() and ms.tobytes() == es.tobytes() and mc.tobytes() == ec.tobytes() and mf.tobytes() == ef.tobytes(), card
# general doubles: every rank the same bytes, each group within 2 ULP·world of the exact sum
keys = rng.integers(0, 5000, n).astype(np.int64) * 1000003
gv = rng.standard_normal(n)
G, mk, ms, mc, mf = merged("f64", keys[lo:hi], gv[lo:hi], lo, n + 1)
ek, es, ec, _nid, ef = o.hash_sum("f64", keys, None, 0, gv, None, 0)
assert mk.tobytes() == ek.tobytes() and mc.tobytes() == ec.tobytes() and mf.tobytes() == ef.tobytes()
assert np.all(np.abs(ms - es) <= 1e-9 * np.maximum(1.0, np.abs(es)))
digests = [None] * world
dist.all_gather_object(digests, ms.tobytes().hex()[:64] + str(hash(ms.tobytes())))
assert len({d[:64] for d in digests}) == 1
# null keys AND the key 0 in one column (the local aggregates hold the null group with key slot 0): two groups, as over the undivided
# column; the null group on every rank, on one rank only, first seen before / after the key 0, values Int64 and general Float64
def merged_with_nulls(kind, keys, kvalid_bits, vals, lo_, hi_, capacity):
    m = hi_ - lo_
    dk = ctx.to_device(keys[lo_:hi_]); dvv = ctx.to_device(vals[lo_:hi_])
    dkv = ctx.to_device(np.packbits(kvalid_bits[lo_:hi_], bitorder="little"), pad=64)
    outs = [ctx.alloc((m + 1) * 8 + 64) for _ in range(4)]
    ng, nid = ctx.hash_sum(kind, dk, dkv, 0, dvv, None, 0, m, outs[0], outs[1], outs[2], outs[3])
    res = [ctx.alloc(capacity * 8 + 64) for _ in range(4)]
    G, gnull = sg.merge_groups(kind == "f64", outs[0], outs[1], outs[2], outs[3], ng, lo_, capacity, *res, null_group_local=nid, with_null_group=True)
    sdt = np.float64 if kind == "f64" else np.int64
    return G, gnull, res[0].download(np.uint64, G), res[1].download(sdt, G), res[2].download(np.int64, G), res[3].download(np.int64, G)


for variant in ("nulls everywhere", "nulls on the last rank only", "null first, key 0 later", "only nulls"):
    kz = rng.integers(0, 50, n).astype(np.int64)             # 0 is one of the keys
    kvb = np.ones(n, bool)
    if variant == "nulls everywhere":
        kvb = rng.random(n) < 0.8
    elif variant == "nulls on the last rank only":
        kvb[n - 100:] = rng.random(100) < 0.5
    elif variant == "null first, key 0 later":
        kvb[0] = False; kz[:1000] = 7; kz[5000] = 0
    else:
        kvb[:] = False
    kpack = np.packbits(kvb, bitorder="little")
    for kind, vals in (("i64", rng.integers(-2**40, 2**40, n, dtype=np.int64)), ("f64", rng.integers(-1000, 1000, n).astype(np.float64))):
        G, gnull, mk, ms, mc, mf = merged_with_nulls(kind, kz, kvb, vals, lo, hi, n + 1)
        ek, es, ec, enull, ef = o.hash_sum(kind, kz, kpack, 0, vals, None, 0)
        assert G == ek.size and gnull == enull, (variant, kind, G, ek.size, gnull, enull)
        assert mk.tobytes() == ek.tobytes() and ms.tobytes() == es.tobytes() and mc.tobytes() == ec.tobytes() and mf.tobytes() == ef.tobytes(), (variant, kind)
    # general doubles: the null group's sum must be the same bytes on every rank and within the group-by tolerance of the oracle's
    gvals = rng.standard_normal(n)
    G, gnull, mk, ms, mc, mf = merged_with_nulls("f64", kz, kvb, gvals, lo, hi, n + 1)
    ek, es, ec, enull, ef = o.hash_sum("f64", kz, kpack, 0, gvals, None, 0)
    assert gnull == enull and mk.tobytes() == ek.tobytes() and mc.tobytes() == ec.tobytes() and mf.tobytes() == ef.tobytes(), variant
    assert np.all(np.abs(ms - es) <= 1e-9 * np.maximum(1.0, np.abs(es))), variant
    digs = [None] * world
    dist.all_gather_object(digs, ms.tobytes().hex())
    assert len(set(digs)) == 1, variant
# ragged: rank 0 contributes no group at all; too small a capacity is an error that names the count
l0, h0 = shard_bounds(n, 0, world)
k2, v2 = keys[h0:], rng.integers(-2**40, 2**40, n, dtype=np.int64)[h0:]
if rank == 0:
    mine_k, mine_v = np.zeros(0, np.int64), np.zeros(0, np.int64)
else:
    mine_k, mine_v = keys[lo:hi], v2[lo - h0:hi - h0]
G, mk, ms, mc, mf = merged("i64", mine_k, mine_v, max(lo - h0, 0), n + 1)
ek, es, ec, _nid, ef = o.hash_sum("i64", k2, None, 0, v2, None, 0)
assert mk.tobytes() == ek.tobytes() and ms.tobytes() == es.tobytes() and mc.tobytes() == ec.tobytes() and mf.tobytes() == ef.tobytes()
try:
    merged("i64", mine_k, mine_v, max(lo - h0, 0), 10)
    raise SystemExit("a capacity of 10 groups was accepted")
except ah.ErrInvalid as e:
    assert str(ek.size) in str(e), str(e)
# ONE rank gets an argument wrong (a null group beyond its group count): its status travels in the first all-gather and EVERY rank
# returns an error — nobody is left waiting in a collective for a rank that went home (the advisor's round-4 finding)
dk = ctx.to_device(keys[lo:hi]); dv = ctx.to_device(np.ones(hi - lo, np.int64))
outs = [ctx.alloc((hi - lo + 1) * 8 + 64) for _ in range(4)]
ng_ok, _ = ctx.hash_sum("i64", dk, None, 0, dv, None, 0, hi - lo, outs[0], outs[1], outs[2], outs[3])
res = [ctx.alloc((n + 1) * 8 + 64) for _ in range(4)]
bad_null = ng_ok + 5 if rank == world - 1 else -1
try:
    sg.merge_groups(False, outs[0], outs[1], outs[2], outs[3], ng_ok, lo, n + 1, *res, null_group_local=bad_null, with_null_group=True)
    raise SystemExit(f"rank {rank}: a call in which rank {world - 1} passed a null group beyond its groups succeeded")
except ah.ErrInvalid as e:
    assert ("null group" in str(e)) if rank == world - 1 else (f"rank {world - 1} failed" in str(e) or world == 1), str(e)
# … and the communicator is still usable afterwards
G, mk, ms, mc, mf = merged("i64", mine_k, mine_v, max(lo - h0, 0), n + 1)
assert mk.tobytes() == ek.tobytes()
assert not tr.errors, tr.errors
comm.close()
dist.barrier()
if rank == 0:
    print(f"dist_gpu_ranks ok: world {world} on one GPU, C4 and C5 through ah_comm_* == oracle over the undivided data")
dist.destroy_process_group()
