"""world_size-2 / 3 gloo tests (CPU) of the record-batch sharding protocol: shard bounds and the owner function
(arrow_go_amd/distributed.py), and — through the executable model of tests/dist_model.py, with a stand-in backed by the CPU
oracle for the per-shard compute — the 16-byte all-reduce of the fused Compare→Filter→Sum partials, the rank-ordered float64
combine, and the key-hash-owner all-to-all merge of the hash group-by, incl. the byte-level block packing of the ah_comm_*
interface.  The product's implementation of the same protocol is C (csrc/ah_comm.hip: ah_comm_cmp_filter_sum_*,
ah_comm_merge_groups) and needs a GPU: tests/test_distributed_gpu.py runs it with 2 and 3 ranks on one device and compares
it with the oracle — and this model is the second opinion on what the bytes must be.
"""
import ctypes
import os
import socket

import numpy as np
import pytest

from arrow_go_amd.distributed import shard_bounds, owner_of, hash_int

GT = 2


def test_shard_bounds_cover_and_balance():
    for n in [0, 1, 7, 8, 1000, 2**27 + 5]:
        for world in [1, 2, 3, 8]:
            spans = [shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_owner_uses_reference_hash(orc):
    keys = np.array([0, 1, 2, 42, 2**63, 2**64 - 1], dtype=np.uint64)
    h = hash_int(keys)
    for k, hv in zip(keys.tolist(), h.tolist()):
        assert hv == orc.hash_int(k)          # internal/hashing/hash_funcs.go:60-67
    own = owner_of(np.arange(100000, dtype=np.uint64), 8)
    assert set(own.tolist()) == set(range(8))
    assert np.bincount(own, minlength=8).min() > 100000 / 8 * 0.9   # balanced partition


class OracleLocal:
    """stand-in for HipLocal: x/valid are host numpy arrays, results are written through the
    raw pointers of the (CPU) torch tensors exactly like the GPU kernels write device memory"""

    def __init__(self):
        from tests import oracle_lib as OL
        self.o = OL.load_oracle()

    def cmp_filter_sum_partial(self, cmpop, x, valid, off, n, thr, dtype, out_sum_ptr, out_count_ptr):
        if np.dtype(dtype) == np.int64:
            s, c = self.o.cmp_filter_sum_i64(cmpop, x[:n], valid, off, thr)
            buf = np.array([s, c], np.int64)
            ctypes.memmove(out_sum_ptr, buf.ctypes.data, 16)
        else:
            # the shard's un-rounded accumulator {s, e, bs, be}: the Python restatement of csrc/ah_ddsum.h over the kept rows
            from tests import ddx_model as DD
            from tests.oracle_lib import unpack_bits
            xs = np.asarray(x[:n], np.float64)
            ok = unpack_bits(valid, off, n).astype(bool) if valid is not None else np.ones(n, bool)
            keep = xs[ok & {0: xs == thr, 1: xs != thr, 2: xs > thr, 3: xs >= thr}[cmpop]]
            acc = DD.accumulate(keep.tolist(), lanes=64)
            ctypes.memmove(out_sum_ptr, np.array(acc, np.float64).ctypes.data, 32)
            ctypes.memmove(out_count_ptr, np.array([keep.size], np.int64).ctypes.data, 8)


    # ---- the three merge steps, numpy on CPU tensors (HipLocal runs them on the GPU) ----
    def partition_by_owner(self, torch, cols, world):
        from arrow_go_amd.distributed import owner_of
        c = cols.numpy()
        own = owner_of(c[0].view(np.uint64), world)
        return [torch.from_numpy(np.ascontiguousarray(c[:, own == r])) for r in range(world)]

    def merge_tuples(self, torch, got, is_float):
        g = got.numpy()
        if g.shape[1] == 0:
            return got
        k = g[0].view(np.uint64)
        order = np.lexsort((np.arange(k.size), k))            # by key, source order inside a key
        ks = k[order]
        starts = np.flatnonzero(np.concatenate([[True], ks[1:] != ks[:-1]]))
        first_pos = order[starts]                                # position of each key's first tuple
        mc = np.add.reduceat(g[2][order], starts)
        if is_float:
            ms = np.add.reduceat(g[1].view(np.float64)[order], starts).view(np.int64)
        else:
            with np.errstate(over="ignore"):
                ms = np.add.reduceat(g[1].view(np.uint64)[order], starts).view(np.int64)
        seen = np.argsort(first_pos, kind="stable")              # first-seen order, like the device hash table
        out = np.stack([ks[starts].view(np.int64)[seen], ms[seen], mc[seen], g[3][first_pos][seen]])
        return torch.from_numpy(np.ascontiguousarray(out))

    def order_by_first(self, torch, rows):
        r = rows.numpy()
        return torch.from_numpy(np.ascontiguousarray(r[:, np.argsort(r[3], kind="stable")]))


class GlooByteComm:
    """stand-in for arrow_go_amd.Comm with the SAME byte-level interface as ah_comm_* (pointers, byte counts, byte offsets),
    carried by gloo point-to-point messages between CPU buffers: what runs under test is AhCommCollectives' packing —
    tuple-major blocks, size table, offsets — for world > 1, which a 1-GPU box cannot exercise over RCCL."""

    def __init__(self, dist, torch):
        self.dist, self.torch = dist, torch
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.ctx = self

    def sync(self):
        pass

    def _view(self, ptr, nbytes):
        return self.torch.frombuffer((ctypes.c_uint8 * max(nbytes, 1)).from_address(ptr if nbytes else ctypes.addressof(ctypes.c_uint8())), dtype=self.torch.uint8)[:nbytes]

    def allreduce_sum(self, type_id, send, recv, count):
        import arrow_go_amd as ah
        dt = {ah._native.INT64: self.torch.int64, ah._native.FLOAT64: self.torch.float64}[type_id]
        t = self._view(recv, count * 8).view(dt)
        if send != recv:
            t.copy_(self._view(send, count * 8).view(dt))
        self.dist.all_reduce(t)

    def allgather(self, send, recv, nbytes):
        parts = [self.torch.zeros(nbytes, dtype=self.torch.uint8) for _ in range(self.world)]
        self.dist.all_gather(parts, self._view(send, nbytes).clone())
        self._view(recv, nbytes * self.world).copy_(self.torch.cat(parts))

    def alltoallv(self, send, send_bytes, send_offs, recv, recv_bytes, recv_offs):
        reqs, keep = [], []
        for r in range(self.world):
            if r == self.rank:
                assert send_bytes[r] == recv_bytes[r]
                if send_bytes[r]:
                    self._view(recv + recv_offs[r], recv_bytes[r]).copy_(self._view(send + send_offs[r], send_bytes[r]).clone())
                continue
            if send_bytes[r]:
                t = self._view(send + send_offs[r], send_bytes[r]).clone(); keep.append(t)
                reqs.append(self.dist.isend(t, r))
            if recv_bytes[r]:
                reqs.append(self.dist.irecv(self._view(recv + recv_offs[r], recv_bytes[r]), r))
        for q in reqs:
            q.wait()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from tests import oracle_lib as OL
    from tests.dist_model import ShardedCompute
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = OL.load_oracle()
        sc = ShardedCompute(dist, torch.device("cpu"), OracleLocal())
        rng = np.random.default_rng(1234)  # identical "table" on every rank; each takes its shard
        n = 100003
        x = rng.integers(-10**9, 10**9, n, dtype=np.int64)
        valid_bits = rng.random(n) < 0.9
        xf = rng.uniform(-1, 1, n)
        lo, hi = shard_bounds(n, rank, world)
        vshard = np.packbits(valid_bits[lo:hi], bitorder="little")
        # C4 int64: exact, must equal the unsharded oracle
        got = sc.cmp_filter_sum(torch, GT, x[lo:hi], vshard, 0, hi - lo, 0, np.int64)
        exp = o.cmp_filter_sum_i64(GT, x, np.packbits(valid_bits, bitorder="little"), 0, 0)
        assert got == exp, (got, exp)
        # C4 float64: rank-ordered sum of the per-shard exact partials, identical on every rank
        gotf = sc.cmp_filter_sum(torch, GT, xf[lo:hi], vshard, 0, hi - lo, 0.25, np.float64)
        parts = []
        for r in range(world):
            l2, h2 = shard_bounds(n, r, world)
            _, s_exact, _c = o.cmp_filter_sum_f64(GT, xf[l2:h2], np.packbits(valid_bits[l2:h2], bitorder="little"), 0, 0.25)
            parts.append(s_exact)
        # one rounding for the whole column whatever the world size: within 1 ULP of the exact sum (the superaccumulator), and — the
        # protocol's point — NOT the rank-ordered sum of rounded partials when those differ
        kept = xf[(xf > 0.25) & valid_bits]
        exact = float(o.sum_float64_xreal(kept))
        assert abs(gotf[0] - exact) <= np.spacing(abs(exact)) and gotf[1] == kept.size, (gotf, exact, parts)
        # ±inf / overflow across ranks follow the extended reals (tests/test_ddsum_host.py has the single-process cases)
        for edits, want in (({3: np.inf}, np.inf), ({3: np.inf, n - 2: -np.inf}, np.nan), ({1: 1e308, n // 2 + 1: 1e308, n - 3: 1e308}, np.inf),
                            ({1: 1e308, 2: 1e308, n - 3: -1e308, n - 4: -1e308}, None)):
            y = xf.copy()
            for i, v in edits.items():
                y[i] = v
            ones = np.full((n + 7) // 8, 0xFF, np.uint8)
            l2, h2 = shard_bounds(n, rank, world)
            gy = sc.cmp_filter_sum(torch, 3, y[l2:h2], None, 0, h2 - l2, -np.inf, np.float64)      # x >= -inf
            w = float(o.sum_float64_xreal(y)) if want is None else want
            assert (np.isnan(w) and np.isnan(gy[0])) or gy[0] == w or (np.isfinite(w) and abs(gy[0] - w) <= np.spacing(abs(w))), (edits, gy, w)
            del ones
        # C5: local aggregate per shard → owner all-to-all → merge → global first-seen order
        keys = rng.integers(0, 777, n).astype(np.int64) * 1000003
        vals = rng.integers(-2**40, 2**40, n, dtype=np.int64)
        lk, ls, lc, _nid, lf = o.hash_sum("i64", keys[lo:hi], None, 0, vals[lo:hi], None, 0)
        mk, ms, mc, mf = sc.merge_groups(torch, lk, ls, lc, lf, lo)
        ek, es, ec, _nid, ef = o.hash_sum("i64", keys, None, 0, vals, None, 0)
        assert mk.tobytes() == ek.tobytes() and ms.tobytes() == es.tobytes()
        assert mc.tobytes() == ec.tobytes() and mf.tobytes() == ef.tobytes()
        # float flavour: integer-valued data → exact in any merge order
        fv = rng.integers(-1000, 1000, n).astype(np.float64)
        lk, ls, lc, _nid, lf = o.hash_sum("f64", keys[lo:hi], None, 0, fv[lo:hi], None, 0)
        mk, ms, mc, mf = sc.merge_groups(torch, lk, ls, lc, lf, lo)
        ek, es, ec, _nid, ef = o.hash_sum("f64", keys, None, 0, fv, None, 0)
        assert mk.tobytes() == ek.tobytes() and ms.tobytes() == es.tobytes() and mc.tobytes() == ec.tobytes()
        # null keys next to the key 0: the null group is merged outside the owner exchange and lands at its first-seen position
        kz = rng.integers(0, 40, n).astype(np.int64)
        for variant in range(3):
            kvb = rng.random(n) < 0.8 if variant == 0 else np.ones(n, bool)
            if variant == 1:
                kvb[n - 50:] = False                           # nulls on the last rank only
            if variant == 2:
                kvb[0] = False; kz[:500] = 7; kz[900] = 0       # null first, key 0 later
            kpack_all = np.packbits(kvb, bitorder="little")
            lk, ls, lc, lnull, lf = o.hash_sum("i64", kz[lo:hi], np.packbits(kvb[lo:hi], bitorder="little"), 0, vals[lo:hi], None, 0)
            mk, ms, mc, mf, mnull = sc.merge_groups(torch, lk, ls, lc, lf, lo, null_group_local=lnull, with_null_group=True)
            ek, es, ec, enull, ef = o.hash_sum("i64", kz, kpack_all, 0, vals, None, 0)
            assert mnull == enull and mk.tobytes() == ek.tobytes() and ms.tobytes() == es.tobytes() and mc.tobytes() == ec.tobytes() and mf.tobytes() == ef.tobytes(), variant
        # the same steps through the C-ABI-shaped provider (AhCommCollectives over a byte-level comm): block packing and
        # offsets for world > 1; ragged on purpose (rank 0 contributes no groups at all in the second round)
        from tests.dist_model import AhCommCollectives
        sc2 = ShardedCompute(AhCommCollectives(GlooByteComm(dist, torch), torch.device("cpu")), torch.device("cpu"), OracleLocal())
        assert sc2.cmp_filter_sum(torch, GT, x[lo:hi], vshard, 0, hi - lo, 0, np.int64) == exp
        assert sc2.cmp_filter_sum(torch, GT, xf[lo:hi], vshard, 0, hi - lo, 0.25, np.float64) == gotf
        lk, ls, lc, _nid, lf = o.hash_sum("i64", keys[lo:hi], None, 0, vals[lo:hi], None, 0)
        mk, ms, mc, mf = sc2.merge_groups(torch, lk, ls, lc, lf, lo)
        ek, es, ec, _nid, ef = o.hash_sum("i64", keys, None, 0, vals, None, 0)
        assert mk.tobytes() == ek.tobytes() and ms.tobytes() == es.tobytes() and mc.tobytes() == ec.tobytes() and mf.tobytes() == ef.tobytes()
        l0, h0 = shard_bounds(n, 0, world)
        k2, v2 = keys[h0:], vals[h0:]                     # rows of rank 0 removed: it has nothing to send
        if rank == 0:
            lk, ls, lc, lf = np.zeros(0, np.uint64), np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.int64)
        else:
            lk, ls, lc, _nid, lf = o.hash_sum("i64", keys[lo:hi], None, 0, vals[lo:hi], None, 0)
        mk, ms, mc, mf = sc2.merge_groups(torch, lk, ls, lc, lf, lo - h0)
        ek, es, ec, _nid, ef = o.hash_sum("i64", k2, None, 0, v2, None, 0)
        assert mk.tobytes() == ek.tobytes() and ms.tobytes() == es.tobytes() and mc.tobytes() == ec.tobytes() and mf.tobytes() == ef.tobytes()
        q.put((rank, "ok"))
    except Exception as e:  # surface the failure in the parent
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_reductions_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in results:
        assert msg == "ok", f"rank {rank}: {msg}"
