"""Record-batch sharding across the GPUs of one node: the host-side driver.

The hot path shards at row-range granularity: Add, Compare, bitmap ops, Filter, Take-by-local-index and Sum partials are
independent per shard — NO data-path collective.  Exactly one exchange step exists, for reductions, and both of its forms are
single calls of the C ABI (include/arrowhip.h, csrc/ah_comm.hip), so a Go host runs them without Python:

  * C4  Compare → Filter → Sum: `ah_comm_cmp_filter_sum_{i64,f64}` — the fused single-pass kernel on the rank's own shard, then one
    16-byte all-reduce (int64: exact in any order) or an all-gather of the partials added in rank order (float64: the same
    bytes on every rank and in every run).
  * C5  hash group-by: `ah_comm_merge_groups` — the rank's local aggregate (ah_hash_sum_*) → groups bucketed by key-hash owner
    on the device → ragged all-to-all of {key, sum, count, first row} tuples (O(groups) bytes, never O(rows)) → the owner
    re-aggregates → ragged all-gather → ordered by global first occurrence.

One process per GPU.  The bytes travel over RCCL / xGMI (`Comm(ctx, rank, world, unique_id)`) or over a transport the host
supplies (`Comm.from_transport(ctx, rank, world, GlooTransport(dist))` — the launcher's own sockets; what the tests use to run 2
and 3 ranks on ONE GPU, which RCCL refuses).  What is left here is the shard arithmetic and thin wrappers; an executable
model of the protocol for machines without a GPU lives in tests/dist_model.py.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

PRIME64_1 = 11400714785074694791  # internal/hashing/hash_funcs.go:60-67


def shard_bounds(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced row range of `rank` (record-batch sharding): the first
    n % world ranks get one extra row."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def hash_int(keys: np.ndarray) -> np.ndarray:
    """The reference's hashInt (bswap64(PRIME·v)) — used only to pick a key's OWNER rank."""
    k = keys.astype(np.uint64, copy=False)
    with np.errstate(over="ignore"):
        prod = k * np.uint64(PRIME64_1)
    return prod.byteswap()


def owner_of(keys: np.ndarray, world: int) -> np.ndarray:
    """owner = top bits of hashInt(key), folded to the world size (ah_hash_partition_u64 computes the same on the device)."""
    return ((hash_int(keys) >> np.uint64(40)) % np.uint64(world)).astype(np.int64)


class GlooTransport:
    """ah_transport callbacks over a torch.distributed process group of CPU tensors (gloo): the host-supplied transport of
    ah_comm_init_transport.  Keep the object alive as long as the communicator."""

    def __init__(self, dist):
        import torch
        from . import _native as N
        self.dist, self.torch = dist, torch
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.errors = []

        def view(ptr, nbytes):
            if nbytes <= 0:
                return torch.zeros(0, dtype=torch.uint8)
            return torch.frombuffer((C.c_uint8 * nbytes).from_address(ptr), dtype=torch.uint8)

        def allgather(_user, send, recv, nbytes):
            try:
                parts = [torch.zeros(nbytes, dtype=torch.uint8) for _ in range(self.world)]
                dist.all_gather(parts, view(send, nbytes).clone())
                view(recv, nbytes * self.world).copy_(torch.cat(parts))
                return 0
            except Exception as e:  # a C caller sees a status, the test sees the text
                self.errors.append(repr(e))
                return 1

        def alltoallv(_user, send, sbytes, soffs, recv, rbytes, roffs):
            try:
                reqs, keep = [], []
                for r in range(self.world):
                    sb, rb = int(sbytes[r]), int(rbytes[r])
                    if r == self.rank:
                        if sb:
                            view(recv + int(roffs[r]), rb).copy_(view(send + int(soffs[r]), sb).clone())
                        continue
                    if sb:
                        t = view(send + int(soffs[r]), sb).clone(); keep.append(t)
                        reqs.append(dist.isend(t, r))
                    if rb:
                        reqs.append(dist.irecv(view(recv + int(roffs[r]), rb), r))
                for q in reqs:
                    q.wait()
                return 0
            except Exception as e:
                self.errors.append(repr(e))
                return 1

        self._ag = N.TRANSPORT_ALLGATHER(allgather)
        self._a2a = N.TRANSPORT_ALLTOALLV(alltoallv)
        self.struct = N.AhTransport(None, self._ag, self._a2a)


class ShardedGpu:
    """This rank's shard of configs C4 / C5: every method is ONE call of the C ABI on `comm` (arrow_go_amd.Comm)."""

    def __init__(self, comm):
        self.comm = comm
        self.rank, self.world = comm.rank, comm.world

    def cmp_filter_sum(self, cmpop: int, x, valid, off: int, n_local: int, thr, dtype):
        """x / valid: device pointers of this rank's shard → the GLOBAL (sum, count), on every rank"""
        if np.dtype(dtype) == np.int64:
            return self.comm.cmp_filter_sum_i64(cmpop, x, valid, off, n_local, int(thr))
        return self.comm.cmp_filter_sum_f64(cmpop, x, valid, off, n_local, float(thr))

    def merge_groups(self, is_f64: bool, keys, sums, counts, first_rows, ngroups_local: int, row_offset: int, capacity: int,
                     out_keys, out_sums, out_counts, out_first_rows, null_group_local: int = -1, with_null_group: bool = False):
        """device pointers in and out; returns the global group count (with_null_group: and the merged null group's position);
        null_group_local = the null group hash_sum reported for this rank's shard, -1 for none"""
        return self.comm.merge_groups(is_f64, keys, sums, counts, first_rows, ngroups_local, row_offset, capacity,
                                      out_keys, out_sums, out_counts, out_first_rows, null_group_local, with_null_group)
