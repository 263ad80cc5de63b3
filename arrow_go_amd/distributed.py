"""Record-batch sharding across the GPUs of one node.

The hot path shards at row-range granularity: Add, Compare, bitmap ops, Filter,
Take-by-local-index and Sum partials are independent per shard, so there is NO data-path
collective for them.  Exactly one exchange step exists, for reductions:

  * Compare → Filter → Sum (config C4): each rank runs the fused single-pass kernel on its
    own shard and leaves {sum, count} in HBM; the global result is ONE RCCL all-reduce of
    16 bytes (int64: exact, order-free) or an all-gather of the per-rank partials added in
    rank order (float64: bit-reproducible for a given world size).
  * hash group-by (config C5): local aggregate per shard → groups bucketed by key-hash
    owner (the reference's hashInt, top bits) → all-to-all of (key, sum, count, first_row)
    tuples — O(groups) bytes on the wire, never O(rows) → owner merges → groups ordered by
    global first occurrence.

One process per GPU (`torchrun`), `torch.distributed` with backend "nccl" (= RCCL over
xGMI) in production, "gloo" in the CPU tests.  torch is plumbing here: process group,
device tensors for the collectives, and its current stream is shared with the ah_ctx so
kernels and collectives order without host synchronisation.

The per-shard compute is a `local` object with the leaf methods used below; in production
it is `HipLocal` (libarrowhip.so on this rank's GPU; construction fails loudly without a
GPU).  Tests inject a stand-in so that the SHARDING / COLLECTIVE / MERGE logic — the code in
this file — runs under world_size-2 gloo on CPU.
"""
from __future__ import annotations

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
    """owner = top bits of hashInt(key), folded to the world size (SURVEY.md §8e plan A)."""
    return ((hash_int(keys) >> np.uint64(40)) % np.uint64(world)).astype(np.int64)


class HipLocal:
    """Per-rank leaf compute on this rank's GPU through the C ABI."""

    def __init__(self, device_id: int, stream: int | None = None):
        import arrow_go_amd as ah
        self.ctx = ah.Context(device_id, stream=stream)
        self.N = ah._native

    def cmp_filter_sum_partial(self, cmpop, x_ptr, valid_ptr, off, n, thr, dtype, out_sum_ptr, out_count_ptr):
        if np.dtype(dtype) == np.int64:
            self.ctx.cmp_filter_sum_i64_dev(cmpop, x_ptr, valid_ptr, off, n, int(thr), out_sum_ptr)  # [sum, count] contiguous
        else:
            self.ctx.cmp_filter_sum_f64_dev(cmpop, x_ptr, valid_ptr, off, n, float(thr), out_sum_ptr, out_count_ptr)


class ShardedCompute:
    """Collective layer over a `local` leaf provider."""

    def __init__(self, dist, device, local):
        self.dist, self.device, self.local = dist, device, local
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    # ---- C4: Compare(op scalar) → Filter(DropNulls) → Sum --------------------------------
    def cmp_filter_sum(self, torch, cmpop: int, x_ptr, valid_ptr, off: int, n_local: int, thr, dtype):
        """x_ptr / valid_ptr: this rank's shard (device pointers for HipLocal).  Returns the
        GLOBAL (sum, count)."""
        if np.dtype(dtype) == np.int64:
            part = torch.zeros(2, dtype=torch.int64, device=self.device)  # [sum, count]
            self.local.cmp_filter_sum_partial(cmpop, x_ptr, valid_ptr, off, n_local, thr, dtype,
                                              part.data_ptr(), part.data_ptr() + 8)
            self.dist.all_reduce(part)  # wrapping int64 sum: exact in any order
            return int(part[0].item()), int(part[1].item())
        s = torch.zeros(1, dtype=torch.float64, device=self.device)
        c = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.local.cmp_filter_sum_partial(cmpop, x_ptr, valid_ptr, off, n_local, thr, dtype, s.data_ptr(), c.data_ptr())
        # float64: all-gather the partials and add them in RANK order on every rank, so the
        # result is bit-identical across ranks and runs (an all-reduce's order is not)
        parts = [torch.zeros(1, dtype=torch.float64, device=self.device) for _ in range(self.world)]
        self.dist.all_gather(parts, s)
        self.dist.all_reduce(c)
        total = 0.0
        for p in parts:
            total += float(p.item())
        return total, int(c.item())

    # ---- C5: hash group-by sum ----------------------------------------------------------------
    def merge_groups(self, torch, keys: np.ndarray, sums: np.ndarray, counts: np.ndarray, first_rows: np.ndarray,
                     row_offset: int):
        """Plan A merge.  Inputs: this rank's LOCAL aggregate (host arrays of equal length:
        group key bit patterns (uint64), partial sum, valid-value count, first local row).
        Returns the global groups owned by... every rank gets the full, globally ordered
        result: (keys, sums, counts) sorted by global first occurrence.  Bytes exchanged are
        O(groups), never O(rows)."""
        world = self.world
        gfirst = first_rows.astype(np.int64) + np.int64(row_offset)
        own = owner_of(keys, world)
        is_float = sums.dtype == np.float64
        # pack tuples per destination: [key, sum(bits), count, first_row] as int64
        send = []
        for r in range(world):
            m = own == r
            buf = np.stack([keys[m].view(np.int64), sums[m].view(np.int64), counts[m].astype(np.int64), gfirst[m]], axis=1)
            send.append(torch.from_numpy(np.ascontiguousarray(buf)).to(self.device))
        # exchange sizes, then payloads (all_to_all of ragged lists)
        sizes = torch.tensor([t.shape[0] for t in send], dtype=torch.int64, device=self.device)
        rsizes = torch.zeros(world, dtype=torch.int64, device=self.device)
        self.dist.all_to_all_single(rsizes, sizes)
        recv = [torch.zeros((int(k), 4), dtype=torch.int64, device=self.device) for k in rsizes.tolist()]
        self._all_to_all(recv, send)
        got = torch.cat(recv, dim=0).cpu().numpy() if recv else np.zeros((0, 4), np.int64)
        # owner-side merge of its keys: deterministic — sort by (key, source order) then segment-reduce;
        # float partials are added in ascending global-first-row order of the contributing shard
        order = np.lexsort((got[:, 3], got[:, 0].view(np.uint64)))
        g = got[order]
        k = g[:, 0].view(np.uint64)
        starts = np.flatnonzero(np.concatenate([[True], k[1:] != k[:-1]])) if len(k) else np.zeros(0, np.int64)
        mk = k[starts]
        mc = np.add.reduceat(g[:, 2], starts) if len(k) else np.zeros(0, np.int64)
        mf = np.minimum.reduceat(g[:, 3], starts) if len(k) else np.zeros(0, np.int64)
        if is_float:
            ms = np.add.reduceat(g[:, 1].view(np.float64), starts) if len(k) else np.zeros(0, np.float64)
        else:
            with np.errstate(over="ignore"):
                ms = np.add.reduceat(g[:, 1].view(np.uint64), starts).view(np.int64) if len(k) else np.zeros(0, np.int64)
        # all-gather the owners' merged groups and order by global first occurrence
        mine = torch.from_numpy(np.ascontiguousarray(np.stack([mk.view(np.int64), ms.view(np.int64), mc, mf], axis=1))).to(self.device)
        n_mine = torch.tensor([mine.shape[0]], dtype=torch.int64, device=self.device)
        all_n = [torch.zeros(1, dtype=torch.int64, device=self.device) for _ in range(world)]
        self.dist.all_gather(all_n, n_mine)
        mx = max(int(t.item()) for t in all_n)
        pad = torch.zeros((mx, 4), dtype=torch.int64, device=self.device)
        pad[: mine.shape[0]] = mine
        gathered = [torch.zeros((mx, 4), dtype=torch.int64, device=self.device) for _ in range(world)]
        self.dist.all_gather(gathered, pad)
        rows = np.concatenate([gathered[r][: int(all_n[r].item())].cpu().numpy() for r in range(world)], axis=0) \
            if mx else np.zeros((0, 4), np.int64)
        rows = rows[np.argsort(rows[:, 3], kind="stable")]
        out_sums = rows[:, 1].view(np.float64) if is_float else rows[:, 1]
        return rows[:, 0].view(np.uint64), out_sums, rows[:, 2], rows[:, 3]

    def _all_to_all(self, recv, send):
        """ragged all-to-all; gloo has no all_to_all for CPU tensors in every build, so fall
        back to point-to-point send/recv pairs there (same bytes on the wire)."""
        try:
            self.dist.all_to_all(recv, send)
            return
        except Exception:
            pass
        reqs = []
        for r in range(self.world):
            if r == self.rank:
                recv[r].copy_(send[r])
                continue
            if send[r].numel():
                reqs.append(self.dist.isend(send[r], r))
            if recv[r].numel():
                reqs.append(self.dist.irecv(recv[r], r))
        for q in reqs:
            q.wait()
