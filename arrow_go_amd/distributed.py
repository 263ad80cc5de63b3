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
    global first occurrence.  Bucketing, owner merge and final ordering are `local` provider
    steps: on the GPU for HipLocal (hash partition + compactions, two group-by passes + a
    gather, radix sort + gathers) — the tuples never visit host memory.

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


    # ---- the three compute steps of the group merge, on device tensors ---------------------------------
    def partition_by_owner(self, torch, cols, world: int):
        """cols [4, g] int64 (row 0 = key bits) → per destination rank a [4, g_r] tensor: owner =
        (hashInt(key) >> 40) mod world (ah_hash_partition_u64), then one compare + four stream
        compactions per destination — all on the GPU."""
        g = cols.shape[1]
        dev = cols.device
        if world == 1 or g == 0:
            return [cols] + [torch.zeros((4, 0), dtype=torch.int64, device=dev) for _ in range(world - 1)]
        owner = torch.empty(g, dtype=torch.int32, device=dev)
        self.ctx.hash_partition(cols[0].data_ptr(), g, world, owner.data_ptr())
        mask = torch.zeros((g + 7) // 8 + 64, dtype=torch.uint8, device=dev)
        out = []
        for r in range(world):
            self.ctx.comparison(self.N.CMP_EQ, self.N.SHAPE_AS, self.N.INT32, owner.data_ptr(), np.array([r], np.int32), mask.data_ptr(), g, 0)
            n_r = self.ctx.filter_count(mask.data_ptr(), None, 0, g, 0)
            part = torch.empty((4, n_r), dtype=torch.int64, device=dev)
            for c in range(4):
                self.ctx.filter_primitive(8, cols[c].data_ptr(), None, 0, mask.data_ptr(), None, 0, g, 0, n_r, part[c].data_ptr(), None,
                                          want_null_count=False)
            out.append(part)
        return out

    def merge_tuples(self, torch, got, is_float: bool):
        """got [4, m]: tuples received for the keys this rank owns, source ranks ascending → [4, g]:
        per key the sum of the partial sums, the sum of the counts, and the first row of the FIRST
        tuple (= the smallest global first row, because lower ranks hold lower row ranges) —
        two device group-by passes (ah_hash_sum_*) and one gather."""
        m = got.shape[1]
        dev = got.device
        if m == 0:
            return got
        ok = torch.empty(m + 1, dtype=torch.int64, device=dev); osum = torch.empty(m + 1, dtype=torch.int64, device=dev)
        oc = torch.empty(m + 1, dtype=torch.int64, device=dev); ofirst = torch.empty(m + 1, dtype=torch.int64, device=dev)
        ng, _ = self.ctx.hash_sum("f64" if is_float else "i64", got[0].data_ptr(), None, 0, got[1].data_ptr(), None, 0, m,
                                  ok.data_ptr(), osum.data_ptr(), oc.data_ptr(), ofirst.data_ptr())
        ok2 = torch.empty(m + 1, dtype=torch.int64, device=dev); csum = torch.empty(m + 1, dtype=torch.int64, device=dev)
        oc2 = torch.empty(m + 1, dtype=torch.int64, device=dev)
        ng2, _ = self.ctx.hash_sum("i64", got[0].data_ptr(), None, 0, got[2].data_ptr(), None, 0, m, ok2.data_ptr(), csum.data_ptr(),
                                   oc2.data_ptr(), None)
        assert ng2 == ng
        out = torch.empty((4, ng), dtype=torch.int64, device=dev)
        out[0] = ok[:ng]; out[1] = osum[:ng]; out[2] = csum[:ng]
        self.ctx.take_primitive(8, got[3].data_ptr(), None, 0, m, 8, True, ofirst.data_ptr(), None, 0, ng, True, out[3].data_ptr(), None)
        return out

    def order_by_first(self, torch, rows):
        """rows [4, G] → the same tuples ascending by row 3 (global first row): device radix sort
        (ah_sort_indices) + four gathers."""
        G = rows.shape[1]
        if G <= 1:
            return rows
        dev = rows.device
        idx = torch.empty(G, dtype=torch.int64, device=dev)
        self.ctx.sort_indices(self.N.INT64, rows[3].data_ptr(), None, 0, G, False, False, idx.data_ptr())
        out = torch.empty_like(rows)
        for c in range(4):
            self.ctx.take_primitive(8, rows[c].data_ptr(), None, 0, G, 8, False, idx.data_ptr(), None, 0, G, False, out[c].data_ptr(), None)
        return out


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
        """Plan A merge, host-array convenience form: this rank's LOCAL aggregate (group key bit
        patterns (uint64), partial sum, valid-value count, first local row) → the global groups in
        order of global first occurrence, on every rank.  The work happens in merge_groups_t."""
        is_float = sums.dtype == np.float64
        cols = np.stack([keys.view(np.int64), sums.view(np.int64), counts.astype(np.int64),
                         first_rows.astype(np.int64) + np.int64(row_offset)])
        rows = self.merge_groups_t(torch, torch.from_numpy(np.ascontiguousarray(cols)).to(self.device), is_float).cpu().numpy()
        out_sums = rows[1].view(np.float64) if is_float else rows[1]
        return rows[0].view(np.uint64), out_sums, rows[2], rows[3]

    def merge_groups_t(self, torch, cols, is_float: bool):
        """cols: [4, g] int64 tensor on this rank's device — rows = key bits, sum bits, count,
        GLOBAL first row of this rank's local groups.  Returns the merged [4, G] tensor (every rank
        gets all groups, ordered by global first occurrence).  Bytes exchanged are O(groups), never
        O(rows); every compute step (owner bucketing, owner-side re-aggregation, final ordering)
        runs through the `local` provider — on the GPU for HipLocal."""
        world = self.world
        # 1. bucket this rank's groups by owner rank
        send = self.local.partition_by_owner(torch, cols, world)          # list of [4, g_r]
        # 2. sizes, then payloads (ragged all-to-all)
        sizes = torch.tensor([t.shape[1] for t in send], dtype=torch.int64, device=self.device)
        rsizes = torch.zeros(world, dtype=torch.int64, device=self.device)
        self.dist.all_to_all_single(rsizes, sizes)
        recv = [torch.zeros((4, int(k)), dtype=torch.int64, device=self.device) for k in rsizes.tolist()]
        self._all_to_all(recv, [t.contiguous() for t in send])
        got = torch.cat(recv, dim=1).contiguous()                         # source ranks in ascending order
        # 3. the owner re-aggregates its keys (sum of partial sums, sum of counts, first of the firsts)
        mine = self.local.merge_tuples(torch, got, is_float)              # [4, g_owned]
        # 4. every rank gets every owner's groups …
        n_mine = torch.tensor([mine.shape[1]], dtype=torch.int64, device=self.device)
        all_n = [torch.zeros(1, dtype=torch.int64, device=self.device) for _ in range(world)]
        self.dist.all_gather(all_n, n_mine)
        counts = [int(t.item()) for t in all_n]
        mx = max(counts) if counts else 0
        pad = torch.zeros((4, mx), dtype=torch.int64, device=self.device)
        pad[:, : mine.shape[1]] = mine
        gathered = [torch.zeros((4, mx), dtype=torch.int64, device=self.device) for _ in range(world)]
        self.dist.all_gather(gathered, pad)
        rows = torch.cat([gathered[r][:, : counts[r]] for r in range(world)], dim=1).contiguous()
        # 5. … ordered by global first occurrence (what a single-process `unique` would produce)
        return self.local.order_by_first(torch, rows)

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
