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

One process per GPU.  The exchanges go through the C ABI — ah_comm_* of libarrowhip.so, RCCL over
xGMI on the ah_ctx's compute stream (AhCommCollectives) — so a Go host can run the same steps
without Python; `torch.distributed` is the injected alternative (TorchCollectives: "gloo" for the
world-2 / 3 CPU tests).  torch is plumbing here: launcher rendezvous and device memory.

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


# ---- collective providers ------------------------------------------------------------------------------------------
# ShardedCompute needs four exchanges; who performs them is injected:
#   AhCommCollectives  — production: ah_comm_* of libarrowhip.so (RCCL over xGMI on the ah_ctx's stream, include/arrowhip.h);
#                        torch only lends the device memory
#   TorchCollectives   — torch.distributed: "gloo" in the CPU tests (world 2 and 3), "nccl" as a cross-check
class TorchCollectives:
    def __init__(self, dist, device):
        self.dist, self.device = dist, device
        self.rank, self.world = dist.get_rank(), dist.get_world_size()

    def all_reduce_sum(self, torch, t):
        self.dist.all_reduce(t)
        return t

    def all_gather_rows(self, torch, t):
        """t: 1-d tensor, same length on every rank → [world, len]"""
        parts = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        return torch.stack(parts)

    def exchange(self, torch, send):
        """ragged all-to-all of [4, g_r] int64 blocks (send[r] goes to rank r) → what arrived, concatenated in source-rank order"""
        world = self.world
        sizes = torch.tensor([t.shape[1] for t in send], dtype=torch.int64, device=self.device)
        rsizes = torch.zeros(world, dtype=torch.int64, device=self.device)
        self.dist.all_to_all_single(rsizes, sizes)
        recv = [torch.zeros((4, int(k)), dtype=torch.int64, device=self.device) for k in rsizes.tolist()]
        send = [t.contiguous() for t in send]
        try:
            self.dist.all_to_all(recv, send)
        except Exception:  # gloo has no all_to_all for CPU tensors in every build: point-to-point pairs, same bytes on the wire
            reqs = []
            for r in range(world):
                if r == self.rank:
                    recv[r].copy_(send[r])
                    continue
                if send[r].numel():
                    reqs.append(self.dist.isend(send[r], r))
                if recv[r].numel():
                    reqs.append(self.dist.irecv(recv[r], r))
            for q in reqs:
                q.wait()
        return torch.cat(recv, dim=1).contiguous()

    def gather_groups(self, torch, mine):
        """ragged all-gather of [4, g] blocks → [4, G], rank order"""
        world = self.world
        counts = [int(v) for v in self.all_gather_rows(torch, torch.tensor([mine.shape[1]], dtype=torch.int64, device=self.device))[:, 0].tolist()]
        mx = max(counts) if counts else 0
        pad = torch.zeros((4, mx), dtype=torch.int64, device=self.device)
        pad[:, : mine.shape[1]] = mine
        gathered = [torch.zeros((4, mx), dtype=torch.int64, device=self.device) for _ in range(world)]
        self.dist.all_gather(gathered, pad)
        return torch.cat([gathered[r][:, : counts[r]] for r in range(world)], dim=1).contiguous()


class AhCommCollectives:
    """The same four exchanges through the C ABI (ah_comm_*).  Sizes cross the host once per exchange (an all-gather of
    world × 8 bytes followed by a stream sync): the blocks themselves never leave the devices."""

    def __init__(self, comm, device):
        self.comm, self.device = comm, device
        self.rank, self.world = comm.rank, comm.world
        import arrow_go_amd as ah
        self.N = ah._native

    def all_reduce_sum(self, torch, t):
        tid = {torch.int64: self.N.INT64, torch.float64: self.N.FLOAT64, torch.int32: self.N.INT32, torch.float32: self.N.FLOAT32}[t.dtype]
        self.comm.allreduce_sum(tid, t.data_ptr(), t.data_ptr(), t.numel())
        return t

    def all_gather_rows(self, torch, t):
        t = t.contiguous()
        out = torch.empty((self.world, t.numel()), dtype=t.dtype, device=self.device)
        self.comm.allgather(t.data_ptr(), out.data_ptr(), t.numel() * t.element_size())
        return out

    def _sizes(self, torch, mine):
        """every rank's size vector: [world, len(mine)] on the host"""
        m = self.all_gather_rows(torch, torch.tensor(mine, dtype=torch.int64, device=self.device))
        self.comm.ctx.sync()
        return m.cpu().numpy()

    def exchange(self, torch, send):
        world, rank = self.world, self.rank
        cnt = [int(t.shape[1]) for t in send]
        table = self._sizes(torch, cnt)                       # table[s][r] = tuples rank s sends to rank r
        rcnt = [int(table[s][rank]) for s in range(world)]
        sbuf = torch.cat([t.t().contiguous() for t in send], dim=0).contiguous() if sum(cnt) else torch.zeros((0, 4), dtype=torch.int64, device=self.device)
        rbuf = torch.empty((sum(rcnt), 4), dtype=torch.int64, device=self.device)   # tuple-major: one contiguous block per peer
        offs = lambda c: [32 * int(v) for v in np.concatenate([[0], np.cumsum(c)[:-1]])]
        self.comm.alltoallv(sbuf.data_ptr(), [32 * c for c in cnt], offs(cnt), rbuf.data_ptr(), [32 * c for c in rcnt], offs(rcnt))
        return rbuf.t().contiguous()

    def gather_groups(self, torch, mine):
        world = self.world
        counts = [int(v) for v in self._sizes(torch, [int(mine.shape[1])])[:, 0]]
        mx = max(counts) if counts else 0
        pad = torch.zeros((mx, 4), dtype=torch.int64, device=self.device)
        pad[: mine.shape[1]] = mine.t()
        out = torch.empty((world, mx, 4), dtype=torch.int64, device=self.device)
        self.comm.allgather(pad.data_ptr(), out.data_ptr(), mx * 32)
        return torch.cat([out[r, : counts[r]] for r in range(world)], dim=0).t().contiguous()


class ShardedCompute:
    """Collective layer over a `local` leaf provider and a collective provider (`dist_or_coll`: a provider object,
    or torch.distributed itself → TorchCollectives)."""

    def __init__(self, dist_or_coll, device, local):
        self.coll = dist_or_coll if hasattr(dist_or_coll, "exchange") else TorchCollectives(dist_or_coll, device)
        self.device, self.local = device, local
        self.rank, self.world = self.coll.rank, self.coll.world

    # ---- C4: Compare(op scalar) → Filter(DropNulls) → Sum --------------------------------
    def cmp_filter_sum(self, torch, cmpop: int, x_ptr, valid_ptr, off: int, n_local: int, thr, dtype):
        """x_ptr / valid_ptr: this rank's shard (device pointers for HipLocal).  Returns the
        GLOBAL (sum, count)."""
        if np.dtype(dtype) == np.int64:
            part = torch.zeros(2, dtype=torch.int64, device=self.device)  # [sum, count]
            self.local.cmp_filter_sum_partial(cmpop, x_ptr, valid_ptr, off, n_local, thr, dtype,
                                              part.data_ptr(), part.data_ptr() + 8)
            self.coll.all_reduce_sum(torch, part)  # wrapping int64 sum: exact in any order — 16 bytes on the wire
            return int(part[0].item()), int(part[1].item())
        s = torch.zeros(1, dtype=torch.float64, device=self.device)
        c = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.local.cmp_filter_sum_partial(cmpop, x_ptr, valid_ptr, off, n_local, thr, dtype, s.data_ptr(), c.data_ptr())
        # float64: all-gather the partials and add them in RANK order on every rank, so the
        # result is bit-identical across ranks and runs (an all-reduce's order is not)
        parts = self.coll.all_gather_rows(torch, s)
        self.coll.all_reduce_sum(torch, c)
        total = 0.0
        for p in parts[:, 0].tolist():
            total += float(p)
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
        # 1. bucket this rank's groups by owner rank
        send = self.local.partition_by_owner(torch, cols, self.world)     # list of [4, g_r]
        # 2. ragged all-to-all: every owner receives the tuples of its keys, source ranks in ascending order
        got = self.coll.exchange(torch, send)
        # 3. the owner re-aggregates its keys (sum of partial sums, sum of counts, first of the firsts)
        mine = self.local.merge_tuples(torch, got, is_float)              # [4, g_owned]
        # 4. every rank gets every owner's groups …
        rows = self.coll.gather_groups(torch, mine)
        # 5. … ordered by global first occurrence (what a single-process `unique` would produce)
        return self.local.order_by_first(torch, rows)
