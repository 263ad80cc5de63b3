"""An executable MODEL of the multi-GPU merge protocol — TEST INFRASTRUCTURE.

The product runs configs C4 / C5 behind the C ABI: ah_comm_cmp_filter_sum_{i64,f64} and ah_comm_merge_groups of
libarrowhip.so (csrc/ah_comm.hip), driven by arrow_go_amd.distributed.  This module restates the same protocol in Python over
injected providers — the exchanges (torch.distributed, or a byte-level comm with the ah_comm_* interface) and the per-rank
compute (a numpy / oracle stand-in) — so that the protocol's bookkeeping (owner bucketing, the size table, block packing and
offsets, source-rank order, the final ordering by global first row) can be exercised with world 2 and 3 over gloo on a machine
WITHOUT a GPU, and so that the GPU tests have a second, independent answer to compare the C implementation with.
Nothing under arrow_go_amd/ imports it.
"""
from __future__ import annotations

import numpy as np

from arrow_go_amd.distributed import shard_bounds, hash_int, owner_of  # noqa: F401


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
        from tests import ddx_model as DD
        s = torch.zeros(4, dtype=torch.float64, device=self.device)       # {s, e, bs, be}: csrc/ah_ddsum.h, un-rounded
        c = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.local.cmp_filter_sum_partial(cmpop, x_ptr, valid_ptr, off, n_local, thr, dtype, s.data_ptr(), c.data_ptr())
        # float64: all-gather the un-rounded accumulators and merge them in RANK order on every rank, round once: bit-identical across
        # ranks and runs (an all-reduce's order is not) and within 1 ULP of the exact sum over the undivided column for any world size
        parts = self.coll.all_gather_rows(torch, s)
        self.coll.all_reduce_sum(torch, c)
        acc = DD.zero()
        for p in parts.tolist():
            DD.merge(acc, p)
        return DD.result(acc), int(c.item())

    # ---- C5: hash group-by sum ----------------------------------------------------------------
    def merge_groups(self, torch, keys: np.ndarray, sums: np.ndarray, counts: np.ndarray, first_rows: np.ndarray,
                     row_offset: int, null_group_local: int = -1, with_null_group: bool = False):
        """Plan A merge, host-array convenience form: this rank's LOCAL aggregate (group key bit
        patterns (uint64), partial sum, valid-value count, first local row) → the global groups in
        order of global first occurrence, on every rank.  The work happens in merge_groups_t.
        null_group_local: index of the null-key group in the local aggregate (-1: none).  It has no key to be owned by: it leaves the
        local columns, every rank's null tuple travels in one small all-gather, is merged in rank order on every rank and joins the
        others before the ordering by first row (ah_comm_merge_groups, csrc/ah_comm.hip)."""
        is_float = sums.dtype == np.float64
        cols = np.stack([keys.view(np.int64), sums.view(np.int64), counts.astype(np.int64),
                         first_rows.astype(np.int64) + np.int64(row_offset)])
        mine = np.zeros(4, np.int64)
        if null_group_local >= 0:
            mine[0], mine[1:] = 1, cols[1:, null_group_local]
            cols = np.delete(cols, null_group_local, axis=1)
        nulls = self.coll.all_gather_rows(torch, torch.from_numpy(mine).to(self.device)).cpu().numpy()      # [world, 4]
        rows = self.merge_groups_t(torch, torch.from_numpy(np.ascontiguousarray(cols)).to(self.device), is_float).cpu().numpy()
        null_pos = -1
        have = nulls[nulls[:, 0] != 0]
        if have.shape[0]:
            if is_float:
                tot = np.float64(0.0)
                for v in have[:, 1].view(np.float64):          # rank order
                    tot = tot + v
                sbits = np.array([tot]).view(np.int64)[0]
            else:
                with np.errstate(over="ignore"):
                    sbits = have[:, 1].view(np.uint64).sum(dtype=np.uint64).view(np.int64) if have.shape[0] > 1 else have[0, 1]
            tup = np.array([0, sbits, have[:, 2].sum(), have[:, 3].min()], np.int64)
            null_pos = int((rows[3] < tup[3]).sum())
            rows = np.concatenate([rows[:, :null_pos], tup[:, None], rows[:, null_pos:]], axis=1)
        out_sums = rows[1].view(np.float64) if is_float else rows[1]
        res = (rows[0].view(np.uint64), out_sums, rows[2], rows[3])
        return res + (null_pos,) if with_null_group else res

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
