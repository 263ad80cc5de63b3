#!/usr/bin/env python3
"""Single-process (world_size 1) check of the multi-GPU path on a real GPU: torch imported
FIRST (its bundled HIP runtime must be the only one in the process), process group on the
"nccl" backend (= RCCL), ah_ctx sharing torch's current stream, fused kernel writing
straight into a torch tensor, RCCL all-reduce, hash group-by + owner merge.  Exits non-zero
on any mismatch with the CPU oracle."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29617")
rank = int(os.environ.get("RANK", "0"))
world = int(os.environ.get("WORLD_SIZE", "1"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local_rank)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

import arrow_go_amd as ah  # noqa: E402  (after torch, on purpose)
from arrow_go_amd.distributed import AhCommCollectives, HipLocal, ShardedCompute, shard_bounds  # noqa: E402
from tests import oracle_lib as OL  # noqa: E402

o = OL.load_oracle()
local = HipLocal(local_rank, stream=torch.cuda.current_stream().cuda_stream)
ctx = local.ctx
# the exchanges go through the C ABI (ah_comm_*: RCCL on the ah_ctx's stream); torch.distributed only carries the 128-byte id
uid = [ah.Comm.unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
comm = ah.Comm(ctx, rank, world, uid[0])
sc = ShardedCompute(AhCommCollectives(comm, torch.device("cuda", local_rank)), torch.device("cuda", local_rank), local)
sc_torch = ShardedCompute(dist, torch.device("cuda", local_rank), local)     # cross-check: the same steps over torch's "nccl" group
rng = np.random.default_rng(1234)
n = 1_000_003
x = rng.integers(-10**9, 10**9, n, dtype=np.int64)
valid_bits = rng.random(n) < 0.9
lo, hi = shard_bounds(n, rank, world)
xt = torch.from_numpy(x[lo:hi].copy()).cuda()                       # torch owns the device memory
vt = torch.from_numpy(np.packbits(valid_bits[lo:hi], bitorder="little")).cuda()
got = sc.cmp_filter_sum(torch, 2, xt.data_ptr(), vt.data_ptr(), 0, hi - lo, 0, np.int64)
exp = o.cmp_filter_sum_i64(2, x, np.packbits(valid_bits, bitorder="little"), 0, 0)
assert got == exp, (got, exp)
assert sc_torch.cmp_filter_sum(torch, 2, xt.data_ptr(), vt.data_ptr(), 0, hi - lo, 0, np.int64) == exp
xf = rng.uniform(-1, 1, n)
xft = torch.from_numpy(xf[lo:hi].copy()).cuda()
gotf = sc.cmp_filter_sum(torch, 2, xft.data_ptr(), vt.data_ptr(), 0, hi - lo, 0.25, np.float64)
assert gotf[1] == int(((xf > 0.25) & valid_bits).sum())
# group-by: local aggregate on the GPU, merge through the collective layer
keys = rng.integers(0, 777, n).astype(np.int64) * 1000003
vals = rng.integers(-2**40, 2**40, n, dtype=np.int64)
kt = torch.from_numpy(keys[lo:hi].copy()).cuda(); vt2 = torch.from_numpy(vals[lo:hi].copy()).cuda()
m = hi - lo
ok = torch.zeros(m + 1, dtype=torch.int64, device="cuda"); osum = torch.zeros(m + 1, dtype=torch.int64, device="cuda")
oc = torch.zeros(m + 1, dtype=torch.int64, device="cuda"); of = torch.zeros(m + 1, dtype=torch.int64, device="cuda")
ng, nid = ctx.hash_sum("i64", kt.data_ptr(), None, 0, vt2.data_ptr(), None, 0, m, ok.data_ptr(), osum.data_ptr(), oc.data_ptr(), of.data_ptr())
torch.cuda.synchronize()
lk = ok[:ng].cpu().numpy().view(np.uint64); ls = osum[:ng].cpu().numpy(); lc = oc[:ng].cpu().numpy(); lf = of[:ng].cpu().numpy()
mk, ms, mc, mf = sc.merge_groups(torch, lk, ls, lc, lf, lo)
ek, es, ec, _nid, ef = o.hash_sum("i64", keys, None, 0, vals, None, 0)
assert mk.tobytes() == ek.tobytes() and ms.tobytes() == es.tobytes() and mc.tobytes() == ec.tobytes() and mf.tobytes() == ef.tobytes()
tk, ts, tc, tf = sc_torch.merge_groups(torch, lk, ls, lc, lf, lo)
assert tk.tobytes() == ek.tobytes() and ts.tobytes() == es.tobytes() and tc.tobytes() == ec.tobytes() and tf.tobytes() == ef.tobytes()
# raw entry points: in-place all-reduce, all-gather, ragged all-to-all (at world 1 the own block is a device copy)
t = torch.arange(5, dtype=torch.float64, device="cuda")
comm.allreduce_sum(ah._native.FLOAT64, t.data_ptr(), t.data_ptr(), 5)
gat = torch.empty(world * 5, dtype=torch.float64, device="cuda")
comm.allgather(t.data_ptr(), gat.data_ptr(), 40)
src = torch.arange(world * 3, dtype=torch.int64, device="cuda") + 100 * rank
dst = torch.zeros(world * 3, dtype=torch.int64, device="cuda")
comm.alltoallv(src.data_ptr(), [24] * world, [24 * r for r in range(world)], dst.data_ptr(), [24] * world, [24 * r for r in range(world)])
ctx.sync()
assert t.tolist() == [float(world * i) for i in range(5)] and gat.tolist() == [float(world * i) for i in range(5)] * world
assert dst.tolist() == [100 * r + 3 * rank + j for r in range(world) for j in range(3)]
# the three merge steps of the collective layer on the GPU, against their numpy restatement: bucket a
# group list for a 3-rank world, re-aggregate tuples with duplicate keys, order by first row
from arrow_go_amd.distributed import owner_of  # noqa: E402
g = 50021
gk = (rng.integers(0, 9000, g).astype(np.int64) * 7919)
cols = np.stack([gk, rng.integers(-2**40, 2**40, g, dtype=np.int64), rng.integers(1, 50, g, dtype=np.int64),
                 np.sort(rng.choice(10**7, g, replace=False)).astype(np.int64)])
ct = torch.from_numpy(cols).cuda()
parts = local.partition_by_owner(torch, ct, 3)
own = owner_of(gk.view(np.uint64), 3)
for r in range(3):
    assert parts[r].cpu().numpy().tobytes() == np.ascontiguousarray(cols[:, own == r]).tobytes(), r
merged = local.merge_tuples(torch, ct, False).cpu().numpy()
uk, first_pos = np.unique(gk, return_index=True)
seen = np.argsort(first_pos, kind="stable")
assert merged[0].tolist() == uk[seen].tolist()
for j, key in enumerate(merged[0][:200].tolist()):
    m_ = gk == key
    assert merged[1][j] == cols[1][m_].sum() and merged[2][j] == cols[2][m_].sum() and merged[3][j] == cols[3][m_][0]
shuf = ct[:, torch.randperm(g, device="cuda")].contiguous()
ordered = local.order_by_first(torch, shuf).cpu().numpy()
assert ordered.tobytes() == cols.tobytes()       # first rows are distinct and ascending in `cols`
comm.close()
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("dist_gpu_check ok: ah_comm_* (RCCL through the C ABI) all-reduce / all-gather / all-to-all + fused kernel + group-by merge, world =", world)
