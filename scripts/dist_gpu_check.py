#!/usr/bin/env python3
"""Single-process (world_size 1) check of the multi-GPU path over RCCL on a real GPU: torch imported FIRST (its bundled HIP
runtime must be the only one in the process), process group on the "nccl" backend (= RCCL) only to carry the 128-byte id,
ah_ctx sharing torch's current stream, then configs C4 / C5 as single calls of the C ABI (ah_comm_cmp_filter_sum_*,
ah_comm_merge_groups) and the raw collectives.  Exits non-zero on any mismatch with the CPU oracle.  (World 2 and 3 run on the
same box through the host-transport communicator: scripts/dist_gpu_ranks.py.)"""
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
from arrow_go_amd.distributed import ShardedGpu, shard_bounds  # noqa: E402
from tests import oracle_lib as OL  # noqa: E402

o = OL.load_oracle()
ctx = ah.Context(local_rank, stream=torch.cuda.current_stream().cuda_stream)
uid = [ah.Comm.unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
comm = ah.Comm(ctx, rank, world, uid[0])
sg = ShardedGpu(comm)
rng = np.random.default_rng(1234)
n = 1_000_003
x = rng.integers(-10**9, 10**9, n, dtype=np.int64)
valid_bits = rng.random(n) < 0.9
lo, hi = shard_bounds(n, rank, world)
xt = torch.from_numpy(x[lo:hi].copy()).cuda()                       # torch owns the device memory
vt = torch.from_numpy(np.packbits(valid_bits[lo:hi], bitorder="little")).cuda()
got = sg.cmp_filter_sum(2, xt.data_ptr(), vt.data_ptr(), 0, hi - lo, 0, np.int64)
exp = o.cmp_filter_sum_i64(2, x, np.packbits(valid_bits, bitorder="little"), 0, 0)
assert got == exp, (got, exp)
xf = rng.uniform(-1, 1, n)
xft = torch.from_numpy(xf[lo:hi].copy()).cuda()
gotf = sg.cmp_filter_sum(2, xft.data_ptr(), vt.data_ptr(), 0, hi - lo, 0.25, np.float64)
assert gotf[1] == int(((xf > 0.25) & valid_bits).sum())
if world == 1:
    _, s_exact, _c = o.cmp_filter_sum_f64(2, xf, np.packbits(valid_bits, bitorder="little"), 0, 0.25)
    assert abs(gotf[0] - s_exact) <= np.spacing(abs(s_exact)), (gotf[0], s_exact)
# group-by: local aggregate on the GPU, merged through ah_comm_merge_groups
keys = rng.integers(0, 777, n).astype(np.int64) * 1000003
vals = rng.integers(-2**40, 2**40, n, dtype=np.int64)
kt = torch.from_numpy(keys[lo:hi].copy()).cuda(); vt2 = torch.from_numpy(vals[lo:hi].copy()).cuda()
m = hi - lo
loc = [torch.zeros(m + 1, dtype=torch.int64, device="cuda") for _ in range(4)]
ng, nid = ctx.hash_sum("i64", kt.data_ptr(), None, 0, vt2.data_ptr(), None, 0, m, *[t.data_ptr() for t in loc])
res = [torch.zeros(n + 1, dtype=torch.int64, device="cuda") for _ in range(4)]
G = sg.merge_groups(False, loc[0].data_ptr(), loc[1].data_ptr(), loc[2].data_ptr(), loc[3].data_ptr(), ng, lo, n + 1, *[t.data_ptr() for t in res])
torch.cuda.synchronize()
mk, ms, mc, mf = (t[:G].cpu().numpy() for t in res)
ek, es, ec, _nid, ef = o.hash_sum("i64", keys, None, 0, vals, None, 0)
assert mk.view(np.uint64).tobytes() == ek.tobytes() and ms.tobytes() == es.tobytes() and mc.tobytes() == ec.tobytes() and mf.tobytes() == ef.tobytes()
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
comm.close()
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("dist_gpu_check ok: ah_comm_* over RCCL through the C ABI — all-reduce / all-gather / all-to-all, C4 (ah_comm_cmp_filter_sum_*) and C5 (ah_comm_merge_groups), world =", world)
