"""The multi-GPU path on a real GPU with world_size 1 (the only size a 1-GPU box allows):
torch.distributed "nccl" (= RCCL) + libarrowhip.so sharing torch's stream.  Runs in a
subprocess because torch must be imported BEFORE libarrowhip.so there (its bundled HIP
runtime has to be the only one in the process)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_path_world1():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dist_gpu_check.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "dist_gpu_check ok" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_comm_entries_several_ranks_on_one_gpu(world):
    """ah_comm_cmp_filter_sum_{i64,f64} and ah_comm_merge_groups with world 2 and 3: one process per rank, all on device 0, the
    bytes carried by the host-transport communicator over gloo (scripts/dist_gpu_ranks.py)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
                        "--master-port", str(29650 + world), os.path.join(ROOT, "scripts", "dist_gpu_ranks.py")], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "dist_gpu_ranks ok" in r.stdout


@pytest.mark.gpu
def test_filter_count_cache_is_off_on_a_shared_stream():
    """ah_ctx_create_on_stream: a foreign kernel (torch, on the shared stream) rewrites the mask between ah_filter_count and
    ah_filter_primitive; the fill must follow the new mask, never the count's stale tile prefixes (scripts/shared_stream_filter_check.py)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "shared_stream_filter_check.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "shared_stream_filter_check ok" in r.stdout


@pytest.mark.gpu
def test_bench_world2_one_gpu(tmp_path):
    """bench.py's OWN multi-rank code (communicator set-up, max-over-ranks timing, the watchdog thread, the C4 / C5 secondary sections with
    the owner merge) with two ranks on one GPU over the host-transport communicator (--transport gloo): exactly ONE JSON line, C4 and C5
    present, and their values == the oracle over the UNDIVIDED data (the concatenation of the ranks' seeded shards)."""
    import json
    import numpy as np
    from tests import oracle_lib as OL
    rows, world = 1 << 22, 2
    dump = str(tmp_path / "bench_world2.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", "29671",
                        os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--rows", str(rows), "--steps", "3", "--warmup", "1", "--transport", "gloo", "--dump", dump],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-3000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == world and res["steps"] == 3 and res["value"] > 0
    c4, c5 = res["c4_filter_aggregate"], res["c5_group_by"]
    assert "error" not in c4 and "error" not in c5, (c4, c5)
    assert "merged by key-hash owner" in c5["workload"] and c5["n_gpus"] == world
    # the same seeded columns bench.py builds (fill_random: one chunk of min(rows, 2^22) rows per column, seeds 10 / 30 / 77 + rank)
    o = OL.load_oracle()
    a = np.concatenate([np.random.default_rng(10 + k).integers(-2**62, 2**62, rows, dtype=np.int64) for k in range(world)])
    x = np.concatenate([np.random.default_rng(30 + k).uniform(-1e6, 1e6, rows) for k in range(world)])
    def bench_keys(k):   # bench.py's C5 keys: the generator's first 2^22 draws go to the tiled block of rounds 1-4, the column is the draws after it
        g = np.random.default_rng(77 + k)
        g.integers(0, 1 << 16, 1 << 22, dtype=np.uint64)
        parts = [g.integers(0, 1 << 16, min(1 << 22, rows - off), dtype=np.uint64) for off in range(0, rows, 1 << 22)]
        return (np.concatenate(parts) * np.uint64(0x9E3779B97F4A7C15)).view(np.int64)
    keys = np.concatenate([bench_keys(k) for k in range(world)])
    d = np.load(dump)
    assert d["merged"][0] == 1
    assert tuple(d["c4"].tolist()) == o.cmp_filter_sum_i64(2, a, None, 0, 0)                    # C4: Σ and count of a > 0 over both shards
    ek, es, ec, _nid, ef = o.hash_sum("f64", keys, None, 0, x, None, 0)                          # C5 over the undivided columns
    assert c5["groups"] == ek.size
    assert d["keys"].tobytes() == ek.tobytes() and d["counts"].tobytes() == ec.tobytes() and d["first_rows"].tobytes() == ef.tobytes()
    assert np.all(np.abs(d["sums"] - es) <= 1e-9 * np.maximum(1.0, np.abs(es)))                 # (the oracle adds in row order; the device sum is the correctly rounded one)
