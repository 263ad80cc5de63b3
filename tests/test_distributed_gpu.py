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
