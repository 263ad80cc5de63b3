cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m pytest tests/test_distributed_gpu.py tests/test_gpu_parity.py::test_hash_sum -m gpu -q -x > gpurun_out/pytest_dist.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_dist.log; tail -30 gpurun_out/pytest_dist.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err; echo "torchrun rc=$?"; cat gpurun_out/bench_torchrun1.json | cut -c1-700; tail -5 gpurun_out/bench_torchrun1.err
