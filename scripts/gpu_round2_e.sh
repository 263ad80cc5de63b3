cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_distributed_gpu.py tests/test_abi.py -q -x > gpurun_out/pytest_dist.log 2>&1; tail -15 gpurun_out/pytest_dist.log
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --steps 10 --warmup 3 --no-kernels --no-cpu-baseline > gpurun_out/bench_torchrun1.json 2> gpurun_out/bench_torchrun1.err; echo rc=$?; cat gpurun_out/bench_torchrun1.json; tail -5 gpurun_out/bench_torchrun1.err
