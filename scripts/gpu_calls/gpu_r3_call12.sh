cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_distributed_gpu.py -m gpu -x -q > gpurun_out/r3c12_pytest_dist.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c12_pytest_dist.log
tail -25 gpurun_out/r3c12_pytest_dist.log
