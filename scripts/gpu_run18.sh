cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hash" > gpurun_out/pytest_hash.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_hash.log
tail -15 gpurun_out/pytest_hash.log
