cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ipc.py -m gpu -q -x > gpurun_out/pytest_ipc.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_ipc.log
tail -25 gpurun_out/pytest_ipc.log
