cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_temporal.py tests/test_ipc.py tests/test_compute_api.py -q -x -m gpu > gpurun_out/pytest_k.log 2>&1; tail -30 gpurun_out/pytest_k.log
