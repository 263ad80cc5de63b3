cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_compute_api.py tests/test_golden.py tests/test_gpu_parity.py -m gpu -q -x -k "round" > gpurun_out/pytest_ext.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_ext.log
tail -30 gpurun_out/pytest_ext.log
