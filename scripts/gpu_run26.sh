cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_compute_api.py tests/test_golden.py -m gpu -q -x -k "validity_and_rounding or floor_ceil or extended_arithmetic" > gpurun_out/pytest_ext.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_ext.log
tail -30 gpurun_out/pytest_ext.log
