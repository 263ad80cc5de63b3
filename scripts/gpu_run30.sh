cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_compute_api.py tests/test_chunked.py tests/test_device_interface.py tests/test_ipc.py tests/test_expressions.py -m gpu -q -x > gpurun_out/pytest_ext.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_ext.log
tail -4 gpurun_out/pytest_ext.log
timeout 300 python scripts/bench_host_call.py 2>&1 | tail -1
ARROWHIP_POOL_BYTES=0 timeout 300 python scripts/bench_host_call.py 2>&1 | tail -1
