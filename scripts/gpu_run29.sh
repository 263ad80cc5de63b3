cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_compute_api.py tests/test_chunked.py -m gpu -q -x -k "sort or hash or take_binary or filter_binary or binary or chunked or record or group" > gpurun_out/pytest_ext.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_ext.log
tail -5 gpurun_out/pytest_ext.log
timeout 200 python scripts/bench_sort.py 2>&1 | tail -1 | cut -c1-600
timeout 200 python scripts/bench_hash.py 10 16 20 24 2>&1 | tail -1 | cut -c1-900
