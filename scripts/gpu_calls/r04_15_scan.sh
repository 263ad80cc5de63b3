cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_compute_api.py -m gpu -q -x -k "cumulative or scan or graph" > gpurun_out/r04_15_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_15_pytest.log
tail -12 gpurun_out/r04_15_pytest.log | cut -c1-250
timeout 600 python -m pytest tests/test_full_size.py -m gpu -q -x -k "cumulative" > gpurun_out/r04_15_pytest_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_15_pytest_full.log
tail -3 gpurun_out/r04_15_pytest_full.log | cut -c1-250
timeout 300 python scripts/bench_scan.py 2>&1 | tail -3 | cut -c1-1500
