# round 5, call 7: the one-call Filter (ah_filter_primitive_once) — every filter test of the suite cross-checks it; timings
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_compute_api.py tests/test_full_size.py -m gpu -q -x -k "filter" > gpurun_out/r05_07_filter.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_07_filter.log
tail -4 gpurun_out/r05_07_filter.log | cut -c1-300
timeout 300 python scripts/bench_filter_once.py | tee gpurun_out/r05_07_filter_once.json
