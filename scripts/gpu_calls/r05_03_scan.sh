# round 5, call 3: one-pass cumulative_sum for checked / null-carrying integer columns — parity and timings
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -x -k "cumulative" > gpurun_out/r05_03_scan.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_03_scan.log
tail -5 gpurun_out/r05_03_scan.log | cut -c1-300
timeout 300 python scripts/bench_scan.py | tee gpurun_out/r05_03_bench_scan.json
