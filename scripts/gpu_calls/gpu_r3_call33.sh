cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python scripts/bench_slices.py 2> /dev/null | tee gpurun_out/r3c33_slices.json
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -x -k "arith or add or sub or mul or unary or checked" > gpurun_out/r3c33_pytest.log 2>&1; tail -3 gpurun_out/r3c33_pytest.log
