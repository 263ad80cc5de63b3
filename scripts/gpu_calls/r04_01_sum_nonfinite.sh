cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_sum_nonfinite.py tests/test_ingest.py tests/test_distributed_gpu.py "tests/test_gpu_parity.py" -m gpu -q -x -k "sum or ingest or ranks or fused or cmp_filter" > gpurun_out/r04_01_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_01_pytest.log
tail -15 gpurun_out/r04_01_pytest.log
timeout 600 python -m pytest tests/test_full_size.py -m gpu -q -x -k "nonfinite or c2_sum or fused" > gpurun_out/r04_01_pytest_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_01_pytest_full.log
tail -8 gpurun_out/r04_01_pytest_full.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 600 python bench.py > gpurun_out/r04_01_bench.json 2> gpurun_out/r04_01_bench.err; tail -c 3000 gpurun_out/r04_01_bench.json
