cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hash_sum or groupby or group or hash_encode" > gpurun_out/r04_12_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_12_pytest.log
tail -6 gpurun_out/r04_12_pytest.log | cut -c1-250
timeout 600 python scripts/bench_gb_mid.py | tee gpurun_out/r04_12_gb_mid.json
bash scripts/gpu_calls/r04_11_bench.sh
