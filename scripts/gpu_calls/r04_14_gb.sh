cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hash_sum or groupby or group" > gpurun_out/r04_14_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_14_pytest.log
tail -6 gpurun_out/r04_14_pytest.log | cut -c1-250
timeout 300 python scripts/bench_gb_small.py | tee gpurun_out/r04_14_gb_small.json
bash scripts/gpu_calls/r04_13_prof.sh 2>&1 | tail -7
