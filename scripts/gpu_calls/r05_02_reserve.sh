# round 5, call 2: the reserving scatter in the library — parity (both pipelines, overflow fallback), timings, kernel trace
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hash_sum or groupby or group" > gpurun_out/r05_02_gb.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_02_gb.log
tail -6 gpurun_out/r05_02_gb.log | cut -c1-300
timeout 300 python scripts/bench_gb_mid.py | tee gpurun_out/r05_02_gb_mid.json
timeout 300 python scripts/bench_gb_small.py | tee gpurun_out/r05_02_gb_small.json
bash scripts/gpu_prof_cmd.sh r05_gb_mid2 scripts/bench_gb_mid.py > /dev/null
