# round 5, call 1: short Float64 Sum parity, group-by parity after the poll / look fixes, group-by timings + kernel trace, the reserving scatter micro
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_sum_short.py tests/test_sum_nonfinite.py -m gpu -q -x > gpurun_out/r05_01_sum.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_01_sum.log
tail -4 gpurun_out/r05_01_sum.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hash_sum or groupby or group" > gpurun_out/r05_01_gb.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_01_gb.log
tail -4 gpurun_out/r05_01_gb.log | cut -c1-250
timeout 300 python scripts/bench_gb_small.py | tee gpurun_out/r05_01_gb_small.json
timeout 300 python scripts/bench_gb_mid.py | tee gpurun_out/r05_01_gb_mid.json
bash scripts/gpu_prof_cmd.sh r05_gb_mid scripts/bench_gb_mid.py | grep -i "quicklook\|aggregate\|scatter\|hist" | cut -c1-200
timeout 120 scripts/micro/scatter_reserve.bin 26 16 2>&1 | tee gpurun_out/r05_01_scatter_reserve.txt
timeout 120 scripts/micro/scatter_reserve.bin 26 20 2>&1 | tee -a gpurun_out/r05_01_scatter_reserve.txt
