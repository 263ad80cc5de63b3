cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "cumulative or sort or take_binary or hash" > gpurun_out/pytest_scan.log 2>&1; tail -5 gpurun_out/pytest_scan.log

timeout 300 python scripts/prof_workloads.py --only hash_sum_2^10,hash_sum_2^16,hash_sum_2^20,hash_sum_2^24,hash_sum_2^20_zipf,cumulative_sum_int64 > gpurun_out/wl_hash.json 2>gpurun_out/wl_hash.err; cat gpurun_out/wl_hash.json; tail -2 gpurun_out/wl_hash.err
