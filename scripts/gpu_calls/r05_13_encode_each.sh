# round 5, call 13: dictionary_encode call by call (the round-end bench line read 4.6 ms at 2^24 keys where every other run read 2.6);
# the group-by sample with its round trips overlapped (bench_gb_mid, reserve 1)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python scripts/bench_encode_each.py 20 24 > gpurun_out/r05_13_encode_each.json 2> gpurun_out/r05_13_encode_each.err; echo "rc=$?"; tail -2 gpurun_out/r05_13_encode_each.err; cat gpurun_out/r05_13_encode_each.json
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hash_sum or groupby or group_by" > gpurun_out/r05_13_gb.log 2>&1; tail -2 gpurun_out/r05_13_gb.log
timeout 300 python scripts/bench_gb_mid.py > gpurun_out/r05_13_gb_mid.json 2> gpurun_out/r05_13_gb_mid.err; echo "rc=$?"; tail -2 gpurun_out/r05_13_gb_mid.err; cat gpurun_out/r05_13_gb_mid.json | cut -c1-1500
