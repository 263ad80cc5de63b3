# round 5, call 5: aggregate pass with the LDS slot lookup spelled out in place, 256-word rank tiles — hashing / group-by parity, timings, timeline
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -x -k "hash or group or unique or dictionary" > gpurun_out/r05_05_hash.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_05_hash.log
tail -4 gpurun_out/r05_05_hash.log | cut -c1-300
timeout 300 python scripts/bench_gb_mid.py | tee gpurun_out/r05_05_gb_mid.json
timeout 300 python scripts/bench_gb_small.py | tee gpurun_out/r05_05_gb_small.json
timeout 300 python scripts/bench_encode_part.py 20 24 | tee gpurun_out/r05_05_encode_part.json
bash scripts/gpu_prof_cmd.sh r05_gb_mid3 scripts/bench_gb_mid.py > /dev/null
