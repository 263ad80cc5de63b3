cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hash" > gpurun_out/pytest_hash.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_hash.log
tail -5 gpurun_out/pytest_hash.log
timeout 300 python scripts/bench_hash.py 16 20 21 22 24 > gpurun_out/bench_hash2.json 2> gpurun_out/bench_hash2.err; cat gpurun_out/bench_hash2.json; tail -3 gpurun_out/bench_hash2.err
bash scripts/gpu_prof_cmd.sh hash24 scripts/bench_hash.py 24
