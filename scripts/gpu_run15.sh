cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 300 python scripts/bench_hash.py > gpurun_out/bench_hash.json 2> gpurun_out/bench_hash.err; cat gpurun_out/bench_hash.json
bash scripts/gpu_prof_cmd.sh hash24 scripts/bench_hash.py 24
bash scripts/gpu_prof_cmd.sh hash16 scripts/bench_hash.py 16
