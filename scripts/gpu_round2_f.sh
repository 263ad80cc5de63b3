cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_all.log
tail -8 gpurun_out/pytest_gpu_all.log
( time timeout 600 python bench.py > gpurun_out/bench_r02f.json 2> gpurun_out/bench_r02f.err ) 2>&1 | grep real
tail -3 gpurun_out/bench_r02f.err; head -c 1500 gpurun_out/bench_r02f.json
