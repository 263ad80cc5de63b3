cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
nproc > gpurun_out/box.txt; grep -m1 "model name" /proc/cpuinfo >> gpurun_out/box.txt
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_full_size.py > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 300 python scripts/bench_take.py > gpurun_out/bench_take.json 2> gpurun_out/bench_take.err; echo "bench_take rc=$?"
cat gpurun_out/bench_take.json; tail -3 gpurun_out/bench_take.err
timeout 1200 python -m pytest tests/test_full_size.py -q --durations=0 > gpurun_out/pytest_full.log 2>&1; echo "pytest full rc=$?" >> gpurun_out/pytest_full.log
tail -40 gpurun_out/pytest_full.log
