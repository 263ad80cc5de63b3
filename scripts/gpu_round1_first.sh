cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(nproc; lscpu | grep "Model name"; rocminfo | grep -E "Marketing|gfx" | head -4) > gpurun_out/box.txt 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 3 > gpurun_out/bench1.json 2> gpurun_out/bench1.err; echo "bench rc=$?" >> gpurun_out/bench1.err
tail -5 gpurun_out/smoke.log; tail -15 gpurun_out/pytest_gpu.log; cat gpurun_out/bench1.json; tail -5 gpurun_out/bench1.err
