cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -40 gpurun_out/pytest_gpu.log
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof_bench.err
cd $R; ls -R gpurun_out/prof_bench | head -20
