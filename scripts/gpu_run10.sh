cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python scripts/cpu_baseline.py > gpurun_out/cpu_baseline.json 2> gpurun_out/cpu_baseline.err; cat gpurun_out/cpu_baseline.json; tail -3 gpurun_out/cpu_baseline.err
