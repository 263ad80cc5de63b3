cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python scripts/bench_extra.py > gpurun_out/bench_extra.json 2> gpurun_out/bench_extra.err; echo "rc=$?"; cat gpurun_out/bench_extra.json; tail -5 gpurun_out/bench_extra.err
