cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python scripts/bench_types.py 2>&1 | tail -12
