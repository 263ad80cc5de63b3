cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python scripts/bench_slices.py > gpurun_out/r3c31_slices.json 2> gpurun_out/r3c31_slices.err; echo "rc=$?"; tail -3 gpurun_out/r3c31_slices.err; cat gpurun_out/r3c31_slices.json
