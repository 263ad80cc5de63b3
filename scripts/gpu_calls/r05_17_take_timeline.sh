# round 5, call 17: kernel trace of twelve identity Takes with nulls (where does the time between the main kernels go?)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp; rm -rf /tmp/prof_t
timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_t -o run -- python $R/scripts/bench_take_timeline.py > /tmp/prof_t.out 2> /tmp/prof_t.err; tail -2 /tmp/prof_t.err
python $R/scripts/kernel_timeline.py /tmp/prof_t/run_results.db 24 > $R/gpurun_out/r05_17_take_timeline.txt; cat $R/gpurun_out/r05_17_take_timeline.txt
