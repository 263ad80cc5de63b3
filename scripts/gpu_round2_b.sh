cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "take" > gpurun_out/pytest_take.log 2>&1; tail -3 gpurun_out/pytest_take.log
timeout 300 python scripts/bench_take.py > gpurun_out/bench_take.json 2> gpurun_out/bench_take.err; cat gpurun_out/bench_take.json; tail -3 gpurun_out/bench_take.err
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_take -o run -- python $R/scripts/bench_take.py --quick > $R/gpurun_out/prof_take.out 2> $R/gpurun_out/prof_take.err
cd $R
python scripts/rocpd_summary.py gpurun_out/prof_take/run_results.db > gpurun_out/prof_take_kernels.csv
head -10 gpurun_out/prof_take_kernels.csv | cut -c1-100,250-330
