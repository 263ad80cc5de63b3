cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "sort" > gpurun_out/pytest_sort.log 2>&1; tail -4 gpurun_out/pytest_sort.log
for k in int normal; do ARROWHIP_DEBUG_MSD=1 python scripts/bench_sort_one.py 27 $k 1 2>&1 | tail -2; done
python scripts/bench_sort_one.py 24 int 1; python scripts/bench_sort_one.py 24 normal 1; python scripts/bench_sort_one.py 22 int 1; python scripts/bench_sort_one.py 26 normal 1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sort -o run -- python $R/scripts/bench_sort_one.py 27 int 1 > /dev/null 2> $R/gpurun_out/prof_sort.err
python $R/scripts/rocpd_summary.py $R/gpurun_out/prof_sort/run_results.db > $R/gpurun_out/prof_sort_kernels.csv
cat $R/gpurun_out/prof_sort_kernels.csv | sed 's/(anonymous namespace):://g' | awk -F'",' '{print substr($1,1,60), $2}' | cut -c1-130 | head -30
rm -rf $R/gpurun_out/prof_sort
