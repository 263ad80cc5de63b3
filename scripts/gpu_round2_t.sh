cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py -q -x -k "sort" > gpurun_out/pytest_t.log 2>&1; tail -4 gpurun_out/pytest_t.log
for k in int normal; do ARROWHIP_DEBUG_MSD=1 python scripts/bench_sort_one.py 27 $k 1 2>&1 | tail -2; done
python scripts/bench_sort_one.py 24 int 1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o run -- python $R/scripts/bench_sort_one.py 27 int 1 > /tmp/prof_s.out 2> /tmp/prof_s.err
python $R/scripts/rocpd_summary.py /tmp/prof_s/run_results.db > $R/gpurun_out/prof_sort_kernels.csv
sed 's/(anonymous namespace):://g' $R/gpurun_out/prof_sort_kernels.csv | awk -F'",' '{print substr($1,1,50), $2}' | head -9
