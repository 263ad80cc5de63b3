cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "sort" > gpurun_out/pytest_sort.log 2>&1; tail -12 gpurun_out/pytest_sort.log
timeout 600 python scripts/bench_sort.py > gpurun_out/bench_sort.out 2> gpurun_out/bench_sort.err; tail -3 gpurun_out/bench_sort.err
python - <<'PY'
import json
for k,v in json.load(open('gpurun_out/bench_sort.json')).items(): print(k, v)
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sort -o run -- python $R/scripts/bench_sort.py > /dev/null 2> $R/gpurun_out/prof_sort.err
python $R/scripts/rocpd_summary.py $R/gpurun_out/prof_sort/run_results.db > $R/gpurun_out/prof_sort_kernels.csv
grep "ms_\|colsum\|bin_prefix\|tile_offs" $R/gpurun_out/prof_sort_kernels.csv | sed 's/(anonymous namespace):://g' | awk -F'",' '{print substr($1,1,48), $2}' | head -16
rm -rf $R/gpurun_out/prof_sort
