cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "hash_sum" > gpurun_out/pytest_hash.log 2>&1; tail -12 gpurun_out/pytest_hash.log
timeout 600 python scripts/bench_groupby.py > gpurun_out/bench_groupby.json 2> gpurun_out/bench_groupby.err; tail -3 gpurun_out/bench_groupby.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_groupby.json'))
for k,v in d['results'].items(): print(k, {a:b for a,b in v.items() if 'encode' not in a and 'i64' not in a})
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_gb_24 -o run -- python $R/scripts/bench_groupby.py --only 24 > $R/gpurun_out/prof_gb_24.out 2> $R/gpurun_out/prof_gb_24.err
python $R/scripts/rocpd_summary.py $R/gpurun_out/prof_gb_24/run_results.db > $R/gpurun_out/prof_gb_24_kernels.csv
grep "gs_\|ms_offs2\|gb_max\|word_prefix\|scan_kernel" $R/gpurun_out/prof_gb_24_kernels.csv | sed 's/(anonymous namespace):://g' | awk -F'",' '{print substr($1,1,45), $2}' | head -12
rm -rf $R/gpurun_out/prof_gb_24
