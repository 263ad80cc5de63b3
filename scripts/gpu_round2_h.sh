cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 120 scripts/micro/lds_atomics64.bin > gpurun_out/lds_atomics64.txt 2>&1; cat gpurun_out/lds_atomics64.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "hash" > gpurun_out/pytest_hash.log 2>&1; tail -5 gpurun_out/pytest_hash.log
timeout 600 python scripts/bench_groupby.py > gpurun_out/bench_groupby.json 2> gpurun_out/bench_groupby.err; tail -3 gpurun_out/bench_groupby.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_groupby.json'))
for k,v in d['results'].items(): print(k, v)
PY
cd /tmp
for c in 16 20 16z 20z; do
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_gb_$c -o run -- python $R/scripts/bench_groupby.py --only $c > $R/gpurun_out/prof_gb_$c.out 2> $R/gpurun_out/prof_gb_$c.err
python $R/scripts/rocpd_summary.py $R/gpurun_out/prof_gb_$c/run_results.db > $R/gpurun_out/prof_gb_${c}_kernels.csv 2>>$R/gpurun_out/prof_gb_sum.err
echo "== $c"; grep "gb_" $R/gpurun_out/prof_gb_${c}_kernels.csv | sed 's/(anonymous namespace):://g' | awk -F'",' '{print substr($1,1,45), $2}' | head -12
rm -rf $R/gpurun_out/prof_gb_$c
done
