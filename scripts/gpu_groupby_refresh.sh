# after a change to the group-by paths: its parity tests, the sweep, the bench line, the per-workload PMC rows and kernel stats
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py -q -x -k "hash" > gpurun_out/pytest_hash.log 2>&1; tail -3 gpurun_out/pytest_hash.log
timeout 600 python scripts/bench_groupby.py > gpurun_out/bench_groupby.json 2> gpurun_out/bench_groupby.err; tail -2 gpurun_out/bench_groupby.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_groupby.json'))
for k,v in d['results'].items(): print(k, v.get('f64_auto_ms'), v.get('i64_auto_ms'))
PY
timeout 600 python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_line.err; tail -c 300 gpurun_out/bench_line.json
bash scripts/gpu_prof_workloads.sh r02 2>&1 | grep "hash_sum\|add_int64"
cd /tmp
for lg in 16 20 24; do
rm -rf /tmp/prof_g; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o run -- python $R/scripts/bench_groupby.py --only $lg > /tmp/prof_g.out 2> /tmp/prof_g.err
python $R/scripts/rocpd_summary.py /tmp/prof_g/run_results.db > $R/gpurun_out/groupby_${lg}_kernel_stats.csv
done
rm -rf $R/gpurun_out/pw_kt $R/gpurun_out/pw_fetch $R/gpurun_out/pw_write
