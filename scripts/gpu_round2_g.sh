cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "hash" > gpurun_out/pytest_hash.log 2>&1; tail -15 gpurun_out/pytest_hash.log
timeout 600 python scripts/bench_groupby.py > gpurun_out/bench_groupby.json 2> gpurun_out/bench_groupby.err; tail -3 gpurun_out/bench_groupby.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_groupby.json'))
for k,v in d['results'].items(): print(k, v)
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_gb -o run -- python $R/scripts/bench_groupby.py --quick > $R/gpurun_out/prof_gb.out 2> $R/gpurun_out/prof_gb.err
cd $R
python scripts/rocpd_summary.py gpurun_out/prof_gb/run_results.db > gpurun_out/prof_gb_kernels.csv 2>gpurun_out/prof_gb_sum.err
head -40 gpurun_out/prof_gb_kernels.csv | cut -c1-60,200-330
