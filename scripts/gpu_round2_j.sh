cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "hash" > gpurun_out/pytest_hash.log 2>&1; tail -5 gpurun_out/pytest_hash.log
timeout 600 python scripts/bench_groupby.py > gpurun_out/bench_groupby.json 2> gpurun_out/bench_groupby.err; tail -3 gpurun_out/bench_groupby.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_groupby.json'))
for k,v in d['results'].items(): print(k, {a:b for a,b in v.items() if 'encode' not in a})
PY
cd /tmp
for c in 16 20 20z; do
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_gb_$c -o run -- python $R/scripts/bench_groupby.py --only $c > $R/gpurun_out/prof_gb_$c.out 2> $R/gpurun_out/prof_gb_$c.err
python $R/scripts/rocpd_summary.py $R/gpurun_out/prof_gb_$c/run_results.db > $R/gpurun_out/prof_gb_${c}_kernels.csv 2>>$R/gpurun_out/prof_gb_sum.err
echo "== $c"; grep "gb_" $R/gpurun_out/prof_gb_${c}_kernels.csv | sed 's/(anonymous namespace):://g' | awk -F'",' '{print substr($1,1,45), $2}' | head -5
rm -rf $R/gpurun_out/prof_gb_$c
done
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/pmc_gb -o run -- python $R/scripts/bench_groupby.py --only 16 > $R/gpurun_out/pmc_gb.out 2> $R/gpurun_out/pmc_gb.err
python $R/scripts/pmc_sq.py $R/gpurun_out/pmc_gb/run_results.db gb_aggregate | tee $R/gpurun_out/pmc_gb_sq.txt
rm -rf $R/gpurun_out/pmc_gb
