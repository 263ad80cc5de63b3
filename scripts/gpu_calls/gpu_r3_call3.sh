# round 3, GPU call 3: partition-first encode parity + timings; take v3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -k "hash or take or unique or dictionary" > gpurun_out/r3c3_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c3_pytest.log
tail -12 gpurun_out/r3c3_pytest.log
timeout 600 python scripts/bench_encode_part.py > gpurun_out/r3c3_encode_part.json 2> gpurun_out/r3c3_encode_part.err; echo "rc=$?"; tail -3 gpurun_out/r3c3_encode_part.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c3_encode_part.json'))
for k,v in d['results'].items(): print(k, v)
PY
timeout 300 python scripts/bench_take_clustered.py > gpurun_out/r3c3_take.json 2> gpurun_out/r3c3_take.err; python -c "import json;[print(k,v) for k,v in json.load(open(\"gpurun_out/r3c3_take.json\")).items()]"
cd /tmp; rm -rf /tmp/prof_e
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_e -o run -- python $R/scripts/bench_encode_part.py 20 > /tmp/prof_e.out 2> /tmp/prof_e.err
python $R/scripts/rocpd_summary.py /tmp/prof_e/run_results.db > $R/gpurun_out/r3c3_encode_20_kernel_stats.csv; head -30 $R/gpurun_out/r3c3_encode_20_kernel_stats.csv | cut -c1-200
