cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hash_encode" > gpurun_out/r3c6_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c6_pytest.log
tail -12 gpurun_out/r3c6_pytest.log
timeout 900 python scripts/bench_encode_part.py 20 22 23 24 25 > gpurun_out/r3c6_encode_part.json 2> gpurun_out/r3c6_encode_part.err; echo "rc=$?"; tail -3 gpurun_out/r3c6_encode_part.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c6_encode_part.json'))
for k,v in d['results'].items(): print(k, v)
PY
cd /tmp; rm -rf /tmp/prof_e
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_e -o run -- python $R/scripts/bench_encode_part.py 24 > /tmp/prof_e.out 2> /tmp/prof_e.err
python $R/scripts/rocpd_summary.py /tmp/prof_e/run_results.db > $R/gpurun_out/r3c6_encode_24_kernel_stats.csv; head -24 $R/gpurun_out/r3c6_encode_24_kernel_stats.csv | cut -c1-150,400-470
