cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_golden.py -m gpu -x -q -k "hash or unique or dictionary or c5" > gpurun_out/r3c22_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c22_pytest.log
tail -4 gpurun_out/r3c22_pytest.log
timeout 900 python scripts/bench_encode_part.py 10 16 18 19 20 21 22 23 24 25 > gpurun_out/r3c22_encode_part.json 2> gpurun_out/r3c22_encode_part.err; echo "rc=$?"; tail -3 gpurun_out/r3c22_encode_part.err
python -c "
import json;d=json.load(open('gpurun_out/r3c22_encode_part.json'))
for k,v in d['results'].items(): print(k,{a:b for a,b in v.items() if 'auto' in a or 'global' in a})"
cd /tmp; rm -rf /tmp/st; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st -o run -- python $R/scripts/bench_encode_one.py 20 > /dev/null 2>&1
python $R/scripts/rocpd_summary.py /tmp/st/run_results.db > $R/gpurun_out/r3c22_encode_20_kernel_stats.csv
rm -rf /tmp/st; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st -o run -- python $R/scripts/bench_encode_one.py 24 > /dev/null 2>&1
python $R/scripts/rocpd_summary.py /tmp/st/run_results.db > $R/gpurun_out/r3c22_encode_24_kernel_stats.csv
