cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hash_encode or hash_fixed" > gpurun_out/r3c20_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c20_pytest.log
tail -4 gpurun_out/r3c20_pytest.log
timeout 900 python scripts/bench_encode_unperm.py 20 22 24 > gpurun_out/r3c20_encode.json 2> gpurun_out/r3c20_encode.err; echo "rc=$?"; tail -3 gpurun_out/r3c20_encode.err
python -c "
import json;d=json.load(open('gpurun_out/r3c20_encode.json'))
for k,v in d['results'].items(): print(k,v)"
cd /tmp; rm -rf /tmp/st; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st -o run -- python $R/scripts/bench_encode_one.py 20 > /dev/null 2>&1
python $R/scripts/rocpd_summary.py /tmp/st/run_results.db > $R/gpurun_out/r3c20_encode_20_kernel_stats.csv
rm -rf /tmp/st; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st -o run -- python $R/scripts/bench_encode_one.py 24 > /dev/null 2>&1
python $R/scripts/rocpd_summary.py /tmp/st/run_results.db > $R/gpurun_out/r3c20_encode_24_kernel_stats.csv
