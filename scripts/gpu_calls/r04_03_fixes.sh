cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_distributed_gpu.py tests/test_gpu_parity.py tests/test_cabi_driver.py -m gpu -q -x -k "ranks or world1 or shared_stream or hash_encode or graph or filter or cabi" > gpurun_out/r04_03_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_03_pytest.log
tail -12 gpurun_out/r04_03_pytest.log
# does the voided-attempt test really void an attempt?  the kernel trace of its first case must show the partition pass AND the global insert after it
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_void -o void -- python -m pytest $GRAFT_REPO_ROOT/tests/test_gpu_parity.py -m gpu -q -x -k voided_partition > /tmp/void.log 2>&1; tail -3 /tmp/void.log
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_void/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
for r in rows:
    n = r['Name']
    if any(k in n for k in ('enc_table', 'enc_unpermute', 'insert_kernel', 'emit_kernel', 'gb_scatter', 'enc_resolve')):
        print(n[:90], r['Calls'], r['TotalDurationNs'])
PY
