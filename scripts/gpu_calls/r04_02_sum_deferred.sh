cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_sum_nonfinite.py tests/test_ingest.py tests/test_distributed_gpu.py "tests/test_gpu_parity.py" -m gpu -q -x -k "sum or ingest or ranks or fused or cmp_filter" > gpurun_out/r04_02_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_02_pytest.log
tail -5 gpurun_out/r04_02_pytest.log
timeout 600 python -m pytest tests/test_full_size.py -m gpu -q -x -k "nonfinite or c2_sum or fused" > gpurun_out/r04_02_pytest_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_02_pytest_full.log
tail -4 gpurun_out/r04_02_pytest_full.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04_02_bench.json 2> gpurun_out/r04_02_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_02_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
k=d.get('kernels',{})
for n in ('sum_float64','sum_int64','add_int64'):
    print(n, k.get(n))
print(d.get('c4_filter_aggregate'))
PY
