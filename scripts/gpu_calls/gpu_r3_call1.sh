# round 3, first GPU call: parity of what changed (filter count cache, vec take, range-aware group sums, copy kernel), then the bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -k "filter or take or hash_sum or sort" > gpurun_out/r3c1_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c1_pytest.log
tail -15 gpurun_out/r3c1_pytest.log
timeout 600 python bench.py --steps 20 --warmup 3 > gpurun_out/r3c1_bench.json 2> gpurun_out/r3c1_bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r3c1_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c1_bench.json'))
print(d['value'], d['roofline'])
k=d.get('kernels',{})
for n,v in k.items():
    if isinstance(v,dict) and any(s in n for s in ('filter','take','ceiling','hash_sum','dictionary','add_int64','sum_float64')):
        print(n, v)
print(d.get('cpu_baseline_mt'))
PY
