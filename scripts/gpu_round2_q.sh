cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_abi.py -q -x -k "hash or dev_flavour or abi or filter or take" > gpurun_out/pytest_q.log 2>&1; tail -5 gpurun_out/pytest_q.log
timeout 300 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_q.json 2> gpurun_out/bench_q.err; tail -2 gpurun_out/bench_q.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_q.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline'])
for k,v in d['kernels'].items():
    if 'hash' in k or 'dict' in k: print(k, v)
PY
