cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "take" > gpurun_out/pytest_take.log 2>&1; tail -4 gpurun_out/pytest_take.log
timeout 900 python -m pytest tests/test_full_size.py -q -x -k "take" > gpurun_out/pytest_take_full.log 2>&1; tail -4 gpurun_out/pytest_take_full.log
timeout 300 python scripts/bench_take.py --quick > gpurun_out/bench_take.json 2> gpurun_out/bench_take.err; tail -2 gpurun_out/bench_take.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_take.json'))
for k,v in d.items():
    if 'auto' in k or 'direct' in k: print(k, v)
PY
