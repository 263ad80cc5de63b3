cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py -q -x -k "hash" > gpurun_out/pytest_v.log 2>&1; tail -4 gpurun_out/pytest_v.log
timeout 600 python scripts/bench_nullkeys.py > gpurun_out/bench_nullkeys.json 2> gpurun_out/bench_nullkeys.err; tail -3 gpurun_out/bench_nullkeys.err
timeout 600 python scripts/bench_groupby.py --quick > gpurun_out/bench_groupby_r.json 2> gpurun_out/bench_groupby_r.err; tail -3 gpurun_out/bench_groupby_r.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_groupby_r.json'))
for k,v in d['results'].items(): print(k, {a:b for a,b in v.items() if 'encode' not in a})
PY
