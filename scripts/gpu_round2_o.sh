cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 600 python bench.py > gpurun_out/bench_r02o.json 2> gpurun_out/bench_r02o.err ) 2>&1 | grep real
tail -3 gpurun_out/bench_r02o.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_r02o.json'))
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['frac'], d['c5_group_by']['ms_per_step'])
for k,v in d['kernels'].items():
    if 'sort' in k or 'take' in k or 'hash' in k or 'dict' in k: print(k, v)
PY
bash scripts/gpu_prof_workloads.sh r02c
