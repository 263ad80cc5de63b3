cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04_11_bench.json 2> gpurun_out/r04_11_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_11_bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'])
for k,v in d.get('kernels',{}).items():
    if 'hash_sum' in k or 'dictionary' in k or 'sum_' in k: print(k, v)
print(d.get('c5_group_by'))
PY
