cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 600 python bench.py > gpurun_out/bench_r02a.json 2> gpurun_out/bench_r02a.err ) 2>&1 | grep real
tail -3 gpurun_out/bench_r02a.err; python -c "
import json; d=json.load(open('gpurun_out/bench_r02a.json'))
print({k:d[k] for k in ('value','ms_per_step','roofline','cpu_baseline','c4_filter_aggregate','c5_group_by')})
for k,v in d['kernels'].items(): print(k, v)
"
bash scripts/gpu_prof_workloads.sh r02a
