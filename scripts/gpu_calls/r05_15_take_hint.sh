# round 5, call 15: Take — one prep launch, the path sample kept per index vector, the clustered kernel counts its own valid rows, one
# finishing launch that adds up and posts: all take tests, then bench.py's kernel lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -k "take or Take or selection or record_batch" > gpurun_out/r05_15_take.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_15_take.log
tail -5 gpurun_out/r05_15_take.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r05_15_bench.json 2> gpurun_out/r05_15_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r05_15_bench.err
python -c "
import json;d=json.load(open('gpurun_out/r05_15_bench.json'));print(d['value'],d['roofline']['frac'])
for k,v in d['kernels'].items():
    if isinstance(v,dict) and ('take' in k or 'ms_each' in v): print(k, v.get('ms'), v.get('GB/s'), v.get('ms_each'))"
