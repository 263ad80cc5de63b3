# round 5, call 14: bench.py's kernel lines with per-repetition timings kept where they spread (which call made 2^24-key encode read 4.6 ms?)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r05_14_bench.json 2> gpurun_out/r05_14_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r05_14_bench.err
python -c "
import json;d=json.load(open('gpurun_out/r05_14_bench.json'));print(d['value'],d['roofline']['frac'])
for k,v in d['kernels'].items():
    if isinstance(v,dict) and ('ms_each' in v or 'encode' in k or 'hash_sum' in k): print(k, v.get('ms'), v.get('ms_each'))"
