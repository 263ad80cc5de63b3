# round 5, call 18: the committed bench.py once more (warm-up counts changed after the round-end run): the line that goes to profiles/
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err; echo "bench rc=$?"; tail -2 gpurun_out/r05_bench_line.err
python -c "
import json;d=json.load(open('gpurun_out/r05_bench_line.json'));print(d['value'],d['roofline']['frac'], d['c5_group_by']['ms_per_step'])
for k,v in d['kernels'].items():
    if isinstance(v,dict) and ('ms_each' in v or 'encode' in k or 'filter_int64_nulls10_sel0.50' == k): print(k, v.get('ms'), v.get('ms_each'))"
