# round 5, call 19: the one-call Filter with its tail reduced to the popcount's two launches (the second posts): every filter test
# cross-checks it against count + fill; bench.py's filter lines (two-phase against one_call_ms)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -x -k "filter or Filter" > gpurun_out/r05_19_filter.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_19_filter.log
tail -3 gpurun_out/r05_19_filter.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r05_19_bench.json 2> gpurun_out/r05_19_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r05_19_bench.err
python -c "
import json;d=json.load(open('gpurun_out/r05_19_bench.json'));print(d['value'],d['roofline']['frac'])
for k,v in d['kernels'].items():
    if isinstance(v,dict) and ('filter' in k or 'ms_each' in v): print(k, v.get('ms'), v.get('fill_only_ms'), v.get('one_call_ms'), v.get('ms_each'))"
timeout 300 python scripts/bench_filter_once.py > gpurun_out/r05_19_filter_once.json 2>/dev/null; cat gpurun_out/r05_19_filter_once.json
