# round 5, call 16: the clustered Take's partial counts spread over 256-byte blocks; take tests (vec path, hint cache) + bench take lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "take" > gpurun_out/r05_16_take.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_16_take.log
tail -3 gpurun_out/r05_16_take.log | cut -c1-300
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r05_16_bench.json 2> gpurun_out/r05_16_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r05_16_bench.err
python -c "
import json;d=json.load(open('gpurun_out/r05_16_bench.json'));print(d['value'],d['roofline']['frac'])
for k,v in d['kernels'].items():
    if isinstance(v,dict) and ('take' in k or 'ms_each' in v): print(k, v.get('ms'), v.get('GB/s'), v.get('ms_each'))"
