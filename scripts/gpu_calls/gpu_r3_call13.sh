cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q -x > gpurun_out/r3c13_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c13_pytest_gpu.log
tail -6 gpurun_out/r3c13_pytest_gpu.log
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3c13_bench.json 2> gpurun_out/r3c13_bench.err; echo "bench rc=$?"
python -c "
import json;d=json.load(open('gpurun_out/r3c13_bench.json'));print(d['value'],d['roofline']['frac']);print({k:v.get('ms') for k,v in d['kernels'].items() if isinstance(v,dict) and any(s in k for s in ('dictionary','hash_sum','take_int64_id','take_int64_rand','filter_count','count_set','sel0.50'))})"
