cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_golden.py tests/test_cabi_driver.py -m gpu -q -x -k "take or cabi" > gpurun_out/r3c34_pytest.log 2>&1; tail -3 gpurun_out/r3c34_pytest.log
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_line.err; echo "bench rc=$?"
python -c "
import json;d=json.load(open('gpurun_out/r03_bench_line.json'));print(d['value'],d['roofline']['frac']);print({k:v.get('ms') for k,v in d['kernels'].items() if isinstance(v,dict) and ('take' in k or 'filter' in k)})"
