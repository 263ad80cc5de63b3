cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_cabi_driver.py tests/test_ingest.py tests/test_full_size.py -m gpu -x -q -k "filter or cabi or ingest" > gpurun_out/r3c17_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c17_pytest.log
tail -4 gpurun_out/r3c17_pytest.log
gcc -O2 -std=c11 -pthread -Iinclude tests/cabi_driver.c -Larrow_go_amd -larrowhip -Wl,-rpath,$R/arrow_go_amd -o /tmp/cabi_driver && /tmp/cabi_driver bench 27 > gpurun_out/r3c17_filter_from_c.json; cat gpurun_out/r3c17_filter_from_c.json
ARROWHIP_NT=0 /tmp/cabi_driver bench 27 | head -c 600; echo
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3c17_bench.json 2> gpurun_out/r3c17_bench.err; echo "bench rc=$?"
python -c "
import json;d=json.load(open('gpurun_out/r3c17_bench.json'));print(d['value'],d['roofline']['frac'])
for k,v in d['kernels'].items():
    if isinstance(v,dict) and ('filter' in k or 'take' in k): print(k, v)"
