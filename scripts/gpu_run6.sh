cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log; tail -4 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 20 --warmup 3 > gpurun_out/bench3.json 2> gpurun_out/bench3.err; python -c "
import json; d=json.load(open('gpurun_out/bench3.json')); print(d['value'], d['roofline']); [print(k, v) for k,v in d['kernels'].items()]; print(d['cpu_baseline'])"
timeout 600 python scripts/bench_extra.py 2>gpurun_out/bench_extra.err | python -c "
import json,sys; d=json.load(sys.stdin); [print(k, v) for k,v in d.items() if 'hash' in k or 'dict' in k]"
