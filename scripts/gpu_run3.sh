cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -5 gpurun_out/pytest_gpu.log
timeout 300 python scripts/sweep_stream.py > gpurun_out/sweep1.txt 2>&1; cat gpurun_out/sweep1.txt
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench2.json 2> gpurun_out/bench2.err; python -c "
import json; d=json.load(open('gpurun_out/bench2.json')); print(d['value'], d['roofline']['achieved']); [print(k, v) for k,v in d['kernels'].items()]"
