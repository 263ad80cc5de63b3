cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "graph or filter_count_cache or mailbox" > gpurun_out/r3c14_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c14_pytest.log
tail -15 gpurun_out/r3c14_pytest.log
timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r3c14_bench.json 2> gpurun_out/r3c14_bench.err; echo "bench rc=$?"
python -c "
import json;d=json.load(open('gpurun_out/r3c14_bench.json'));print(d['value'],d['roofline']['frac']);print(d.get('graph_replay'))"
tail -5 gpurun_out/r3c14_bench.err
