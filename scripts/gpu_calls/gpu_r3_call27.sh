cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_golden.py -m gpu -q -x -k "take" > gpurun_out/r3c27_pytest.log 2>&1; tail -3 gpurun_out/r3c27_pytest.log
timeout 600 python scripts/bench_take_clustered.py > gpurun_out/r3c27_take_clustered.json 2> gpurun_out/r3c27_take_clustered.err; echo "rc=$?"
python -c "
import json;d=json.load(open('gpurun_out/r3c27_take_clustered.json'))
for k,v in d.items():
    if 'nulls' in k: print(k,v)"
