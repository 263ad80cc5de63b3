cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python scripts/bench_take_nt.py > gpurun_out/r3c15_take_nt.json 2> gpurun_out/r3c15_take_nt.err; echo "rc=$?"
python -c "
import json;d=json.load(open('gpurun_out/r3c15_take_nt.json'))
for k,v in d.items(): print(k,v)"
tail -3 gpurun_out/r3c15_take_nt.err
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "take" > gpurun_out/r3c15_pytest.log 2>&1; tail -3 gpurun_out/r3c15_pytest.log
