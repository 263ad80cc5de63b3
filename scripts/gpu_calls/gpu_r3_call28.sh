cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python scripts/bench_xcd_map.py > gpurun_out/r3c28_xcd_map.json 2> gpurun_out/r3c28_xcd_map.err; echo "rc=$?"; tail -3 gpurun_out/r3c28_xcd_map.err
python -c "
import json;d=json.load(open('gpurun_out/r3c28_xcd_map.json'))
for k,v in d.items(): print(k,v)"
ARROWHIP_ARITH_XCD_MAP=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "arith or add or graph" > gpurun_out/r3c28_pytest.log 2>&1; tail -3 gpurun_out/r3c28_pytest.log
