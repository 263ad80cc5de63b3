cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
ARROWHIP_ENCODE_DICT_COMPACT=2 timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -k "hash or unique or dictionary" > gpurun_out/r3c23_pytest_compact2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c23_pytest_compact2.log
tail -4 gpurun_out/r3c23_pytest_compact2.log
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py -m gpu -x -q -k "hash_encode or c5_hash_2_26" > gpurun_out/r3c23_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c23_pytest.log
tail -4 gpurun_out/r3c23_pytest.log
timeout 900 python scripts/bench_encode_dict.py > gpurun_out/r3c23_encode_dict.json 2> gpurun_out/r3c23_encode_dict.err; echo "rc=$?"; tail -3 gpurun_out/r3c23_encode_dict.err
python -c "
import json;d=json.load(open('gpurun_out/r3c23_encode_dict.json'))
for k,v in d['results'].items(): print(k,{a:b for a,b in v.items() if 'round1' in a})"
