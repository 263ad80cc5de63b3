cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hash_encode_partitioned" > gpurun_out/r3c18_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c18_pytest.log
tail -4 gpurun_out/r3c18_pytest.log
timeout 900 python scripts/bench_encode_unperm.py > gpurun_out/r3c18_encode_unperm.json 2> gpurun_out/r3c18_encode_unperm.err; echo "rc=$?"; tail -3 gpurun_out/r3c18_encode_unperm.err
python -c "
import json;d=json.load(open('gpurun_out/r3c18_encode_unperm.json'))
for k,v in d['results'].items(): print(k,v)"
