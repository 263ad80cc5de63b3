cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py -m gpu -x -q -k "hash_encode or fixed_width or c5_hash_2_26" > gpurun_out/r3c9_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c9_pytest.log
tail -6 gpurun_out/r3c9_pytest.log
timeout 900 python scripts/bench_encode_part.py 22 23 24 25 > gpurun_out/r3c9_encode_part.json 2> gpurun_out/r3c9_encode_part.err; echo "rc=$?"; tail -3 gpurun_out/r3c9_encode_part.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c9_encode_part.json'))
for k,v in d['results'].items(): print(k, v)
PY
