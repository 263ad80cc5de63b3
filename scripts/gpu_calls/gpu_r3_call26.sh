cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py -m gpu -x -q -k "hash_sum or c5" > gpurun_out/r3c26_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c26_pytest.log
tail -12 gpurun_out/r3c26_pytest.log
timeout 600 python scripts/bench_gb_guess.py > gpurun_out/r3c26_gb_guess.json 2> gpurun_out/r3c26_gb_guess.err; echo "rc=$?"; tail -3 gpurun_out/r3c26_gb_guess.err
python -c "
import json;d=json.load(open('gpurun_out/r3c26_gb_guess.json'))
for k,v in d.items(): print(k,v)"
