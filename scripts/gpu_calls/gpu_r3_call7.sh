cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py -m gpu -x -q -k "hash_encode or c5_hash" > gpurun_out/r3c7_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c7_pytest.log
tail -6 gpurun_out/r3c7_pytest.log
timeout 900 python scripts/bench_encode_part.py 10 16 18 19 20 21 22 23 24 25 > gpurun_out/r3c7_encode_part.json 2> gpurun_out/r3c7_encode_part.err; echo "rc=$?"; tail -3 gpurun_out/r3c7_encode_part.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c7_encode_part.json'))
for k,v in d['results'].items(): print(k, {a:b for a,b in v.items() if 'global' in a or 'auto' in a})
PY
bash scripts/gpu_prof_workloads.sh r03 2>&1 | tail -40
cd $R
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r3c7_bench.json 2> gpurun_out/r3c7_bench.err; echo "bench rc=$?"; tail -2 gpurun_out/r3c7_bench.err
python -c "
import json;d=json.load(open('gpurun_out/r3c7_bench.json'));print(d['value'],d['roofline'])"
