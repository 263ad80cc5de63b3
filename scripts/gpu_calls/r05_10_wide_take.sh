# round 5, call 10: 16- / 32-byte slots through ah_take_primitive (C ABI against the oracle) and through the registry's take / filter
# for FixedSizeBinary / Decimal128 / Decimal256 (against Arrow C++); the distributed tests after bench.py's C5 key change; the C5 lines
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -k "take" tests/test_compute_api.py tests/test_distributed_gpu.py -m gpu -q -x > gpurun_out/r05_10_take.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_10_take.log
tail -12 gpurun_out/r05_10_take.log | cut -c1-400
timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r05_10_bench.json 2> gpurun_out/r05_10_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r05_10_bench.err
python -c "
import json;d=json.load(open('gpurun_out/r05_10_bench.json'));print(d['value'],d['roofline']);print(d.get('c5_group_by'))"
