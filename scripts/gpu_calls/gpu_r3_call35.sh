cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
ARROWHIP_ENCODE_UNPERM2_GROUP=4 timeout 120 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hash_encode_partitioned and (11 or 13)" 2>&1 | tail -3 | tee gpurun_out/r3c35_pytest_small.log
timeout 60 python scripts/bench_encode_unperm2.py 2>&1 | tail -1 | tee gpurun_out/r3c35_unperm2.json
ARROWHIP_ENCODE_UNPERM2_GROUP=4 timeout 100 python -m pytest tests/test_full_size.py -m gpu -q -x -k "c5_hash_2_26" 2>&1 | tail -2 | tee gpurun_out/r3c35_pytest_full.log
