cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 100 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -x -k "hash_encode or unique or dictionary or hash_fixed" 2>&1 | tail -2 | tee gpurun_out/r3c36_pytest.log
