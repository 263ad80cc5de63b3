cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -x -k "arithmetic_ext or divide_vectors or shift_vectors or bitwise_vectors or abs_negate or sqrt_vectors" > gpurun_out/pytest_ext.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_ext.log
tail -30 gpurun_out/pytest_ext.log
