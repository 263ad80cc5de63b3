cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_compute_api.py tests/test_golden.py -m gpu -q -x -k "binary or unique or dictionary" > gpurun_out/pytest_hash.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_hash.log
tail -25 gpurun_out/pytest_hash.log
