cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_chunked.py -m gpu -q -x > gpurun_out/pytest_chunked.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_chunked.log
tail -40 gpurun_out/pytest_chunked.log
