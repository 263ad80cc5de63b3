cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_expressions.py -m gpu -q > gpurun_out/r04_05_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_05_pytest.log
tail -60 gpurun_out/r04_05_pytest.log | cut -c1-300
