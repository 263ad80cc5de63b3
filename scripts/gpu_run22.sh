cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_expressions.py -m gpu -q -x > gpurun_out/pytest_expr.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_expr.log
tail -30 gpurun_out/pytest_expr.log
