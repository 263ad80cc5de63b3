cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r03_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_pytest_gpu.log
tail -4 gpurun_out/r03_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
