# round 5, call 6: the whole GPU suite + smoke on the current tree (regressions caught before the round's last calls)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r05_06_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_06_pytest_gpu.log
tail -8 gpurun_out/r05_06_pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
