# round 3, GPU call 5: the new full-size / C-driver tests first, then the whole GPU suite
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_cabi_driver.py tests/test_full_size.py -m gpu -x -q -k "cabi or c4_fused or general_doubles or clustered_vec or c5_hash" > gpurun_out/r3c5_pytest_new.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c5_pytest_new.log
tail -25 gpurun_out/r3c5_pytest_new.log
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r3c5_pytest_all.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c5_pytest_all.log
tail -15 gpurun_out/r3c5_pytest_all.log
