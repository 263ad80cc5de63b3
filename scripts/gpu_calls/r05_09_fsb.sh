# round 5, call 9: FixedSizeBinary / Decimal keys through the registry; the rank-failure case of ah_comm_merge_groups on 2 and 3 ranks
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_compute_api.py tests/test_distributed_gpu.py tests/test_expressions.py -m gpu -q -x > gpurun_out/r05_09_api.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_09_api.log
tail -15 gpurun_out/r05_09_api.log | cut -c1-400
