cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_distributed_gpu.py -m gpu -q -x > gpurun_out/r04_04_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_04_pytest.log
tail -15 gpurun_out/r04_04_pytest.log
bash scripts/gpu_prof_cmd.sh r04_void scripts/void_attempt_trace.py
cat gpurun_out/prof_r04_void.out | tail -2
