# (record of the call that ended the round's GPU budget: the survey variant of the group-by — scripts/micro/gq_survey_variant.hip.txt —
#  hung in gq_global_slot on a column of 4096 keys; pytest and the mid-range bench ran into their timeouts, 34 GPU-minutes.  The
#  timeouts below were far too generous for a first run of a new kernel: use ≤ 120 s for the first test of anything that loops.)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "hash_sum or groupby or group" > gpurun_out/r04_17_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_17_pytest.log
tail -6 gpurun_out/r04_17_pytest.log | cut -c1-250
timeout 300 python scripts/bench_gb_small.py | tee gpurun_out/r04_17_gb_small.json
timeout 600 python scripts/bench_gb_mid.py | tee gpurun_out/r04_17_gb_mid.json
timeout 600 python -m pytest tests/test_full_size.py -m gpu -q -x -k "hash or c5" > gpurun_out/r04_17_pytest_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_17_pytest_full.log
tail -4 gpurun_out/r04_17_pytest_full.log | cut -c1-250
