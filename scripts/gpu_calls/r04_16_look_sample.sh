# (look and sample as two sets of workgroups of one launch: passed everything, measured slower — DESIGN.md §3.2 — and was reverted)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q > gpurun_out/r04_16_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_16_pytest.log
tail -12 gpurun_out/r04_16_pytest.log | cut -c1-250
timeout 600 python -m pytest tests/test_full_size.py -m gpu -q -x -k "hash or c5" > gpurun_out/r04_16_pytest_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04_16_pytest_full.log
tail -4 gpurun_out/r04_16_pytest_full.log | cut -c1-250
timeout 300 python scripts/bench_gb_small.py | tee gpurun_out/r04_16_gb_small.json
timeout 600 python scripts/bench_gb_mid.py | tee gpurun_out/r04_16_gb_mid.json
