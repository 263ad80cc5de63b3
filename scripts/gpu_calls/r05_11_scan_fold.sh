# round 5, call 11: cumulative_sum — output validity + null count written by the one-pass kernel itself, bit fetch by v_readlane,
# left neighbour by wave_shr DPP: parity (all cumulative tests), then scripts/bench_scan.py (fold against option 2 = separate launches)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "cumulative or cumsum or scan" > gpurun_out/r05_11_scan.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_11_scan.log
tail -6 gpurun_out/r05_11_scan.log | cut -c1-400
timeout 300 python scripts/bench_scan.py > gpurun_out/r05_11_bench_scan.json 2> gpurun_out/r05_11_bench_scan.err; echo "bench rc=$?"; tail -2 gpurun_out/r05_11_bench_scan.err
cat gpurun_out/r05_11_bench_scan.json
