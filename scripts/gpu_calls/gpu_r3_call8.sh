cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_cabi_driver.py -m gpu -x -q -k "hash_sum or hash_encode or cabi" > gpurun_out/r3c8_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c8_pytest.log
tail -5 gpurun_out/r3c8_pytest.log
gcc -O2 -std=c11 -pthread -Iinclude tests/cabi_driver.c -Larrow_go_amd -larrowhip -Wl,-rpath,$R/arrow_go_amd -o /tmp/cabi_driver && /tmp/cabi_driver bench 27 > gpurun_out/r3c8_filter_from_c.json; cat gpurun_out/r3c8_filter_from_c.json
timeout 600 python scripts/bench_hash.py 10 13 16 20 > gpurun_out/r3c8_bench_hash.json 2> gpurun_out/r3c8_bench_hash.err; cat gpurun_out/r3c8_bench_hash.json
cd /tmp; rm -rf /tmp/prof_g
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o run -- python $R/scripts/bench_hash.py 10 > /tmp/prof_g.out 2> /tmp/prof_g.err
python $R/scripts/rocpd_summary.py /tmp/prof_g/run_results.db | head -12 | cut -c1-120,300-380
