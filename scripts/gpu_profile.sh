cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_kt -o bench -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_kt.json 2> $R/gpurun_out/prof_kt.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/prof_fetch -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_fetch.json 2> $R/gpurun_out/prof_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/prof_write -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_write.json 2> $R/gpurun_out/prof_write.err
cd $R
python scripts/rocpd_summary.py gpurun_out/prof_kt/bench_results.db | head -8 | cut -c1-160
python scripts/rocpd_pmc_summary.py gpurun_out/prof_fetch/bench_results.db | head -12 | cut -c1-200
python scripts/rocpd_pmc_summary.py gpurun_out/prof_write/bench_results.db | head -12 | cut -c1-200
tail -2 gpurun_out/prof_fetch.err
