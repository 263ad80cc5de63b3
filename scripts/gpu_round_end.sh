# what is run at the end of a round: full GPU suite, smoke, bench line, per-workload PMC, group-by sweep, kernel stats of the
# sort / group-by paths.  Outputs under gpurun_out/ (copied to profiles/ by hand, named per round).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
bash scripts/gpu_final_check.sh
bash scripts/gpu_prof_workloads.sh r02 2>&1 | tail -30
cd $R
timeout 600 python scripts/bench_groupby.py > gpurun_out/bench_groupby.json 2> gpurun_out/bench_groupby.err; tail -2 gpurun_out/bench_groupby.err
timeout 300 python scripts/bench_nullkeys.py > gpurun_out/bench_nullkeys.json 2> gpurun_out/bench_nullkeys.err
cd /tmp
for c in "27 int" "27 normal"; do
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_s -o run -- python $R/scripts/bench_sort_one.py $c 1 > /tmp/prof_s.out 2> /tmp/prof_s.err
done
python $R/scripts/rocpd_summary.py /tmp/prof_s/run_results.db > $R/gpurun_out/sort_msd_kernel_stats.csv
for lg in 16 20 24; do
rm -rf /tmp/prof_g; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_g -o run -- python $R/scripts/bench_groupby.py --only $lg > /tmp/prof_g.out 2> /tmp/prof_g.err
python $R/scripts/rocpd_summary.py /tmp/prof_g/run_results.db > $R/gpurun_out/groupby_${lg}_kernel_stats.csv
done
rm -rf $R/gpurun_out/pw_kt $R/gpurun_out/pw_fetch $R/gpurun_out/pw_write
