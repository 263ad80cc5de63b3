cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
for lg in 16 10; do
rm -rf /tmp/prof_n; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_n -o run -- python $R/scripts/bench_nullkeys.py --only $lg > /tmp/prof_n.out 2> /tmp/prof_n.err; tail -1 /tmp/prof_n.err
python $R/scripts/rocpd_summary.py /tmp/prof_n/run_results.db > $R/gpurun_out/nullkeys_${lg}_kernel_stats.csv
sed 's/(anonymous namespace):://g' $R/gpurun_out/nullkeys_${lg}_kernel_stats.csv | awk -F'",' '{print substr($1,1,60), $2}' | head -14
done
