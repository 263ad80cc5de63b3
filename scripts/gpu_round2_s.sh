cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_gb_16 -o run -- python $R/scripts/bench_groupby.py --only 16 > $R/gpurun_out/prof_gb_16.out 2> $R/gpurun_out/prof_gb_16.err
python $R/scripts/rocpd_summary.py $R/gpurun_out/prof_gb_16/run_results.db > $R/gpurun_out/prof_gb_16_kernels.csv
sed 's/(anonymous namespace):://g' $R/gpurun_out/prof_gb_16_kernels.csv | awk -F'",' '{print substr($1,1,60), $2}' | head -24
rm -rf $R/gpurun_out/prof_gb_16
