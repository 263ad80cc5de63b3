# kernel-trace of one many-groups workload: bash scripts/gpu_calls/r06_prof_gb.sh <lg> <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; LG=${1:-24}; TAG=${2:-gb}
cd /tmp; rm -rf /tmp/prof_$TAG
ENC=${ENC:-0} timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o run -- python $R/scripts/bench_gb_many.py $LG > /tmp/prof_$TAG.out 2>/tmp/prof_$TAG.err
tail -1 /tmp/prof_$TAG.out
python $R/scripts/rocpd_summary.py /tmp/prof_$TAG/run_results.db > $R/gpurun_out/r06_${TAG}_kernel_stats.csv; head -30 $R/gpurun_out/r06_${TAG}_kernel_stats.csv | cut -c1-200
