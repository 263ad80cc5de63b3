cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_hash -o hash -- python $R/scripts/bench_hash_profile.py > $R/gpurun_out/prof_hash.log 2>&1
cd $R && python scripts/rocpd_summary.py gpurun_out/prof_hash/hash_results.db | cut -c1-150 | head -14
