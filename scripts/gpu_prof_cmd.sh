# usage: bash scripts/gpu_prof_cmd.sh <tag> <python script + args...>   → gpurun_out/prof_<tag>/, prints the kernel table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; shift
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o run -- python $R/"$@" > $R/gpurun_out/prof_$TAG.out 2> $R/gpurun_out/prof_$TAG.err
cd $R
python scripts/rocpd_summary.py gpurun_out/prof_$TAG/run_results.db > gpurun_out/prof_${TAG}_kernels.csv
head -14 gpurun_out/prof_${TAG}_kernels.csv | cut -c1-220
