# kernel timeline of the last call of a script: bash scripts/gpu_calls/r06_timeline.sh <n kernels from the end> <script> [args]
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; N=$1; shift
cd /tmp; rm -rf /tmp/prof_tl
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_tl -o run -- python $R/"$@" > /tmp/prof_tl.out 2>/tmp/prof_tl.err
tail -1 /tmp/prof_tl.out
python $R/scripts/kernel_timeline.py /tmp/prof_tl/run_results.db $N
