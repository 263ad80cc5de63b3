# usage: bash scripts/gpu_prof_workloads.sh <tag> [--only a,b]  → gpurun_out/<tag>_pmc_by_workload.json
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=$1; shift
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/pw_kt -o run -- python $R/scripts/prof_workloads.py "$@" > $R/gpurun_out/${TAG}_workloads.json 2> $R/gpurun_out/pw_kt.err
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pw_fetch -o run -- python $R/scripts/prof_workloads.py "$@" > /dev/null 2> $R/gpurun_out/pw_fetch.err
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pw_write -o run -- python $R/scripts/prof_workloads.py "$@" > /dev/null 2> $R/gpurun_out/pw_write.err
cd $R
python scripts/pmc_by_workload.py gpurun_out/${TAG}_workloads.json gpurun_out/pw_kt/run_results.db gpurun_out/pw_fetch/run_results.db gpurun_out/pw_write/run_results.db > gpurun_out/${TAG}_pmc_by_workload.json
tail -3 gpurun_out/pw_kt.err
python -c "
import json; d=json.load(open('gpurun_out/${TAG}_pmc_by_workload.json'))
for k,v in d['workloads'].items(): print(k, v['per_call'])
"
