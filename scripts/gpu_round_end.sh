# what is run at the end of a round: full GPU suite, smoke, bench line, per-workload PMC.  Outputs under gpurun_out/ (copied to
# profiles/ by hand, named per round).   bash scripts/gpu_round_end.sh r03
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=${1:-rXX}
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/${TAG}_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_gpu.log
tail -8 gpurun_out/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/gpu_prof_workloads.sh ${TAG} 2>&1 | tail -36
cd $R
cp gpurun_out/${TAG}_pmc_by_workload.json profiles/${TAG}_pmc_by_workload.json   # so that the bench line below carries this session's counters
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_line.err; echo "bench rc=$?"; tail -2 gpurun_out/${TAG}_bench_line.err
python -c "
import json;d=json.load(open('gpurun_out/${TAG}_bench_line.json'));print(d['value'],d['roofline']);print({k:v.get('ms') for k,v in d['kernels'].items() if isinstance(v,dict)})"
cd /tmp; rm -rf /tmp/prof_b
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_b -o run -- python $R/bench.py --steps 20 --warmup 3 --no-kernels --no-cpu-baseline > /tmp/prof_b.out 2> /tmp/prof_b.err
python $R/scripts/rocpd_summary.py /tmp/prof_b/run_results.db > $R/gpurun_out/${TAG}_bench_kernel_stats.csv; head -8 $R/gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-160
cat /tmp/prof_b.out > $R/gpurun_out/${TAG}_bench_line_under_rocprof.json
