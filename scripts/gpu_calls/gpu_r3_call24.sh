cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 1 --steps 10 --warmup 2 --no-kernels --no-cpu-baseline > gpurun_out/r3c24_bench_dist1.json 2> gpurun_out/r3c24_bench_dist1.err; echo "bench dist rc=$?"; tail -2 gpurun_out/r3c24_bench_dist1.err
python -c "
import json;d=json.load(open('gpurun_out/r3c24_bench_dist1.json'));print(d['value'],d['config']['collectives']);print(d['c4_filter_aggregate']);print(d['c5_group_by'])"
timeout 600 python bench.py --steps 5 --warmup 2 --no-kernels --no-cpu-baseline > gpurun_out/r3c24_bench_quick.json 2> gpurun_out/r3c24_bench_quick.err; echo "bench rc=$?"; wc -l gpurun_out/r3c24_bench_quick.json
cd /tmp; rm -rf /tmp/st; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st -o run -- python $R/scripts/bench_encode_one.py 20 > /dev/null 2>&1
python $R/scripts/rocpd_summary.py /tmp/st/run_results.db > $R/gpurun_out/r3c24_encode_20_kernel_stats.csv
rm -rf /tmp/st; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st -o run -- python $R/scripts/bench_encode_one.py 24 > /dev/null 2>&1
python $R/scripts/rocpd_summary.py /tmp/st/run_results.db > $R/gpurun_out/r3c24_encode_24_kernel_stats.csv
