# round 3, GPU call 4: C4/C5 single calls (RCCL world 1, gloo-transport world 2 and 3 on one GPU), encode auto, take steps=1
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_distributed_gpu.py -m gpu -x -q > gpurun_out/r3c4_pytest_dist.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c4_pytest_dist.log
tail -25 gpurun_out/r3c4_pytest_dist.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "hash_encode or take_vec" > gpurun_out/r3c4_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3c4_pytest.log
tail -5 gpurun_out/r3c4_pytest.log
timeout 600 python scripts/bench_encode_part.py 10 16 18 19 20 21 22 24 > gpurun_out/r3c4_encode_part.json 2> gpurun_out/r3c4_encode_part.err; echo "rc=$?"; tail -3 gpurun_out/r3c4_encode_part.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3c4_encode_part.json'))
for k,v in d['results'].items(): print(k, {a:b for a,b in v.items() if 'global' in a or 'auto' in a})
PY
timeout 300 python scripts/bench_take_clustered.py > gpurun_out/r3c4_take.json 2> gpurun_out/r3c4_take.err; python -c "import json;[print(k,v) for k,v in json.load(open('gpurun_out/r3c4_take.json')).items() if 'vec1' in k]"
# the bench under the launcher at world 1: the C4 / C5 lines go through ah_comm_* over RCCL
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 1 --steps 10 --warmup 2 --no-kernels --no-cpu-baseline > gpurun_out/r3c4_bench_dist1.json 2> gpurun_out/r3c4_bench_dist1.err; echo "bench dist rc=$?"; tail -2 gpurun_out/r3c4_bench_dist1.err
python -c "
import json;d=json.load(open('gpurun_out/r3c4_bench_dist1.json'));print(d['value'],d['config']['collectives']);print(d['c4_filter_aggregate']);print(d['c5_group_by'])"
