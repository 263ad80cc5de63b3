# round 5, call 4: host-resident arguments streamed through the device by the host mirror's CallFunction
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_host_resident.py -m gpu -q -x -s > gpurun_out/r05_04_host.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_04_host.log
tail -25 gpurun_out/r05_04_host.log | cut -c1-400
timeout 600 python -m pytest tests/test_compute_api.py tests/test_ingest.py -m gpu -q -x > gpurun_out/r05_04_api.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_04_api.log
tail -4 gpurun_out/r05_04_api.log | cut -c1-300
