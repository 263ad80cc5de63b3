cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 scripts/micro/add_variants.hip -o /tmp/add_variants 2>&1 | tail -5
timeout 300 /tmp/add_variants > gpurun_out/micro_add.txt 2>&1; cat gpurun_out/micro_add.txt
