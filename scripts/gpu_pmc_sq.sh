# SQ counters (instruction mix, issue / wait time) of the instruction-bound kernels: group-by passes and the MSD sort passes.
# Two --pmc passes per workload (8 counters each); summaries under gpurun_out/pmc_sq_*.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"
B="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
run() {  # name filter cmd...
  name=$1; flt=$2; shift 2
  : > $R/gpurun_out/pmc_sq_$name.txt
  for set in "$A" "$B"; do
    rm -rf /tmp/pmc_$name
    timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_$name -o run -- "$@" > /tmp/pmc_$name.out 2> /tmp/pmc_$name.err || tail -3 /tmp/pmc_$name.err
    python $R/scripts/pmc_sq.py /tmp/pmc_$name/run_results.db "$flt" >> $R/gpurun_out/pmc_sq_$name.txt
  done
  cat $R/gpurun_out/pmc_sq_$name.txt | cut -c1-400
}
run groupby_16 gb_ python $R/scripts/bench_groupby.py --only 16
run sort_27 ms_ python $R/scripts/bench_sort_one.py 27 int 1
