cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"
B="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
run() {
  name=$1; flt=$2; shift 2
  : > $R/gpurun_out/pmc_sq_$name.txt
  for set in "$A" "$B"; do
    rm -rf /tmp/pmc_$name
    timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_$name -o run -- "$@" > /tmp/pmc_$name.out 2> /tmp/pmc_$name.err || tail -3 /tmp/pmc_$name.err
    python $R/scripts/pmc_sq.py /tmp/pmc_$name/run_results.db "$flt" >> $R/gpurun_out/pmc_sq_$name.txt
  done
  cat $R/gpurun_out/pmc_sq_$name.txt | cut -c1-600
}
run encode_20 enc_ python $R/scripts/bench_encode_one.py 20
rm -rf /tmp/st; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st -o run -- python $R/scripts/bench_encode_one.py 20 > /dev/null 2>&1
python $R/scripts/rocpd_summary.py /tmp/st/run_results.db > $R/gpurun_out/r3c19_encode_20_kernel_stats.csv; cut -c1-90,200-400 $R/gpurun_out/r3c19_encode_20_kernel_stats.csv | head -24
