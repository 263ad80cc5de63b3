# SQ counters of the partitioned group-by at 2^16 groups (hash_sum Float64, 2^26 rows): gpurun_out/r06_pmc_sq_groupby_16.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES"
B="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"
: > $R/gpurun_out/r06_pmc_sq_groupby_16.txt
for set in "$A" "$B"; do
  rm -rf /tmp/pmc_g16
  ENC=0 timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_g16 -o run -- python $R/scripts/bench_gb_many.py 16 > /tmp/pmc_g16.out 2> /tmp/pmc_g16.err || tail -3 /tmp/pmc_g16.err
  python $R/scripts/pmc_sq.py /tmp/pmc_g16/run_results.db gb_ >> $R/gpurun_out/r06_pmc_sq_groupby_16.txt
done
cut -c1-420 $R/gpurun_out/r06_pmc_sq_groupby_16.txt
