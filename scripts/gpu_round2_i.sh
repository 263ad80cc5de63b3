cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d $R/gpurun_out/pmc_gb -o run -- python $R/scripts/bench_groupby.py --only 16 > $R/gpurun_out/pmc_gb.out 2> $R/gpurun_out/pmc_gb.err
tail -3 $R/gpurun_out/pmc_gb.err
python $R/scripts/pmc_sq.py $R/gpurun_out/pmc_gb/run_results.db gb_ | tee $R/gpurun_out/pmc_gb_sq.txt
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d $R/gpurun_out/pmc_gb2 -o run -- python $R/scripts/bench_groupby.py --only 16 > $R/gpurun_out/pmc_gb2.out 2> $R/gpurun_out/pmc_gb2.err
tail -3 $R/gpurun_out/pmc_gb2.err
python $R/scripts/pmc_sq.py $R/gpurun_out/pmc_gb2/run_results.db gb_ | tee -a $R/gpurun_out/pmc_gb_sq.txt
rm -rf $R/gpurun_out/pmc_gb $R/gpurun_out/pmc_gb2
