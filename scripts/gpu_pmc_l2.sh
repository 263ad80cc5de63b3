# L2 counters of the two gather-rate-bound passes: the re-packed-table probe of dictionary_encode (2^16 keys) and the window gather
# of the binned Take.  → gpurun_out/pmc_l2.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
cd /tmp
: > $R/gpurun_out/pmc_l2.txt
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  rm -rf /tmp/pmc_l2; timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_l2 -o run -- python $R/scripts/prof_workloads.py --only dict_encode_2^16,take_random > /tmp/pmc_l2.out 2> /tmp/pmc_l2.err || tail -3 /tmp/pmc_l2.err
  python $R/scripts/pmc_sq.py /tmp/pmc_l2/run_results.db insert_compact >> $R/gpurun_out/pmc_l2.txt
  python $R/scripts/pmc_sq.py /tmp/pmc_l2/run_results.db bin_gather >> $R/gpurun_out/pmc_l2.txt
done
cat $R/gpurun_out/pmc_l2.txt | cut -c1-400
