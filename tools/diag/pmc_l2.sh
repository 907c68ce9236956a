# L2 hit / miss counters of one NT GEMM shape (run on the GPU box): bash tools/diag/pmc_l2.sh M N K
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_l2
echo "pass start" > $R/gpurun_out/pmc_l2.log
timeout -k 10 120 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $R/gpurun_out/pmc_l2 -- python3 $R/tools/nt_one.py $1 $2 $3 0 8 >> $R/gpurun_out/pmc_l2.log 2>&1 || echo "pass failed"
python3 $R/tools/pmc_generic.py $(ls $R/gpurun_out/pmc_l2/*/*counter_collection.csv) --match gemm_nt > $R/gpurun_out/r2_pmc_l2_$1_$2_$3.txt
rm -rf $R/gpurun_out/pmc_l2
cat $R/gpurun_out/r2_pmc_l2_$1_$2_$3.txt
