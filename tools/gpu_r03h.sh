cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
bash tools/fused_ts.sh > $O/fused_ts.txt 2>&1
bash tools/pmc_pass.sh r03h_lds "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" > $O/pmc_lds.txt 2>&1
grep "fused_v2 cycles" $O/fused_ts.txt | head -3; tail -22 $O/pmc_lds.txt | cut -c1-120
