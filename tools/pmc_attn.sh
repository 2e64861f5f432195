cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_at_a -- python3 $R/tools/bench_attn.py 3 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_at_b -- python3 $R/tools/bench_attn.py 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $R/gpurun_out/pmc_at_c -- python3 $R/tools/bench_attn.py 3 > /dev/null 2>&1
find $R/gpurun_out/pmc_at_a $R/gpurun_out/pmc_at_b $R/gpurun_out/pmc_at_c -name "*agent_info*" -delete
ls $R/gpurun_out/pmc_at_*/*
