# SQ counters of the attention kernels at the ViT shape (tools/bench_attn.py), three PMC passes; prints per-kernel averages.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_at_a $R/gpurun_out/pmc_at_b $R/gpurun_out/pmc_at_c
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_at_a -- python3 $R/tools/bench_attn.py 3 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_at_b -- python3 $R/tools/bench_attn.py 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $R/gpurun_out/pmc_at_c -- python3 $R/tools/bench_attn.py 3 > /dev/null 2>&1
cd $R
for d in a b c; do
  for c in $(python3 - <<PY
import csv,glob
s=set()
for f in glob.glob("gpurun_out/pmc_at_$d/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)): s.add(r["Counter_Name"])
print(" ".join(sorted(s)))
PY
); do echo "-- $c"; python3 tools/pmc_one.py gpurun_out/pmc_at_$d $c attn_bwd_d | cut -c1-150; done
done
find gpurun_out/pmc_at_a gpurun_out/pmc_at_b gpurun_out/pmc_at_c -type f -delete
