# SQ counters of the 256 x 256 NT kernel on one shape (tools/bench_gemm_pmc.py), two PMC passes; sums per kernel over 16 launches.
# usage (on the GPU box): bash tools/pmc_nt256.sh M N K > gpurun_out/pmc_nt256.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_nt_a $R/gpurun_out/pmc_nt_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_nt_a -- python3 $R/tools/bench_gemm_pmc.py $1 $2 $3 5 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/pmc_nt_b -- python3 $R/tools/bench_gemm_pmc.py $1 $2 $3 5 > /dev/null 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_WAVE32_LDS SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_nt_c -- python3 $R/tools/bench_gemm_pmc.py $1 $2 $3 5 > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for d in ("a", "b", "c"):
    for f in glob.glob(f"gpurun_out/pmc_nt_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "gemm_nt_256" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"): n[(k, r["Counter_Name"])] += 1
for k, c in acc.items():
    la = max(n[(k, "SQ_WAVE_CYCLES")], 1)
    print(f"== {k}  launches {la}")
    wc = c["SQ_WAVE_CYCLES"]
    for name in sorted(c):
        extra = f"  ({c[name] / wc:.3f} of wave cycles)" if name.startswith("SQ_WAIT") or name.startswith("SQ_ACTIVE") or name.startswith("SQ_INST_") else ""
        print(f"   {name:28s} {c[name] / la:14.0f} per launch{extra}")
PY
find gpurun_out/pmc_nt_a gpurun_out/pmc_nt_b gpurun_out/pmc_nt_c -type f -delete
