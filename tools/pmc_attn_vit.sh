# SQ counters of the ViT-shape attention kernels (tools/bench_attn_vit.py), two PMC passes; per-kernel sums over the run.
# usage (on the GPU box): XFM_ATTN_VIT_BWD=3 bash tools/pmc_attn_vit.sh > gpurun_out/pmc_attn_vit.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_av_a $R/gpurun_out/pmc_av_b
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_av_a -- python3 $R/tools/bench_attn_vit.py 3 > /dev/null 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/pmc_av_b -- python3 $R/tools/bench_attn_vit.py 3 > /dev/null 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for d in ("a", "b"):
    for f in glob.glob(f"gpurun_out/pmc_av_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "attn" not in k: continue
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES", "GRBM_GUI_ACTIVE"): n[(k, r["Counter_Name"])] += 1
for k, c in acc.items():
    la = max(n[(k, "SQ_WAVE_CYCLES")], 1)
    print(f"== {k}  launches {la}")
    wc = c["SQ_WAVE_CYCLES"]
    for name in sorted(c):
        extra = f"  ({c[name] / wc:.3f} of wave cycles)" if name.startswith("SQ_WAIT") or name.startswith("SQ_ACTIVE") else ""
        print(f"   {name:28s} {c[name] / la:14.0f} per launch{extra}")
PY
find gpurun_out/pmc_av_a gpurun_out/pmc_av_b -type f -delete
