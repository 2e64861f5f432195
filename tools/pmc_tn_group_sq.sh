# SQ counters of the grouped weight-gradient kernel and, beside it, of the persistent NT kernel on a K = 3072 problem (same 256 x 256
# tile, same 8 waves): where do their K-steps differ?  Run on the GPU box.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_tng_a $R/gpurun_out/pmc_tng_b
cat > /tmp/nt_big.py <<PY
import sys; sys.path.insert(0, "$R")
import torch
from xfm_amd import functional as Fx
g = torch.Generator(device="cuda").manual_seed(1)
a = torch.randn(21760, 3072, device="cuda", generator=g).bfloat16(); b = (torch.randn(768, 3072, device="cuda", generator=g) * 0.05).bfloat16()
out = torch.empty(21760, 768, device="cuda", dtype=torch.bfloat16)
for _ in range(6): Fx.gemm_nt(a, b, out=out, tile_hint=5)
torch.cuda.synchronize()
PY
for prog in "tools/tn_group_bench.py" "/tmp/nt_big.py"; do
  P=$prog; case $prog in tools/*) P=$R/$prog;; esac
  tag=$(basename $prog .py)
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_tng_a_$tag -- python3 $P > /dev/null 2>&1
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 --kernel-trace --output-format csv -d $R/gpurun_out/pmc_tng_b_$tag -- python3 $P > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for d in ("$R/gpurun_out/pmc_tng_a_$tag", "$R/gpurun_out/pmc_tng_b_$tag"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_tn_group_kernel" in r["Kernel_Name"] or "gemm_nt_256_kernel" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print("== $tag")
wc = tot.get("SQ_WAVE_CYCLES", 1.0)
for k in sorted(tot):
    print("  %-32s per launch %.4g   / SQ_WAVE_CYCLES %.3f" % (k, tot[k] / max(n[k], 1), tot[k] / n[k] / (wc / n["SQ_WAVE_CYCLES"])))
PY
  find $R/gpurun_out/pmc_tng_a_$tag $R/gpurun_out/pmc_tng_b_$tag -type f -delete
done
