"""MFMA utilisation from SQ counters per kernel family (tools/pmc_mfma.sh).

usage: mfma_util.py <counter_dir>  -> JSON on stdout
busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): the share of SIMD-cycles in which the matrix
pipe was executing, summed over the launches of the family (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles, rocprofv3
reports GRBM_GUI_ACTIVE summed over the 8 XCDs).  A kernel at the dense bf16 peak has busy fraction 1.0 at that launch's clock."""
import collections
import csv
import glob
import json
import sys


def family(name):
    n = name.replace("void ", "").split("(")[0]
    if "gemm_nt_256" in n or "gemm_tn_256" in n:
        return n
    if "gemm_tn_group" in n:
        return "gemm_tn_group_kernel (grouped weight gradients)"
    if "gemm_nt" in n:
        return "gemm_nt (small tiles)"
    if "gemm_tn" in n:
        return "gemm_tn (128 x 128)"
    if "attn" in n:
        return "attention (all)"
    return None


def main():
    d = sys.argv[1]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            fam = family(r["Kernel_Name"])
            if fam is not None:
                acc[fam][r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    acc[fam]["launches"] += 1
    out = {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -- python3 bench.py --steps 2 --warmup 1 "
                     "--no-cpu-baseline --no-fusion-probe with XFM_WGRAD_STREAM=0 XFM_TEXT_STREAM=0 (one kernel at a time)",
           "definition": "mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)", "families": {}}
    for fam, c in sorted(acc.items()):
        simd_cycles = c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
        out["families"][fam] = {"launches": int(c["launches"]), "mfma_busy_fraction": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, 4),
                                "gpu_active_cycles_per_launch": round(c["GRBM_GUI_ACTIVE"] / 8.0 / max(c["launches"], 1))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
