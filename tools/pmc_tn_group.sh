# FETCH_SIZE of the grouped weight-gradient kernel with / without the XCD progress throttle (tools/tn_group_bench.py); run on the GPU box.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for w in 0 4 16; do
  rm -rf $R/gpurun_out/pmc_tng_$w
  XFM_TN_SYNC_WINDOW=$w rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_tng_$w -- python3 $R/tools/tn_group_bench.py > $R/gpurun_out/pmc_tng_$w.log 2>&1
  python3 - <<PY
import csv, glob
tot, n = 0.0, 0
for f in glob.glob("$R/gpurun_out/pmc_tng_$w/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "gemm_tn_group_kernel" in r["Kernel_Name"]:
            tot += float(r["Counter_Value"]); n += 1
print("window $w: gemm_tn_group_kernel launches", n, "read bytes per launch (2 x FETCH_SIZE KB): %.2f GB" % (2 * tot * 1024 / max(n, 1) / 1e9))
PY
  find $R/gpurun_out/pmc_tng_$w -type f -delete
done
