cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for gm in 1 8; do for hint in 1 4; do
  XFM_GEMM_GROUP_M=$gm rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/l2_${gm}_${hint} -- python3 $R/tools/bench_gemm_pmc.py 25216 2304 768 $hint > /dev/null 2>&1
done; done
ls $R/gpurun_out | grep l2_
