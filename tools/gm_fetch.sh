# FETCH_SIZE (beyond-L2 read requests, KB at 64 B per request) and duration of the 256x256 NT kernel per tile-group height, on the
# ViT shapes of the step with cold-rotated operands.  Run on the GPU box.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shape in "25216 768 768" "25216 2304 768" "25216 768 3072" "25216 3072 768"; do
  for gm in 1 2 4 8 16; do
    rm -rf $R/gpurun_out/gmf
    XFM_GEMM_GROUP_M=$gm rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/gmf -- python3 $R/tools/bench_gemm_pmc.py $shape 5 > /dev/null 2>&1
    echo "shape $shape GROUP_M=$gm: $(python3 $R/tools/pmc_one.py $R/gpurun_out/gmf FETCH_SIZE gemm_nt)"
  done
done
rm -rf $R/gpurun_out/gmf
