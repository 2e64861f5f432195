# MFMA pipe utilisation per kernel family from SQ counters (one PMC pass, kernels one at a time) -> gpurun_out/mfma_util.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_mfma
export XFM_WGRAD_STREAM=0 XFM_TEXT_STREAM=0
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-clocks --no-fusion-probe > $R/gpurun_out/pmc_mfma.log 2>&1
cd $R && python3 tools/mfma_util.py gpurun_out/pmc_mfma > gpurun_out/mfma_util.json
find gpurun_out/pmc_mfma -type f -delete
cat gpurun_out/mfma_util.json
