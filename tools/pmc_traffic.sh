# HBM traffic per kernel family: two separate PMC passes (FETCH_SIZE, WRITE_SIZE) over a short bench run, then
# tools/hbm_traffic.py sums them per family into profiles/<round>_hbm_traffic.json.  Run on the GPU box via gpurun.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-clocks --no-fusion-probe > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-clocks --no-fusion-probe > $R/gpurun_out/pmc_write.log 2>&1
cd $R && python3 tools/hbm_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 4 > gpurun_out/hbm_traffic.json
find gpurun_out/pmc_fetch gpurun_out/pmc_write -type f -delete
cat gpurun_out/hbm_traffic.json | head -50
