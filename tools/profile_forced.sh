# rocprofv3 kernel-trace summary of the pre-training step with the N > 1 code path forced in a group of one (XFM_DDP_FORCE=1):
# the RCCL kernel rows and the per-chunk grouped weight gradients -> gpurun_out/prof_forced/b_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_forced
cd $R && XFM_DDP_FORCE=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_forced -o b -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-clocks --no-fusion-probe > gpurun_out/prof_forced.log 2>&1
find gpurun_out/prof_forced -name "*kernel_trace*" -delete
tail -1 gpurun_out/prof_forced.log | cut -c1-200
