# rocprofv3 kernel-trace summary of the bench workload with the side streams off (every kernel alone on the GPU: isolated
# per-kernel durations, the measure of machine work) -> gpurun_out/prof_serial/b_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_serial
export XFM_WGRAD_STREAM=0 XFM_TEXT_STREAM=0
cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_serial -o b -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-clocks --no-fusion-probe > gpurun_out/prof_serial.log 2>&1
find gpurun_out/prof_serial -name "*kernel_trace*" -delete
tail -1 gpurun_out/prof_serial.log | cut -c1-200
python3 tools/profile_summary.py gpurun_out/prof_serial/b_kernel_stats.csv 8 60 > gpurun_out/prof_serial_summary.md
