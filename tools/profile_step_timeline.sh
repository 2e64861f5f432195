# Per-queue occupancy of the default (multi-stream) bench step: gpurun_out/step_timeline.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_tl
cd $R && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -o t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fusion-probe > gpurun_out/prof_tl.log 2>&1
T=$(find gpurun_out/prof_tl -name "*kernel_trace.csv" | head -1)
LIST=${LIST:-0:2.6} WAKE=${WAKE:-0.5} GANTT=${GANTT:-1} STEP_KERNEL=${STEP_KERNEL:-adamw_kernel:3:4} python3 tools/timeline.py $T 0.45 30 > gpurun_out/step_timeline.txt 2>&1
find gpurun_out/prof_tl -type f -delete
head -12 gpurun_out/step_timeline.txt
