cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_fusion
cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_fusion -o f -- python3 tools/fusion_probe.py 8 $1 > gpurun_out/prof_fusion.log 2>&1
T=$(find gpurun_out/prof_fusion -name "*kernel_trace.csv" | head -1)
python3 tools/timeline.py $T 0.3 45 > gpurun_out/fusion_timeline$1.txt 2>&1
find gpurun_out/prof_fusion -name "*kernel_trace*" -delete
tail -1 gpurun_out/prof_fusion.log | cut -c1-300
head -8 gpurun_out/fusion_timeline$1.txt
