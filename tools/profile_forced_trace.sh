cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_ft
cd $R && XFM_DDP_FORCE=${FORCE:-1} rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ft -o t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-clocks --no-fusion-probe > gpurun_out/prof_ft.log 2>&1
T=$(find gpurun_out/prof_ft -name "*kernel_trace.csv" | head -1)
python3 tools/queue_map.py $T 2 > gpurun_out/queue_map_${FORCE:-1}.txt 2>&1
find gpurun_out/prof_ft -type f -delete
