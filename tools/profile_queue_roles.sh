cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_qr
cd $R && XFM_DDP_FORCE=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_qr -o t -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-clocks --no-fusion-probe > gpurun_out/prof_qr.log 2>&1
T=$(find gpurun_out/prof_qr -name "*kernel_trace.csv" | head -1)
python3 tools/queue_roles.py $T
find gpurun_out/prof_qr -type f -delete
