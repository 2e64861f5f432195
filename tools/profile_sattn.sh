# per-kernel times of the small self-attention micro-benchmark, packed and one-row-per-workgroup
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/sattn1 $R/gpurun_out/sattn0
cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sattn1 -o p -- python3 tools/bench_sattn.py > gpurun_out/sattn1.log 2>&1
export XFM_ATTN_PACK=0
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sattn0 -o p -- python3 tools/bench_sattn.py > gpurun_out/sattn0.log 2>&1
find gpurun_out/sattn1 gpurun_out/sattn0 -name "*kernel_trace*" -delete
for d in sattn1 sattn0; do f=$(find gpurun_out/$d -name "*kernel_stats.csv"); head -6 $f | cut -c1-140; done
