# rocprofv3 kernel-trace summary of a bench workload -> gpurun_out/prof_stats/b_kernel_stats.csv (copy it to profiles/).
# usage: bash tools/profile_bench.sh [workload]   (default: the pre-training step; imagenet | retrieval | vqa | glue)
W=${1:-pretrain}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_stats
cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -o b -- python3 bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-clocks --no-fusion-probe > gpurun_out/prof_stats.log 2>&1
find gpurun_out/prof_stats -name "*kernel_trace*" -delete
tail -1 gpurun_out/prof_stats.log | cut -c1-160
