# Every tracked artifact of a round in one go (run on the GPU box; copies land in gpurun_out/round/, to be moved into profiles/):
#   bash tools/round_profiles.sh A   -> headline: bench line, kernel stats (default + single stream), gemm shapes
#   bash tools/round_profiles.sh B   -> headline: HBM traffic + MFMA utilisation (PMC passes)
#   bash tools/round_profiles.sh C   -> secondary workloads: bench lines + kernel stats
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round
mkdir -p $O
case "$1" in
A)
  cd $R && XFM_BENCH_GEMM_SHAPES=$O/gemm_shapes.txt python3 bench.py > $O/bench_pretrain.json 2> $O/bench_pretrain.err
  bash tools/profile_bench.sh pretrain && cp gpurun_out/prof_stats/b_kernel_stats.csv $O/bench_kernel_stats.csv
  bash tools/profile_serial.sh && cp gpurun_out/prof_serial/b_kernel_stats.csv $O/bench_kernel_stats_single_stream.csv
  ;;
B)
  cd $R && bash tools/pmc_traffic.sh > /dev/null 2>&1; cp gpurun_out/hbm_traffic.json $O/hbm_traffic.json
  bash tools/pmc_mfma.sh > /dev/null 2>&1; cp gpurun_out/mfma_util.json $O/mfma_util.json
  ;;
C)
  cd $R
  for w in imagenet retrieval vqa glue; do
    python3 bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err
    bash tools/profile_bench.sh $w && cp gpurun_out/prof_stats/b_kernel_stats.csv $O/${w}_kernel_stats.csv
  done
  ;;
esac
ls -la $O
