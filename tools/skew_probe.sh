# start-time skew of the persistent GEMM (XFM_GEMM_SKEW_US = spread in us per K-step): isolated timings of the step's big shapes
for s in 0 0.5 1.0 1.5 2.0 3.0; do
  echo "== XFM_GEMM_SKEW_US=$s"
  XFM_GEMM_SKEW_US=$s python tools/bench_nt256.py 2>&1 | grep -v "amdgpu.ids\|PERSIST" | head -8
done
