for t in 0 35 55 70; do echo "== tail split max $t%"; XFM_GEMM_TAIL_SPLIT=$t XFM_TUNE_NT_SHAPES=1 timeout -k 10 200 python tools/tune_gemm.py cold 2>&1 | grep -v amdgpu | cut -c1-50; done
