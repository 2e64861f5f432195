cd $GRAFT_REPO_ROOT
for gm in 1000000 4 8 16 32; do echo "GROUP_M=$gm"; XFM_GEMM_GROUP_M=$gm python tools/tune_gemm.py cold 2>&1 | grep -E "^ (25216|75648|11520) " | head -9 | cut -c1-140; done
