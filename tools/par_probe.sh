python tools/fusion_host.py > gpurun_out/par_a.txt 2>&1 &
python tools/fusion_host.py > gpurun_out/par_b.txt 2>&1 &
wait
grep "iters 64" gpurun_out/par_a.txt gpurun_out/par_b.txt
python tools/fusion_host.py 2>&1 | grep "iters 64"
