# sample clocks / power while the bench workload runs (is the step power-limited?)
R=$GRAFT_REPO_ROOT
cd $R
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | head -6
python tools/soak.py 300 > gpurun_out/soak_clk.log 2>&1 &
PID=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -i "sclk\|Average Graphics\|Socket Power\|GPU use" | tr '\n' ' '; echo; sleep 2; done
wait $PID
tail -2 gpurun_out/soak_clk.log
