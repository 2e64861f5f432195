# The ATen / runtime kernels inside ONE traced training step (the last of the run): count, time, and which of our kernels they sit between.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ktrace
cd $R && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktrace -o k -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-fusion-probe --no-clocks > gpurun_out/ktrace.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/ktrace/**/k_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# the last step = from the last-but-one adamw's end to the last adamw's end
ad = [i for i, r in enumerate(rows) if 'adamw_kernel' in r['Kernel_Name']]
lo, hi = ad[-4] + 1, ad[-1] + 1      # three adamw launches per step
seg = rows[lo:hi]
t0, t1 = int(seg[0]['Start_Timestamp']), int(seg[-1]['End_Timestamp'])
ours = lambda n: not ('at::native' in n or 'rocclr' in n or 'rocprim' in n or n.startswith('void (anonymous'))
cnt = collections.Counter(); dur = collections.Counter()
for r in seg:
    n = r['Kernel_Name']
    if not ours(n):
        key = n.split('<')[0].split('(')[0][-40:] + ' ' + (n.split('native::')[-1][:50] if 'native::' in n else '')
        cnt[key] += 1; dur[key] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print(f"step span {(t1 - t0) / 1e6:.2f} ms, {len(seg)} kernels, of them {sum(cnt.values())} ATen / runtime kernels taking {sum(dur.values()) / 1e6:.3f} ms")
for k, c in cnt.most_common(25):
    print(f"  {c:4d}  {dur[k] / 1e3:8.1f} us  {k}")
PY
rm -rf gpurun_out/ktrace
