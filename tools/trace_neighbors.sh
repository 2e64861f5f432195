# which kernels surround the blit copies (__amd_rocclr_copyBuffer) in stream order?  -> gpurun_out/copy_neighbors.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ktrace
export XFM_WGRAD_STREAM=0 XFM_TEXT_STREAM=0
cd $R && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ktrace -o k -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fusion-probe > gpurun_out/ktrace.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/ktrace/**/k_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0][:70] for r in rows]
# last third of the trace = the last (instrumented) step
n = len(names)
seg = names[2 * n // 3:]
ctx = collections.Counter()
for i, k in enumerate(seg):
    if 'copyBuffer' in k:
        prev = next((seg[j] for j in range(i - 1, -1, -1) if 'copyBuffer' not in seg[j]), '-')
        nxt = next((seg[j] for j in range(i + 1, len(seg)) if 'copyBuffer' not in seg[j]), '-')
        ctx[(prev, nxt)] += 1
with open('gpurun_out/copy_neighbors.txt', 'w') as out:
    for (p, q), c in ctx.most_common(40):
        out.write(f"{c:4d}  after {p}  |  before {q}\n")
print(open('gpurun_out/copy_neighbors.txt').read()[:3000])
PY
rm -rf gpurun_out/ktrace
