cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ktrace
export XFM_WGRAD_STREAM=0 XFM_TEXT_STREAM=0
cd $R && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/ktrace -o k -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-fusion-probe --no-clocks > gpurun_out/ktrace.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/ktrace/**/k_kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
n = len(rows)
seg = rows[2 * n // 3:]
out = []
run = 0
for i, r in enumerate(seg):
    k = r['Kernel_Name'].split('(')[0][:60]
    if 'copyBuffer' in k:
        run += 1
        continue
    if run:
        out.append(f"   [{run} x copyBuffer]")
        run = 0
    out.append(k)
# print only neighbourhoods of copy runs
idx = [i for i, l in enumerate(out) if 'copyBuffer' in l]
shown = set()
for i in idx:
    for j in range(max(0, i - 2), min(len(out), i + 2)):
        if j not in shown:
            shown.add(j)
for j in sorted(shown):
    print(out[j])
mc = glob.glob('gpurun_out/ktrace/**/k_memory_copy_trace.csv', recursive=True)
if mc:
    rows = list(csv.DictReader(open(mc[0])))
    import collections
    c = collections.Counter((r.get('Direction'), ) for r in rows)
    print("memory copies in the whole run:", dict(c), "columns", list(rows[0].keys()) if rows else None)
PY
rm -rf gpurun_out/ktrace
