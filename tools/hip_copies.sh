# Which HIP runtime calls produce the blit-copy kernels of a step?  HIP API trace of a short bench run: counts of the memcpy / memset
# entry points per step and their byte sizes.  usage (GPU box): bash tools/hip_copies.sh > gpurun_out/hip_copies.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/hiptrace
cd $R && rocprofv3 --hip-trace --output-format csv -d /tmp/hiptrace -o h -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-fusion-probe > /tmp/hiptrace.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/hiptrace/**/h_hip_api_trace.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
print("columns:", list(rows[0].keys()))
c = collections.Counter(r["Function"] for r in rows)
for k, v in c.most_common(24):
    print(f"{v:8d}  {k}")
# between consecutive hipDeviceSynchronize calls (the bench brackets its timed region with them)
syncs = sorted(int(r["Start_Timestamp"]) for r in rows if r["Function"] == "hipDeviceSynchronize")
edges = [0] + syncs + [1 << 62]
for a, b in zip(edges[:-1], edges[1:]):
    seg = [r for r in rows if a <= int(r["Start_Timestamp"]) < b]
    cc = collections.Counter(r["Function"] for r in seg)
    print(f"segment of {len(seg):7d} calls: launches {cc['hipLaunchKernel']:6d}  memcpyAsync {cc['hipMemcpyAsync']:5d}  memcpyWithStream {cc['hipMemcpyWithStream']:5d}  "
          f"eventRecord {cc['hipEventRecord']:5d}  streamWaitEvent {cc['hipStreamWaitEvent']:5d}  malloc {cc['hipMalloc']:4d}")
PY
