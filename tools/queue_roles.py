"""Which streams share a hardware queue?  From a rocprofv3 --kernel-trace CSV: per Queue_Id, the marker kernels of each stream of the step
(text tower: emb_bwd_kernel; RCCL: oneRankReduce / ncclDevKernel; weight-gradient side streams: gemm_tn_group_kernel; operand refresh:
cast_transpose_batch_kernel; launch stream: adamw_kernel; zero stream: the large FillFunctor<float> launches).  usage: queue_roles.py <csv>"""
import csv
import sys
from collections import defaultdict

roles = {"emb_bwd_kernel": "TEXT", "oneRankReduce": "RCCL", "ncclDevKernel": "RCCL", "gemm_tn_group_kernel": "WGRAD-side", "cast_transpose_batch_kernel": "CAST",
         "adamw_kernel": "MAIN", "emb_fwd_kernel": "TEXT(fwd)"}
seen = defaultdict(lambda: defaultdict(int))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    for k, v in roles.items():
        if k in n:
            seen[r["Queue_Id"]][v] += 1
    if "FillFunctor<float>" in n and int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 50000:
        seen[r["Queue_Id"]]["ZERO(big fills)"] += 1
for q in sorted(seen):
    print(f"queue {q}: " + ", ".join(f"{k} x{v}" for k, v in sorted(seen[q].items())))
