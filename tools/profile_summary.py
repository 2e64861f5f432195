"""Markdown table of a rocprofv3 kernel_stats.csv: per-step calls / ms / average us / share (steps = launches of the run)."""
import csv
import sys

path, steps = sys.argv[1], float(sys.argv[2])
rows = list(csv.DictReader(open(path)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"Total kernel time per step: {tot / steps / 1e6:.1f} ms over {steps:.0f} steps\n")
print("| kernel | calls/step | ms/step | avg us | % |\n|---|---|---|---|---|")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    name = r["Name"].replace("void ", "")
    name = name.split("(")[0] if not name.startswith("at::") else name[:60]
    print(f"| `{name}` | {int(r['Calls']) / steps:.0f} | {int(r['TotalDurationNs']) / steps / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | "
          f"{100.0 * int(r['TotalDurationNs']) / tot:.1f} |")
