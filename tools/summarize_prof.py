#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into a small markdown summary:
per-kernel time statistics from the kernel trace and per-dispatch averages of
every PMC counter collected, for the demod kernels."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


print(f"# rocprofv3 summary: {os.path.basename(out)}\n")
for f in find("trace/**/*kernel_stats.csv"):
    print("## kernel stats (rocprofv3 --kernel-trace --stats)\n")
    print("| kernel | calls | total ns | avg ns | min ns | max ns | % |")
    print("|---|---|---|---|---|---|---|")
    for row in csv.DictReader(open(f)):
        name = row.get("Name", "")[:90]
        print(f"| {name} | {row.get('Calls')} | {row.get('TotalDurationNs')} | {row.get('AverageNs')} | "
              f"{row.get('MinNs')} | {row.get('MaxNs')} | {row.get('Percentage')} |")
    print()
for f in find("trace/**/*kernel_trace.csv"):
    rows = list(csv.DictReader(open(f)))
    byk = defaultdict(list)
    for r in rows:
        byk[r["Kernel_Name"]].append(r)
    print("## dispatch resources (kernel trace)\n")
    print("| kernel | grid | workgroup | VGPR | accum VGPR | SGPR | LDS | scratch |")
    print("|---|---|---|---|---|---|---|---|")
    for k, rs in byk.items():
        r = rs[-1]
        print(f"| {k[:90]} | {r.get('Grid_Size')} | {r.get('Workgroup_Size')} | {r.get('VGPR_Count')} | "
              f"{r.get('Accum_VGPR_Count')} | {r.get('SGPR_Count')} | {r.get('LDS_Block_Size')} | {r.get('Scratch_Size')} |")
    print()
print("## PMC counters, average per dispatch\n")
print("| pass | kernel | counter | dispatches | avg value |")
print("|---|---|---|---|---|")
for d in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    for f in find(f"{d}/**/*counter_collection.csv"):
        acc = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"][:70], r["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
        for (k, c), (n, v) in sorted(acc.items()):
            if "demod" in k:
                print(f"| {d} | {k} | {c} | {n} | {v / n:.6g} |")
