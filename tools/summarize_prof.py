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
    # A fresh process runs its first dispatches at lower clocks than it then holds (the first ~30 of a bench.py run
    # take 0.7-1.05 ms against 0.66-0.72 in steady state): the all-dispatch average rocprofv3 --stats prints understates
    # the steady state by about 1 %.  Both are given; bench.py's own figure is taken after its warm-up.
    print("## demod kernels: all dispatches against steady state (kernel trace, dispatches in start order)\n")
    print("| kernel | dispatches | avg ns, all | avg ns after the first 32 | median ns | min ns |")
    print("|---|---|---|---|---|---|")
    for k, rs in byk.items():
        if "demod" not in k:
            continue
        rs = sorted(rs, key=lambda r: int(r["Start_Timestamp"]))
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs]
        steady = dur[32:] if len(dur) > 48 else dur
        med = sorted(dur)[len(dur) // 2]
        print(f"| {k[:90]} | {len(dur)} | {sum(dur) / len(dur):.0f} | {sum(steady) / len(steady):.0f} ({len(steady)}) | {med} | {min(dur)} |")
    print()
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

# HBM traffic per dispatch of the streaming kernel, as bench.py reports it (profiles/traffic.json): FETCH_SIZE and WRITE_SIZE
# are KiB; FETCH_SIZE counts 32 B for a 64 B request on gfx950 with 16-byte-per-lane streaming reads and is doubled
# (MI355X_MICROARCH.md, HBM / rocprofv3 section).
fetch = write = None
for d, name in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for f in find(f"{d}/**/*counter_collection.csv"):
        n, v = 0, 0.0
        for r in csv.DictReader(open(f)):
            if "demod_rows" in r["Kernel_Name"] and r["Counter_Name"] == name:
                n += 1
                v += float(r["Counter_Value"])
        if n:
            if name == "FETCH_SIZE":
                fetch = v / n
            else:
                write = v / n
if fetch is not None and write is not None:
    hbm = (2.0 * fetch + write) * 1024.0
    print(f"\n## HBM traffic per demod_rows dispatch\n\nFETCH_SIZE {fetch:.6g} KiB (raw; doubled per the gfx950 correction), WRITE_SIZE {write:.6g} KiB: "
          f"{hbm:.6g} bytes per dispatch")
    import json
    print("\ntraffic.json entry: " + json.dumps({"fetch_size_kib_raw": round(fetch, 1), "write_size_kib": round(write, 1), "hbm_bytes_per_launch": int(round(hbm))}))
