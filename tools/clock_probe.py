#!/usr/bin/env python3
"""Sustained shader clock per demod_rows dispatch, from one `rocprofv3 --pmc GRBM_GUI_ACTIVE` run.

    python tools/clock_probe.py <dir with *_counter_collection.csv> [label]

GRBM_GUI_ACTIVE of a dispatch is the sum over the 8 XCDs of the cycles the dispatch kept them busy; the CSV also
carries the dispatch's start and end timestamps (ns), so cycles / 8 / duration is the clock the chip held for it.
The first 32 dispatches of a process (clock ramp) are left out.  Prints one line per kernel."""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
label = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(d.rstrip("/"))
rows = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or "demod_rows" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("demod_rows")[1].split("(")[0]
        rows[name].append((int(r["Start_Timestamp"]), float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for name, v in rows.items():
    v.sort()
    v = v[32:] if len(v) > 64 else v
    cyc = sum(x[1] for x in v) / len(v) / 8.0
    ns = sum(x[2] for x in v) / len(v)
    per_sample = 3 if ", true, " in name else 4      # algorithmic bytes per sample: int16 output writes 1 B instead of 2
    print(f"{label:14s} demod_rows{name:34s} {len(v):4d} dispatches  {ns / 1e6:.4f} ms  {cyc:.4e} cycles per XCD  {cyc / ns:.3f} GHz  "
          f"frac {per_sample * (1 << 30) / (ns * 1e-9) / 8e12:.3f} (2^30 samples per dispatch assumed)")
