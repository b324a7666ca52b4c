#!/usr/bin/env python3
"""Per-kernel average of a rocprofv3 --pmc counter_collection CSV (one counter).
usage: summarize_pmc.py <pf_counter_collection.csv> [COUNTER] [kernel-name substring]
FETCH_SIZE is in KiB and, on gfx950 with this rocprofv3, tallies 128-B requests at 64 B for wide
coalesced streams (MI355X_MICROARCH.md, HBM section): the x2-corrected column is the HBM read traffic."""
import collections
import csv
import statistics
import sys

path = sys.argv[1]
counter = sys.argv[2] if len(sys.argv) > 2 else "FETCH_SIZE"
only = sys.argv[3] if len(sys.argv) > 3 else ""
d = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    if r["Counter_Name"] == counter and only in r["Kernel_Name"]:
        d[r["Kernel_Name"]].append((float(r["Counter_Value"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
print(f"kernel,calls,avg_{counter},avg_{counter}_x2_MB,avg_duration_ns")
for k, v in sorted(d.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
    fs = statistics.mean(x[0] for x in v)
    ns = statistics.mean(x[1] for x in v)
    print(f"\"{k}\",{len(v)},{fs:.1f},{fs * 2 * 1024 / 1e6:.2f},{ns:.0f}")
