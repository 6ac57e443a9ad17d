#!/usr/bin/env python3
"""Prints the kernels of the LAST large-topic sort in a rocprofv3 kernel-trace CSV, in time order.
    python tools/trace_timeline.py /tmp/kt
"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "la::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "build_keys" in r["Kernel_Name"]][-1]
prev_end = None
tot = {}
for r in rows[idx:]:
    n = r["Kernel_Name"].replace("void ", "").replace("la::", "").replace("(anonymous namespace)::", "").split("(")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0
    print("%-24s %8.1f us  gap %5.1f" % (n, (e - s) / 1e3, gap))
    prev_end = e
    tot[n] = tot.get(n, 0) + (e - s) / 1e3
print(tot)
print("span %.1f us" % ((prev_end - int(rows[idx]["Start_Timestamp"])) / 1e3))
