#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc output directories into one JSON (stdout).

    python tools/pmc_parse.py DIR [DIR ...] [--cal-n N]

For every kernel: mean of each counter per dispatch and the dispatch count.  FETCH_SIZE / WRITE_SIZE are
calibrated on lag_kernel_vec2 (known bytes: 16 B read + 8 B written per partition in LATEST mode), as
MI355X_MICROARCH.md's HBM section prescribes (gfx950's FETCH_SIZE under-reports wide streaming reads);
the factors are reported and applied to the other kernels ("*_bytes_calibrated").
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name)


def main():
    args = sys.argv[1:]
    cal_n = 1 << 24
    if "--cal-n" in args:
        i = args.index("--cal-n")
        cal_n = int(args[i + 1])
        del args[i:i + 2]
    sums = defaultdict(lambda: defaultdict(float))
    counts = defaultdict(lambda: defaultdict(int))
    durs = defaultdict(list)
    for d in args:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    c = row["Counter_Name"]
                    sums[k][c] += float(row["Counter_Value"])
                    counts[k][c] += 1
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    durs[short(row["Kernel_Name"])].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    out = {"kernels": {}, "calibration": {}}
    for k in sums:
        e = {c: sums[k][c] / counts[k][c] for c in sums[k]}
        e["dispatches"] = max(counts[k].values())
        if durs.get(k):
            e["avg_ns_under_pmc"] = sum(durs[k]) / len(durs[k])
        out["kernels"][k] = e
    cal = next((k for k in out["kernels"] if "lag_kernel" in k), None)
    if cal:
        e = out["kernels"][cal]
        if e.get("FETCH_SIZE"):
            out["calibration"]["fetch_bytes_per_count"] = 16.0 * cal_n / e["FETCH_SIZE"]
        if e.get("WRITE_SIZE"):
            out["calibration"]["write_bytes_per_count"] = 8.0 * cal_n / e["WRITE_SIZE"]
        out["calibration"]["kernel"] = cal
        out["calibration"]["known_bytes"] = {"read": 16 * cal_n, "written": 8 * cal_n}
        for k, e in out["kernels"].items():
            if "FETCH_SIZE" in e and "fetch_bytes_per_count" in out["calibration"]:
                e["fetch_bytes_calibrated"] = e["FETCH_SIZE"] * out["calibration"]["fetch_bytes_per_count"]
            if "WRITE_SIZE" in e and "write_bytes_per_count" in out["calibration"]:
                e["write_bytes_calibrated"] = e["WRITE_SIZE"] * out["calibration"]["write_bytes_per_count"]
    json.dump(out, sys.stdout, indent=1, sort_keys=True)
    print()


if __name__ == "__main__":
    main()
