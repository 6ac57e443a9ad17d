#!/usr/bin/env python3
"""Copies what should be judged from gpurun_out/<tag>/ (scratch) into profiles/ (tracked) and writes
profiles/traffic.json, the file bench.py reads its roofline.traffic from.

    python tools/publish_profiles.py r01d r01_d

FETCH_SIZE / WRITE_SIZE come from separate rocprofv3 --pmc passes (tools/gpu_session.sh) and are calibrated
on lag_kernel_vec2, whose HBM bytes are known exactly (MI355X_MICROARCH.md, HBM section).
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag, name = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    # kernel stats (truncate torch's kilometre-long kernel names)
    for f in glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv")):
        rows = list(csv.reader(open(f)))
        with open(os.path.join(dst, name + "_kernel_stats.csv"), "w", newline="") as fh:
            w = csv.writer(fh, quoting=csv.QUOTE_ALL)
            for r in rows:
                r[0] = r[0][:160]
                w.writerow(r)
    for b in ("bench", "bench_driver", "bench_sort", "bench_strong_cfg4", "bench_latest", "bench_wide", "bench_cfg4"):
        p = os.path.join(src, b + ".json")
        if os.path.exists(p):
            lines = [l for l in open(p) if l.startswith("{")]
            if lines:
                open(os.path.join(dst, "%s_%s.json" % (name, b)), "w").write(lines[-1])
    # the other steps of tools/gpu_session.sh: kernel stats of the cfg5 / block / sort probes, text records
    for sub, label in (("stats_cfg5", "cfg5"), ("stats_block", "block"), ("stats_sort", "sort")):
        for f in glob.glob(os.path.join(src, sub, "*", "*kernel_stats.csv")):
            rows = list(csv.reader(open(f)))
            with open(os.path.join(dst, "%s_%s_kernel_stats.csv" % (name, label)), "w", newline="") as fh:
                w = csv.writer(fh, quoting=csv.QUOTE_ALL)
                for r in rows:
                    r[0] = r[0][:160]
                    w.writerow(r)
    for txt, out in (("cfg5_probe.txt", "cfg5_probe.txt"), ("block_probe.txt", "block_probe.txt"), ("stress.txt", "stress.txt"),
                     ("latency.txt", "latency_probe.txt"), ("latency_c.txt", "latency_c.txt"), ("sort_timeline.txt", "sort_timeline.txt"),
                     ("bench_digest.txt", "bench_digest.txt"), ("pmc_block_summary.json", "block_pmc_summary.json")):
        if os.path.exists(os.path.join(src, txt)):
            shutil.copyfile(os.path.join(src, txt), os.path.join(dst, "%s_%s" % (name, out)))
    if os.path.exists(os.path.join(src, "pytest.log")):
        tail = [l for l in open(os.path.join(src, "pytest.log")) if any(k in l for k in ("passed", "failed", "FAILED", "rror", "pytest exit"))]
        open(os.path.join(dst, name + "_gpu_tests.txt"), "w").write("".join(tail[-20:]))
    if not os.path.exists(os.path.join(src, "pmc_summary.json")):
        return
    summ = json.load(open(os.path.join(src, "pmc_summary.json")))
    ours = {k: v for k, v in summ["kernels"].items() if "la::" in k}
    json.dump({"calibration": summ["calibration"], "kernels": ours},
              open(os.path.join(dst, name + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)
    # traffic.json: the packed tile kernel on the bench's default workload (+ the empty wide kernel)
    entries = []
    pk = next((v for k, v in ours.items() if "wave_tile_packed_kernel<32, 8" in k), None)
    wk = next((v for k, v in ours.items() if "wave_tile_wide_kernel<32, 8, false>" in k), None)
    if pk and "fetch_bytes_calibrated" in pk and "write_bytes_calibrated" in pk:
        hbm = pk["fetch_bytes_calibrated"] + pk["write_bytes_calibrated"]
        if wk:
            hbm += wk.get("fetch_bytes_calibrated", 0) + wk.get("write_bytes_calibrated", 0)
        entries.append({
            "topics": 100000, "partitions": 256, "consumers": 32, "reset_mode": "earliest", "algo": "auto",
            "hbm_bytes_per_launch": round(hbm),
            "read_bytes": round(pk["fetch_bytes_calibrated"]), "written_bytes": round(pk["write_bytes_calibrated"]),
            "source": "profiles/%s_pmc_summary.json: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate "
                      "passes (tools/gpu_session.sh), calibrated on lag_kernel_vec2's known bytes "
                      "(fetch x%.1f B/count, write x%.1f B/count)" % (
                          name, summ["calibration"]["fetch_bytes_per_count"],
                          summ["calibration"]["write_bytes_per_count"]),
        })
    # keep what other sessions put there (the sort-phase entry of tools/gpu_session.sh ... sort)
    try:
        old = json.load(open(os.path.join(dst, "traffic.json"))).get("entries", [])
    except (OSError, ValueError):
        old = []
    entries += [e for e in old if e.get("kind") == "sort_phase"]
    json.dump({"entries": entries}, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    print(json.dumps(entries, indent=1))


if __name__ == "__main__":
    main()
