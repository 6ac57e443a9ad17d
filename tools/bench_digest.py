#!/usr/bin/env python3
"""The few numbers of a bench.py JSON line one looks at first (used by tools/gpu_session.sh)."""
import json
import sys


def main(path):
    lines = [l for l in open(path) if l.startswith("{")]
    if not lines:
        print("no JSON line in", path)
        return
    d = json.loads(lines[-1])
    r = d.get("roofline") or {}
    print("value %.4g  ms/step %s" % (d["value"], d["ms_per_step"]),
          {k: r.get(k) for k in ("frac", "kernel_ms", "frac_same_buffers", "no_bounds_ms", "frac_moved", "frac_cold")})
    for k in ("all_none", "latest_mode", "wire_out", "sort_phase"):
        v = r.get(k)
        if isinstance(v, dict):
            print(" ", k, {a: b for a, b in v.items() if not isinstance(b, str) or a == "error"})
    cfgs = d.get("configs") or {}
    print("  configs (ms, frac, bit_exact, sha256 == frozen, copies)", {k: (v.get("ms_per_call"), v.get("frac"), v.get("bit_exact"), v.get("sha256_matches_frozen_literal_oracle"),
                            (v.get("rotation") or {}).get("sets")) for k, v in cfgs.items() if isinstance(v, dict) and k != "caught_up"})
    cu = cfgs.get("caught_up") or {}
    if cu:
        print("  caught up (ms, ms with every partition lagging, bit_exact)", {k: (v.get("ms_per_call"), v.get("config_ms_per_call"), v.get("bit_exact"), v.get("error"))
                                                                          for k, v in cu.items() if isinstance(v, dict)})
    sc = d.get("small_call") or {}
    if "rows" in sc:
        print("  small_call", [(x["partitions"], x["gpu_call_us"], x["cpu_oracle_us"]) for x in sc["rows"]],
              [(x["partitions"], x["grouped_us"], x["cpu_oracle_us"]) for x in (sc.get("c_abi") or {}).get("rows", [])])
        cold = (sc.get("c_abi") or {}).get("cold")
        if cold:
            print("  cold small call (C ABI)", {k: v for k, v in cold.items() if k != "what"})
    sp = d.get("sort_phase") or {}
    print("  sort_phase", sp.get("kernel_ms"), sp.get("frac"), sp.get("frac_moved"), sp.get("error"),
          "| parity", d.get("parity"), "| cpu", (d.get("cpu_baseline") or {}).get("value"))
    hb = d.get("host_boundary") or {}
    print("  host", {k: hb.get(k) for k in ("ms", "pinned_ms", "sparse_begin_ms", "grouped_ms", "error")})


if __name__ == "__main__":
    main(sys.argv[1])
