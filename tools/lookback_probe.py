#!/usr/bin/env python3
"""How far the look-back walks of the single-kernel radix passes go (development build with -DLA_LOOKBACK_STATS):
walks per sort, hops per walk, empty polls, the longest walk, cycles spent walking.
    LA_EXTRA_HIPCC_FLAGS=-DLA_LOOKBACK_STATS python -m kafka_lag_based_assignor_amd.build --force
    python tools/lookback_probe.py [--partitions 33554432] [--reps 5]
"""
import argparse, ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--partitions", type=int, default=1 << 25)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import torch
    import bench
    from kafka_lag_based_assignor_amd import _native as N
    dev = torch.device("cuda", 0)
    ctx = N.Context(0)
    lib = N.load()
    stream = torch.cuda.current_stream().cuda_stream
    stats = (ctypes.c_ulonglong * 8)()
    have = hasattr(lib, "la_debug_lookback_stats")
    if have:
        lib.la_debug_lookback_stats(stats, 1)
    clk = (ctypes.c_ulonglong * 12)()
    have_clk = hasattr(lib, "la_debug_sweep_clocks")          # -DLA_SWEEP_CLOCKS
    if have_clk:
        lib.la_debug_sweep_clocks(clk, 1)
    sp = bench.run_sort_phase(torch, N, ctx, dev, args.partitions, args.reps, stream, "single")
    print("sort phase: %.4f ms, frac %.4f, sorted_ok %s" % (sp["kernel_ms"], sp["frac"], sp["sorted_ok"]))
    if have:
        lib.la_debug_lookback_stats(stats, 0)
        walks, hops, empty, longest, cycles = (int(stats[i]) for i in range(5))
        print("walks %d  hops/walk %.2f  empty polls/walk %.2f  longest walk %d hops  cycles/walk %.0f (%.2f us at 100 MHz clock64)"
              % (walks, hops / max(walks, 1), empty / max(walks, 1), longest, cycles / max(walks, 1), cycles / max(walks, 1) / 100.0))
    if have_clk:
        lib.la_debug_sweep_clocks(clk, 0)
        tiles = max(int(clk[11]), 1)
        names = ["ticket + loads issued", "loads landed + ranks (wavefront 0)", "... slowest wavefront", "counts, bin starts, stage 1",
                 "walk (wavefront 0)", "... slowest walk", "scatter 1, stage 2, scatter 2"]
        print("tiles %d; us per tile (thread 0, 100 MHz clock): " % tiles +
              "; ".join("%s %.2f" % (n, int(clk[i]) / tiles / 100.0) for i, n in enumerate(names)) +
              "; total %.2f" % (sum(int(clk[i]) for i in range(7)) / tiles / 100.0))
    ctx.close()


if __name__ == "__main__":
    main()
