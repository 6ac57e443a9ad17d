#!/usr/bin/env python3
"""Long randomized parity sweep on the GPU (developer tool; the committed tests run a short version).
    python tools/stress_gpu.py [n_tile_seeds] [n_large_seeds]
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util
spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
tgp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tgp)
from kafka_lag_based_assignor_amd import _native as N

def main():
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    nl = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    ctx = N.Context(0)
    bad = 0
    for seed in range(100, 100 + nt):
        try:
            tgp.test_fuzz_tile_batches.__wrapped__(ctx, seed) if hasattr(tgp.test_fuzz_tile_batches, "__wrapped__") else tgp.test_fuzz_tile_batches(ctx, seed)
        except AssertionError as e:
            bad += 1; print("TILE seed", seed, "FAILED:", str(e)[:200])
    for seed in range(100, 100 + nl):
        try:
            tgp.test_fuzz_large_topics(ctx, seed)
        except AssertionError as e:
            bad += 1; print("LARGE seed", seed, "FAILED:", str(e)[:200])
    print("stress done: %d tile + %d large seeds, %d failures" % (nt, nl, bad))
    ctx.close()

if __name__ == "__main__":
    main()
