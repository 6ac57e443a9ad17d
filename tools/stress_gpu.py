#!/usr/bin/env python3
"""Long randomized parity sweep on the GPU (developer tool; the committed tests run a short version).
    python tools/stress_gpu.py [n_tile_seeds] [n_large_seeds] [n_block_seeds]
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
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 300
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
    for seed in range(100, 100 + nb):
        try:
            tgp.test_fuzz_block_topics(ctx, seed)
        except AssertionError as e:
            bad += 1; print("BLOCK seed", seed, "FAILED:", str(e)[:200])
    for seed in range(100, 100 + nb // 10):
        rng = np.random.default_rng(seed)
        try:
            tgp.test_block_batches_mixed_with_tile_and_large_topics(ctx, seed, int(rng.choice([600, 1500, 3000, 9000])),
                                                                    int(rng.choice([70, 100, 300, 2500])))
            tgp.test_ragged_tile_batch_by_shape_class(ctx, int(rng.integers(300, 9000)), int(rng.choice([2, 3, 50])))
        except AssertionError as e:
            bad += 1; print("MIXED seed", seed, "FAILED:", str(e)[:200])
    # large path with more than 1 024 consumers: sample-sorted greedy rounds, full network, both interleaved
    ns = int(sys.argv[4]) if len(sys.argv) > 4 else 60
    for seed in range(100, 100 + ns):
        rng = np.random.default_rng(seed)
        c = int(rng.integers(1025, 8193))
        p = int(rng.integers(c, min(40 * c, int(4e8) // c) + 1))
        kind = str(rng.choice(["u40", "ties", "zero", "u63", "pareto"]))
        try:
            tgp.test_large_sample_sort_rounds(ctx, p, c, kind)
        except AssertionError as e:
            bad += 1; print("SAMPLE seed", seed, "p", p, "c", c, kind, "FAILED:", str(e)[:200])
    print("stress done: %d tile + %d large + %d block + %d sample-sort seeds, %d failures" % (nt, nl, nb, ns, bad))
    ctx.close()

if __name__ == "__main__":
    main()
