#!/usr/bin/env python3
"""Randomized check of the rounds behind a topic's last lag (round 6: zero_tail_rounds in la_block.hip, the zero-tail search in
la_large.hip, the tile path's ordered rounds): batches of topics of every path whose lags are zero for a random share of the
partitions, against the oracle.
    python tools/stress_zero_tail.py [cases] [first seed]
"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from kafka_lag_based_assignor_amd import _native as N, synth
from oracle import oracle
from oracle.round_form import round_form
from gpu_helpers import _device_call


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    ctx = N.Context(0)
    bad = 0
    for seed in range(s0, s0 + cases):
        rng = np.random.default_rng(seed)
        t = int(rng.integers(1, 7))
        shapes = []
        for _ in range(t):
            kind = rng.integers(0, 4)
            if kind == 0: p, c = int(rng.integers(1, 1025)), int(rng.integers(1, 65))             # tile
            elif kind == 1: p, c = int(rng.integers(1025, 16385)), int(rng.integers(1, 257))      # block, one-wavefront greedy forms
            elif kind == 2: p, c = int(rng.integers(1025, 8193)), int(rng.integers(257, 2049))    # block, multi-wavefront
            else: p, c = int(rng.integers(16385, 120000)), int(rng.choice([3, 70, 600, 3000, 5000, 8192]))   # large
            shapes.append((p, c))
        part_off = np.concatenate([[0], np.cumsum([s[0] for s in shapes])]).astype(np.int64)
        cons_off = np.concatenate([[0], np.cumsum([s[1] for s in shapes])]).astype(np.int64)
        lag = np.zeros(int(part_off[-1]), np.int64)
        for i, (p, c) in enumerate(shapes):
            how = rng.integers(0, 5)
            k = [0, int(rng.integers(0, c + 2)), int(rng.integers(0, p + 1)), max(0, p - int(rng.integers(0, c + 2))),
                 (int(rng.integers(0, p // c + 1)) * c)][how]
            k = min(k, p)
            seg = np.zeros(p, np.int64)
            seg[:k] = rng.integers(1, int(rng.choice([3, 1000, 1 << 31, 1 << 40])), k)
            lag[part_off[i]:part_off[i + 1]] = rng.permutation(seg)
        pid = np.concatenate([rng.permutation(s[0]) for s in shapes]).astype(np.int32)
        ranks = np.concatenate([np.sort(rng.choice(3 * s[1] + 5, s[1], replace=False)) for s in shapes]).astype(np.int32)
        w = synth.Workload("zero tails", t, part_off, pid, np.zeros(lag.size, np.int64), lag.copy(), np.zeros(lag.size, np.int64), lag,
                           cons_off, ranks, max(s[0] for s in shapes), max(s[1] for s in shapes))
        exp = round_form(part_off, pid, lag, cons_off, ranks)
        got = _device_call(ctx, w)
        ok = all(np.array_equal(g, e) for g, e in zip(got, exp))
        if ok and seed % 10 == 0 and sum(p * c for p, c in shapes) < 2e8:
            ok = all(np.array_equal(g, e) for g, e in zip(exp, oracle.assign_flat(part_off, pid, lag, cons_off, ranks)))
        if not ok:
            bad += 1
            print("seed", seed, "shapes", shapes, "MISMATCH")
    print("zero-tail stress: %d cases from seed %d, %d failures" % (cases, s0, bad))
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
