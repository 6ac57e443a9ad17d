#!/usr/bin/env python3
"""What a greedy round of cfg5 (1 topic x 1 048 576 partitions x 8 192 consumers, Pareto lags) looks like, on the CPU: per round
the number of ascending runs of the bins after the add, how many bins change places when the round is sorted, the largest
distance a bin travels and where the first descents are.  The numbers behind moved_sort_bins / merge_runs_bins (la_large.hip).
    python tools/cfg5_rounds.py [--config cfg5] > profiles/archive/r04_cfg5_rounds.txt
"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kafka_lag_based_assignor_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg5")
    args = ap.parse_args()
    w = synth.config(args.config)
    lag = np.maximum(w.end - np.where(w.committed >= 0, w.committed, w.begin), 0)
    order = np.lexsort((w.partition_id, ~lag))
    sl = lag[order]
    C = int(w.cons_off[1])
    P = sl.size
    tot = np.zeros(C, dtype=np.int64)
    idx = np.arange(C)
    print("# %s: %d partitions, %d consumers, lags %d .. %d" % (args.config, P, C, sl[-1], sl[0]))
    print("# row q = the bins AFTER round q's add, i.e. what round q + 1 has to sort\n# round: runs, distinct lags, bins handed to the small sort (not a prefix maximum and suffix minimum), bins that change places\n#        (first .. last position), largest distance, first descents")
    for q in range((P + C - 1) // C):
        L = sl[q * C:(q + 1) * C]
        o = np.lexsort((idx, tot))
        tot, idx = tot[o], idx[o]
        tot[:L.size] += L
        key = tot * C + idx
        fin = np.argsort(key, kind="stable")
        moved = np.nonzero(fin != np.arange(C))[0]
        # what moved_sort_bins hands in: every bin that is not above all bins before it and below all bins behind it
        pm = np.maximum.accumulate(np.concatenate(([-1], key[:-1])))
        sm = np.minimum.accumulate(np.concatenate((key[1:], [np.iinfo(np.int64).max]))[::-1])[::-1]
        handed = int(((key < pm) | (key > sm)).sum())
        d = np.nonzero(key[1:] < key[:-1])[0]
        print("%3d: %4d runs, %4d distinct lags, %4d handed in, %4d move (%s), distance <= %d, descents at %s" %
              (q, d.size + 1, np.unique(L).size, handed, moved.size,
               "%d .. %d" % (moved.min(), moved.max()) if moved.size else "-",
               int(np.abs(fin - np.arange(C)).max()), d[:8].tolist()))


if __name__ == "__main__":
    main()
