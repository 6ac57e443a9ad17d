#!/usr/bin/env python3
"""Soak of the zero-copy host calls: the same few workloads through la_assign_batch_grouped / la_assign_batch over and over, every
result compared with the oracle's (computed once).  What it is after: an ordering slip between the kernels' stores into mapped
host memory, the fused end of the call (last workgroup / last block stores the completion word) and the spinning host thread
would show up as a rare wrong or stale word, not as a failing unit test.
    python tools/soak_small_calls.py [seconds]
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kafka_lag_based_assignor_amd import _native as N, synth
from oracle import oracle


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
    ctx = N.Context(0)
    shapes = [(1, 3, 2), (10, 10, 3), (40, 50, 5), (100, 20, 4), (7, 300, 33), (2, 1000, 64), (30, 100, 8), (100, 100, 8),
              (300, 100, 8), (1000, 16, 4), (3, 4000, 100), (1000, 50, 5)]
    work = []
    for i, (t, p, c) in enumerate(shapes):
        w = synth.make_uniform("soak", 900 + i, t, p, c, "uniform40" if i % 2 else "zipf")
        lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
        e_pid, e_rank, e_tot = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
        order = np.argsort(e_rank, kind="stable")
        n_members = c + 1
        first = np.searchsorted(e_rank[order], np.arange(n_members + 1)).astype(np.int64)
        topic = (np.searchsorted(w.part_off, order, side="right") - 1).astype(np.int32)
        a = (w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
        hb = N.offset_bounds(w.begin, w.end, w.committed, w.partition_id)
        work.append((a, n_members, (first, topic, e_pid[order], e_tot), (e_pid, e_rank, e_tot), hb, w.n_partitions))
    calls = bad = 0
    t_end = time.time() + seconds
    rng = np.random.default_rng(1)
    while time.time() < t_end:
        for _ in range(200):
            a, n_members, want_g, want_a, hb, n = work[int(rng.integers(0, len(work)))]
            kind = int(rng.integers(0, 3))
            if hb is not None and rng.random() < 0.5:
                ctx.hint_next_call(hb)
            if kind < 2:
                g = ctx.assign_batch_grouped(*a, n_members)
                ok = all(np.array_equal(x, y) for x, y in zip(g, want_g))
            else:
                r = ctx.assign_batch(*a)
                ok = all(np.array_equal(x, y) for x, y in zip(r, want_a))
            calls += 1
            if not ok:
                bad += 1
                print("MISMATCH at call", calls, "partitions", n, "kind", kind, "pipeline", ctx.last_pipeline(), flush=True)
    print("soak: %d calls in %.0f s, %d mismatches" % (calls, seconds, bad))
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
