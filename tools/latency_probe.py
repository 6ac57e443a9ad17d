#!/usr/bin/env python3
"""Latency of ONE host-buffer call on small, realistic batches (a Kafka group leader's rebalance is one such call).
    python tools/latency_probe.py
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kafka_lag_based_assignor_amd import _native as N, synth


def main():
    ctx = N.Context(0)
    for (t, p, c) in [(1, 3, 2), (10, 10, 3), (100, 20, 4), (1000, 50, 5), (1000, 256, 32), (10000, 64, 8)]:
        w = synth.make_uniform("lat", 20, t, p, c, "uniform40")
        a = (w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
        out = ctx.assign_batch(*a)
        for _ in range(20):
            ctx.assign_batch(*a, out=out)
        ts = []
        for _ in range(300):
            t0 = time.perf_counter()
            ctx.assign_batch(*a, out=out)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e6
        # + the grouped flow the hosts use
        tg = []
        for _ in range(100):
            t0 = time.perf_counter()
            ctx.assign_batch(*a, keep_on_device=True)
            ctx.group_last_by_member(w.n_partitions, c)
            tg.append(time.perf_counter() - t0)
        tg = np.array(tg) * 1e6
        # ... and as ONE call (la_assign_batch_grouped: the lists ride in the small call's one download)
        t1 = []
        ref = ctx.group_last_by_member(w.n_partitions, c)
        for _ in range(100):
            t0 = time.perf_counter()
            got = ctx.assign_batch_grouped(*a, c)
            t1.append(time.perf_counter() - t0)
        t1 = np.array(t1) * 1e6
        same = all(np.array_equal(x, y) for x, y in zip(got[:3], ref))
        print("%6d topics x %4d partitions x %3d consumers (%8d partitions): assign median %.1f us (p10 %.1f, p90 %.1f); "
              "assign + group_last %.1f us; assign_batch_grouped %.1f us (same lists: %s)"
              % (t, p, c, w.n_partitions, np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90), np.median(tg),
                 np.median(t1), same))
    ctx.close()


if __name__ == "__main__":
    main()
