#!/usr/bin/env python3
"""Latency of ONE host-buffer call on small, realistic batches (a Kafka group leader's rebalance is one such call), with the C
oracle's time for the same call on one host core beside it, and which pipeline the library chose.
    python tools/latency_probe.py            # LA_ZERO_COPY_BYTES=0 python tools/latency_probe.py: the copying form for A/B
"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kafka_lag_based_assignor_amd import _native as N, synth
from oracle import oracle

PIPE = {0: "one copy", 1: "lanes", 2: "streams", 3: "zero copy", 4: "mapped"}


def med(f, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        r = f()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e6, r


def main():
    ctx = N.Context(0)
    print("zero-copy threshold: LA_ZERO_COPY_BYTES=%s" % os.environ.get("LA_ZERO_COPY_BYTES", "default (12 MB)"))
    rows = [(1, 3, 2), (10, 10, 3), (40, 50, 5), (100, 20, 4), (100, 100, 8), (1000, 16, 4), (1000, 50, 5), (1000, 256, 32), (10000, 64, 8)]
    if os.environ.get("LAT_ROWS"):                     # e.g. LAT_ROWS=100x100x8,1000x50x5
        rows = [tuple(int(v) for v in r.split("x")) for r in os.environ["LAT_ROWS"].split(",")]
    for (t, p, c) in rows:
        w = synth.make_uniform("lat", 20, t, p, c, "uniform40")
        a = (w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
        out = ctx.assign_batch(*a)
        for _ in range(20):
            ctx.assign_batch(*a, out=out)
        reps = 300 if w.n_partitions <= 50000 else 60
        t_assign, _ = med(lambda: ctx.assign_batch(*a, out=out), reps)
        pipe = ctx.last_pipeline()
        t_two, ref = med(lambda: (ctx.assign_batch(*a, keep_on_device=True), ctx.group_last_by_member(w.n_partitions, c))[1], reps // 3)
        t_grouped, got = med(lambda: ctx.assign_batch_grouped(*a, c), reps // 3)
        idx, val = N.sparse_begin(w.begin, w.committed)
        t_sparse, got_s = med(lambda: ctx.assign_batch_grouped_sparse(w.part_off, w.partition_id, w.end, w.committed, N.LA_RESET_EARLIEST,
                                                                    idx, val, w.cons_off, w.cons_rank, c), reps // 3)
        same = all(np.array_equal(x, y) for x, y in zip(got[:3], ref)) and all(np.array_equal(x, y) for x, y in zip(got, got_s))

        def cpu():
            lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
            e = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
            return e, np.argsort(e[1], kind="stable")
        t_cpu, _ = med(cpu, max(5, reps // 10))
        # the same calls on pinned arrays from la_host_alloc -- what the Java host's direct ByteBuffers are made of -- with the
        # marshaller's bounds hinted before every call (la_hint_next_call), results into pinned arrays as well
        pa = [x if isinstance(x, int) else ctx.host_alloc(x.shape, x.dtype) for x in a]
        for dst, src in zip(pa, a):
            if not isinstance(src, int):
                dst[...] = src
        pout = tuple(ctx.host_alloc(o.shape, o.dtype) for o in out)
        gout = (ctx.host_alloc((c + 1,), np.int64), ctx.host_alloc((w.n_partitions,), np.int32), ctx.host_alloc((w.n_partitions,), np.int32))
        hb = N.offset_bounds(w.begin, w.end, w.committed, w.partition_id)

        def pinned():
            ctx.hint_next_call(hb)
            return ctx.assign_batch(*pa, out=pout)

        def pinned_grouped():
            ctx.hint_next_call(hb)
            ctx.assign_batch(*pa, keep_on_device=True, want_totals=False)       # (totals would come back in a pageable array)
            return ctx.group_last_by_member(w.n_partitions, c, out=gout)
        def pinned_grouped_call():                     # what the Java host calls: one grouped call on its direct buffers
            ctx.hint_next_call(hb)
            return ctx.assign_batch_grouped(*pa, c, want_totals=False)   # (totals would come back in a pageable array)
        for _ in range(5):
            pinned()
        t_pin, got_p = med(pinned, reps)
        pipe_p, launches_p = ctx.last_pipeline(), ctx.last_launches()
        t_pin_g, got_pg = med(pinned_grouped, reps // 3)
        t_pin_gc, got_pgc = med(pinned_grouped_call, reps // 3)
        pipe_gc = ctx.last_pipeline()
        same = same and all(np.array_equal(x, y) for x, y in zip(got_pgc[:3], ref))
        same = same and all(np.array_equal(x, y) for x, y in zip(got_p, out)) and all(np.array_equal(x, y) for x, y in zip(got_pg, ref))
        print("%6d topics x %4d partitions x %3d consumers (%8d partitions) [%s]: assign %.1f us; assign + group_last %.1f us; "
              "assign_batch_grouped %.1f us (sparse begin %.1f us; same lists: %s); pinned arrays [%s, %d launches]: assign %.1f us, "
              "assign + group_last %.1f us, assign_batch_grouped %.1f us [%s]; C oracle + sort by member on one core %.1f us"
              % (t, p, c, w.n_partitions, PIPE.get(pipe, pipe), t_assign, t_two, t_grouped, t_sparse, same, PIPE.get(pipe_p, pipe_p),
                 launches_p, t_pin, t_pin_g, t_pin_gc, PIPE.get(pipe_gc, pipe_gc), t_cpu))
    ctx.close()


if __name__ == "__main__":
    main()
