#!/usr/bin/env python3
"""K large topics in ONE call: side by side (la::large_topics_launch, the default) against one after another
(LA_FLAG_SERIAL_LARGE, round 3's form).  VERDICT r3 #4: 64 topics x 65 536 partitions x 4 096 consumers should cost about
what ONE such topic costs.  Device-resident, HIP events around settled calls; K = 8 is checked against oracle/round_form.py.

    python tools/large_many_probe.py [--partitions 65536 --consumers 4096]
"""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kafka_lag_based_assignor_amd import _native as N, synth  # noqa: E402


def device_batch(torch, dev, w, flags):
    d = {k: torch.from_numpy(np.ascontiguousarray(getattr(w, k))).to(dev) for k in
         ("part_off", "partition_id", "begin", "end", "committed", "cons_off", "cons_rank")}
    out = (torch.zeros(w.n_partitions, device=dev, dtype=torch.int32), torch.zeros(w.n_partitions, device=dev, dtype=torch.int32),
           torch.zeros(w.cons_rank.size, device=dev, dtype=torch.int64))
    b = N.DeviceBatch()
    b.n_topics, b.reset_mode, b.algo, b.flags = w.n_topics, N.LA_RESET_EARLIEST, N.LA_ALGO_AUTO, flags
    b.n_partitions, b.n_consumers = w.n_partitions, w.cons_rank.size
    b.max_partitions_per_topic, b.max_consumers_per_topic = w.max_partitions, w.max_consumers
    b.d_part_off, b.d_partition_id = d["part_off"].data_ptr(), d["partition_id"].data_ptr()
    b.d_begin_off, b.d_end_off, b.d_committed_off = d["begin"].data_ptr(), d["end"].data_ptr(), d["committed"].data_ptr()
    b.d_cons_off, b.d_cons_rank = d["cons_off"].data_ptr(), d["cons_rank"].data_ptr()
    b.d_out_partition, b.d_out_member_rank, b.d_out_total_lag = out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr()
    po, co = np.ascontiguousarray(w.part_off), np.ascontiguousarray(w.cons_off)
    b.h_part_off = po.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    b.h_cons_off = co.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    return b, d, out, (po, co)


def timed(torch, ctx, b, stream, calls):
    for _ in range(2):
        ctx.assign_batch_device(b, stream)
    ctx.sync(stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(calls):
        ctx.assign_batch_device(b, stream)
    e1.record()
    ctx.sync(stream)
    return e0.elapsed_time(e1) / calls


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--partitions", type=int, default=65536)
    ap.add_argument("--consumers", type=int, default=4096)
    a = ap.parse_args()
    import torch
    from oracle import oracle
    from oracle.round_form import round_form
    dev = torch.device("cuda", 0)
    ctx = N.Context(0)
    stream = torch.cuda.current_stream().cuda_stream
    print("%d partitions x %d consumers per topic, device-resident, earliest mode" % (a.partitions, a.consumers))
    base = None
    for dist in ("uniform40", "pareto"):
        for k in (1, 8, 64):
            w = synth.make_uniform("many", 30 + k, k, a.partitions, a.consumers, dist)
            row = []
            for name, flags in (("side by side", 0), ("serial", N.LA_FLAG_SERIAL_LARGE)):
                b, d, out, keep = device_batch(torch, dev, w, flags)
                ms = timed(torch, ctx, b, stream, 20 if k * (1 if flags == 0 else k) <= 64 else 5)
                row.append((name, ms))
                if k == 8 and flags == 0:
                    lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
                    t0 = time.perf_counter()
                    e = round_form(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
                    ok = all(np.array_equal(x.cpu().numpy(), y) for x, y in zip(out, e))
                    print("    K = 8 %s: bit-exact vs round_form: %s (checker %.1f s)" % (dist, ok, time.perf_counter() - t0))
            if k == 1:
                base = row[0][1]
            print("  %-9s K = %2d: %s  |  side by side = %.2f x one topic; %.3g partition-assignments/s"
                  % (dist, k, ", ".join("%s %.3f ms" % r for r in row), row[0][1] / base, k * a.partitions / (row[0][1] * 1e-3)))
    ctx.close()


if __name__ == "__main__":
    main()
