#!/usr/bin/env python3
"""Can the kernels read a big batch straight out of pinned host memory (and write the results straight into it) as fast as
the copy engines move it?  The device entry point is given HOST-MAPPED pointers (la_host_alloc = hipHostMalloc: mapped,
same address on both sides): no hipMemcpy, no chunks -- the tile kernel's loads cross PCIe themselves.
    python tools/mapped_probe.py [--workload target]
"""
import argparse, ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kafka_lag_based_assignor_amd import _native as N, synth
from oracle import oracle


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="target")
    ap.add_argument("--scale", type=float, default=1.0)
    a = ap.parse_args()
    import torch
    w = synth.config(a.workload, a.scale)
    ctx = N.Context(0)
    stream = ctx.stream
    pin = {}
    for k in ("part_off", "partition_id", "begin", "end", "committed", "cons_off", "cons_rank"):
        src = np.ascontiguousarray(getattr(w, k))
        pin[k] = ctx.host_alloc(src.shape, src.dtype)
        pin[k][...] = src
    out_p = ctx.host_alloc((w.n_partitions,), np.int32)
    out_m = ctx.host_alloc((w.n_partitions,), np.int32)
    out_t = ctx.host_alloc((w.cons_rank.size,), np.int64)
    for latest in (False, True):
        b = N.DeviceBatch()
        b.n_topics, b.reset_mode, b.algo, b.flags = w.n_topics, (N.LA_RESET_LATEST if latest else N.LA_RESET_EARLIEST), N.LA_ALGO_AUTO, 0
        b.n_partitions, b.n_consumers = w.n_partitions, w.cons_rank.size
        b.max_partitions_per_topic, b.max_consumers_per_topic = w.max_partitions, w.max_consumers
        b.d_part_off, b.d_partition_id = pin["part_off"].ctypes.data, pin["partition_id"].ctypes.data
        b.d_begin_off = None if latest else pin["begin"].ctypes.data
        b.d_end_off, b.d_committed_off = pin["end"].ctypes.data, pin["committed"].ctypes.data
        b.d_cons_off, b.d_cons_rank = pin["cons_off"].ctypes.data, pin["cons_rank"].ctypes.data
        b.d_out_partition, b.d_out_member_rank, b.d_out_total_lag = out_p.ctypes.data, out_m.ctypes.data, out_t.ctypes.data
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            ctx.assign_batch_device(b, stream)
            ctx.sync(stream)
            ts.append((time.perf_counter() - t0) * 1e3)
        lag = oracle.compute_lags(w.begin, w.end, w.committed, latest)
        k = min(w.n_topics, 2000)
        e = oracle.assign_flat(w.part_off[:k + 1], w.partition_id[:w.part_off[k]], lag[:w.part_off[k]], w.cons_off[:k + 1], w.cons_rank[:w.cons_off[k]])
        ok = np.array_equal(out_p[:w.part_off[k]], e[0]) and np.array_equal(out_m[:w.part_off[k]], e[1]) and np.array_equal(out_t[:w.cons_off[k]], e[2])
        nbytes = w.n_partitions * (20 if latest else 28 * 0 + 20 + 8 * 0.01) + w.cons_rank.size * 4
        print("%s %s: kernels on host-mapped arrays %s ms per call (bit-exact on the first %d topics: %s); ~%.0f MB read over PCIe -> %.1f GB/s"
              % (a.workload, "latest" if latest else "earliest", ["%.2f" % t for t in ts], k, ok, nbytes / 1e6, nbytes / (min(ts[1:]) * 1e-3) / 1e9))
    # the copy pipeline on the same pinned arrays, for the same box
    a_ = (pin["part_off"], pin["partition_id"], pin["begin"], pin["end"], pin["committed"], N.LA_RESET_EARLIEST, pin["cons_off"], pin["cons_rank"])
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        ctx.assign_batch(*a_, out=(out_p, out_m, out_t))
        ts.append((time.perf_counter() - t0) * 1e3)
    print("la_assign_batch on the same pinned arrays (three streams): %s ms, pipeline %d" % (["%.2f" % t for t in ts], ctx.last_pipeline()))
    idx, val = N.sparse_begin(w.begin, w.committed)
    p_idx, p_val = ctx.host_alloc(idx.shape, np.int64), ctx.host_alloc(val.shape, np.int64)
    p_idx[...] = idx; p_val[...] = val
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        ctx.assign_batch_sparse(pin["part_off"], pin["partition_id"], pin["end"], pin["committed"], N.LA_RESET_EARLIEST, p_idx, p_val,
                                pin["cons_off"], pin["cons_rank"], out=(out_p, out_m, out_t))
        ts.append((time.perf_counter() - t0) * 1e3)
    print("la_assign_batch_sparse on the same pinned arrays: %s ms" % ["%.2f" % t for t in ts])
    ctx.close()


if __name__ == "__main__":
    main()
