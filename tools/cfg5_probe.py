#!/usr/bin/env python3
"""cfg5 (1 topic x 1 048 576 partitions x 8 192 consumers) through the device entry point: per-phase times of the large
path with the sample-sorted greedy rounds, with the full network, and the parity of both against the oracle's round form.
    python tools/cfg5_probe.py [--consumers 8192] [--partitions 1048576] [--reps 5]
"""
import argparse, ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--partitions", type=int, default=1048576)
    ap.add_argument("--consumers", type=int, default=8192)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--quick", action="store_true", help="only the default form")
    ap.add_argument("--dist", default="pareto")
    args = ap.parse_args()
    import torch
    from kafka_lag_based_assignor_amd import _native as N, synth
    from round_form import round_form
    w = synth.make_uniform("cfg5", 5, 1, args.partitions, args.consumers, args.dist)
    dev = torch.device("cuda", 0)
    d = {k: torch.from_numpy(getattr(w, k)).to(dev) for k in
         ("part_off", "partition_id", "begin", "end", "committed", "cons_off", "cons_rank")}
    out_pid = torch.empty(w.n_partitions, device=dev, dtype=torch.int32)
    out_rank = torch.empty(w.n_partitions, device=dev, dtype=torch.int32)
    out_total = torch.empty(w.cons_rank.size, device=dev, dtype=torch.int64)
    ctx = N.Context(0)
    b = N.DeviceBatch()
    b.n_topics = 1; b.reset_mode = N.LA_RESET_EARLIEST; b.algo = N.LA_ALGO_AUTO
    b.n_partitions = w.n_partitions; b.n_consumers = w.cons_rank.size
    b.max_partitions_per_topic = w.max_partitions; b.max_consumers_per_topic = w.max_consumers
    b.d_part_off = d["part_off"].data_ptr(); b.d_partition_id = d["partition_id"].data_ptr()
    b.d_begin_off = d["begin"].data_ptr(); b.d_end_off = d["end"].data_ptr(); b.d_committed_off = d["committed"].data_ptr()
    b.d_cons_off = d["cons_off"].data_ptr(); b.d_cons_rank = d["cons_rank"].data_ptr()
    b.d_out_partition = out_pid.data_ptr(); b.d_out_member_rank = out_rank.data_ptr(); b.d_out_total_lag = out_total.data_ptr()
    b.h_part_off = w.part_off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    b.h_cons_off = w.cons_off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    stream = torch.cuda.current_stream().cuda_stream
    lag = np.maximum(w.end - np.where(w.committed >= 0, w.committed, w.begin), 0)
    exp = round_form(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
    forms = ((0, "default"), (N.LA_FLAG_NO_MOVED_SORT, "run merge"),
             (N.LA_FLAG_NO_MOVED_SORT | N.LA_FLAG_NO_RUN_MERGE, "sample sort"), (N.LA_FLAG_NO_SAMPLE_SORT, "full network"),
             (N.LA_FLAG_SAMPLE_TIGHT | N.LA_FLAG_NO_RUN_MERGE, "tight"))
    for flags, what in (forms[:1] if args.quick else forms):
        b.flags = flags | N.LA_FLAG_PROFILE
        ctx.assign_batch_device(b, stream); ctx.sync(stream)
        ts = []
        t0 = time.perf_counter()
        for _ in range(args.reps):
            ctx.assign_batch_device(b, stream)
            ctx.sync(stream)
            t = ctx.last_phase_times()
            ts.append((t.keys_ms, t.sort_ms, t.greedy_ms))
        wall = (time.perf_counter() - t0) / args.reps * 1e3
        ok = all(np.array_equal(a, e) for a, e in zip((out_pid.cpu().numpy(), out_rank.cpu().numpy(), out_total.cpu().numpy()), exp))
        k, s, g = np.mean(ts, axis=0)
        if hasattr(ctx._lib, "la_debug_round_clocks"):      # development build (-DLA_ROUND_CLOCKS)
            clk = (ctypes.c_ulonglong * 16)()
            ctx._lib.la_debug_round_clocks(clk, 1)
            names = ("sample sort", "bucket search", "slots+scan", "stage", "rank walk", "final order", "add+stores",
                     "moved: who stays", "moved: across waves", "moved: hand in", "moved: samples", "moved: buckets",
                     "moved: scan", "moved: stage", "moved: walk", "moved: back")
            tot = float(sum(clk[:16])) or 1.0
            print("   cycles per phase (thread 0, %d calls): " % (args.reps + 1) +
                  ", ".join("%s %.1f%%" % (n, 100.0 * clk[i] / tot) for i, n in enumerate(names) if clk[i]) +
                  "; total %.3g = %.0f per round" % (tot, tot / (args.reps + 1) / 127.0))
        if hasattr(ctx._lib, "la_debug_round_stamps") and os.environ.get("LA_ROUND_STAMPS"):
            st = (ctypes.c_ulonglong * 520)()
            ctx._lib.la_debug_round_stamps(st)
            rounds = (args.partitions + args.consumers - 1) // args.consumers
            names = {0: "in order", 1: "moved", 2: "run merge", 3: "sample", 4: "network", 9: "end"}
            # stamp q is taken when round q's bins are in order: the difference to the stamp before is round q - 1's add and stores plus round q's sort
            print("   round: cycles since the round before was in order, how this round's bins were ordered, bins that moved")
            for q in range(1, min(rounds, 258) + 1):
                print("   %3d %7d %-9s %5d" % (q, st[q] - st[q - 1], names.get(st[260 + q] >> 32, "?"), st[260 + q] & 0xFFFFFFFF))
        print("%-13s keys %.3f ms, sort %.3f ms (%d id + %d key passes), ids+greedy %.3f ms, call %.3f ms wall; bit-exact vs round form: %s"
              % (what, k, s, t.id_passes, t.key_passes, g, wall, ok))
    ctx.close()


if __name__ == "__main__":
    main()
