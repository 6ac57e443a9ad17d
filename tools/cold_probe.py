#!/usr/bin/env python3
"""What a REAL rebalance pays: one host-buffer call after the GPU idled (a group leader rebalances once in a while, not back
to back).  For each batch: the warm median (back-to-back calls), the call after `idle` seconds of nothing, and the same cold
call when la_wake went out `lead` milliseconds earlier (what a host issues at the top of assign(), before the broker round
trips that fetch the offsets); controls: the host core kept busy with the device idle, a spin kernel on another stream.
    python tools/cold_probe.py [--idle 0.2 1.0] [--lead 0.5 2 10]
"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kafka_lag_based_assignor_amd import _native as N, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--idle", type=float, nargs="*", default=[0.05, 0.3, 1.0])
    ap.add_argument("--lead", type=float, nargs="*", default=[0.2, 1.0, 5.0])
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--wake", choices=["wake", "lag", "call"], default="wake")
    ap.add_argument("--spin", type=float, nargs="*", default=[100, 1000, 4000], help="device spin kernels of this many us as the wake-up")
    args = ap.parse_args()
    ctx = N.Context(0)
    one = np.array([5], np.int64)
    dummy = (np.array([0, 1], np.int64), np.array([0], np.int32), np.array([0], np.int64), one, np.array([1], np.int64), N.LA_RESET_EARLIEST,
             np.array([0, 1], np.int64), np.array([0], np.int32))
    if args.spin:
        import torch
        torch.zeros(1, device="cuda")
    for (t, p, c) in [(10, 10, 3), (100, 20, 4), (1000, 256, 32)]:
        w = synth.make_uniform("cold", 21, t, p, c, "uniform40")
        a = (w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)

        def call():
            t0 = time.perf_counter()
            ctx.assign_batch_grouped(*a, c)
            return (time.perf_counter() - t0) * 1e6
        for _ in range(30):
            call()
        warm = float(np.median([call() for _ in range(100)]))
        line = "%7d partitions (%d x %d x %d): warm %6.1f us" % (w.n_partitions, t, p, c, warm)
        for idle in args.idle:
            cold = []
            for _ in range(args.reps):
                time.sleep(idle)
                cold.append(call())
            line += " | idle %.2fs: %6.1f" % (idle, float(np.median(cold)))
        idle = args.idle[-1]
        # controls: the host alone kept busy for `lead` ms after the idle time (no device work): what of the cold penalty is the
        # host's (a sleeping core, cold caches); and a device kept busy for a while (torch's spin kernel) instead of one tiny kernel
        for lead in args.lead[:2]:
            ctl = []
            for _ in range(args.reps):
                time.sleep(idle)
                t_w = time.perf_counter()
                while (time.perf_counter() - t_w) * 1e3 < lead:
                    pass
                ctl.append(call())
            line += " | host busy %.1f ms, device idle: %6.1f" % (lead, float(np.median(ctl)))
        if args.spin:
            import torch
            for spin_us in args.spin:
                for lead in (max(args.lead[0], spin_us / 1000.0 + 0.1), 5.0):
                    sp = []
                    for _ in range(args.reps):
                        time.sleep(idle)
                        torch.cuda._sleep(int(spin_us * 2100))              # ~cycles at 2.1 GHz, asynchronous
                        t_w = time.perf_counter()
                        while (time.perf_counter() - t_w) * 1e3 < lead:
                            pass
                        sp.append(call())
                    line += " | device spun %d us, call %.1f ms after its launch: %6.1f" % (spin_us, lead, float(np.median(sp)))
        for lead in args.lead:
            woke = []
            for _ in range(args.reps):
                time.sleep(idle)
                if args.wake == "call":
                    ctx.assign_batch_grouped(*dummy, 1)                     # a one-partition rebalance through the real path, waited for
                elif args.wake == "lag":
                    ctx.compute_lag(None, one, one, N.LA_RESET_LATEST)      # (round 6's first probe: one tiny kernel + its copies + a wait)
                else:
                    ctx.wake()                                              # la_wake
                t_w = time.perf_counter()
                while (time.perf_counter() - t_w) * 1e3 < lead:
                    pass
                woke.append(call())
            line += " | woken %.1f ms before: %6.1f" % (lead, float(np.median(woke)))
        print(line, flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
