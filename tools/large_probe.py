#!/usr/bin/env python3
"""One large topic through the device-resident entry point (radix-sort path), for rocprofv3.
    python tools/large_probe.py --partitions 33554432 --consumers 1024 [--launches 3] [--check]
"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--partitions", type=int, default=1 << 25)
    ap.add_argument("--consumers", type=int, default=1024)
    ap.add_argument("--launches", type=int, default=3)
    ap.add_argument("--dist", default="uniform40")
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    import torch
    import bench
    from kafka_lag_based_assignor_amd import _native as N
    dev = torch.device("cuda", 0)
    ctx = N.Context(0)
    P, C = args.partitions, args.consumers
    from kafka_lag_based_assignor_amd import synth
    hw = bench.sort_phase_workload(P, torch, dev) if C == 0 else synth.make_uniform("large", 12, 1, P, C, args.dist)
    sh = bench.DeviceShard(torch, N, dev, hw, 0, 1, False, "auto")
    b = sh.batch
    stream = torch.cuda.current_stream().cuda_stream
    ctx.assign_batch_device(b, stream); ctx.sync(stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.launches):
        ctx.assign_batch_device(b, stream)
    ctx.sync(stream)
    dt = (time.perf_counter() - t0) / args.launches
    print("large topic: %d partitions x %d consumers: %.3f ms per launch, %.3g assignments/s" % (P, C, dt * 1e3, P / dt))
    if args.check:
        from oracle import oracle
        lag = oracle.compute_lags(hw.begin, hw.end, hw.committed, False)
        e = oracle.assign_flat(hw.part_off, hw.partition_id, lag, hw.cons_off, hw.cons_rank)
        got = (sh.out_pid[:P].cpu().numpy(), sh.out_rank[:P].cpu().numpy(), sh.out_total[:C].cpu().numpy())
        ok = all(np.array_equal(a, b) for a, b in zip(e, got))
        print("bit-exact vs oracle:", ok)
    ctx.close()


if __name__ == "__main__":
    main()
