#!/usr/bin/env python3
"""Workload for the rocprofv3 --pmc passes (run under rocprofv3, one counter group per pass).

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT/fetch -- python tools/pmc_probe.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d OUT/write -- python tools/pmc_probe.py
    python tools/pmc_parse.py OUT/fetch OUT/write > profiles/rNN_traffic.json

Launches, in this order:
  1. the calibration stream: lag_kernel_vec2 (la_compute_lag) over CAL_N partitions -- a kernel whose
     HBM bytes are known exactly (LATEST: 16 B read + 8 B written per partition; working set > 256 MiB
     so the Infinity Cache cannot hide reads), as MI355X_MICROARCH.md's HBM section prescribes;
  2. the hot path on the bench workload (--topics x --partitions x --consumers), `--launches` times.
No torch import: plain ctypes + hipMalloc through the library's own host entry points would copy;
here the device-resident entry point is used with torch tensors only to hold device memory.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CAL_N = 1 << 24


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--topics", type=int, default=100000)
    ap.add_argument("--partitions", type=int, default=256)
    ap.add_argument("--consumers", type=int, default=32)
    ap.add_argument("--reset-mode", choices=["latest", "earliest"], default="earliest")
    ap.add_argument("--algo", choices=["auto", "wide", "argmin"], default="auto")
    ap.add_argument("--launches", type=int, default=5)
    ap.add_argument("--large-partitions", type=int, default=0,
                    help="also run ONE large topic of this many partitions (radix-sort path)")
    ap.add_argument("--large-consumers", type=int, default=1024)
    ap.add_argument("--none-frac", type=float, default=None,
                    help="share of partitions without a committed offset (synth.config none_frac; default: the fixed 1 %%)")
    args = ap.parse_args()

    import torch
    import bench
    from kafka_lag_based_assignor_amd import _native as N
    from kafka_lag_based_assignor_amd import synth

    dev = torch.device("cuda", 0)
    ctx = N.Context(0)

    # 1. calibration (host buffers in, host buffer out; only the kernel dispatch is counted)
    rng = np.random.default_rng(1)
    end = rng.integers(0, 1 << 40, CAL_N).astype(np.int64)
    com = rng.integers(0, 1 << 20, CAL_N).astype(np.int64)
    for _ in range(3):
        ctx.compute_lag(None, end, com, N.LA_RESET_LATEST)
    del end, com

    # 2. the hot path
    T, P, C = args.topics, args.partitions, args.consumers
    if T > 0:
        # the bench's own vectors when the shape is the target's (synth.config), the same generator otherwise
        w = (synth.config("target", none_frac=args.none_frac) if (T, P, C) == (100000, 256, 32) else
             synth.make_uniform("custom", 11, T, P, C, "zipf", none_frac=args.none_frac))
        sh = bench.DeviceShard(torch, N, dev, w, 0, T, args.reset_mode == "latest", args.algo)
        b = sh.batch
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(args.launches):
            ctx.assign_batch_device(b, stream)
        ctx.sync(stream)

    # 3. optional: one large topic (device radix sort + one-workgroup greedy)
    if args.large_partitions > 0:
        P2, C2 = args.large_partitions, args.large_consumers
        w2 = bench.sort_phase_workload(P2, torch, dev) if C2 == 0 else synth.make_uniform("large", 12, 1, P2, C2, "uniform40")
        sh2 = bench.DeviceShard(torch, N, dev, w2, 0, 1, args.reset_mode == "latest", "auto")
        b2 = sh2.batch
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(2):
            ctx.assign_batch_device(b2, stream)
        ctx.sync(stream)
    ctx.close()
    print("pmc_probe done")


if __name__ == "__main__":
    main()
