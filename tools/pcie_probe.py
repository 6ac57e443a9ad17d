#!/usr/bin/env python3
"""What the host link of this box sustains: pinned and pageable H2D / D2H copies of the target batch's sizes, alone and
both directions at once -- the floor of the host-buffer entry point (la_assign_batch)."""
import time
import torch

dev = torch.device("cuda", 0)
H2D, D2H = 717 * 10**6, 230 * 10**6


def bench(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    return best


for pinned in (True, False):
    h_in = torch.empty(H2D, dtype=torch.uint8).pin_memory() if pinned else torch.empty(H2D, dtype=torch.uint8)
    h_in.fill_(1)
    h_out = torch.empty(D2H, dtype=torch.uint8).pin_memory() if pinned else torch.empty(D2H, dtype=torch.uint8)
    h_out.fill_(1)
    d_in = torch.empty(H2D, dtype=torch.uint8, device=dev)
    d_out = torch.empty(D2H, dtype=torch.uint8, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    t = bench(lambda: d_in.copy_(h_in, non_blocking=True))
    print("%s H2D %d MB: %.2f ms = %.1f GB/s" % ("pinned" if pinned else "pageable", H2D // 10**6, t * 1e3, H2D / t / 1e9))
    t = bench(lambda: h_out.copy_(d_out, non_blocking=True))
    print("%s D2H %d MB: %.2f ms = %.1f GB/s" % ("pinned" if pinned else "pageable", D2H // 10**6, t * 1e3, D2H / t / 1e9))

    def both():
        with torch.cuda.stream(s1):
            d_in.copy_(h_in, non_blocking=True)
        with torch.cuda.stream(s2):
            h_out.copy_(d_out, non_blocking=True)
    t = bench(both)
    print("%s both directions at once: %.2f ms (H2D alone would be the floor)" % ("pinned" if pinned else "pageable", t * 1e3))
    if pinned:
        # chunked: 16 chunks on two alternating streams (what a pipeline does)
        n = 16
        step = H2D // n

        def chunked():
            for i in range(n):
                with torch.cuda.stream(s1 if i % 2 == 0 else s2):
                    d_in[i * step:(i + 1) * step].copy_(h_in[i * step:(i + 1) * step], non_blocking=True)
        t = bench(chunked)
        print("pinned H2D in %d chunks on 2 streams: %.2f ms = %.1f GB/s" % (n, t * 1e3, H2D / t / 1e9))
