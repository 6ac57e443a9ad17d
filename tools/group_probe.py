#!/usr/bin/env python3
"""la_group_by_member_device alone (device-resident, HIP events): the one-workgroup form against the radix form, by size.
    python tools/group_probe.py            # LA_NO_SMALL_GROUP=1: the radix form; LA_LIB_PATH=...: another build
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kafka_lag_based_assignor_amd import _native as N


def main():
    import torch
    dev = torch.device("cuda", 0)
    ctx = N.Context(0)
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(1)
    tag = "radix form" if os.environ.get("LA_NO_SMALL_GROUP") else os.path.basename(os.environ.get("LA_LIB_PATH", "default build"))
    row = []
    for n, m in ((100, 3), (1000, 5), (2000, 5), (4096, 8), (8192, 8), (16384, 8), (16384, 4000), (30000, 8)):
        t = max(1, n // 50)
        part_off = torch.from_numpy(np.linspace(0, n, t + 1).astype(np.int64)).to(dev)
        out_p = torch.from_numpy(rng.integers(0, 1 << 20, n).astype(np.int32)).to(dev)
        out_m = torch.from_numpy(rng.integers(-1, m, n).astype(np.int32)).to(dev)
        off = torch.zeros(m + 1, dtype=torch.int64, device=dev)
        g_t = torch.zeros(n, dtype=torch.int32, device=dev)
        g_p = torch.zeros(n, dtype=torch.int32, device=dev)
        call = lambda: ctx.group_by_member_device(t, n, part_off.data_ptr(), out_p.data_ptr(), out_m.data_ptr(), m, off.data_ptr(),
                                                  g_t.data_ptr(), g_p.data_ptr(), stream)
        for _ in range(20):
            call()
        ctx.sync(stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            call()
        e1.record()
        ctx.sync(stream)
        if hasattr(ctx._lib, "la_debug_group_clocks"):      # lab build (-DLA_GROUP_CLOCKS on la_large): phase times of the one-workgroup form
            import ctypes
            clk = (ctypes.c_ulonglong * 12)()
            ctx._lib.la_debug_group_clocks(clk, 1)
            for _ in range(50):
                call()
            ctx.sync(stream)
            ctx._lib.la_debug_group_clocks(clk, 1)
            names = ["zero+loads+heads", "A ballots", "topic scan", "counts scan", "B cursors", "C places", "lists out"]
            print("   n=%d m=%d us per phase: " % (n, m) + ", ".join("%s %.2f" % (nm, clk[i] / 50 / 100.0) for i, nm in enumerate(names)))
        order = np.argsort(out_m.cpu().numpy(), kind="stable")
        ok = np.array_equal(g_p.cpu().numpy(), out_p.cpu().numpy()[order])
        row.append("%d x %d: %.1f us%s" % (n, m, e0.elapsed_time(e1) / 200 * 1e3, "" if ok else " WRONG"))
    print("%-28s %s" % (tag, " | ".join(row)))
    ctx.close()


if __name__ == "__main__":
    main()
