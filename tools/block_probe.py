"""Time batches of block-path topics (device entry, inputs resident): ms per call and assignments/s."""
import sys, time, ctypes
import numpy as np
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from kafka_lag_based_assignor_amd import _native as N

dev = torch.device("cuda", 0)
ctx = N.Context(0)
rng = np.random.default_rng(1)

def run(t, p, c, algo=N.LA_ALGO_AUTO, reps=5, lag_bits=int(__import__("os").environ.get("LAG_BITS", "34")), offsets=False):
    n, k = t * p, t * c
    part_off = np.arange(t + 1, dtype=np.int64) * p
    cons_off = np.arange(t + 1, dtype=np.int64) * c
    pid = torch.argsort(torch.rand(t, p, device=dev), dim=1).to(torch.int32).reshape(-1).contiguous()
    lag = torch.randint(0, 1 << lag_bits, (n,), device=dev, dtype=torch.int64)
    zf = float(__import__("os").environ.get("ZERO_FRAC", "0"))        # a consumer group that has caught up on this share of the partitions
    if zf > 0:
        lag = torch.where(torch.rand(n, device=dev) < zf, torch.zeros_like(lag), lag)
    ranks = torch.arange(c, device=dev, dtype=torch.int32).repeat(t).contiguous()
    d_po, d_co = torch.from_numpy(part_off).to(dev), torch.from_numpy(cons_off).to(dev)
    out_pid = torch.empty(n, device=dev, dtype=torch.int32); out_rank = torch.empty(n, device=dev, dtype=torch.int32)
    out_total = torch.empty(max(k, 1), device=dev, dtype=torch.int64)
    b = N.DeviceBatch()
    b.n_topics, b.reset_mode, b.algo, b.flags = t, N.LA_RESET_LATEST, algo, 0
    b.n_partitions, b.n_consumers = n, k
    b.max_partitions_per_topic, b.max_consumers_per_topic = p, c
    b.d_part_off, b.d_partition_id = d_po.data_ptr(), pid.data_ptr()
    b.d_begin_off = b.d_end_off = b.d_committed_off = None
    b.d_lag = lag.data_ptr()
    if offsets:                      # offsets in: committed (2 % none), end = committed + lag, begin = 0; earliest
        com = torch.randint(0, 1 << 20, (n,), device=dev, dtype=torch.int64)
        end = com + lag
        com = torch.where(torch.rand(n, device=dev) < 0.02, torch.full_like(com, -1), com)
        beg = torch.zeros(n, device=dev, dtype=torch.int64)
        b.d_lag = None
        b.reset_mode = N.LA_RESET_EARLIEST
        b.d_begin_off, b.d_end_off, b.d_committed_off = beg.data_ptr(), end.data_ptr(), com.data_ptr()
    b.d_cons_off, b.d_cons_rank = d_co.data_ptr(), ranks.data_ptr()
    b.d_out_partition, b.d_out_member_rank, b.d_out_total_lag = out_pid.data_ptr(), out_rank.data_ptr(), out_total.data_ptr()
    b.h_part_off = part_off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    b.h_cons_off = cons_off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    st = torch.cuda.current_stream().cuda_stream
    ctx.assign_batch_device(b, st); ctx.sync(st)
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.assign_batch_device(b, st); ctx.sync(st)
        best = min(best, time.perf_counter() - t0)
    return best

if hasattr(ctx._lib, "la_debug_block_clocks"):          # development build (-DLA_BLOCK_CLOCKS): stage times of workgroup 0
    names = ["loads", "decide", "sort", "slots + out_partition", "greedy rounds", "member ranks out", "key32: network", "key32: rest of the rounds",
             "radix: counts", "radix: prefix over wavefronts", "radix: scan", "radix: scatter", "radix: read back", "-", "-", "-"]
    shapes = [(200, 8000, 16), (200, 8000, 64), (200, 8000, 4), (1000, 2000, 16)]
    if len(sys.argv) > 1:                               # python tools/block_probe.py T,P,C [T,P,C ..]
        shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
    for (t, p, c) in shapes:
        clk = (ctypes.c_ulonglong * 16)()
        run(t, p, c, reps=1)
        ctx._lib.la_debug_block_clocks(clk, 1)
        ms = run(t, p, c, reps=9) * 1e3                 # 1 warm-up + 9 timed calls
        ctx._lib.la_debug_block_clocks(clk, 1)
        print("T=%d P=%d C=%d: %.3f ms per call; workgroup 0, us per stage (100 MHz clock): " % (t, p, c, ms) +
              ", ".join("%s %.1f" % (n, v / 10 / 100.0) for n, v in zip(names, clk) if n != "-"))
    sys.exit(0)

shapes = [(1000, 2000, 100), (5000, 200, 100), (200, 8000, 16), (2000, 1000, 500), (64, 8192, 2048),
          (1, 2000, 100), (1, 8192, 2048), (20000, 100, 65), (300, 5000, 3), (1, 100, 65), (1, 1025, 8), (20000, 300, 10)]
if len(sys.argv) > 1:                                   # python tools/block_probe.py T,P,C [T,P,C ..]
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
for (t, p, c) in shapes:
    ms = run(t, p, c) * 1e3
    ms_off = run(t, p, c, offsets=True) * 1e3
    print("T=%6d P=%5d C=%5d : %8.3f ms  %.3e assignments/s; from offsets %8.3f ms" % (t, p, c, ms, t * p / ms * 1e3, ms_off), flush=True)
if len(sys.argv) > 1:
    sys.exit(0)
print("-- P=8000 sweep over C")
for c in (2000, 300, 100, 64, 16, 4):
    ms = run(200, 8000, c) * 1e3
    print("T=200 P=8000 C=%5d : %8.3f ms (%d rounds)" % (c, ms, -(-8000 // c)), flush=True)
