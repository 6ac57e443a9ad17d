"""Ragged tile batch: single launch at the widest shape vs shape classes (LA_FLAG_RAGGED)."""
import sys, time, ctypes
import numpy as np
import torch
sys.path.insert(0, ".")
from kafka_lag_based_assignor_amd import _native as N

dev = torch.device("cuda", 0)
ctx = N.Context(0)

def batch(t, big_every, seed=1):
    rng = np.random.default_rng(seed)
    ps = rng.integers(1, 40, t); cs = rng.integers(1, 7, t)
    mid = rng.random(t) < 0.1
    ps[mid] = rng.integers(65, 256, int(mid.sum())); cs[mid] = rng.integers(1, 33, int(mid.sum()))
    if big_every:
        ps[::big_every] = rng.integers(600, 1025, ps[::big_every].size); cs[::big_every] = rng.integers(1, 65, cs[::big_every].size)
    return ps, cs

def run(ps, cs, flags, reps=5):
    t = ps.size
    part_off = np.concatenate([[0], np.cumsum(ps)]).astype(np.int64); cons_off = np.concatenate([[0], np.cumsum(cs)]).astype(np.int64)
    n, k = int(part_off[-1]), int(cons_off[-1])
    pid = torch.from_numpy(np.concatenate([np.arange(p, dtype=np.int32)[::-1] for p in ps])).to(dev)
    torch.manual_seed(5)
    lag = torch.randint(0, 1 << 34, (n,), device=dev, dtype=torch.int64)
    ranks = torch.from_numpy(np.concatenate([np.arange(c, dtype=np.int32) for c in cs])).to(dev)
    d_po, d_co = torch.from_numpy(part_off).to(dev), torch.from_numpy(cons_off).to(dev)
    out_pid = torch.empty(n, device=dev, dtype=torch.int32); out_rank = torch.empty(n, device=dev, dtype=torch.int32)
    out_total = torch.empty(max(k, 1), device=dev, dtype=torch.int64)
    b = N.DeviceBatch()
    b.n_topics, b.reset_mode, b.algo, b.flags = t, N.LA_RESET_LATEST, N.LA_ALGO_AUTO, flags
    b.n_partitions, b.n_consumers = n, k
    b.max_partitions_per_topic, b.max_consumers_per_topic = int(ps.max()), int(cs.max())
    b.d_part_off, b.d_partition_id = d_po.data_ptr(), pid.data_ptr()
    b.d_begin_off = b.d_end_off = b.d_committed_off = None
    b.d_lag = lag.data_ptr()
    b.d_cons_off, b.d_cons_rank = d_co.data_ptr(), ranks.data_ptr()
    b.d_out_partition, b.d_out_member_rank, b.d_out_total_lag = out_pid.data_ptr(), out_rank.data_ptr(), out_total.data_ptr()
    b.h_part_off = part_off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    b.h_cons_off = cons_off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    st = torch.cuda.current_stream().cuda_stream
    ctx.assign_batch_device(b, st); ctx.sync(st)
    best = 1e9; enq = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ctx.assign_batch_device(b, st); t1 = time.perf_counter(); ctx.sync(st)
        best = min(best, time.perf_counter() - t0); enq = min(enq, t1 - t0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); ctx.assign_batch_device(b, st); e1.record(); torch.cuda.synchronize()
    print("   flags=%d enqueue %.3f ms, gpu (events) %.3f ms" % (flags, enq * 1e3, e0.elapsed_time(e1)))
    return best * 1e3, n, (out_pid.clone(), out_rank.clone(), out_total.clone())

for t, be in [(100000, 100), (100000, 0), (1000000, 1000), (20000, 3)]:
    ps, cs = batch(t, be)
    m0, n, r0 = run(ps, cs, 0)
    m1, n, r1 = run(ps, cs, N.LA_FLAG_RAGGED)
    same = all(torch.equal(a, b) for a, b in zip(r0[:2], r1[:2]))
    print("T=%7d big every %4d  N=%9d : one shape %.3f ms, shape classes %.3f ms (%.1fx), outputs equal %s" % (t, be, n, m0, m1, m0 / m1, same), flush=True)
