"""Per-call wall time of la_assign_batch on the target batch in a FRESH process: which calls pay what.

    python tools/pin_seq_probe.py [--pageable] [--tiny-first N] [--gap SECONDS] [--calls N]
"""
import argparse, sys, time
import numpy as np
sys.path.insert(0, ".")
from kafka_lag_based_assignor_amd import _native as N, synth

ap = argparse.ArgumentParser()
ap.add_argument("--pageable", action="store_true")
ap.add_argument("--tiny-first", type=int, default=0)
ap.add_argument("--gap", type=float, default=0.0)
ap.add_argument("--calls", type=int, default=8)
a = ap.parse_args()
w = synth.config("target")
ctx = N.Context(0)
def hold(x):
    if a.pageable: return np.ascontiguousarray(x)
    o = ctx.host_alloc(x.shape, x.dtype); o[...] = x; return o
def outs(n, k):
    if a.pageable: return (np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(k, np.int64))
    o = (ctx.host_alloc((n,), np.int32), ctx.host_alloc((n,), np.int32), ctx.host_alloc((k,), np.int64))
    for x in o: x[...] = 0
    return o
P = [hold(getattr(w, k)) for k in ("part_off", "partition_id", "begin", "end", "committed")]
co, cr = hold(w.cons_off), hold(w.cons_rank)
out = outs(w.n_partitions, w.cons_rank.size)
if a.tiny_first:
    t1 = synth.make_uniform("tiny", 3, 4, 256, 32, "zipf")
    Q = [hold(getattr(t1, k)) for k in ("part_off", "partition_id", "begin", "end", "committed")]
    qo, qr = hold(t1.cons_off), hold(t1.cons_rank)
    qout = outs(t1.n_partitions, t1.cons_rank.size)
    tt = []
    for i in range(a.tiny_first):
        t = time.perf_counter()
        ctx.assign_batch(Q[0], Q[1], Q[2], Q[3], Q[4], N.LA_RESET_EARLIEST, qo, qr, out=qout)
        tt.append((time.perf_counter() - t) * 1e3)
    print("tiny calls ms:", " ".join("%.2f" % x for x in tt[:4]), "...", "%.2f" % tt[-1])
ts = []
for i in range(a.calls):
    if a.gap: time.sleep(a.gap)
    t = time.perf_counter()
    ctx.assign_batch(P[0], P[1], P[2], P[3], P[4], N.LA_RESET_EARLIEST, co, cr, out=out)
    ts.append((time.perf_counter() - t) * 1e3)
print("%s tiny_first=%d gap=%.1f: " % ("pageable" if a.pageable else "pinned", a.tiny_first, a.gap) +
      " ".join("%.1f" % x for x in ts), "pipeline", ctx.last_pipeline())
