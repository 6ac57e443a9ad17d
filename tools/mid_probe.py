import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from kafka_lag_based_assignor_amd import _native as N, synth
ctx = N.Context(0)
for (t, p, c) in [(1000, 256, 32), (10000, 64, 8)]:
    w = synth.make_uniform("lat", 20, t, p, c, "uniform40")
    a = (w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
    out = ctx.assign_batch(*a)
    for _ in range(20):
        ctx.assign_batch(*a, out=out)
    ts = []
    for _ in range(80):
        t0 = time.perf_counter(); ctx.assign_batch(*a, out=out); ts.append(time.perf_counter() - t0)
    print("%d x %d x %d: median %.1f us min %.1f us pipeline %d shards %s launches %d" % (t, p, c, np.median(ts) * 1e6, min(ts) * 1e6, ctx.last_pipeline(), ctx.last_shard_bounds(), ctx.last_launches()))
