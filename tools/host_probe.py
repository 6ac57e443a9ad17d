"""Time la_assign_batch on host buffers (target batch): python wrapper vs pre-touched outputs."""
import sys, time, ctypes
import numpy as np
sys.path.insert(0, ".")
from kafka_lag_based_assignor_amd import _native as N
from kafka_lag_based_assignor_amd import synth

w = synth.config("target")
ctx = N.Context(0)
lib = ctx._lib
def p64(a): return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
def p32(a): return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
for rep in range(3):
    t = time.perf_counter()
    r = ctx.assign_batch(w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
    print("wrapper, fresh outputs: %.1f ms" % ((time.perf_counter() - t) * 1e3))
op = np.zeros(w.n_partitions, np.int32); om = np.zeros(w.n_partitions, np.int32); ot = np.zeros(w.cons_rank.size, np.int64)
for rep in range(4):
    t = time.perf_counter()
    rc = lib.la_assign_batch(ctx._h, w.n_topics, p64(w.part_off), p32(w.partition_id), p64(w.begin), p64(w.end), p64(w.committed),
                             N.LA_RESET_EARLIEST, p64(w.cons_off), p32(w.cons_rank), p32(op), p32(om), p64(ot))
    dt = time.perf_counter() - t
    print("C call, touched outputs: rc=%d %.1f ms  (%.2e assignments/s)  pipeline %d" % (rc, dt * 1e3, w.n_partitions / dt, ctx.last_pipeline()))
print("same result:", np.array_equal(op, r[0]), np.array_equal(om, r[1]), np.array_equal(ot, r[2]))
# pinned host arrays (la_host_alloc: what the Java host's direct buffers are made of)
def pinned(a):
    out = ctx.host_alloc(a.shape, a.dtype)
    out[...] = a
    return out
P = {k: pinned(getattr(w, k)) for k in ("part_off", "partition_id", "begin", "end", "committed", "cons_off", "cons_rank")}
pop, pom, pot = pinned(op * 0), pinned(om * 0), pinned(ot * 0)
for rep in range(4):
    t = time.perf_counter()
    rc = lib.la_assign_batch(ctx._h, w.n_topics, p64(P["part_off"]), p32(P["partition_id"]), p64(P["begin"]), p64(P["end"]),
                             p64(P["committed"]), N.LA_RESET_EARLIEST, p64(P["cons_off"]), p32(P["cons_rank"]),
                             p32(pop), p32(pom), p64(pot))
    dt = time.perf_counter() - t
    print("C call, pinned buffers: rc=%d %.1f ms  (%.2e assignments/s)  pipeline %d" % (rc, dt * 1e3, w.n_partitions / dt, ctx.last_pipeline()))
print("same result:", np.array_equal(pop, r[0]), np.array_equal(pom, r[1]), np.array_equal(pot, r[2]))

# the Java host's flow: assign, then every member's list.  (a) download the result, upload it again for the
# grouping call; (b) keep it on the device and download only the grouped form
M = 32
gt, gp = np.zeros(w.n_partitions, np.int32), np.zeros(w.n_partitions, np.int32)
off = np.zeros(M + 1, np.int64)
lib.la_group_last_by_member.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.POINTER(ctypes.c_int64),
                                        ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]
for rep in range(3):
    t = time.perf_counter()
    rc1 = lib.la_assign_batch(ctx._h, w.n_topics, p64(w.part_off), p32(w.partition_id), p64(w.begin), p64(w.end), p64(w.committed),
                              N.LA_RESET_EARLIEST, p64(w.cons_off), p32(w.cons_rank), p32(op), p32(om), p64(ot))
    rc2 = lib.la_group_by_member(ctx._h, w.n_topics, p64(w.part_off), p32(op), p32(om), M, p64(off), p32(gt), p32(gp))
    dt = time.perf_counter() - t
    print("assign + group_by_member (two round trips): rc=%d,%d %.1f ms" % (rc1, rc2, dt * 1e3))
ref = (off.copy(), gt.copy(), gp.copy())
for rep in range(3):
    t = time.perf_counter()
    rc1 = lib.la_assign_batch(ctx._h, w.n_topics, p64(w.part_off), p32(w.partition_id), p64(w.begin), p64(w.end), p64(w.committed),
                              N.LA_RESET_EARLIEST, p64(w.cons_off), p32(w.cons_rank), None, None, p64(ot))
    rc2 = lib.la_group_last_by_member(ctx._h, M, p64(off), p32(gt), p32(gp))
    dt = time.perf_counter() - t
    print("assign (results stay) + group_last_by_member: rc=%d,%d %.1f ms  (%.2e assignments/s)" % (rc1, rc2, dt * 1e3, w.n_partitions / dt))
print("same lists:", np.array_equal(off, ref[0]), np.array_equal(gt, ref[1]), np.array_equal(gp, ref[2]))

pgt, pgp, poff = pinned(gt * 0), pinned(gp * 0), pinned(off * 0)
for rep in range(3):
    t = time.perf_counter()
    rc1 = lib.la_assign_batch(ctx._h, w.n_topics, p64(P["part_off"]), p32(P["partition_id"]), p64(P["begin"]), p64(P["end"]),
                              p64(P["committed"]), N.LA_RESET_EARLIEST, p64(P["cons_off"]), p32(P["cons_rank"]), None, None, p64(pot))
    rc2 = lib.la_group_last_by_member(ctx._h, M, p64(poff), p32(pgt), p32(pgp))
    dt = time.perf_counter() - t
    print("pinned: assign (results stay) + group_last_by_member: rc=%d,%d %.1f ms  (%.2e assignments/s)" % (rc1, rc2, dt * 1e3, w.n_partitions / dt))
print("same lists:", np.array_equal(poff, ref[0]), np.array_equal(pgt, ref[1]), np.array_equal(pgp, ref[2]))
