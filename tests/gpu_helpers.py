"""Helpers shared by the per-component GPU test files (test_tile_gpu / test_block_gpu / test_large_gpu / test_host_calls_gpu):
workload builders, the device-entry call, comparisons.  Not a test module; the fresh-process tests import it by name."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from kafka_lag_based_assignor_amd import _native as N
from kafka_lag_based_assignor_amd import sharding, synth
from oracle import oracle
from oracle.round_form import round_form

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- sparse begin -----------------------------------------------------------------------------------------------------
def _expected(w, latest, begin=None):
    lag = oracle.compute_lags(w.begin if begin is None else begin, w.end, w.committed, latest)
    return oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)


def _workload(seed, frac_none, topics=300, big=False):
    w = synth.ragged(seed, topics, 3000 if big else 300, 40)
    rng = np.random.default_rng(seed)
    n = w.n_partitions
    com = rng.integers(0, 1 << 20, n).astype(np.int64)
    com[rng.random(n) < frac_none] = -1
    w.committed = com
    w.begin = rng.integers(0, 1 << 19, n).astype(np.int64)                  # non-zero: a dropped entry would show
    with np.errstate(over="ignore"):
        w.end = np.maximum(com, w.begin) + np.maximum(w.lag, 0)
    return w


def _pinned(c, a):
    a = np.ascontiguousarray(a)
    p = c.host_alloc(a.shape, a.dtype)
    p[...] = a
    return p


def _batch_of(shapes, seed, kinds=None, negative=False):
    """A batch of topics with the given (partitions, consumers) shapes; lags per `kinds` (default: mixed)."""
    rng = np.random.default_rng(seed)
    ps = [s[0] for s in shapes]
    cs = [s[1] for s in shapes]
    part_off = np.concatenate([[0], np.cumsum(ps)]).astype(np.int64)
    cons_off = np.concatenate([[0], np.cumsum(cs)]).astype(np.int64)
    pid = np.concatenate([rng.permutation(p) if i % 3 else np.arange(p) for i, p in enumerate(ps)] + [np.empty(0, np.int64)]).astype(np.int32)
    lags = []
    for i, p in enumerate(ps):
        kind = (kinds or ["u40", "ties", "pareto", "zero", "u20", "full"])[i % (len(kinds) if kinds else 6)]
        if kind == "u40":
            l = rng.integers(0, 1 << 40, p)
        elif kind == "u20":
            l = rng.integers(0, 1 << 20, p)
        elif kind == "ties":
            l = rng.integers(0, 7, p) * 1000
        elif kind == "zero":
            l = np.zeros(p, np.int64)
        elif kind == "pareto":
            l = np.floor(np.minimum(float(1 << 40), 1000.0 * (1.0 - rng.random(p)) ** (-1.0 / 1.5))).astype(np.int64)
        else:
            l = rng.integers(-(1 << 63), (1 << 63) - 1, p)
            if not negative:
                l = np.where(l < 0, ~l, l)
        lags.append(np.asarray(l, np.int64))
    lag = np.concatenate(lags + [np.empty(0, np.int64)])
    ranks = np.concatenate([np.sort(rng.choice(3 * c + 5, c, replace=False)) for c in cs] + [np.empty(0, np.int64)]).astype(np.int32)
    n = int(part_off[-1])
    return synth.Workload("batch", len(shapes), part_off, pid, np.zeros(n, np.int64), lag.copy(), np.zeros(n, np.int64), lag,
                          cons_off, ranks, max(ps) if ps else 0, max(cs) if cs else 0)


def _device_call(ctx, w, flags=0, algo=N.LA_ALGO_AUTO, want_totals=True):
    import ctypes
    import torch
    dev = torch.device("cuda", 0)
    d = {k: torch.from_numpy(np.ascontiguousarray(getattr(w, k))).to(dev) for k in ("part_off", "partition_id", "lag", "cons_off", "cons_rank")}
    out_pid = torch.full((max(w.n_partitions, 1),), -7, device=dev, dtype=torch.int32)
    out_rank = torch.full((max(w.n_partitions, 1),), -7, device=dev, dtype=torch.int32)
    out_total = torch.full((max(w.cons_rank.size, 1),), -7, device=dev, dtype=torch.int64)
    b = N.DeviceBatch()
    b.n_topics, b.reset_mode, b.algo, b.flags = w.n_topics, N.LA_RESET_LATEST, algo, flags
    b.n_partitions, b.n_consumers = w.n_partitions, w.cons_rank.size
    b.max_partitions_per_topic, b.max_consumers_per_topic = w.max_partitions, w.max_consumers
    b.d_part_off, b.d_partition_id, b.d_lag = d["part_off"].data_ptr(), d["partition_id"].data_ptr(), d["lag"].data_ptr()
    b.d_cons_off, b.d_cons_rank = d["cons_off"].data_ptr(), d["cons_rank"].data_ptr()
    b.d_out_partition, b.d_out_member_rank = out_pid.data_ptr(), out_rank.data_ptr()
    b.d_out_total_lag = out_total.data_ptr() if want_totals else None
    po, co = np.ascontiguousarray(w.part_off, np.int64), np.ascontiguousarray(w.cons_off, np.int64)
    b.h_part_off = po.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    b.h_cons_off = co.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    stream = torch.cuda.current_stream().cuda_stream
    ctx.assign_batch_device(b, stream)
    ctx.sync(stream)
    return out_pid.cpu().numpy()[: w.n_partitions], out_rank.cpu().numpy()[: w.n_partitions], out_total.cpu().numpy()[: w.cons_rank.size]


def _same3(got, exp, what=""):
    for g, e, name in zip(got, exp, ("partition order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg="%s %s" % (name, what))


# ---- keys-first sorts with tie repair ----------------------------------------------------------------------
def _sort_topic(n, kind, seed, shuffled=True, dup_ids=False):
    rng = np.random.default_rng(seed)
    if kind == "wide":
        lag = rng.integers(0, 1 << 40, n)
    elif kind == "runs":                                    # thousands of short runs
        lag = rng.integers(0, max(2, n // 300), n)
    elif kind == "run4096":                                 # one run of exactly 4 096 (fits anywhere), the rest distinct
        lag = rng.permutation(n).astype(np.int64) + 10
        lag[rng.choice(n, 4096, replace=False)] = 5
    elif kind == "run9000":                                 # one run too long for a workgroup: the redo slots
        lag = rng.permutation(n).astype(np.int64) + 10
        lag[rng.choice(n, 9000, replace=False)] = 7
    elif kind == "run20000_top":                            # ... at the top of the order, next to short runs
        lag = rng.integers(0, n // 50, n)
        lag[rng.choice(n, 20000, replace=False)] = 1 << 41
    elif kind == "pairs":                                   # runs of two, three and four (settled by the scan itself) and of five
        lag = rng.permutation(np.repeat(rng.permutation(n), rng.integers(1, 6, n))[:n] + 3)
    elif kind == "equal":
        lag = np.full(n, 12345)
    elif kind == "full":
        lag = rng.integers(-(1 << 63), (1 << 63) - 1, n)
        lag[rng.choice(n, n // 10, replace=False)] = -3      # ties among negative lags
    else:
        raise ValueError(kind)
    pid = rng.permutation(n) if shuffled else np.arange(n)
    if dup_ids:
        pid = pid // 3
    lag = np.asarray(lag, np.int64)
    pid = pid.astype(np.int32)
    return synth.Workload("sort", 1, np.array([0, n], np.int64), pid, np.zeros(n, np.int64), lag.copy(), np.zeros(n, np.int64), lag,
                          np.array([0, 0], np.int64), np.zeros(0, np.int32), n, 0)


# ---- la_hint_next_call: the caller's bounds reach the host-buffer entry points -------------------------
def _pinned_copy(ctx, arrays):
    out = []
    for a in arrays:
        if a is None or isinstance(a, int):
            out.append(a)
        else:
            p = ctx.host_alloc(a.shape, a.dtype)
            p[...] = a
            out.append(p)
    return out


# ---- one launch for a small rebalance -------------------------------------------------------------------
def _grouped_expect(w, n_members):
    lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
    e_pid, e_rank, e_tot = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
    order = np.argsort(e_rank, kind="stable")                     # member by member, inside a member in the reference's order
    first = np.searchsorted(e_rank[order], np.arange(n_members + 1))
    topic = (np.searchsorted(w.part_off, order, side="right") - 1).astype(np.int32)
    return first.astype(np.int64), topic, e_pid[order], e_tot, (e_pid, e_rank)


# ---- block path: 65 .. 256 consumers, the greedy's bins ordered through 32-bit keys -------------------------
def _one_topic(P, C, lag, seed):
    rng = np.random.default_rng(seed)
    pid = rng.permutation(P).astype(np.int32)
    ranks = np.sort(rng.choice(3 * C + 5, C, replace=False)).astype(np.int32)
    lag = np.asarray(lag, np.int64)
    return synth.Workload("k32", 1, np.array([0, P], np.int64), pid, np.zeros(P, np.int64), lag.copy(), np.zeros(P, np.int64), lag,
                          np.array([0, C], np.int64), ranks, P, C)


# ---- LA_FLAG_WIRE_OUT: the all-gather's wire elements straight from the assignment kernels ------------------
def _wire_call(ctx, w, fmt, bounds, latest=False, flags=0, hint=None):
    import torch
    from kafka_lag_based_assignor_amd import sharding
    dev = torch.device("cuda", 0)
    d = {k: torch.from_numpy(np.ascontiguousarray(getattr(w, k))).to(dev) for k in
         ("part_off", "partition_id", "begin", "end", "committed", "cons_off", "cons_rank")}
    n = w.n_partitions
    wire = torch.zeros(max(n, 1) * fmt.elem_bytes + 16, device=dev, dtype=torch.uint8)
    out_total = torch.full((max(w.cons_rank.size, 1),), -7, device=dev, dtype=torch.int64)
    b = N.DeviceBatch()
    b.n_topics, b.reset_mode, b.algo = w.n_topics, (N.LA_RESET_LATEST if latest else N.LA_RESET_EARLIEST), N.LA_ALGO_AUTO
    b.flags = N.LA_FLAG_WIRE_OUT | flags
    b.n_partitions, b.n_consumers = n, w.cons_rank.size
    mp, mc = hint or (w.max_partitions, w.max_consumers)
    b.max_partitions_per_topic, b.max_consumers_per_topic = mp, mc
    b.d_part_off, b.d_partition_id = d["part_off"].data_ptr(), d["partition_id"].data_ptr()
    b.d_begin_off, b.d_end_off, b.d_committed_off = d["begin"].data_ptr(), d["end"].data_ptr(), d["committed"].data_ptr()
    b.d_cons_off, b.d_cons_rank = d["cons_off"].data_ptr(), d["cons_rank"].data_ptr()
    b.d_out_partition = b.d_out_member_rank = None
    b.d_out_total_lag = out_total.data_ptr()
    if bounds is not None:
        b.flags |= N.LA_FLAG_BOUNDS
        b.max_lag_hint, b.max_partition_id_hint = bounds
    b.d_out_wire = wire.data_ptr() + 2                              # element-aligned only: 2 bytes off a 16-byte boundary
    b.wire_elem_bytes, b.wire_id_bits = fmt.elem_bytes, fmt.id_bits
    stream = torch.cuda.current_stream().cuda_stream
    ctx.assign_batch_device(b, stream)
    ctx.sync(stream)
    raw = wire.cpu().numpy()[2:2 + n * fmt.elem_bytes].view(fmt.dtype)
    return raw, out_total.cpu().numpy()[: w.cons_rank.size]


__all__ = ['ROOT', '_expected', '_workload', '_pinned', '_batch_of', '_device_call', '_same3', '_sort_topic', '_pinned_copy', '_grouped_expect', '_one_topic', '_wire_call']
