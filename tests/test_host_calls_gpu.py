"""Host-buffer entry points (la_assign_batch*, la_hint_next_call, la_group_*: pipelines, hints, sparse begin, the one-launch small rebalance) through the C ABI, bit-exact against the oracle."""
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
from gpu_helpers import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("frac_none", [0.0, 0.01, 0.3, 1.0])
@pytest.mark.parametrize("kind", ["one_copy", "lanes", "streams", "mapped", "shards", "mapped_shards"])
def test_sparse_begin_equals_the_dense_call(frac_none, kind):
    big = kind != "one_copy"
    w = _workload(int(frac_none * 100) + len(kind), frac_none, topics=400 if big else 60, big=big)
    idx, val = N.sparse_begin(w.begin, w.committed)
    assert idx.size == int((w.committed < 0).sum())
    flags = {"one_copy": 0, "lanes": N.LA_CREATE_SPLIT_ALWAYS | 3, "streams": 0, "mapped": 0, "shards": N.LA_CREATE_SPLIT_ALWAYS,
             "mapped_shards": N.LA_CREATE_SPLIT_ALWAYS}[kind]
    dev = [0, 0, 0] if kind in ("shards", "mapped_shards") else 0
    if kind == "one_copy":
        os.environ["LA_ZERO_COPY_BYTES"] = "0"                  # (read at la_create; every staged call is zero-copy by default)
    try:
        c_made = N.Context(dev, flags=flags)
    finally:
        os.environ.pop("LA_ZERO_COPY_BYTES", None)
    with c_made as c:
        if kind == "streams":
            os.environ["LA_CHUNK_PARTITIONS"] = "20000"
            os.environ["LA_NO_MAPPED_PIPELINE"] = "1"
        try:
            exp = _expected(w, False)
            args = (w.part_off, w.partition_id, w.end, w.committed, N.LA_RESET_EARLIEST)
            if kind in ("streams", "mapped", "mapped_shards"):
                # every array pinned: the thread-less three-stream pipeline, or -- the default -- the kernels on the arrays in place
                with N.Context(dev, flags=flags) as cp:                           # (a context created under the chunk override)
                    pin = lambda a: _pinned(cp, a)                                # noqa: E731
                    out = (cp.host_alloc((w.n_partitions,), np.int32), cp.host_alloc((w.n_partitions,), np.int32),
                           cp.host_alloc((w.cons_rank.size,), np.int64))
                    got = cp.assign_batch_sparse(pin(w.part_off), pin(w.partition_id), pin(w.end), pin(w.committed),
                                                 N.LA_RESET_EARLIEST, pin(idx), pin(val), pin(w.cons_off), pin(w.cons_rank), out=out)
                    assert cp.last_pipeline() == (N.LA_PIPELINE_STREAMS if kind == "streams" else N.LA_PIPELINE_MAPPED)
                    # the dense call on the same pinned arrays, and the results left on the device + grouped
                    dense = cp.assign_batch(pin(w.part_off), pin(w.partition_id), pin(w.begin), pin(w.end), pin(w.committed),
                                            N.LA_RESET_EARLIEST, pin(w.cons_off), pin(w.cons_rank), out=out)
                    for g, e in zip(dense, exp):
                        np.testing.assert_array_equal(g, e)
                    n_members = int(w.cons_rank.max()) + 1 if w.cons_rank.size else 0
                    cp.assign_batch_sparse(pin(w.part_off), pin(w.partition_id), pin(w.end), pin(w.committed), N.LA_RESET_EARLIEST,
                                           pin(idx), pin(val), pin(w.cons_off), pin(w.cons_rank), keep_on_device=True)
                    g_off, g_t, g_p = cp.group_last_by_member(w.n_partitions, n_members)
                    order = np.argsort(exp[1], kind="stable")
                    np.testing.assert_array_equal(g_p, exp[0][order])
            else:
                got = c.assign_batch_sparse(*args, idx, val, w.cons_off, w.cons_rank)
                assert c.last_pipeline() == (N.LA_PIPELINE_ONE_COPY if kind == "one_copy" else N.LA_PIPELINE_LANES)
            for g, e, what in zip(got, exp, ("order", "member", "totals")):
                np.testing.assert_array_equal(g, e, err_msg=what)
            # and the dense call on the same context agrees (same bits either way)
            dense = c.assign_batch(w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
            for g, e in zip(dense, exp):
                np.testing.assert_array_equal(g, e)
        finally:
            os.environ.pop("LA_CHUNK_PARTITIONS", None)
            os.environ.pop("LA_NO_MAPPED_PIPELINE", None)


def test_sparse_begin_semantics(ctx):
    w = _workload(5, 0.2, topics=50)
    idx, val = N.sparse_begin(w.begin, w.committed)
    a = (w.part_off, w.partition_id, w.end, w.committed)
    # `latest` never reads begin: the list is ignored, may be absent
    exp = _expected(w, True)
    for lst in ((idx, val), (None, None)):
        got = ctx.assign_batch_sparse(*a, N.LA_RESET_LATEST, lst[0], lst[1], w.cons_off, w.cons_rank)
        np.testing.assert_array_equal(got[1], exp[1])
    # an unlisted partition without a committed offset has begin 0 (getOrDefault(tp, 0L), Main.java:350-351)
    keep = np.arange(idx.size) % 2 == 0
    begin0 = np.zeros_like(w.begin)
    begin0[idx[keep]] = val[keep]
    exp = _expected(w, False, begin0)
    got = ctx.assign_batch_sparse(*a, N.LA_RESET_EARLIEST, idx[keep], val[keep], w.cons_off, w.cons_rank)
    for g, e in zip(got, exp):
        np.testing.assert_array_equal(g, e)
    # entries for partitions that HAVE a committed offset are harmless
    extra_idx = np.arange(w.n_partitions, dtype=np.int64)
    got = ctx.assign_batch_sparse(*a, N.LA_RESET_EARLIEST, extra_idx, w.begin, w.cons_off, w.cons_rank)
    for g, e in zip(got, _expected(w, False)):
        np.testing.assert_array_equal(g, e)
    # the grouped form: the same lists as the dense grouped call
    n_members = int(w.cons_rank.max()) + 1
    g_dense = ctx.assign_batch_grouped(w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off,
                                       w.cons_rank, n_members)
    g_sparse = ctx.assign_batch_grouped_sparse(*a, N.LA_RESET_EARLIEST, idx, val, w.cons_off, w.cons_rank, n_members)
    for x, y in zip(g_dense, g_sparse):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("flags", [0, N.LA_CREATE_SPLIT_ALWAYS | 3])
def test_sparse_begin_bad_lists_are_errors(flags):
    w = _workload(9, 0.1, topics=200, big=flags != 0)
    idx, val = N.sparse_begin(w.begin, w.committed)
    a = (w.part_off, w.partition_id, w.end, w.committed, N.LA_RESET_EARLIEST)
    with N.Context(0, flags=flags) as c:
        swapped = idx.copy()
        swapped[[0, -1]] = swapped[[-1, 0]]                                        # not ascending (across chunks)
        beyond = idx.copy()
        beyond[-1] = w.n_partitions                                                # outside the batch
        negative = idx.copy()
        negative[0] = -1
        for bad in (swapped, beyond, negative):
            with pytest.raises(N.LagAssignError) as e:
                c.assign_batch_sparse(*a, bad, val, w.cons_off, w.cons_rank)
            assert e.value.code == N.LA_EINVAL and "none_index" in str(e.value)
        good = c.assign_batch_sparse(*a, idx, val, w.cons_off, w.cons_rank)        # the context is usable afterwards
        np.testing.assert_array_equal(good[1], _expected(w, False)[1])
        with pytest.raises(N.LagAssignError):
            c.assign_batch_sparse(*a, idx, None, w.cons_off, w.cons_rank)          # a null array with n_none > 0


def test_version_follows_the_header():
    header = open(os.path.join(ROOT, "include", "lagassign.h")).read()
    v = int(re.search(r"#define\s+LA_VERSION\s+(\d+)", header).group(1))
    assert N.load().la_version() == v >= 300


def test_calls_leave_the_current_device_alone(ctx, torch_dev):
    torch, dev = torch_dev
    before = torch.cuda.current_device()                                              # torch owns the HIP runtime: ask torch
    w = synth.config("cfg3", 0.01)
    ctx.assign_batch(w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
    ctx.device_features(0)
    assert torch.cuda.current_device() == before


# ---- zero-copy small calls ------------------------------------------------------------------------------------
def test_smallest_calls_run_zero_copy_and_agree_with_the_copying_forms():
    """Staged calls (layouts up to 12 MB; 128 KB until round 5): the kernels read the inputs from coherent host memory in place
    and write results / totals / lists into it; no hipMemcpy, no stream wait.  Same results as the one-copy form
    (LA_ZERO_COPY_BYTES=0) on every entry point, across the old threshold, alternating on one context; errors surface and
    leave the context usable."""
    os.environ["LA_ZERO_COPY_BYTES"] = "0"
    try:
        ref_ctx = N.Context(0)
    finally:
        os.environ.pop("LA_ZERO_COPY_BYTES", None)
    with N.Context(0) as c, ref_ctx:
        seen, seen_ref = set(), set()
        for seed, (t, p, cc) in enumerate([(1, 3, 2), (10, 10, 3), (40, 50, 5), (3, 700, 90), (1, 2500, 3), (1, 1800, 300),
                                           (60, 64, 8), (300, 100, 7), (1, 20, 0), (5, 0, 3)]):
            w = synth.ragged(100 + seed, t, p, cc)
            n_members = int(w.cons_rank.max()) + 1 if w.cons_rank.size else 0
            for mode in (N.LA_RESET_EARLIEST, N.LA_RESET_LATEST):
                a = (w.part_off, w.partition_id, None if mode == N.LA_RESET_LATEST else w.begin, w.end, w.committed, mode,
                     w.cons_off, w.cons_rank)
                exp = _expected(w, mode == N.LA_RESET_LATEST)
                got = c.assign_batch(*a)
                seen.add(c.last_pipeline())
                _same3(got, exp, "zero copy? %d" % c.last_pipeline())
                _same3(ref_ctx.assign_batch(*a), exp, "one copy")
                seen_ref.add(ref_ctx.last_pipeline())
                g = c.assign_batch_grouped(*a, n_members)
                g_ref = ref_ctx.assign_batch_grouped(*a, n_members)
                for x, y in zip(g, g_ref):
                    np.testing.assert_array_equal(x, y)
                # results kept on the device, grouped by a second call
                c.assign_batch(*a, keep_on_device=True)
                g2 = c.group_last_by_member(w.n_partitions, n_members)
                for x, y in zip(g2, (g[0], g[1], g[2])):
                    np.testing.assert_array_equal(x, y)
            idx, val = N.sparse_begin(w.begin, w.committed)
            got = c.assign_batch_sparse(w.part_off, w.partition_id, w.end, w.committed, N.LA_RESET_EARLIEST, idx, val, w.cons_off, w.cons_rank)
            _same3(got, _expected(w, False), "sparse")
            _same3(c.assign_batch_lags(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank),
                   oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank), "lags")
        # (the batches without partitions take no staged form at all)
        assert N.LA_PIPELINE_ZERO_COPY in seen and N.LA_PIPELINE_ONE_COPY not in seen, seen
        assert N.LA_PIPELINE_ONE_COPY in seen_ref and N.LA_PIPELINE_ZERO_COPY not in seen_ref, seen_ref
        # errors: unsorted ranks (validated on the host for a small call), then the context still works
        with pytest.raises(N.LagAssignError) as e:
            c.assign_batch_lags([0, 2], [0, 1], [5, 6], [0, 2], [3, 1])
        assert e.value.code == N.LA_EINVAL
        p, m, t = c.assign_batch_lags([0, 3], [0, 1, 2], [100000, 50000, 60000], [0, 2], [0, 1])      # README.md:42-57
        assert c.last_pipeline() == N.LA_PIPELINE_ZERO_COPY
        assert p.tolist() == [0, 2, 1] and m.tolist() == [0, 1, 1] and t.tolist() == [100000, 110000]


def test_hinted_host_call_is_one_tile_launch_per_chunk(ctx):
    """A batch too large to be resident at once (40 000 x 256 x 32: the inline single-launch form of small batches does not
    apply) through la_assign_batch on pinned, mapped arrays: without a hint the tile path is two launches (packed records +
    the wide-record kernel over an empty list), with the marshaller's bounds ONE; + one launch for the consumer-rank check.
    Same results either way, equal to the device-resident path's; the hint is one-shot."""
    w = synth.make_uniform("hint", 31, 40000, 256, 32, "zipf")
    a = (w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
    pa = _pinned_copy(ctx, a)
    pout = (ctx.host_alloc((w.n_partitions,), np.int32), ctx.host_alloc((w.n_partitions,), np.int32),
            ctx.host_alloc((w.cons_rank.size,), np.int64))
    bounds = N.offset_bounds(w.begin, w.end, w.committed, w.partition_id)
    assert bounds is not None
    ref = [x.copy() for x in ctx.assign_batch(*pa, out=pout)]
    assert ctx.last_pipeline() == N.LA_PIPELINE_MAPPED
    plain = ctx.last_launches()
    ctx.hint_next_call(bounds)
    got = [x.copy() for x in ctx.assign_batch(*pa, out=pout)]
    hinted = ctx.last_launches()
    assert (plain, hinted) == (3, 2), (plain, hinted)
    _same3(got, ref, "hinted")
    ctx.assign_batch(*pa, out=pout)                                # one-shot: the next call has no hint again
    assert ctx.last_launches() == plain
    # pageable arrays: chunks over the lanes.  A chunk that is resident at once takes the inline single-launch form with or
    # without bounds, so the hint can only ever remove launches
    ctx.assign_batch(*a, out=tuple(np.empty_like(x) for x in ref))
    plain_l = ctx.last_launches()
    assert ctx.last_pipeline() == N.LA_PIPELINE_LANES
    ctx.hint_next_call(bounds)
    got = ctx.assign_batch(*a, out=tuple(np.empty_like(x) for x in ref))
    assert ctx.last_launches() <= plain_l
    _same3(got, ref, "hinted, lanes")
    # the oracle on a slice of the batch (the whole batch through the device path is covered elsewhere)
    t = 500
    p1, k1 = int(w.part_off[t]), int(w.cons_off[t])
    lag = oracle.compute_lags(w.begin[:p1], w.end[:p1], w.committed[:p1], False)
    e = oracle.assign_flat(w.part_off[:t + 1], w.partition_id[:p1], lag, w.cons_off[:t + 1], w.cons_rank[:k1])
    _same3((ref[0][:p1], ref[1][:p1], ref[2][:k1]), e, "oracle slice")


def test_violated_hint_is_einval_never_a_different_result(ctx):
    """The kernels decide per wavefront, from the data, whether a tile's records pack; the bounds only prove that the
    wide-record launch behind them has nothing to do.  A tile that does NOT pack although the bounds said it would (an end
    offset of 2^56 under a promise of 2^31) is reported as LA_EINVAL; a bound that is wrong but harmless changes nothing."""
    w = synth.make_uniform("hint", 32, 40000, 256, 32, "zipf")
    end = w.end.copy()
    end[123457] = 1 << 56                                          # one partition with a lag of ~2^56: its tile needs wide records
    a = (w.part_off, w.partition_id, w.begin, end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
    pa = _pinned_copy(ctx, a)
    pout = (ctx.host_alloc((w.n_partitions,), np.int32), ctx.host_alloc((w.n_partitions,), np.int32),
            ctx.host_alloc((w.cons_rank.size,), np.int64))
    good = [x.copy() for x in ctx.assign_batch(*pa, out=pout)]     # no hint: the tile goes through the wide-record kernel
    assert ctx.last_pipeline() == N.LA_PIPELINE_MAPPED and ctx.last_launches() == 3
    t = 123457 // 256
    p0, p1, k0, k1 = t * 256, (t + 2) * 256, t * 32, (t + 2) * 32
    lag = oracle.compute_lags(w.begin[p0:p1], end[p0:p1], w.committed[p0:p1], False)
    e = oracle.assign_flat(w.part_off[t:t + 3] - p0, w.partition_id[p0:p1], lag, w.cons_off[t:t + 3] - k0, w.cons_rank[k0:k1])
    _same3((good[0][p0:p1], good[1][p0:p1], good[2][k0:k1]), e, "oracle on the wide tile and its neighbour")
    ctx.hint_next_call(((1 << 31), 255))                           # the promise the data breaks
    with pytest.raises(N.LagAssignError) as ei:
        ctx.assign_batch(*pa, out=pout)
    assert ei.value.code == N.LA_EINVAL and "bounds" in str(ei.value)
    _same3(ctx.assign_batch(*pa, out=pout), good, "after the failure, no hint")
    ctx.hint_next_call(N.offset_bounds(w.begin, end, w.committed, w.partition_id))   # the honest bounds prove nothing here:
    _same3(ctx.assign_batch(*pa, out=pout), good, "honest bounds")                   # two launches, same result
    assert ctx.last_launches() == 3
    ctx.hint_next_call(((1 << 57), 100))                           # wrong about the ids, harmless: every tile still packs or defers
    _same3(ctx.assign_batch(*pa, out=pout), good, "harmless wrong bound")
    with pytest.raises(N.LagAssignError):
        ctx.hint_next_call((-1, 5))


def test_grouped_and_sparse_calls_take_the_hint(ctx):
    w = synth.make_uniform("hint", 33, 40000, 256, 32, "zipf")
    idx, val = N.sparse_begin(w.begin, w.committed)
    n_members = int(w.cons_rank.max()) + 1
    bounds = N.offset_bounds(w.begin, w.end, w.committed, w.partition_id)
    base = ctx.assign_batch_grouped_sparse(w.part_off, w.partition_id, w.end, w.committed, N.LA_RESET_EARLIEST, idx, val,
                                           w.cons_off, w.cons_rank, n_members)
    plain = ctx.last_launches()
    ctx.hint_next_call(bounds)
    got = ctx.assign_batch_grouped_sparse(w.part_off, w.partition_id, w.end, w.committed, N.LA_RESET_EARLIEST, idx, val,
                                          w.cons_off, w.cons_rank, n_members)
    assert ctx.last_launches() <= plain          # (pageable arrays: chunks small enough to be resident are one launch anyway)
    for g, e in zip(got, base):
        np.testing.assert_array_equal(g, e)
    lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
    ctx.hint_next_call(N.offset_bounds(None, None, None, w.partition_id, lag=lag))
    r = ctx.assign_batch_lags(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
    e = ctx.assign_batch(w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
    _same3(r, e, "lags entry with a hint")


@pytest.mark.parametrize("t,p,c", [(1, 3, 2), (10, 10, 3), (40, 25, 5), (3, 300, 33), (1, 1000, 64), (25, 40, 8), (1, 1, 1),
                                   (40, 50, 5), (7, 300, 33), (2, 1000, 64), (60, 40, 8)])
def test_small_rebalance_is_one_launch(ctx, t, p, c):
    """la_assign_batch_grouped on a rebalance of up to 1 024 partitions: the tile kernel's last workgroup builds every member's
    list and stores the completion word -- ONE launch (la_last_launches), zero copies; beyond (up to 2 560) the one-workgroup
    grouping is its own launch, which is faster there (1 024 threads against the tail's 256; profiles/r05_ae_fused_tail.txt),
    and so is the plain finishing launch for a call without lists.  The lists equal the stable sort by member of the oracle's
    assignment.  Repeated calls (the tail's counter resets itself)."""
    w = synth.make_uniform("small", 40 + t, t, p, c, "uniform40")
    n_members = c + 2
    a = (w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
    off, topic, part, tot, (e_pid, e_rank) = _grouped_expect(w, n_members)
    for _ in range(3):
        g_off, g_t, g_p, g_tot = ctx.assign_batch_grouped(*a, n_members)
        assert ctx.last_pipeline() == N.LA_PIPELINE_ZERO_COPY
        assert ctx.last_launches() == (1 if w.n_partitions <= 1024 else 2), ctx.last_launches()
        np.testing.assert_array_equal(g_off, off)
        np.testing.assert_array_equal(g_p, part)
        np.testing.assert_array_equal(g_t, topic)
        np.testing.assert_array_equal(g_tot, tot)
        got = ctx.assign_batch(*a)
        assert ctx.last_launches() == 2                              # (assignment + the finishing launch)
        _same3(got, (e_pid, e_rank, tot), "ungrouped")


def test_small_rebalance_falls_back_to_separate_launches_when_it_must(ctx):
    """More entries than one workgroup groups, more members than the tail's LDS holds, or a topic beyond the tile path in the
    batch: the lists come from their own launch(es), same results."""
    for (t, p, c, n_members) in [(30, 100, 8, 10), (4, 50, 5, 3000)]:
        w = synth.make_uniform("small", 60 + t, t, p, c, "uniform40")
        a = (w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
        off, topic, part, tot, _ = _grouped_expect(w, n_members)
        g_off, g_t, g_p, g_tot = ctx.assign_batch_grouped(*a, n_members)
        # 3 000 entries: assignment + the two-launch counting sort, whose last block also ends the call; 3 000 members: + one workgroup
        assert ctx.last_launches() == (3 if n_members == 10 else 2), ctx.last_launches()
        np.testing.assert_array_equal(g_off, off)
        np.testing.assert_array_equal(g_p, part)
        np.testing.assert_array_equal(g_t, topic)
    # a block-path topic (1 100 partitions) beside tile topics: the tile launch is not the batch's last
    import gpu_helpers as t4
    w = t4._batch_of([(20, 4), (1100, 5), (30, 3)], 9, kinds=["u40"])
    lag = w.lag
    n_members = int(w.cons_rank.max()) + 1
    e_pid, e_rank, e_tot = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
    zeros = np.zeros_like(lag)
    g_off, g_t, g_p, g_tot = ctx.assign_batch_grouped(w.part_off, w.partition_id, zeros, lag, zeros, N.LA_RESET_EARLIEST,
                                                      w.cons_off, w.cons_rank, n_members)
    assert ctx.last_launches() >= 3
    order = np.argsort(e_rank, kind="stable")
    np.testing.assert_array_equal(g_p, e_pid[order])
    np.testing.assert_array_equal(g_tot, e_tot)


def test_fused_tail_off_gives_the_same_lists_in_a_fresh_process():
    code = r"""
import numpy as np, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from kafka_lag_based_assignor_amd import _native as N, synth
import gpu_helpers as t
ctx = N.Context(0)
for (tt, p, c) in ((10, 10, 3), (40, 50, 5), (2, 1000, 64)):
    w = synth.make_uniform("small", 40 + tt, tt, p, c, "uniform40")
    off, topic, part, tot, _ = t._grouped_expect(w, c + 2)
    g = ctx.assign_batch_grouped(w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank, c + 2)
    assert ctx.last_launches() == WANT, ctx.last_launches()
    for x, y in zip(g, (off, topic, part, tot)):
        np.testing.assert_array_equal(x, y)
    r = ctx.assign_batch(w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
    assert ctx.last_launches() == WANT, ctx.last_launches()
    e = t._grouped_expect(w, c + 2)[4]
    np.testing.assert_array_equal(r[0], e[0]); np.testing.assert_array_equal(r[1], e[1])
print("ok")
"""
    # never fused (two launches), and round 5's first form: every staged call that can ends inside the tile kernel (one launch,
    # lists of up to 2 560 entries and calls without lists included) -- the default fuses lists up to 1 024 entries only
    for env_add, want in (({"LA_NO_FUSED_TAIL": "1"}, 2), ({"LA_FUSED_TAIL": "all"}, 1)):
        env = dict(os.environ, **env_add)
        out = subprocess.run([sys.executable, "-c", code.replace("WANT", str(want)) % (ROOT, os.path.join(ROOT, "tests"))], env=env,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "ok" in out.stdout, (env_add, out.stdout[-1500:], out.stderr[-1500:])


# ---- member lists of a mid-size rebalance: the two-launch counting sort (la_group_small.h, group_mid_*) ---------------------------
@pytest.mark.parametrize("n_topics,max_p,members", [(300, 256, 32), (40, 3000, 5), (5000, 9, 3), (2, 70000, 510), (700, 700, 509),
                                                   (1, 2561, 1), (9, 4096, 64), (60000, 4, 2), (3, 100000, 511)])
def test_member_lists_of_a_mid_size_rebalance(ctx, n_topics, max_p, members):
    """la_group_by_member between 2 560 and 65 536 entries with at most 510 members runs as two launches (per-block counts, then
    placement); 511 members take the radix form.  Ragged topics, empty topics, topics without consumers (rank -1), blocks that end
    in the middle of a topic: equal to a stable sort by member of the same arrays."""
    rng = np.random.default_rng(n_topics + members)
    sizes = rng.integers(0, max_p + 1, n_topics)
    sizes[rng.integers(0, n_topics, max(1, n_topics // 7))] = 0                 # empty topics
    part_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(part_off[-1])
    out_p = rng.integers(0, 1 << 20, n).astype(np.int32)
    out_m = rng.integers(0, members, n).astype(np.int32)
    for t in rng.integers(0, n_topics, max(1, n_topics // 9)):                  # topics without consumers
        out_m[part_off[t]:part_off[t + 1]] = -1
    off, g_t, g_p = ctx.group_by_member(part_off, out_p, out_m, members)
    launches = ctx.last_launches()
    order = np.argsort(out_m, kind="stable")
    counts = np.bincount(out_m + 1, minlength=members + 1)
    np.testing.assert_array_equal(off, np.cumsum(counts)[: members + 1])
    np.testing.assert_array_equal(g_p, out_p[order])
    np.testing.assert_array_equal(g_t, (np.searchsorted(part_off, order, side="right") - 1).astype(np.int32))
    if 2560 < n <= 2048 * 32 and members <= 510:
        assert launches == 2, launches


# ---- which staged calls are zero-copy, and from where mapped caller arrays are read in place ---------------------------------------
@pytest.mark.parametrize("topics,p,c,grouped,pipeline", [(100, 100, 8, False, "ZERO_COPY"), (1000, 30, 5, False, "ZERO_COPY"),
                                                        (1000, 50, 5, False, "MAPPED"), (1000, 50, 5, True, "ZERO_COPY"),
                                                        (1000, 100, 8, True, "MAPPED")])
def test_mid_size_calls_staged_or_read_in_place(ctx, topics, p, c, grouped, pipeline):
    """Pageable arrays: every layout up to 12 MB is packed into the mapped staging buffer and read there (zero-copy).
    la_host_alloc arrays (what the Java host's direct buffers are): from 1.25 MB on (3 MB with the lists aboard) nothing is packed,
    the kernels read the caller's arrays.  Either way the result is the oracle's."""
    w = synth.make_uniform("mid", topics + c, topics, p, c, "uniform40")
    a = (w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
    pa = _pinned_copy(ctx, a)
    want = getattr(N, "LA_PIPELINE_" + pipeline)
    lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
    e = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
    if grouped:
        ref = ctx.assign_batch_grouped(*a, c, want_totals=False)
        assert ctx.last_pipeline() == N.LA_PIPELINE_ZERO_COPY
        got = ctx.assign_batch_grouped(*pa, c, want_totals=False)
        assert ctx.last_pipeline() == want, ctx.last_pipeline()
        for x, y in zip(got[:3], ref[:3]):
            np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(got[2], e[0][np.argsort(e[1], kind="stable")])
    else:
        pout = (ctx.host_alloc((w.n_partitions,), np.int32), ctx.host_alloc((w.n_partitions,), np.int32),
                ctx.host_alloc((w.cons_rank.size,), np.int64))
        ref = ctx.assign_batch(*a)
        assert ctx.last_pipeline() == N.LA_PIPELINE_ZERO_COPY
        _same3(ref, e, "oracle, pageable")
        got = ctx.assign_batch(*pa, out=pout)
        assert ctx.last_pipeline() == want, ctx.last_pipeline()
        _same3(got, e, "oracle, pinned")


def test_a_staged_call_at_the_limit_and_just_beyond(ctx):
    """335 000 / 365 000 partitions (11.8 / 12.8 MB of layout) on either side of the 12 MB limit: zero-copy (packed and unpacked
    with the parked threads' help), then the lanes."""
    for topics, want in ((1340, N.LA_PIPELINE_ZERO_COPY), (1460, N.LA_PIPELINE_LANES)):
        w = synth.make_uniform("edge", topics, topics, 250, 16, "zipf")
        got = ctx.assign_batch(w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
        assert ctx.last_pipeline() == want, (topics, ctx.last_pipeline())
        lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
        _same3(got, oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank), "oracle")


def test_a_call_that_fails_before_it_starts_spends_the_hint(ctx):
    """la_hint_next_call is one-shot: the hint is forgotten when the next host-buffer assign call returns, WHATEVER it returns --
    also when that call returns before it looks at the hint (a NULL lag array, a negative member count, n_topics
    = 0 used to leave it pending, and the stale bound then failed the next, unrelated call with LA_EINVAL)."""
    import ctypes
    w = synth.make_uniform("spend", 41, 3000, 256, 32, "zipf")
    big = w.end.copy()
    big[1234] = 1 << 56                                            # breaks a bound of 2^31 (its tile needs wide records)
    exp = oracle.assign_flat(w.part_off, w.partition_id, oracle.compute_lags(w.begin, big, w.committed, False),
                             w.cons_off, w.cons_rank)
    lib, h = ctx._lib, ctx._h
    n_members = int(w.cons_rank.max()) + 1
    p64 = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))     # noqa: E731
    p32 = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))     # noqa: E731
    po, co = np.ascontiguousarray(w.part_off), np.ascontiguousarray(w.cons_off)
    pid, cr = np.ascontiguousarray(w.partition_id), np.ascontiguousarray(w.cons_rank)
    out_p, out_m = np.empty(pid.size, np.int32), np.empty(pid.size, np.int32)
    out_t = np.zeros(cr.size, np.int64)

    def early_failures():
        # (1) the lags entry with lag = NULL  (2) a grouped call with a negative member count  (3) grouped, n_topics = 0
        yield lambda: lib.la_assign_batch_lags(h, w.n_topics, p64(po), p32(pid), None, p64(co), p32(cr), p32(out_p), p32(out_m),
                                               p64(out_t)), N.LA_EINVAL
        moff = np.zeros(n_members + 1, np.int64)
        yield lambda: lib.la_assign_batch_grouped(h, w.n_topics, p64(po), p32(pid), p64(w.begin), p64(big), p64(w.committed),
                                                  N.LA_RESET_EARLIEST, p64(co), p32(cr), -1, p64(moff), None, p32(out_p),
                                                  p64(out_t)), N.LA_EINVAL
        yield lambda: lib.la_assign_batch_grouped(h, 0, p64(po), p32(pid), p64(w.begin), p64(big), p64(w.committed),
                                                  N.LA_RESET_EARLIEST, p64(co), p32(cr), n_members, p64(moff), None, p32(out_p),
                                                  p64(out_t)), N.LA_OK
    for call, want in early_failures():
        ctx.hint_next_call(((1 << 31), 255))                       # a promise the NEXT-but-one call's data would break
        assert int(call()) == want
        got = ctx.assign_batch(w.part_off, w.partition_id, w.begin, big, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
        _same3(got, exp, "unhinted call after an early return")
    # and the control: the same hint, pending, does fail that call
    ctx.hint_next_call(((1 << 31), 255))
    with pytest.raises(N.LagAssignError):
        ctx.assign_batch(w.part_off, w.partition_id, w.begin, big, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)


def test_la_wake_leaves_the_next_call_alone(ctx):
    """la_wake (ABI 0.5.0): a one-partition rebalance through the real small-call path, for the moment a host enters assign().
    Before a call, twice, between a hint and its call, with no call behind it, on a multi-shard context, on a context that has
    not run anything yet: the results are the oracle's, the pending hint and the last call's diagnostics survive it (the wake is
    not the caller's assign call), results kept on the device do not."""
    import time
    assert N.load().la_version() >= 500
    w = synth.make_uniform("wake", 51, 300, 100, 8, "uniform40")
    a = (w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
    exp = oracle.assign_flat(w.part_off, w.partition_id, oracle.compute_lags(w.begin, w.end, w.committed, False), w.cons_off, w.cons_rank)
    fresh = N.Context(0)
    try:
        fresh.wake()                                               # the very first thing a context is asked to do
        _same3(fresh.assign_batch(*a), exp, "fresh context")
    finally:
        fresh.close()
    ctx.wake()
    ctx.wake()
    _same3(ctx.assign_batch(*a), exp, "after two wakes")
    pipe, launches = ctx.last_pipeline(), ctx.last_launches()
    ctx.wake()
    assert (ctx.last_pipeline(), ctx.last_launches()) == (pipe, launches)
    z = synth.make_uniform("wake", 52, 3000, 256, 32, "zipf")
    big = z.end.copy()
    big[1234] = 1 << 56                                            # its tile needs wide records: breaks a bound of 2^31
    zb = (z.part_off, z.partition_id, z.begin, big, z.committed, N.LA_RESET_EARLIEST, z.cons_off, z.cons_rank)
    zexp = oracle.assign_flat(z.part_off, z.partition_id, oracle.compute_lags(z.begin, big, z.committed, False), z.cons_off, z.cons_rank)
    ctx.hint_next_call(((1 << 31), 255))                           # a promise the next call's data breaks ...
    ctx.wake()                                                     # ... still pending behind the wake
    with pytest.raises(N.LagAssignError):
        ctx.assign_batch(*zb)
    _same3(ctx.assign_batch(*zb), zexp, "and spent by that call")
    ctx.assign_batch(*a, keep_on_device=True)
    ctx.wake()
    with pytest.raises(N.LagAssignError):                          # the kept results went with the staging buffers
        ctx.group_last_by_member(w.n_partitions, 8)
    t0 = time.perf_counter()
    for _ in range(100):
        ctx.wake()
    assert (time.perf_counter() - t0) / 100 < 500e-6               # (warm: a one-partition call)
    _same3(ctx.assign_batch(*a), exp, "behind 100 wakes")
    multi = N.Context([0, 0], flags=N.LA_CREATE_SPLIT_ALWAYS | 2)
    try:
        multi.wake()
        _same3(multi.assign_batch(*a), exp, "two shards")
    finally:
        multi.close()
    assert N.load().la_wake(None) == N.LA_EINVAL
