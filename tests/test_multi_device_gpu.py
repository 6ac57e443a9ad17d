"""Multi-device contexts (la_create_multi) and the chunked host-buffer pipeline, bit-exact against the oracle.

A gpurun box has ONE MI355X, so the shard split is exercised with several logical shards mapped to device 0
(an id may repeat in la_create_multi) and LA_CREATE_SPLIT_ALWAYS, which makes the library shard and chunk
batches of any size: same planner, same threads, same per-shard buffers and offsets as on an 8-GPU node; only
the device ids differ.  Everything goes through the C ABI.
"""
import numpy as np
import pytest

from kafka_lag_based_assignor_amd import _native as N
from kafka_lag_based_assignor_amd import synth
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx4():
    c = N.Context([0, 0, 0, 0], flags=N.LA_CREATE_SPLIT_ALWAYS)
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx1_chunked():
    c = N.Context(0, flags=N.LA_CREATE_SPLIT_ALWAYS | 3)        # one shard, three lanes, three chunks per call
    yield c
    c.close()


@pytest.fixture(scope="module")
def ctx1():
    c = N.Context(0)
    yield c
    c.close()


def _same(got, exp, what=""):
    for g, e, name in zip(got, exp, ("partition order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg="%s %s" % (name, what))


def _mixed_batch(seed):
    """tile, block and large topics, topics without partitions, topics without consumers -- in one batch"""
    rng = np.random.default_rng(seed)
    ps = [10, 2000, 0, 300, 1500, 64, 5000, 7, 0, 20000, 130, 1, 0, 900, 12]
    cs = [2, 5, 3, 64, 100, 0, 200, 8, 0, 10, 300, 1, 2500, 33, 0]
    order = rng.permutation(len(ps))
    ps = [ps[i] for i in order]
    cs = [cs[i] for i in order]
    part_off = np.concatenate([[0], np.cumsum(ps)]).astype(np.int64)
    cons_off = np.concatenate([[0], np.cumsum(cs)]).astype(np.int64)
    pid = np.concatenate([rng.permutation(p) for p in ps]).astype(np.int32)
    lag = rng.integers(0, 1 << 30, part_off[-1]).astype(np.int64)
    ranks = np.concatenate([np.sort(rng.choice(4000, c, replace=False)) for c in cs]).astype(np.int32)
    return part_off, pid, lag, cons_off, ranks


def test_create_multi_shapes(ctx4):
    assert ctx4.shard_count == 4 and [ctx4.shard_device(i) for i in range(4)] == [0, 0, 0, 0]
    assert N.device_count() >= 1
    every = N.Context("all")                                   # n_devices = 0: every device of the node
    assert every.shard_count == N.device_count()
    every.close()
    with pytest.raises(N.LagAssignError) as ei:
        N.Context([0, 99])
    assert ei.value.code == N.LA_ENODEV


def test_distinct_devices_when_the_box_has_them():
    """Fires the day a box with >= 2 GPUs runs the tests: la_create_multi over DISTINCT device ids, the form an 8-GPU
    Kafka group leader uses.  Same checks as the logical-shard tests: cfg4 at full size, a mixed batch, the lists."""
    n = N.device_count()
    if n < 2:
        pytest.skip("one GPU on this box: distinct-device shards are exercised as logical shards on device 0")
    c = N.Context(list(range(n)), flags=N.LA_CREATE_SPLIT_ALWAYS)
    try:
        assert c.shard_count == n and [c.shard_device(i) for i in range(n)] == list(range(n))
        w = synth.config("cfg4")
        lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
        exp = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
        got = c.assign_batch(w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST,
                             w.cons_off, w.cons_rank)
        _same(got, exp, "cfg4 on %d devices" % n)
        assert list(c.last_shard_bounds()) == list(N.plan_shards(w.part_off, n))
        part_off, pid, lag2, cons_off, ranks = _mixed_batch(5)
        _same(c.assign_batch_lags(part_off, pid, lag2, cons_off, ranks),
              oracle.assign_flat(part_off, pid, lag2, cons_off, ranks), "mixed batch on %d devices" % n)
    finally:
        c.close()


def test_four_shards_cfg4_full_size(ctx4):
    w = synth.config("cfg4")                                   # 100 000 topics x 64 partitions x 8 consumers
    for mode in (N.LA_RESET_LATEST, N.LA_RESET_EARLIEST):
        latest = mode == N.LA_RESET_LATEST
        lag = oracle.compute_lags(w.begin, w.end, w.committed, latest)
        exp = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
        got = ctx4.assign_batch(w.part_off, w.partition_id, None if latest else w.begin, w.end, w.committed, mode,
                                w.cons_off, w.cons_rank)
        _same(got, exp, "cfg4 on 4 shards")
    # the split the call used is the planner's
    np.testing.assert_array_equal(ctx4.last_shard_bounds(), N.plan_shards(w.part_off, 4))
    assert ctx4.last_shard_bounds().tolist() == [0, 25000, 50000, 75000, 100000]



def _device_shard(torch, w, t0, t1, dev):
    """Topics [t0, t1) of w resident on `dev`, with result buffers and the la_device_batch over them."""
    import ctypes
    from kafka_lag_based_assignor_amd import sharding
    po, co, ps, cs = sharding.shard_slices(w.part_off, w.cons_off, t0, t1)
    po, co = np.ascontiguousarray(po), np.ascontiguousarray(co)
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)      # noqa: E731
    d = dict(po=up(po), co=up(co), pid=up(w.partition_id[ps]), begin=up(w.begin[ps]), end=up(w.end[ps]),
             com=up(w.committed[ps]), cr=up(w.cons_rank[cs]))
    n, k = int(po[-1]), int(co[-1])
    out = dict(pid=torch.full((max(n, 1),), -7, device=dev, dtype=torch.int32),
               rank=torch.full((max(n, 1),), -7, device=dev, dtype=torch.int32),
               total=torch.zeros(max(k, 1), device=dev, dtype=torch.int64))
    b = N.DeviceBatch()
    b.n_topics = t1 - t0
    b.reset_mode = N.LA_RESET_EARLIEST
    b.algo = N.LA_ALGO_AUTO
    b.n_partitions, b.n_consumers = n, k
    b.max_partitions_per_topic = int(np.diff(po).max()) if t1 > t0 else 0
    b.max_consumers_per_topic = int(np.diff(co).max()) if t1 > t0 else 0
    b.d_part_off, b.d_cons_off = d["po"].data_ptr(), d["co"].data_ptr()
    b.d_partition_id, b.d_begin_off, b.d_end_off = d["pid"].data_ptr(), d["begin"].data_ptr(), d["end"].data_ptr()
    b.d_committed_off, b.d_cons_rank = d["com"].data_ptr(), d["cr"].data_ptr()
    b.d_out_partition, b.d_out_member_rank = out["pid"].data_ptr(), out["rank"].data_ptr()
    b.d_out_total_lag = out["total"].data_ptr()
    b.h_part_off = po.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    b.h_cons_off = co.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    return dict(b=b, d=d, out=out, n=n, k=k, keep=(po, co))


def test_device_entry_points_on_every_shard(ctx4):
    """la_assign_batch_device_on / la_sync_on / la_shard_stream: a caller whose data is already in HBM drives every shard
    of one context -- cfg4 at full size split by la_plan_shards, each range enqueued on its shard's own stream back to
    back from this one thread, then every shard synced; the concatenation is the oracle's result."""
    import torch
    w = synth.config("cfg4")
    lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
    exp = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
    S = ctx4.shard_count
    bounds = N.plan_shards(w.part_off, S)
    shards = []
    for i in range(S):
        dev = torch.device("cuda", ctx4.shard_device(i))
        shards.append(_device_shard(torch, w, int(bounds[i]), int(bounds[i + 1]), dev))
    torch.cuda.synchronize()
    streams = [ctx4.shard_stream(i) for i in range(S)]
    assert len(set(streams)) == S and all(streams)
    for i in range(S):                                          # enqueue everywhere first ...
        ctx4.assign_batch_device(shards[i]["b"], streams[i], shard=i)
    for i in range(S):                                          # ... then wait
        ctx4.sync(streams[i], shard=i)
    got = tuple(np.concatenate([sh["out"][key][: sh[cnt]].cpu().numpy() for sh in shards])
                for key, cnt in (("pid", "n"), ("rank", "n"), ("total", "k")))
    _same(got, exp, "device entry points on 4 shards")
    # a mixed batch (tile + block + large topics) on the LAST shard only: scratch is per shard
    part_off, pid, lag2, cons_off, ranks = _mixed_batch(11)
    z = np.zeros_like(lag2)
    wm = synth.Workload("mixed", len(part_off) - 1, part_off, pid, z, lag2.copy(), z, lag2, cons_off, ranks, 0, 0)
    one = _device_shard(torch, wm, 0, wm.n_topics, torch.device("cuda", ctx4.shard_device(S - 1)))
    ctx4.assign_batch_device(one["b"], streams[S - 1], shard=S - 1)
    ctx4.sync(streams[S - 1], shard=S - 1)
    got = tuple(one["out"][key][: one[cnt]].cpu().numpy() for key, cnt in (("pid", "n"), ("rank", "n"), ("total", "k")))
    _same(got, oracle.assign_flat(part_off, pid, lag2, cons_off, ranks), "mixed batch on the last shard")
    with pytest.raises(N.LagAssignError) as ei:
        ctx4.assign_batch_device(one["b"], 0, shard=S)
    assert ei.value.code == N.LA_EINVAL


def test_native_rccl_allgather_of_the_results(ctx4):
    """la_allgather_results: the north star's single RCCL all-gather as a native call site (ncclCommInitAll + one
    ncclAllGather per shard in a group), for a process that drives all the node's GPUs itself.  A one-GPU box can run it
    at one rank -- RCCL init, the collective on the shard's stream behind the kernels, la_sync_on -- and with every GPU
    of a bigger box (distinct devices); several shards on one device are refused with a clear error, as RCCL would."""
    import torch
    n_dev = N.device_count()
    c = N.Context(list(range(n_dev)))                              # one shard per physical device
    try:
        S = c.shard_count
        w = synth.config("cfg4", 0.05)
        lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
        exp = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
        bounds = N.plan_shards(w.part_off, S)
        counts = [int(w.part_off[bounds[i + 1]] - w.part_off[bounds[i]]) for i in range(S)]
        cap = max(counts)
        shards, packed, recv = [], [], []
        for i in range(S):
            dev = torch.device("cuda", c.shard_device(i))
            sh = _device_shard(torch, w, int(bounds[i]), int(bounds[i + 1]), dev)
            buf = torch.zeros(2 * cap, device=dev, dtype=torch.int32)          # [2, cap]: partition order | member rank
            sh["b"].d_out_partition = buf.data_ptr()
            sh["b"].d_out_member_rank = buf.data_ptr() + 4 * cap
            shards.append(sh); packed.append(buf)
            recv.append(torch.full((S * 2 * cap,), -9, device=dev, dtype=torch.int32))
        torch.cuda.synchronize()
        for i in range(S):
            c.assign_batch_device(shards[i]["b"], c.shard_stream(i), shard=i)
        c.allgather_results(2 * cap, [p.data_ptr() for p in packed], [r.data_ptr() for r in recv])   # behind the kernels
        for i in range(S):
            c.sync(c.shard_stream(i), shard=i)
        for i in range(S):                                          # every device holds the global assignment
            g = recv[i].cpu().numpy().reshape(S, 2, cap)
            pid = np.concatenate([g[r, 0, :counts[r]] for r in range(S)])
            rank = np.concatenate([g[r, 1, :counts[r]] for r in range(S)])
            np.testing.assert_array_equal(pid, exp[0], err_msg="gathered partition order on device %d" % i)
            np.testing.assert_array_equal(rank, exp[1], err_msg="gathered member ranks on device %d" % i)
        # the same reassembly through the narrow wire format (round 4): pack on every shard, ONE all-gather of 2-byte elements per
        # shard in a group, unpack on every device -- the whole native N > 1 step of a process that drives all GPUs itself
        fmt = N.wire_format_for(int(w.partition_id.max()), int(w.cons_rank.max()) + 1)
        assert fmt.elem_bytes == 2
        capw = (cap + 7) // 8 * 8
        wire_s, wire_r, outs = [], [], []
        for i in range(S):
            dev = torch.device("cuda", c.shard_device(i))
            wire_s.append(torch.zeros(capw * 2, device=dev, dtype=torch.uint8))
            wire_r.append(torch.full((S * capw * 2,), 0xEE, device=dev, dtype=torch.uint8))
            outs.append(torch.full((2 * S * capw,), -9, device=dev, dtype=torch.int32))
        for i in range(S):
            c.assign_batch_device(shards[i]["b"], c.shard_stream(i), shard=i)
            c.pack_results(counts[i], packed[i].data_ptr(), packed[i].data_ptr() + 4 * cap, fmt, wire_s[i].data_ptr(),
                           c.shard_stream(i), shard=i)
        c.allgather_packed(capw, fmt.elem_bytes, [x.data_ptr() for x in wire_s], [x.data_ptr() for x in wire_r])
        for i in range(S):
            c.unpack_results(S * capw, wire_r[i].data_ptr(), fmt, outs[i].data_ptr(), outs[i].data_ptr() + 4 * S * capw,
                             c.shard_stream(i), shard=i)
            c.sync(c.shard_stream(i), shard=i)
        for i in range(S):
            g = outs[i].cpu().numpy().reshape(2, S, capw)
            np.testing.assert_array_equal(np.concatenate([g[0, r, :counts[r]] for r in range(S)]), exp[0])
            np.testing.assert_array_equal(np.concatenate([g[1, r, :counts[r]] for r in range(S)]), exp[1])
    finally:
        c.close()
    # several shards on ONE device: refused
    d = torch.zeros(8, device="cuda:0", dtype=torch.int32)
    with pytest.raises(N.LagAssignError) as ei:
        ctx4.allgather_results(2, [d.data_ptr()] * 4, [d.data_ptr()] * 4)
    assert ei.value.code == N.LA_EINVAL and "distinct device" in str(ei.value)
    with pytest.raises(N.LagAssignError) as ei:
        ctx4.allgather_packed(2, 2, [d.data_ptr()] * 4, [d.data_ptr()] * 4)
    assert ei.value.code == N.LA_EINVAL and "distinct device" in str(ei.value)
    with pytest.raises(N.LagAssignError):
        ctx4.allgather_packed(2, 3, [d.data_ptr()] * 4, [d.data_ptr()] * 4)          # an element is 2, 4 or 8 bytes
    # argument errors of the per-shard entry points: codes, not crashes
    import ctypes
    lib = ctx4._lib
    assert lib.la_sync_on(ctx4._h, 7, None) == N.LA_EINVAL
    assert lib.la_allgather_results(ctx4._h, -1, None, None) == N.LA_EINVAL
    assert lib.la_allgather_results(None, 1, None, None) == N.LA_EINVAL
    assert lib.la_shard_stream(ctx4._h, 9) is None and lib.la_device_features(ctx4._h, -1) == N.LA_EINVAL
    assert lib.la_last_pipeline(None) == N.LA_EINVAL
    one = N.Context(0)
    try:
        null = (ctypes.c_void_p * 1)(None)
        assert lib.la_allgather_results(one._h, 4, null, null) == N.LA_EINVAL        # a shard's buffer is NULL
        assert lib.la_allgather_results(one._h, 0, null, null) == N.LA_OK            # nothing to gather
    finally:
        one.close()


def test_compute_lag_and_group_by_member_across_shards(ctx4, ctx1):
    """la_compute_lag and la_group_by_member split large inputs over the shards (element ranges / topic ranges); the
    results equal the one-device ones."""
    w = synth.config("cfg4", 0.2)
    for latest in (True, False):
        exp = oracle.compute_lags(w.begin, w.end, w.committed, latest)
        mode = N.LA_RESET_LATEST if latest else N.LA_RESET_EARLIEST
        np.testing.assert_array_equal(ctx4.compute_lag(None if latest else w.begin, w.end, w.committed, mode), exp)
        np.testing.assert_array_equal(ctx1.compute_lag(None if latest else w.begin, w.end, w.committed, mode), exp)
    lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
    pid, rank, _ = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
    a = ctx4.group_by_member(w.part_off, pid, rank, 8)
    b = ctx1.group_by_member(w.part_off, pid, rank, 8)
    for x, y, what in zip(a, b, ("member_off", "grouped_topic", "grouped_partition")):
        np.testing.assert_array_equal(x, y, err_msg=what)
    order = np.argsort(rank, kind="stable")                     # the reference's lists: stable by member
    np.testing.assert_array_equal(a[2], pid[order])


@pytest.mark.parametrize("seed", range(6))
def test_four_shards_mixed_paths(ctx4, seed):
    part_off, pid, lag, cons_off, ranks = _mixed_batch(seed)
    exp = oracle.assign_flat(part_off, pid, lag, cons_off, ranks)
    _same(ctx4.assign_batch_lags(part_off, pid, lag, cons_off, ranks), exp, "mixed batch, seed %d" % seed)


def test_four_shards_ragged_all_three_paths(ctx4, ctx1_chunked):
    from test_gpu_parity import _ragged_skewed
    w = _ragged_skewed(5, 8000, 5)
    rng = np.random.default_rng(6)
    extra_p = [3000, 150, 20000]
    extra_c = [40, 300, 10]
    part_off = np.concatenate([w.part_off, w.part_off[-1] + np.cumsum(extra_p)]).astype(np.int64)
    cons_off = np.concatenate([w.cons_off, w.cons_off[-1] + np.cumsum(extra_c)]).astype(np.int64)
    pid = np.concatenate([w.partition_id] + [rng.permutation(p).astype(np.int32) for p in extra_p])
    lag = np.concatenate([w.lag, rng.integers(0, 1 << 30, sum(extra_p)).astype(np.int64)])
    ranks = np.concatenate([w.cons_rank] + [np.arange(c, dtype=np.int32) for c in extra_c])
    exp = oracle.assign_flat(part_off, pid, lag, cons_off, ranks)
    _same(ctx4.assign_batch_lags(part_off, pid, lag, cons_off, ranks), exp, "4 shards")
    _same(ctx1_chunked.assign_batch_lags(part_off, pid, lag, cons_off, ranks), exp, "1 shard, 3 chunks")


def test_named_configs_on_shards_and_chunks(ctx4, ctx1_chunked):
    for name, scale in (("cfg1", 1.0), ("cfg2b", 1.0), ("cfg3", 0.2), ("cfg5", 1.0 / 64), ("block_b", 0.1)):
        w = synth.config(name, scale)
        lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
        exp = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
        for c, what in ((ctx4, "4 shards"), (ctx1_chunked, "3 chunks")):
            got = c.assign_batch(w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST,
                                 w.cons_off, w.cons_rank)
            _same(got, exp, "%s %s" % (name, what))


def test_fewer_topics_than_shards(ctx4):
    # 1, 2, 3 topics on 4 shards; and the README example
    w = synth.config("cfg1")
    p, m, t = ctx4.assign_batch_lags(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    assert p.tolist() == [0, 2, 1] and m.tolist() == [0, 1, 1] and t.tolist() == [100000, 110000]
    p, m, t = ctx4.assign_batch_lags([0, 4, 6], [0, 1, 2, 3, 0, 1], [100000, 100000, 500, 1, 900000, 100000],
                                     [0, 2, 3], [0, 1, 0])                     # Test.java:82-132, flat form
    assert p.tolist() == [0, 1, 2, 3, 0, 1] and m.tolist() == [0, 1, 0, 1, 0, 0] and t.tolist() == [100500, 100001, 1000000]


def test_group_last_by_member_across_shards(ctx4, ctx1):
    # every member's list is the concatenation, in shard order, of the shards' lists: compare with the grouping of the
    # downloaded global arrays on one device
    cases = [synth.ragged(31, 400, 300, 40, negative=True), synth.config("block_b", 0.1), synth.config("cfg3", 0.2)]
    part_off, pid, lag, cons_off, ranks = _mixed_batch(2)
    cases.append(synth.Workload("mixed", part_off.size - 1, part_off, pid, np.zeros_like(lag), lag.copy(),
                                np.zeros_like(lag), lag, cons_off, ranks, 20000, 2500))
    for w in cases:
        n_members = int(w.cons_rank.max()) + 1 if w.cons_rank.size else 0
        exp_p, exp_m, exp_t = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
        want = ctx1.group_by_member(w.part_off, exp_p, exp_m, n_members)
        p, m, t = ctx4.assign_batch_lags(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank, keep_on_device=True)
        assert p is None and m is None
        np.testing.assert_array_equal(t, exp_t)
        assert ctx4.last_shard_bounds().size - 1 == min(4, w.part_off.size - 1)
        got = ctx4.group_last_by_member(int(w.part_off[-1]), n_members)
        for g, e, what in zip(got, want, ("member_off", "grouped_topic", "grouped_partition")):
            np.testing.assert_array_equal(g, e, err_msg="%s %s" % (w.name, what))


def test_reference_lists_across_shards(ctx4):                  # Test.java:82-132: the exact lists, 2 topics on 2 shards
    p, m, t = ctx4.assign_batch_lags([0, 4, 6], [0, 1, 2, 3, 0, 1], [100000, 100000, 500, 1, 900000, 100000],
                                     [0, 2, 3], [0, 1, 0], keep_on_device=True)
    off, g_t, g_p = ctx4.group_last_by_member(6, 2)
    lists = [[(int(a), int(b)) for a, b in zip(g_t[off[r]:off[r + 1]], g_p[off[r]:off[r + 1]])] for r in range(2)]
    assert lists[0] == [(0, 0), (0, 2), (1, 0), (1, 1)]        # consumer-1: topic1-0, topic1-2, topic2-0, topic2-1
    assert lists[1] == [(0, 1), (0, 3)]                        # consumer-2: topic1-1, topic1-3


def test_errors_from_a_late_chunk_are_reported(ctx4, ctx1_chunked):
    w = synth.ragged(7, 300, 100, 12)
    bad = w.cons_rank.copy()
    k = int(w.cons_off[-2])                                    # first consumer of the LAST topic
    if w.cons_off[-1] - k >= 2:
        bad[k], bad[k + 1] = bad[k + 1], bad[k]
    else:
        pytest.skip("last topic has fewer than two consumers")
    for c in (ctx4, ctx1_chunked):
        with pytest.raises(N.LagAssignError) as ei:
            c.assign_batch_lags(w.part_off, w.partition_id, w.lag, w.cons_off, bad)
        assert ei.value.code == N.LA_EINVAL
        # the context is usable afterwards
        exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
        _same(c.assign_batch_lags(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank), exp, "after an error")


def test_default_context_pipelines_a_large_batch(ctx1):
    # 8 M partitions: 16 chunks over 3 lanes (H2D of one chunk under the kernels / D2H of others)
    w = synth.config("cfg4", 1.25)
    assert w.n_partitions == 8_000_000
    lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
    exp = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
    got = ctx1.assign_batch(w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST,
                            w.cons_off, w.cons_rank)
    _same(got, exp, "pipelined")
    assert ctx1.last_shard_bounds().tolist() == [0, w.n_topics]


@pytest.fixture(params=["mapped", "streams"])
def pinned_pipeline(request):
    """Pinned caller arrays take one of two forms: `mapped` (the default since round 4: the kernels read and write the arrays
    in place over PCIe, no copies) or `streams` (LA_NO_MAPPED_PIPELINE=1: the three-stream copy pipeline of round 3)."""
    import os
    if request.param == "streams":
        os.environ["LA_NO_MAPPED_PIPELINE"] = "1"
    yield N.LA_PIPELINE_MAPPED if request.param == "mapped" else N.LA_PIPELINE_STREAMS
    os.environ.pop("LA_NO_MAPPED_PIPELINE", None)


def test_pinned_host_arrays(ctx1, pinned_pipeline):
    w = synth.config("cfg3", 2.0)
    pin = {k: ctx1.host_alloc(getattr(w, k).shape, getattr(w, k).dtype) for k in
           ("part_off", "partition_id", "begin", "end", "committed", "cons_off", "cons_rank")}
    for k, a in pin.items():
        a[...] = getattr(w, k)
    out = (ctx1.host_alloc((w.n_partitions,), np.int32), ctx1.host_alloc((w.n_partitions,), np.int32),
           ctx1.host_alloc((w.cons_rank.size,), np.int64))
    lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
    exp = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
    got = ctx1.assign_batch(pin["part_off"], pin["partition_id"], pin["begin"], pin["end"], pin["committed"],
                            N.LA_RESET_EARLIEST, pin["cons_off"], pin["cons_rank"], out=out)
    _same(got, exp, "pinned")
    assert ctx1.last_pipeline() == pinned_pipeline              # every array pinned: no worker threads
    got = ctx1.assign_batch(w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
    _same(got, exp, "pageable")
    assert ctx1.last_pipeline() == N.LA_PIPELINE_LANES
    # one pageable array among pinned ones is enough to need the threads
    got = ctx1.assign_batch(pin["part_off"], pin["partition_id"], w.begin, pin["end"], pin["committed"],
                            N.LA_RESET_EARLIEST, pin["cons_off"], pin["cons_rank"], out=out)
    _same(got, exp, "mixed pinned / pageable")
    assert ctx1.last_pipeline() == N.LA_PIPELINE_LANES


def _pin(ctx, a):
    out = ctx.host_alloc(a.shape, a.dtype)
    out[...] = a
    return out


@pytest.mark.parametrize("which", ["4 shards", "1 shard, 3 chunks", "default context"])
def test_stream_pipeline_on_pinned_arrays(ctx4, ctx1_chunked, ctx1, which, pinned_pipeline):
    """The two forms for pinned caller arrays (in place / three streams) over shards x chunks: a mixed batch through tile, block and large
    paths, the target shape, results left on the device + grouped lists, LATEST mode without begin, and an unsorted
    cons_rank segment in a late chunk -- same results and errors as the lanes form."""
    c = {"4 shards": ctx4, "1 shard, 3 chunks": ctx1_chunked, "default context": ctx1}[which]
    part_off, pid, lag, cons_off, ranks = _mixed_batch(3)
    exp = oracle.assign_flat(part_off, pid, lag, cons_off, ranks)
    P = [_pin(c, np.ascontiguousarray(a)) for a in (part_off, pid, lag, cons_off, ranks)]
    out = (c.host_alloc((pid.size,), np.int32), c.host_alloc((pid.size,), np.int32), c.host_alloc((ranks.size,), np.int64))
    _same(c.assign_batch_lags(*P, out=out), exp, "mixed batch, pinned, " + which)
    if which != "default context":                              # (the default context sends a batch this small in one copy)
        assert c.last_pipeline() == pinned_pipeline
    w = synth.config("target", 0.03 if which == "default context" else 0.01)
    names = ("part_off", "partition_id", "begin", "end", "committed", "cons_off", "cons_rank")
    pw = {k: _pin(c, getattr(w, k)) for k in names}
    for latest in (False, True):
        mode = N.LA_RESET_LATEST if latest else N.LA_RESET_EARLIEST
        lags = oracle.compute_lags(w.begin, w.end, w.committed, latest)
        e = oracle.assign_flat(w.part_off, w.partition_id, lags, w.cons_off, w.cons_rank)
        o = (c.host_alloc((w.n_partitions,), np.int32), c.host_alloc((w.n_partitions,), np.int32),
             c.host_alloc((w.cons_rank.size,), np.int64))
        got = c.assign_batch(pw["part_off"], pw["partition_id"], None if latest else pw["begin"], pw["end"], pw["committed"],
                             mode, pw["cons_off"], pw["cons_rank"], out=o)
        _same(got, e, "target, pinned, latest=%s, %s" % (latest, which))
        assert c.last_pipeline() == pinned_pipeline
    # results stay on the device, only the grouped lists come back
    c.assign_batch(pw["part_off"], pw["partition_id"], pw["begin"], pw["end"], pw["committed"], N.LA_RESET_EARLIEST,
                   pw["cons_off"], pw["cons_rank"], want_totals=False, keep_on_device=True)
    off, g_t, g_p = c.group_last_by_member(w.n_partitions, 32)
    lags = oracle.compute_lags(w.begin, w.end, w.committed, False)
    e_p, e_m, _ = oracle.assign_flat(w.part_off, w.partition_id, lags, w.cons_off, w.cons_rank)
    order = np.argsort(e_m, kind="stable")
    np.testing.assert_array_equal(g_p, e_p[order])
    np.testing.assert_array_equal(off, np.searchsorted(e_m[order], np.arange(33)))
    # an unsorted consumer segment in the LAST topic: reported, nothing left in flight, the next call works
    bad = pw["cons_rank"].copy()
    badp = _pin(c, bad)
    badp[-1], badp[-2] = bad[-2], bad[-1]
    with pytest.raises(N.LagAssignError) as ei:
        c.assign_batch(pw["part_off"], pw["partition_id"], pw["begin"], pw["end"], pw["committed"], N.LA_RESET_EARLIEST,
                       pw["cons_off"], badp, out=o)
    assert ei.value.code == N.LA_EINVAL and c.last_pipeline() == pinned_pipeline
    got = c.assign_batch(pw["part_off"], pw["partition_id"], pw["begin"], pw["end"], pw["committed"], N.LA_RESET_EARLIEST,
                         pw["cons_off"], pw["cons_rank"], out=o)
    np.testing.assert_array_equal(got[0], e_p)
    np.testing.assert_array_equal(got[1], e_m)


def test_large_path_topic_without_partitions_reports_zero_totals(ctx1):
    # a topic with no partition metadata and more consumers than the block path holds
    part_off = np.array([0, 50, 50, 80], dtype=np.int64)
    cons_off = np.array([0, 4, 3004, 3010], dtype=np.int64)
    rng = np.random.default_rng(9)
    pid = np.concatenate([rng.permutation(50), rng.permutation(30)]).astype(np.int32)
    lag = rng.integers(0, 1 << 20, 80).astype(np.int64)
    ranks = np.concatenate([np.arange(4), np.arange(3000), np.arange(6)]).astype(np.int32)
    exp = oracle.assign_flat(part_off, pid, lag, cons_off, ranks)
    assert not exp[2][4:3004].any()
    for _ in range(2):                                         # twice: the scratch holds the first run's totals
        _same(ctx1.assign_batch_lags(part_off, pid, lag, cons_off, ranks), exp, "empty large topic")


def test_inline_launch_between_deferring_launches():
    # big batch A defers tiles into one counter of the pair, small batch B takes the single-launch form
    # (no wide kernel to zero the other counter), big batch C must start counting at zero again
    from test_gpu_parity import _deferring_workload, _run_device
    c = N.Context(0)
    big = _deferring_workload()
    small = synth.ragged(3, 40, 8, 8, negative=True)
    exp_big = oracle.assign_flat(big.part_off, big.partition_id, big.lag, big.cons_off, big.cons_rank)
    exp_small = oracle.assign_flat(small.part_off, small.partition_id, small.lag, small.cons_off, small.cons_rank)
    for rep in range(3):
        _same(_run_device(c, big, N.LA_ALGO_AUTO, use_lag=True), exp_big, "A/C rep %d" % rep)
        _same(_run_device(c, small, N.LA_ALGO_AUTO, use_lag=True), exp_small, "B rep %d" % rep)
    # a different, smaller deferring batch after the big one: a stale count would walk entries of the old list
    half = synth.Workload("defer/2", 50000, big.part_off[:50001], big.partition_id[:big.part_off[50000]],
                          big.begin[:big.part_off[50000]], big.end[:big.part_off[50000]],
                          big.committed[:big.part_off[50000]], big.lag[:big.part_off[50000]],
                          big.cons_off[:50001], big.cons_rank[:big.cons_off[50000]], 8, 8)
    exp_half = oracle.assign_flat(half.part_off, half.partition_id, half.lag, half.cons_off, half.cons_rank)
    _same(_run_device(c, big, N.LA_ALGO_AUTO, use_lag=True), exp_big, "A again")
    _same(_run_device(c, small, N.LA_ALGO_AUTO, use_lag=True), exp_small, "B again")
    _same(_run_device(c, half, N.LA_ALGO_AUTO, use_lag=True), exp_half, "C = half of A")
    c.close()


def test_small_batches_take_the_one_copy_form_and_agree_with_the_pipeline(ctx1, ctx1_chunked):
    """A call whose inputs and results fit 2 MB travels as one H2D + one D2H through a staging buffer (la_api.hip,
    assign_small); everything else takes the chunked pipeline.  Same results either way, around the threshold, with
    offsets or lags, with the results kept on the device for la_group_last_by_member, and when calls of both kinds
    alternate on one context."""
    for n_topics in (1, 40, 900, 1100, 1400, 3000):               # x 50 partitions x 5 consumers: 2 MB is ~1 130 topics
        w = synth.make_uniform("small", 21, n_topics, 50, 5, "uniform40")
        lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
        exp = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
        for c, what in ((ctx1, "default context"), (ctx1_chunked, "pipeline forced")):
            got = c.assign_batch(w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST,
                                 w.cons_off, w.cons_rank)
            _same(got, exp, "%d topics, offsets, %s" % (n_topics, what))
            _same(c.assign_batch_lags(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank), exp,
                  "%d topics, lags, %s" % (n_topics, what))
            lag_latest = oracle.compute_lags(w.begin, w.end, w.committed, True)
            e_p, e_m, e_t = oracle.assign_flat(w.part_off, w.partition_id, lag_latest, w.cons_off, w.cons_rank)
            want = ctx1.group_by_member(w.part_off, e_p, e_m, 5)      # (a host-buffer call: before the one under test)
            p, m, t = c.assign_batch(w.part_off, w.partition_id, None, w.end, w.committed, N.LA_RESET_LATEST,
                                     w.cons_off, w.cons_rank, keep_on_device=True)
            assert p is None and m is None
            np.testing.assert_array_equal(t, e_t)
            got_g = c.group_last_by_member(w.n_partitions, 5)
            for g, e, name in zip(got_g, want, ("member_off", "grouped_topic", "grouped_partition")):
                np.testing.assert_array_equal(g, e, err_msg="%s, %d topics, %s" % (name, n_topics, what))


def test_small_batch_errors_and_mixed_shapes(ctx1):
    # unsorted ranks are reported by the small form too, and the context works afterwards
    with pytest.raises(N.LagAssignError) as ei:
        ctx1.assign_batch_lags([0, 2], [0, 1], [5, 6], [0, 2], [3, 1])
    assert ei.value.code == N.LA_EINVAL
    # tile + block + large topics in one small call
    part_off, pid, lag, cons_off, ranks = _mixed_batch(4)
    exp = oracle.assign_flat(part_off, pid, lag, cons_off, ranks)
    for _ in range(2):
        _same(ctx1.assign_batch_lags(part_off, pid, lag, cons_off, ranks), exp, "mixed shapes, small form")
    # a big call between two small ones: the results la_group_last_by_member groups are the LAST call's
    big = synth.config("cfg4", 0.2)
    small = synth.config("cfg3", 0.004)
    for w in (small, big, small):
        lag_w = oracle.compute_lags(w.begin, w.end, w.committed, False)
        e_p, e_m, e_t = oracle.assign_flat(w.part_off, w.partition_id, lag_w, w.cons_off, w.cons_rank)
        ctx1.assign_batch(w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off,
                          w.cons_rank, keep_on_device=True)
        n_members = int(w.cons_rank.max()) + 1
        want = (np.concatenate([[0], np.cumsum(np.bincount(e_m, minlength=n_members))]),)
        got_g = ctx1.group_last_by_member(w.n_partitions, n_members)
        np.testing.assert_array_equal(got_g[0], want[0], err_msg=w.name)
        order = np.argsort(e_m, kind="stable")
        np.testing.assert_array_equal(got_g[2], e_p[order], err_msg=w.name)
