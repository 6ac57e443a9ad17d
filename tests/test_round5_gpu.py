"""Round-5 additions, through the C ABI, bit-exact against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from kafka_lag_based_assignor_amd import _native as N
from kafka_lag_based_assignor_amd import synth
from oracle import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    c = N.Context(0)
    yield c
    c.close()


def _same3(got, exp, what=""):
    for g, e, name in zip(got, exp, ("partition order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg="%s %s" % (name, what))


# ---- block_sort_radix, a padding sentinel against the record whose examined bits are all ones ------------
def test_block_radix_sentinel_against_all_ones_record_in_fresh_process():
    """The workgroup's digit sort pads with all-ones sentinels and looks at ceil((lbw + sh) / 8) digits.  When lbw + sh is a
    multiple of 8, the record (lag 0, id 2^sh - 1) has the sentinel's digits; a sentinel of an earlier wavefront may then sort
    before it.  P is not a multiple of 64 x 8, the special record sits in the second wavefront, lbw + sh in {8, 16, 24}; every
    topic through the digits (LA_BLOCK_RADIX=3, read once per process), against the oracle."""
    code = r"""
import numpy as np, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from kafka_lag_based_assignor_amd import _native as N, synth
from oracle import oracle
import test_round4_gpu as t
ctx = N.Context(0)
rng = np.random.default_rng(5)
for P, C, sh, lbw in ((924, 65, 10, 6), (924, 70, 10, 14), (924, 65, 4, 4), (1500, 3, 11, 5), (3000, 100, 12, 12), (9000, 8, 14, 10),
                      (924, 65, 10, 7)):
    for at in (64, 100, 127, P - 1):
        ids = rng.permutation((1 << sh) - 1)[:P] if (1 << sh) - 1 >= P else rng.integers(0, (1 << sh) - 1, P)
        ids = ids.astype(np.int32)
        lag = rng.integers(0, 1 << lbw, P).astype(np.int64)
        lag[0] = (1 << lbw) - 1                       # the OR of the lags has lbw bits
        ids[at] = (1 << sh) - 1                       # the record whose lbw + sh bits are all ones ...
        lag[at] = 0                                   # ... : lag_max - 0 = all ones, id all ones
        part_off = np.array([0, P], np.int64); cons_off = np.array([0, C], np.int64)
        ranks = np.arange(C, dtype=np.int32) * 2
        w = synth.Workload("s", 1, part_off, ids, np.zeros(P, np.int64), lag.copy(), np.zeros(P, np.int64), lag, cons_off, ranks, P, C)
        exp = oracle.assign_flat(part_off, ids, lag, cons_off, ranks)
        t._same3(t._device_call(ctx, w), exp, what=str((P, C, sh, lbw, at)))
print("ok")
"""
    env = dict(os.environ, LA_BLOCK_RADIX="3")
    out = subprocess.run([sys.executable, "-c", code % (ROOT, os.path.join(ROOT, "tests"))], env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0 and "ok" in out.stdout, (out.stdout[-1500:], out.stderr[-1500:])


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


# ---- one launch for a small rebalance -------------------------------------------------------------------
def _grouped_expect(w, n_members):
    lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
    e_pid, e_rank, e_tot = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
    order = np.argsort(e_rank, kind="stable")                     # member by member, inside a member in the reference's order
    first = np.searchsorted(e_rank[order], np.arange(n_members + 1))
    topic = (np.searchsorted(w.part_off, order, side="right") - 1).astype(np.int32)
    return first.astype(np.int64), topic, e_pid[order], e_tot, (e_pid, e_rank)


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
    import test_round4_gpu as t4
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
import test_round5_gpu as t
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


# ---- block path: 65 .. 256 consumers, the greedy's bins ordered through 32-bit keys -------------------------
def _one_topic(P, C, lag, seed):
    rng = np.random.default_rng(seed)
    pid = rng.permutation(P).astype(np.int32)
    ranks = np.sort(rng.choice(3 * C + 5, C, replace=False)).astype(np.int32)
    lag = np.asarray(lag, np.int64)
    return synth.Workload("k32", 1, np.array([0, P], np.int64), pid, np.zeros(P, np.int64), lag.copy(), np.zeros(P, np.int64), lag,
                          np.array([0, C], np.int64), ranks, P, C)


@pytest.mark.parametrize("P,C", [(10000, 128), (1100, 65), (2049, 100), (4097, 129), (8193, 200), (16384, 256), (300, 70), (5000, 255)])
@pytest.mark.parametrize("kind", ["u40", "bigties", "zero", "pareto", "u20", "nearties"])
def test_block_greedy_through_32_bit_keys(ctx, P, C, kind):
    """greedy_one_wave_key32 against the literal oracle: uniform 40-bit lags (bits dropped from the key, shared truncated totals
    rare), many EQUAL large lags (every round meets tied totals with bits dropped: the exact re-ordering runs), lags that differ
    only below the dropped bits, all-zero and small lags (drop == 0: the key is exact, memberId breaks the ties), a Pareto tail."""
    import test_round4_gpu as t4
    rng = np.random.default_rng(P + C)
    if kind == "u40":
        lag = rng.integers(0, 1 << 40, P)
    elif kind == "bigties":
        lag = (1 << 39) + rng.integers(0, 3, P) * (1 << 20)
    elif kind == "nearties":
        lag = (1 << 41) + rng.integers(0, 64, P)
    elif kind == "zero":
        lag = np.zeros(P, np.int64)
    elif kind == "u20":
        lag = rng.integers(0, 1 << 20, P)
    else:
        lag = np.floor(np.minimum(float(1 << 40), 1000.0 * (1.0 - rng.random(P)) ** (-1.0 / 1.5))).astype(np.int64)
    w = _one_topic(P, C, lag, P * 7 + C)
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    _same3(t4._device_call(ctx, w), exp, "%s %d x %d" % (kind, P, C))


def test_block_greedy_forms_agree_in_a_fresh_process():
    """LA_BLOCK_KEY32=0 (the 64-bit bins through the networks, rounds 3-4) and =2 (32-bit keys for 256 bins too) give what the
    default gives: the oracle's assignment.  With bigties / nearties lags the exact re-ordering of tied rounds runs in mode 2."""
    code = r"""
import numpy as np, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from kafka_lag_based_assignor_amd import _native as N
from oracle import oracle
import test_round4_gpu as t4, test_round5_gpu as t5
ctx = N.Context(0)
rng = np.random.default_rng(3)
for (P, C) in ((10000, 128), (3000, 200), (16000, 256), (1500, 66)):
    for lag in (rng.integers(0, 1 << 40, P), (1 << 39) + rng.integers(0, 3, P) * (1 << 20), (1 << 41) + rng.integers(0, 64, P)):
        w = t5._one_topic(P, C, lag, P + C)
        exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
        t4._same3(t4._device_call(ctx, w), exp, what=str((P, C)))
print("ok")
"""
    for mode in ("0", "2", "dense0"):                              # 2: the 32-bit-key form for 256 bins as well (default: 128 only)
        env = dict(os.environ, LA_BLOCK_KEY32=mode)
        if mode == "dense0":                                       # ... and without the one-scatter placement of dense ids in front of the digits
            env = dict(os.environ, LA_BLOCK_DENSE_IDS="0")
        out = subprocess.run([sys.executable, "-c", code % (ROOT, os.path.join(ROOT, "tests"))], env=env, capture_output=True,
                             text=True, timeout=600)
        assert out.returncode == 0 and "ok" in out.stdout, (mode, out.stdout[-1500:], out.stderr[-1500:])


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


@pytest.mark.parametrize("topics,p,c,dist", [(30000, 256, 32, "zipf"), (60000, 64, 8, "uniform40"), (20000, 1000, 64, "zipf"),
                                             (50000, 37, 5, "zipf"), (300000, 7, 3, "uniform40"), (9000, 256, 32, "zipf")])
def test_wire_out_equals_the_packed_results(ctx, topics, p, c, dist):
    """The wire elements the tile kernels write themselves are what la_pack_results_on makes of the two int32 arrays (checked
    against sharding.wire_pack_numpy of the ORACLE's arrays on a slice and of the plain device call's arrays in full); totals as
    usual.  Topic starts that are odd multiples of the element size (37, 7 partitions per topic): the 8-byte stores are only
    element-aligned.  A batch small enough to be resident at once (9 000 topics) still takes the one-launch wire form."""
    from kafka_lag_based_assignor_amd import sharding
    w = synth.make_uniform("wire", topics % 97, topics, p, c, dist)
    bounds = N.offset_bounds(w.begin, w.end, w.committed, w.partition_id)
    n_members = int(w.cons_rank.max()) + 1
    fmt = N.wire_format_for(int(w.partition_id.max()), n_members)
    assert fmt.elem_bytes in (2, 4)
    raw, tot = _wire_call(ctx, w, fmt, bounds)
    assert ctx.last_launches() == 1
    ref_p, ref_m, ref_t = ctx.assign_batch(w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off,
                                           w.cons_rank)
    np.testing.assert_array_equal(raw, sharding.pack_results_numpy(ref_p, ref_m, fmt.elem_bytes, fmt.id_bits))
    np.testing.assert_array_equal(tot, ref_t)
    t = min(topics, 300)
    p1, k1 = int(w.part_off[t]), int(w.cons_off[t])
    lag = oracle.compute_lags(w.begin[:p1], w.end[:p1], w.committed[:p1], False)
    e_pid, e_rank, _ = oracle.assign_flat(w.part_off[:t + 1], w.partition_id[:p1], lag, w.cons_off[:t + 1], w.cons_rank[:k1])
    np.testing.assert_array_equal(raw[:p1], sharding.pack_results_numpy(e_pid, e_rank, fmt.elem_bytes, fmt.id_bits))


def test_wire_out_topics_without_consumers_and_ragged_sizes(ctx):
    import test_round4_gpu as t4
    from kafka_lag_based_assignor_amd import sharding
    rng = np.random.default_rng(4)
    shapes = [(int(rng.integers(0, 257)), int(rng.integers(0, 33))) for _ in range(40000)]
    w0 = t4._batch_of(shapes, 12, kinds=["u20", "zero", "ties"])
    w = synth.Workload("ragged", w0.n_topics, w0.part_off, w0.partition_id, np.zeros_like(w0.lag), w0.lag.copy(), np.zeros_like(w0.lag),
                       w0.lag, w0.cons_off, w0.cons_rank, 256, 32)
    fmt = N.wire_format_for(255, int(w.cons_rank.max()) + 1)
    raw, tot = _wire_call(ctx, w, fmt, (1 << 20, 255), hint=(256, 32))
    e_pid, e_rank, e_tot = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    np.testing.assert_array_equal(raw, sharding.pack_results_numpy(e_pid, e_rank, fmt.elem_bytes, fmt.id_bits))   # rank -1 -> 0
    np.testing.assert_array_equal(tot, e_tot)


def test_wire_out_refuses_what_it_cannot_do(ctx):
    w = synth.make_uniform("wire", 5, 20000, 256, 32, "zipf")
    bounds = N.offset_bounds(w.begin, w.end, w.committed, w.partition_id)
    fmt = N.wire_format_for(255, 32)
    for kwargs in ({"bounds": None}, {"bounds": (1 << 60, 255)}, {"bounds": bounds, "flags": N.LA_FLAG_RAGGED},
                   {"bounds": bounds, "hint": (2000, 32)}):
        with pytest.raises(N.LagAssignError) as ei:
            _wire_call(ctx, w, fmt, kwargs.get("bounds"), flags=kwargs.get("flags", 0), hint=kwargs.get("hint"))
        assert ei.value.code == N.LA_EINVAL and "LA_FLAG_WIRE_OUT" in str(ei.value)
    small = N.WireFormat(2, 12)                                    # 4 bits above the id: member ranks up to 14 only
    with pytest.raises(N.LagAssignError) as ei:
        _wire_call(ctx, w, small, bounds)
    assert ei.value.code == N.LA_EINVAL and "wire format" in str(ei.value)
    raw, _ = _wire_call(ctx, w, fmt, bounds)                       # and the context is fine afterwards
    assert raw.size == w.n_partitions


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


# ---- under-filled tile launches run a wider shape; the narrow shapes at small sizes keep their tests -------------------------------
def test_tile_tests_with_the_narrow_shapes_in_a_fresh_process():
    """wave_tile_widen (la_wave_tile.hip) gives every topic of a small batch twice / four times the lanes: the whole suite's small tile
    batches now run the wide shapes.  The narrowest shape of every (partitions, consumers) -- what large batches run, and everything
    ran until round 5 -- is kept under test by running the tile tests once more with LA_NO_TILE_WIDEN=1."""
    env = dict(os.environ, LA_NO_TILE_WIDEN="1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-m", "gpu", "-x",
                          "-k", "tile or target_shape or ragged or grouped", "-p", "no:cacheprovider"],
                         env=env, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-1000:])


def test_widened_and_narrow_tile_shapes_agree(ctx):
    """The same small batches through the default (widened) pick and the oracle; shapes chosen so that the pick widens by 2x and 4x and
    not at all (64 consumers; one record per lane)."""
    import test_round4_gpu as t4
    for (t, p, c) in [(1000, 256, 32), (300, 64, 8), (50, 1000, 9), (2000, 100, 5), (7, 1024, 64), (400, 8, 8), (3000, 30, 3)]:
        w = synth.make_uniform("widen", t + p, t, p, c, "zipf")
        exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)     # (_device_call hands over w.lag)
        _same3(t4._device_call(ctx, w), exp, "%d x %d x %d" % (t, p, c))
