"""Parity of the HIP path (through the C ABI) against the CPU oracle, bit-exact.

Runs only on a real MI355X (`-m gpu`).  The oracle is the checker; the product path is
kafka_lag_based_assignor_amd._native -> liblagassign.so -> HIP kernels.
"""
import os

import numpy as np
import pytest

from kafka_lag_based_assignor_amd import _native as N
from kafka_lag_based_assignor_amd import synth
from oracle import oracle
from round_form import round_form as _round_form

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = N.Context(0)
    yield c
    c.close()


def _check_lags(ctx, w, what=""):
    """assign on precomputed lags == static assign(Map,Map) seam."""
    exp_p, exp_m, exp_t = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    got_p, got_m, got_t = ctx.assign_batch_lags(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    np.testing.assert_array_equal(got_p, exp_p, err_msg="partition order " + what)
    np.testing.assert_array_equal(got_m, exp_m, err_msg="member " + what)
    np.testing.assert_array_equal(got_t, exp_t, err_msg="totals " + what)


def _check_offsets(ctx, w, mode, what=""):
    latest = mode == N.LA_RESET_LATEST
    lag = oracle.compute_lags(w.begin, w.end, w.committed, latest)
    exp_p, exp_m, exp_t = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
    got_p, got_m, got_t = ctx.assign_batch(w.part_off, w.partition_id, None if latest else w.begin, w.end,
                                           w.committed, mode, w.cons_off, w.cons_rank)
    np.testing.assert_array_equal(got_p, exp_p, err_msg="partition order " + what)
    np.testing.assert_array_equal(got_m, exp_m, err_msg="member " + what)
    np.testing.assert_array_equal(got_t, exp_t, err_msg="totals " + what)


# ---- the reference's own vectors through the native path -----------------------------------
def test_compute_lag_reference_vectors(ctx):             # Test.java:21-80
    b = [1111, 0, 1111, 1111]
    e = [9999, 0, 9999, 9999]
    c = [5555, 5555, -1, -1]
    assert ctx.compute_lag(b, e, c, N.LA_RESET_EARLIEST).tolist() == [4444, 0, 8888, 8888]
    assert ctx.compute_lag(None, e, c, N.LA_RESET_LATEST).tolist() == [4444, 0, 0, 0]


def test_readme_example(ctx):                            # README.md:42-57
    w = synth.config("cfg1")
    p, m, t = ctx.assign_batch_lags(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    assert p.tolist() == [0, 2, 1] and m.tolist() == [0, 1, 1] and t.tolist() == [100000, 110000]


def test_assign_reference_vector(ctx):                   # Test.java:82-132, flat form
    # topic1: consumers {consumer-1 (rank 0), consumer-2 (rank 1)}; topic2: {consumer-1}
    p, m, t = ctx.assign_batch_lags([0, 4, 6], [0, 1, 2, 3, 0, 1], [100000, 100000, 500, 1, 900000, 100000],
                                    [0, 2, 3], [0, 1, 0])
    assert p.tolist() == [0, 1, 2, 3, 0, 1]
    assert m.tolist() == [0, 1, 0, 1, 0, 0]
    assert t.tolist() == [100500, 100001, 1000000]


def test_zero_and_skewed_vectors(ctx):                   # Test.java:134-228
    p, m, _ = ctx.assign_batch_lags([0, 7], list(range(7)), [0] * 7, [0, 2], [0, 1])
    assert m.tolist() == [0, 1, 0, 1, 0, 1, 0]
    lags = [360, 359, 230, 118, 444, 122, 65, 111, 455000, 424000]
    p, m, t = ctx.assign_batch_lags([0, 10], list(range(10)), lags, [0, 3], [0, 1, 2])
    assert p.tolist() == [8, 9, 4, 0, 1, 2, 5, 3, 7, 6]
    assert m.tolist() == [0, 1, 2, 2, 1, 0, 2, 1, 0, 2]
    assert t.tolist() == [455341, 424477, 991]


# ---- lag kernel ---------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 2, 3, 255, 4097, 1 << 20])
def test_lag_kernel_fuzz(ctx, n):
    rng = np.random.default_rng(n)
    b = rng.integers(-(1 << 63), (1 << 63) - 1, n)
    e = rng.integers(-(1 << 63), (1 << 63) - 1, n)
    c = rng.integers(-(1 << 62), (1 << 63) - 1, n)
    b[: n // 3] = rng.integers(0, 1000, n // 3)
    e[: n // 3] = rng.integers(0, 5000, n // 3)
    c[: n // 3] = rng.integers(-1, 5000, n // 3)
    for mode, latest in ((N.LA_RESET_LATEST, True), (N.LA_RESET_EARLIEST, False)):
        np.testing.assert_array_equal(ctx.compute_lag(b, e, c, mode), oracle.compute_lags(b, e, c, latest))


# ---- wave-tile kernel: every tile class, ragged shapes ----------------------------------------------
@pytest.mark.parametrize("max_p,max_c", [(1, 1), (3, 2), (8, 8), (17, 5), (64, 8), (100, 16), (128, 3),
                                         (256, 32), (300, 33), (512, 64), (777, 20), (1024, 64), (1024, 1)])
def test_tile_ragged_lags(ctx, max_p, max_c):
    w = synth.ragged(1000 + max_p * 7 + max_c, 300, max_p, max_c, negative=True)
    _check_lags(ctx, w, "ragged %dx%d" % (max_p, max_c))


@pytest.mark.parametrize("max_p,max_c", [(64, 8), (256, 32), (1000, 50)])
@pytest.mark.parametrize("mode", [N.LA_RESET_LATEST, N.LA_RESET_EARLIEST])
def test_tile_ragged_offsets(ctx, max_p, max_c, mode):
    w = synth.ragged(77 + max_p + mode, 200, max_p, max_c)
    _check_offsets(ctx, w, mode)


@pytest.mark.parametrize("dist", ["zero", "ties", "u63", "full"])
def test_tile_hard_distributions(ctx, dist):
    w = synth.ragged(5, 400, 256, 32, dist=dist, negative=True)
    _check_lags(ctx, w, dist)


# ---- full tiles: the form without clamps, validity selects and sentinels (chosen per wavefront) -----------------
def _full_tile_batch(seed, t, p, c, kind):
    """T topics of exactly P partitions each (P = lanes x records of some tile shape) -- except every 5th topic in the
    "mixed" kinds, which is one partition short, so wavefronts of both forms and wavefronts holding one topic of each
    sit in the same launch."""
    rng = np.random.default_rng(seed)
    sizes = np.full(t, p, dtype=np.int64)
    if kind.startswith("mixed"):
        sizes[::5] = max(p - 1, 1)
    part_off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(part_off[-1])
    pid = np.empty(n, dtype=np.int32)
    lag = np.empty(n, dtype=np.int64)
    for i in range(t):
        q = int(sizes[i])
        ids = rng.permutation(q).astype(np.int64)
        if kind == "sparse":
            ids = ids * 7 + 5                                    # ids beyond the tile: keys carry the load slot
        elif kind == "dup" and q > 1:
            ids[rng.integers(0, q)] = ids[rng.integers(0, q)]    # a duplicate id: two records in one slot, the check must fail
        pid[part_off[i]:part_off[i + 1]] = ids
        if kind == "ties":
            l = rng.integers(0, 3, q) * 1000
        elif kind == "zero":
            l = np.zeros(q, dtype=np.int64)
        elif kind == "u63":
            l = rng.integers(0, (1 << 63) - 1, q)                # does not pack: the wide kernel's list
        elif kind == "negative":
            l = rng.integers(0, 1 << 30, q)
            l[rng.integers(0, q)] = -5
        elif kind == "tiny":
            l = rng.integers(0, 100, q)                          # the key is the whole record: no fetch, no check
        else:
            l = rng.integers(0, 1 << 40, q)
        lag[part_off[i]:part_off[i + 1]] = l
    cons_off = np.arange(t + 1, dtype=np.int64) * c
    ranks = np.tile(np.arange(c, dtype=np.int32) * 2 + 1, t)
    com = rng.integers(0, 1 << 20, n).astype(np.int64)
    end = com + np.maximum(lag, 0)
    com = np.where(rng.random(n) < 0.03, np.int64(-1), com)
    begin = rng.integers(0, 1 << 10, n).astype(np.int64)
    return synth.Workload("full", t, part_off, pid, begin, end, com, lag, cons_off, ranks, p, c)


@pytest.mark.parametrize("p,c", [(8, 8), (16, 3), (32, 32), (64, 8), (128, 16), (128, 64), (256, 32), (512, 33), (1024, 64)])
@pytest.mark.parametrize("kind", ["u40", "mixed", "sparse", "dup", "ties", "zero", "tiny", "u63", "negative"])
def test_tile_full_tiles(ctx, p, c, kind):
    w = _full_tile_batch(7 * p + c, 37, p, c, kind)               # 37: the last wavefront of a launch is partly empty
    _check_lags(ctx, w, "full tiles %d x %d %s" % (p, c, kind))


@pytest.mark.parametrize("p,c", [(64, 8), (256, 32), (1024, 20)])
@pytest.mark.parametrize("mode", [N.LA_RESET_LATEST, N.LA_RESET_EARLIEST])
def test_tile_full_tiles_offsets(ctx, p, c, mode):
    for kind in ("u40", "mixed"):
        w = _full_tile_batch(p + c + mode, 41, p, c, kind)
        w.lag = None                                              # offsets in: the lag is computed in the kernel
        _check_offsets(ctx, w, mode, "full tiles %d x %d %s" % (p, c, kind))


# ---- packed 64-bit record format (chosen per wavefront) vs the wide format -----------------------------
def _uniform_batch(seed, t, p, c, lag_hi, pid_hi, lag_lo=0):
    """T topics x P partitions x C consumers; lags uniform on [lag_lo, lag_hi], ids distinct in [0, pid_hi]."""
    rng = np.random.default_rng(seed)
    lag = rng.integers(lag_lo, lag_hi, t * p, endpoint=True).astype(np.int64)
    lag[rng.integers(0, t * p, max(1, t * p // 50))] = lag_hi            # make sure the bound is hit
    pid = np.empty(t * p, dtype=np.int64)
    for i in range(t):
        ids = np.append(rng.choice(min(pid_hi, 8 * p + 8), p - 1, replace=False), pid_hi)
        pid[i * p:(i + 1) * p] = rng.permutation(ids)
    zeros = np.zeros(t * p, dtype=np.int64)
    return synth.Workload("uniform", t, np.arange(t + 1, dtype=np.int64) * p, pid.astype(np.int32), zeros, lag.copy(),
                          zeros, lag, np.arange(t + 1, dtype=np.int64) * c,
                          np.tile(np.arange(c, dtype=np.int32) * 3 + 1, t), p, c)


@pytest.mark.parametrize("dist", ["zero", "ties", "small", "u40"])
@pytest.mark.parametrize("max_p,max_c", [(8, 8), (30, 3), (64, 8), (128, 16), (256, 32), (500, 64), (1024, 40)])
def test_tile_packed_records_ragged(ctx, dist, max_p, max_c):
    # narrow lags: (almost) every wavefront takes the packed format; the wide one must agree
    w = synth.ragged(31 * max_p + max_c, 260, max_p, max_c, dist=dist)
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    for algo in (N.LA_ALGO_AUTO, N.LA_ALGO_ROUNDS_WIDE):
        got = _run_device(ctx, w, algo, use_lag=True)
        for g, e, what in zip(got, exp, ("partition order", "member", "totals")):
            np.testing.assert_array_equal(g, e, err_msg="%s algo %d" % (what, algo))


@pytest.mark.parametrize("p,c", [(64, 8), (256, 32), (1024, 64)])
@pytest.mark.parametrize("pid_bits", [0, 12, 20, 31])      # 0: ids are exactly 0..P-1
def test_tile_packed_boundary(ctx, p, c, pid_bits):
    # the packing rule: lag < 2^min(63 - sh, 57 - log2(tile)) with sh = bits of the largest id.
    # One below the limit packs, the limit itself must fall back to wide records; both exact.
    cap_bits = int(np.ceil(np.log2(p)))
    pid_bits = pid_bits or cap_bits
    lb = min(63 - pid_bits, 57 - cap_bits)
    for lag_hi in ((1 << lb) - 1, 1 << lb, (1 << lb) + 12345):
        w = _uniform_batch(pid_bits * 1000 + p, 40, p, c, lag_hi, (1 << pid_bits) - 1, lag_lo=lag_hi // 2)
        exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
        for algo in (N.LA_ALGO_AUTO, N.LA_ALGO_ROUNDS_WIDE):
            got = _run_device(ctx, w, algo, use_lag=True)
            for g, e, what in zip(got, exp, ("partition order", "member", "totals")):
                np.testing.assert_array_equal(g, e, err_msg="%s lag_hi=%d algo %d" % (what, lag_hi, algo))


def test_tile_packed_negative_id_or_lag_falls_back(ctx):
    w = _uniform_batch(77, 64, 256, 32, 1 << 30, 255)
    w.partition_id[5] = -3                      # one negative id in the first wavefront
    w.lag[256 * 9 + 17] = -1                    # one negative lag in another
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    got = _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=True)
    for g, e in zip(got, exp):
        np.testing.assert_array_equal(g, e)


@pytest.mark.parametrize("p,c", [(64, 8), (256, 32), (1000, 48)])
@pytest.mark.parametrize("ids", ["dense", "sparse", "offset"])
def test_tile_many_equal_lags_among_large_ones(ctx, p, c, ids):
    # The fast sort orders 32-bit keys and then checks the full records.  Equal lags (here: most
    # partitions at 0, the auto.offset.reset=latest situation) next to lags that need > 23 bits are
    # exactly what a truncated key cannot order by itself: the id tie-break and the fallback must.
    rng = np.random.default_rng(p * 31 + len(ids))
    t = 120
    lag = np.where(rng.random(t * p) < 0.6, 0, rng.integers(1, 1 << 40, t * p)).astype(np.int64)
    lag[rng.integers(0, t * p, t * p // 10)] = 123456789012          # equal non-zero lags too
    pid = np.concatenate([rng.permutation(p) for _ in range(t)]).astype(np.int64)
    if ids == "sparse":
        pid = pid * 1009 + 7
    elif ids == "offset":
        pid = pid + 5000
    zeros = np.zeros(t * p, dtype=np.int64)
    w = synth.Workload("ties", t, np.arange(t + 1, dtype=np.int64) * p, pid.astype(np.int32), zeros, lag.copy(), zeros,
                       lag, np.arange(t + 1, dtype=np.int64) * c, np.tile(np.arange(c, dtype=np.int32), t), p, c)
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    for algo in (N.LA_ALGO_AUTO, N.LA_ALGO_ROUNDS_WIDE):
        got = _run_device(ctx, w, algo, use_lag=True)
        for g, e, what in zip(got, exp, ("partition order", "member", "totals")):
            np.testing.assert_array_equal(g, e, err_msg="%s algo %d" % (what, algo))


def test_tile_duplicate_partition_ids(ctx):
    # not something Kafka produces, but the reference would simply sort and assign the records it is
    # given; two records with one id collide in the id-indexed slice and must be caught by the check
    rng = np.random.default_rng(99)
    t, p, c = 64, 256, 32
    pid = np.concatenate([rng.permutation(p) for _ in range(t)]).astype(np.int32)
    pid[rng.integers(0, t * p, 200)] = 7
    lag = rng.integers(0, 1 << 33, t * p).astype(np.int64)
    zeros = np.zeros(t * p, dtype=np.int64)
    w = synth.Workload("dups", t, np.arange(t + 1, dtype=np.int64) * p, pid, zeros, lag.copy(), zeros, lag,
                       np.arange(t + 1, dtype=np.int64) * c, np.tile(np.arange(c, dtype=np.int32), t), p, c)
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    got = _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=True)
    for g, e in zip(got, exp):
        np.testing.assert_array_equal(g, e)


def test_last_element_of_batch_is_a_lone_pair_tail(ctx):
    # loads are 2-element and clamped to the batch: the very last partition of the batch can sit in the
    # second half of a clamped pair; odd sizes at the end of the arrays exercise that
    for p_last in (1, 2, 3, 9, 17, 255):
        ps = [256, 31, p_last]
        cs = [32, 5, 4]
        rng = np.random.default_rng(p_last)
        part_off = np.concatenate([[0], np.cumsum(ps)]).astype(np.int64)
        cons_off = np.concatenate([[0], np.cumsum(cs)]).astype(np.int64)
        n = int(part_off[-1])
        pid = np.concatenate([rng.permutation(q) for q in ps]).astype(np.int32)
        com = rng.integers(-1, 1000, n).astype(np.int64)
        end = com.clip(0) + rng.integers(0, 1 << 30, n)
        begin = rng.integers(0, 50, n).astype(np.int64)
        ranks = np.concatenate([np.arange(q) for q in cs]).astype(np.int32)
        w = synth.Workload("tail", 3, part_off, pid, begin, end.astype(np.int64), com, np.zeros(n, dtype=np.int64),
                           cons_off, ranks, max(ps), max(cs))
        for latest in (True, False):
            lag = oracle.compute_lags(w.begin, w.end, w.committed, latest)
            exp = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
            got = _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=False, latest=latest)
            for g, e in zip(got, exp):
                np.testing.assert_array_equal(g, e)


def test_target_shape_packed_equals_wide_equals_oracle(ctx):
    w = synth.config("target", 0.05)             # 5 000 topics x 256 x 32
    for latest in (True, False):
        lag = oracle.compute_lags(w.begin, w.end, w.committed, latest)
        exp = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
        for algo in (N.LA_ALGO_AUTO, N.LA_ALGO_ROUNDS_WIDE):
            got = _run_device(ctx, w, algo, use_lag=False, latest=latest)
            for g, e in zip(got, exp):
                np.testing.assert_array_equal(g, e)


def test_named_configs_scaled(ctx):
    for name, scale in (("cfg3", 1.0), ("cfg4", 0.02), ("target", 0.01)):
        w = synth.config(name, scale)
        _check_offsets(ctx, w, N.LA_RESET_LATEST, name)
        _check_offsets(ctx, w, N.LA_RESET_EARLIEST, name)


def test_unsorted_consumers_rejected(ctx):
    with pytest.raises(N.LagAssignError) as ei:
        ctx.assign_batch_lags([0, 2], [0, 1], [5, 6], [0, 2], [3, 1])
    assert ei.value.code == N.LA_EINVAL


# ---- large path: device radix sort + one-workgroup greedy ----------------------------------------------
def _single_topic(seed, p, c, kind, shuffled=True, negative=False):
    rng = np.random.default_rng(seed)
    pid = rng.permutation(p).astype(np.int32) if shuffled else np.arange(p, dtype=np.int32)
    if kind == "u63":
        lag = rng.integers(0, (1 << 63) - 1, p)
    elif kind == "u40":
        lag = rng.integers(0, 1 << 40, p)
    elif kind == "ties":
        lag = rng.integers(0, 5, p) * 1000
    elif kind == "zero":
        lag = np.zeros(p, dtype=np.int64)
    else:
        lag = rng.integers(-(1 << 63), (1 << 63) - 1, p)
    if not negative:
        lag = np.where(lag < 0, ~lag, lag)
    ranks = np.sort(rng.choice(3 * c + 1, c, replace=False)).astype(np.int32)
    return [0, p], pid, lag.astype(np.int64), [0, c], ranks


@pytest.mark.parametrize("p,c,kind", [
    (1025, 3, "u40"), (2000, 65, "u40"), (4096, 64, "ties"), (5000, 100, "zero"), (10000, 128, "u63"),
    (10000, 128, "u40"), (12345, 1000, "full"), (50, 100, "u40"), (70000, 2048, "u40"), (8193, 8192, "ties"),
])
def test_large_single_topic(ctx, p, c, kind):
    po, pid, lag, co, ranks = _single_topic(p + c, p, c, kind, negative=(kind == "full"))
    exp = oracle.assign_flat(po, pid, lag, co, ranks)
    got = ctx.assign_batch_lags(po, pid, lag, co, ranks)
    for g, e, what in zip(got, exp, ("partition order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg=what)


def test_large_ids_already_sorted_and_sparse(ctx):
    po, pid, lag, co, ranks = _single_topic(9, 3000, 70, "ties", shuffled=False)
    for ids in (pid, (pid * 1000 - 500000).astype(np.int32)):       # dense sorted, sparse with negatives
        exp = oracle.assign_flat(po, ids, lag, co, ranks)
        got = ctx.assign_batch_lags(po, ids, lag, co, ranks)
        for g, e in zip(got, exp):
            np.testing.assert_array_equal(g, e)


def test_mixed_batch_small_and_large_topics(ctx):
    rng = np.random.default_rng(3)
    ps = [10, 2000, 0, 300, 1500, 64, 5000, 7]
    cs = [2, 5, 3, 64, 100, 0, 200, 8]
    part_off = np.concatenate([[0], np.cumsum(ps)])
    cons_off = np.concatenate([[0], np.cumsum(cs)])
    pid = np.concatenate([rng.permutation(p) for p in ps]).astype(np.int32)
    lag = rng.integers(0, 1 << 30, part_off[-1]).astype(np.int64)
    ranks = np.concatenate([np.sort(rng.choice(1000, c, replace=False)) for c in cs]).astype(np.int32)
    exp = oracle.assign_flat(part_off, pid, lag, cons_off, ranks)
    got = ctx.assign_batch_lags(part_off, pid, lag, cons_off, ranks)
    for g, e in zip(got, exp):
        np.testing.assert_array_equal(g, e)


def test_cfg2_full_size_offsets(ctx):
    for name in ("cfg2a", "cfg2b"):
        w = synth.config(name)
        if name == "cfg2a":
            _check_lags(ctx, w, name)          # totals overflow by design: lag entry point
        else:
            _check_offsets(ctx, w, N.LA_RESET_LATEST, name)
            _check_offsets(ctx, w, N.LA_RESET_EARLIEST, name)


def test_cfg5_scaled(ctx):
    w = synth.config("cfg5", 1.0 / 16)            # 65 536 partitions x 8 192 consumers
    _check_offsets(ctx, w, N.LA_RESET_EARLIEST, "cfg5/16")


# ---- BASELINE.json's configurations at FULL size ---------------------------------------------------------
def test_cfg4_full_size(ctx):
    w = synth.config("cfg4")                       # 100 000 topics x 64 partitions x 8 consumers
    _check_offsets(ctx, w, N.LA_RESET_LATEST, "cfg4 latest")
    _check_offsets(ctx, w, N.LA_RESET_EARLIEST, "cfg4 earliest")


def test_target_full_size(ctx):
    w = synth.config("target")                     # 100 000 topics x 256 partitions x 32 consumers, 25.6 M partitions
    lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
    exp_p, exp_m, exp_t = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
    got_p, got_m, got_t = ctx.assign_batch(w.part_off, w.partition_id, w.begin, w.end, w.committed,
                                           N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
    assert np.array_equal(got_p, exp_p) and np.array_equal(got_m, exp_m) and np.array_equal(got_t, exp_t)
    # the quality metric of BASELINE.json follows from the totals, so it is the oracle's exactly
    np.testing.assert_array_equal(synth.lag_ratio(got_t, w.cons_off), synth.lag_ratio(exp_t, w.cons_off))
    # size-independent properties: a permutation per topic, sorted by (lag desc, id asc), every round a
    # permutation of the consumers, totals = sums of what each consumer received
    t, p, c = w.n_topics, 256, 32
    assert np.array_equal(np.sort(got_p.reshape(t, p), axis=1), np.tile(np.arange(p, dtype=np.int32), (t, 1)))
    lag_by_id = np.empty_like(lag).reshape(t, p)
    np.put_along_axis(lag_by_id, w.partition_id.reshape(t, p).astype(np.int64), lag.reshape(t, p), axis=1)
    sl = np.take_along_axis(lag_by_id, got_p.reshape(t, p).astype(np.int64), axis=1)
    assert (sl[:, 1:] <= sl[:, :-1]).all()
    tie = sl[:, 1:] == sl[:, :-1]
    gp = got_p.reshape(t, p)
    assert (gp[:, 1:][tie] > gp[:, :-1][tie]).all()
    assert np.array_equal(np.sort(got_m.reshape(t, p // c, c), axis=2),
                          np.broadcast_to(np.arange(c, dtype=np.int32), (t, p // c, c)))
    sums = np.zeros((t, c), dtype=np.int64)
    np.add.at(sums, (np.repeat(np.arange(t), p), got_m.astype(np.int64)), sl.reshape(-1))
    assert np.array_equal(sums.reshape(-1), got_t)


def test_cfg5_full_size(ctx):
    w = synth.config("cfg5")                       # 1 topic, 1 048 576 partitions, 8 192 consumers, 128 rounds
    lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
    got = ctx.assign_batch(w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST,
                           w.cons_off, w.cons_rank)
    for g, e, what in zip(got, _round_form(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank),
                          ("partition order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg="round form: " + what)
    # the literal per-step Collections.min of the oracle: 8.6e9 comparator steps, about 15 s of one host core
    for g, e, what in zip(got, oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank),
                          ("partition order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg="oracle: " + what)


def test_target_full_size_other_modes(ctx):
    """The 25.6 M-partition target in the forms test_target_full_size leaves out: auto.offset.reset=latest
    (Main.java:391-392), 64-bit element indexing, and wide (96-bit) records forced -- each bit-exact against the
    literal oracle on every partition."""
    w = synth.config("target")
    exp = {}
    for latest in (True, False):
        lag = oracle.compute_lags(w.begin, w.end, w.committed, latest)
        exp[latest] = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
    assert not np.array_equal(exp[True][2], exp[False][2])          # the reset mode matters on this workload
    for latest, algo, flags, what in ((True, N.LA_ALGO_AUTO, 0, "latest"),
                                      (False, N.LA_ALGO_AUTO, N.LA_FLAG_INDEX64, "earliest, INDEX64"),
                                      (True, N.LA_ALGO_AUTO, N.LA_FLAG_INDEX64, "latest, INDEX64"),
                                      (False, N.LA_ALGO_ROUNDS_WIDE, 0, "earliest, wide records"),
                                      (True, N.LA_ALGO_ROUNDS_WIDE, 0, "latest, wide records")):
        got = _run_device(ctx, w, algo, use_lag=False, latest=latest, flags=flags)
        for g, e, name in zip(got, exp[latest], ("partition order", "member", "totals")):
            np.testing.assert_array_equal(g, e, err_msg="target %s: %s" % (what, name))
    # the host-buffer entry point in latest mode (begin_off = NULL is legal there)
    got = ctx.assign_batch(w.part_off, w.partition_id, None, w.end, w.committed, N.LA_RESET_LATEST, w.cons_off, w.cons_rank)
    for g, e in zip(got, exp[True]):
        np.testing.assert_array_equal(g, e)


def test_cfg5_full_size_latest(ctx):
    """cfg5 (1 048 576 partitions x 8 192 consumers) with auto.offset.reset=latest: the 1 % of partitions without a
    committed offset have lag 0 there (Main.java:391-392), which moves them to the tail of the sorted order."""
    w = synth.config("cfg5")
    lag = oracle.compute_lags(w.begin, w.end, w.committed, True)
    assert not np.array_equal(lag, oracle.compute_lags(w.begin, w.end, w.committed, False))
    got = ctx.assign_batch(w.part_off, w.partition_id, None, w.end, w.committed, N.LA_RESET_LATEST, w.cons_off, w.cons_rank)
    for g, e, what in zip(got, _round_form(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank),
                          ("partition order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg="round form: " + what)
    for g, e, what in zip(got, oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank),
                          ("partition order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg="oracle: " + what)


def test_cfg5_full_size_other_forms(ctx):
    """cfg5 at full size through the device entry point in the forms the default call does not take: greedy rounds that
    never merge their runs, four-kernel radix passes, both, and the full bitonic network in every round -- each equal to the
    round form of the oracle (which test_cfg5_full_size checks against the literal per-step min)."""
    w = synth.config("cfg5")
    lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
    exp = _round_form(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
    for flags, what in ((0, "default"), (N.LA_FLAG_NO_RUN_MERGE, "runs sorted, not merged"),
                        (N.LA_FLAG_SORT_MULTIKERNEL, "four-kernel radix passes"),
                        (N.LA_FLAG_NO_RUN_MERGE | N.LA_FLAG_SORT_MULTIKERNEL | N.LA_FLAG_SAMPLE_TIGHT, "all three hooks"),
                        (N.LA_FLAG_NO_SAMPLE_SORT | N.LA_FLAG_NO_RUN_MERGE, "full network every round")):
        got = _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=False, latest=False, flags=flags)
        for g, e, name in zip(got, exp, ("partition order", "member", "totals")):
            np.testing.assert_array_equal(g, e, err_msg="cfg5, %s: %s" % (what, name))


# ---- device-resident entry point; round form == literal wavefront argmin ------------------------------
def _run_device(ctx, w, algo, use_lag=False, latest=True, flags=0, host_offsets=True):
    import ctypes
    import torch
    dev = torch.device("cuda", 0)
    d = {k: torch.from_numpy(np.ascontiguousarray(getattr(w, k))).to(dev) for k in
         ("part_off", "partition_id", "begin", "end", "committed", "lag", "cons_off", "cons_rank")}
    out_pid = torch.full((w.n_partitions,), -7, device=dev, dtype=torch.int32)
    out_rank = torch.full((w.n_partitions,), -7, device=dev, dtype=torch.int32)
    out_total = torch.zeros(w.cons_rank.size, device=dev, dtype=torch.int64)
    b = N.DeviceBatch()
    b.n_topics = w.n_topics
    b.reset_mode = N.LA_RESET_LATEST if latest else N.LA_RESET_EARLIEST
    b.algo = algo
    b.flags = flags
    b.n_partitions = w.n_partitions
    b.n_consumers = w.cons_rank.size
    b.max_partitions_per_topic = w.max_partitions
    b.max_consumers_per_topic = w.max_consumers
    b.d_part_off = d["part_off"].data_ptr()
    b.d_partition_id = d["partition_id"].data_ptr()
    b.d_begin_off = d["begin"].data_ptr()
    b.d_end_off = d["end"].data_ptr()
    b.d_committed_off = d["committed"].data_ptr()
    b.d_lag = d["lag"].data_ptr() if use_lag else None
    b.d_cons_off = d["cons_off"].data_ptr()
    b.d_cons_rank = d["cons_rank"].data_ptr()
    b.d_out_partition = out_pid.data_ptr()
    b.d_out_member_rank = out_rank.data_ptr()
    b.d_out_total_lag = out_total.data_ptr()
    po = np.ascontiguousarray(w.part_off, dtype=np.int64)
    co = np.ascontiguousarray(w.cons_off, dtype=np.int64)
    if host_offsets:
        b.h_part_off = po.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
        b.h_cons_off = co.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))
    stream = torch.cuda.current_stream().cuda_stream
    ctx.assign_batch_device(b, stream)
    ctx.sync(stream)
    return out_pid.cpu().numpy(), out_rank.cpu().numpy(), out_total.cpu().numpy()


def _deferring_workload():
    """100 000 tiny topics (beyond one round of resident workgroups), a sprinkling of lags the packed format
    cannot hold: the packed kernel defers those tiles to the wide kernel's list."""
    rng = np.random.default_rng(21)
    t = 100000
    ps = rng.integers(0, 9, t)
    cs = rng.integers(0, 9, t)
    part_off = np.concatenate([[0], np.cumsum(ps)]).astype(np.int64)
    cons_off = np.concatenate([[0], np.cumsum(cs)]).astype(np.int64)
    n = int(part_off[-1])
    pid = np.concatenate([rng.permutation(int(p)) for p in ps]).astype(np.int32)
    lag = rng.integers(0, 1 << 30, n).astype(np.int64)
    wide = rng.integers(0, n, 400)
    lag[wide[:200]] = -rng.integers(1, 1 << 40, 200)                 # negative lags: cannot be packed
    lag[wide[200:]] = rng.integers(1 << 60, (1 << 63) - 1, 200)      # totals would overflow the packed bins
    ranks = np.concatenate([np.sort(rng.choice(64, int(c), replace=False)) for c in cs]).astype(np.int32)
    zeros = np.zeros(n, dtype=np.int64)
    return synth.Workload("defer", t, part_off, pid, zeros, lag.copy(), zeros, lag, cons_off, ranks, 8, 8)


@pytest.mark.parametrize("which", ["one launch", "packed + wide kernels with deferred tiles"])
def test_device_entry_is_graph_capturable(ctx, which):
    # after one warm-up call (scratch allocated) the plain tile path is launches only: it can be captured in a
    # HIP graph and replayed -- also when tiles are deferred (a captured launch clears its counter with a memset node)
    import ctypes
    import torch
    dev = torch.device("cuda", 0)
    w = synth.config("cfg3", 0.3) if which == "one launch" else _deferring_workload()
    d = {k: torch.from_numpy(np.ascontiguousarray(getattr(w, k))).to(dev) for k in
         ("part_off", "partition_id", "lag", "cons_off", "cons_rank")}
    out_pid = torch.zeros(w.n_partitions, device=dev, dtype=torch.int32)
    out_rank = torch.zeros(w.n_partitions, device=dev, dtype=torch.int32)
    out_total = torch.zeros(w.cons_rank.size, device=dev, dtype=torch.int64)
    b = N.DeviceBatch()
    b.n_topics, b.reset_mode, b.algo, b.flags = w.n_topics, N.LA_RESET_LATEST, N.LA_ALGO_AUTO, 0
    b.n_partitions, b.n_consumers = w.n_partitions, w.cons_rank.size
    b.max_partitions_per_topic, b.max_consumers_per_topic = w.max_partitions, w.max_consumers
    b.d_part_off, b.d_partition_id, b.d_lag = d["part_off"].data_ptr(), d["partition_id"].data_ptr(), d["lag"].data_ptr()
    b.d_cons_off, b.d_cons_rank = d["cons_off"].data_ptr(), d["cons_rank"].data_ptr()
    b.d_out_partition, b.d_out_member_rank, b.d_out_total_lag = out_pid.data_ptr(), out_rank.data_ptr(), out_total.data_ptr()
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ctx.assign_batch_device(b, s.cuda_stream)
        ctx.sync(s.cuda_stream)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        ctx.assign_batch_device(b, s.cuda_stream)
    for rep in range(3):
        out_pid.zero_(); out_rank.zero_(); out_total.zero_()
        g.replay()
        torch.cuda.synchronize()
        for got, e in zip((out_pid, out_rank, out_total), exp):
            np.testing.assert_array_equal(got.cpu().numpy(), e)


@pytest.mark.parametrize("max_p,max_c", [(64, 8), (256, 32), (700, 64)])
def test_device_entry_rounds_equals_argmin_equals_oracle(ctx, max_p, max_c):
    w = synth.ragged(max_p + 11, 150, max_p, max_c, negative=True)
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    for algo in (N.LA_ALGO_ROUNDS, N.LA_ALGO_ARGMIN):
        got = _run_device(ctx, w, algo, use_lag=True)
        for g, e in zip(got, exp):
            np.testing.assert_array_equal(g, e)


def test_device_entry_large_argmin(ctx):
    po, pid, lag, co, ranks = _single_topic(4, 3000, 300, "u40")
    w = synth.Workload("one", 1, np.array(po), pid, np.zeros_like(lag), lag, np.zeros_like(lag), lag,
                       np.array(co), ranks, 3000, 300)
    exp = oracle.assign_flat(po, pid, lag, co, ranks)
    for algo in (N.LA_ALGO_ROUNDS, N.LA_ALGO_ARGMIN):
        got = _run_device(ctx, w, algo, use_lag=True)
        for g, e in zip(got, exp):
            np.testing.assert_array_equal(g, e)


def test_device_entry_fetches_offsets_when_no_host_copy_is_given(ctx):
    # shapes beyond a wave tile need the topics' sizes on the host; without h_part_off / h_cons_off the library
    # copies the offsets back itself
    w = synth.ragged(77, 60, 3000, 200, negative=True)
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    got = _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=True, host_offsets=False)
    for g, e, what in zip(got, exp, ("partition order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg=what)


def test_device_entry_shape_hint_violation_is_reported(ctx):
    w = synth.ragged(2, 50, 100, 16)
    w.max_partitions = 8           # lie about the shape
    with pytest.raises(N.LagAssignError) as ei:
        _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=True)
    assert ei.value.code == N.LA_ESHAPE


# ---- assignment -> per-member lists (la_group_by_member) -------------------------------------------------
def _expected_groups(part_off, out_p, out_m, n_members):
    order = np.argsort(out_m, kind="stable")                 # rank -1 first, then members in rank order
    topic_of = np.searchsorted(part_off, np.arange(out_p.size), side="right") - 1
    counts = np.bincount(out_m + 1, minlength=n_members + 1)
    return np.cumsum(counts)[: n_members + 1].astype(np.int64), topic_of[order].astype(np.int32), out_p[order]


@pytest.mark.parametrize("shape", [(1, 3, 2), (50, 64, 8), (300, 256, 32), (3, 5000, 300)])
def test_group_by_member_matches_stable_sort(ctx, shape):
    t, max_p, max_c = shape
    w = synth.ragged(7 * t + max_p, t, max_p, max_c, dist="small")
    out_p, out_m, _ = ctx.assign_batch_lags(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    n_members = int(w.cons_rank.max()) + 1 if w.cons_rank.size else 0
    off, g_t, g_p = ctx.group_by_member(w.part_off, out_p, out_m, n_members)
    e_off, e_t, e_p = _expected_groups(np.asarray(w.part_off), out_p, out_m, n_members)
    np.testing.assert_array_equal(off, e_off)
    np.testing.assert_array_equal(g_t, e_t)
    np.testing.assert_array_equal(g_p, e_p)


def test_group_by_member_reference_vector(ctx):              # Test.java:82-132: the exact lists
    out_p, out_m, _ = ctx.assign_batch_lags([0, 4, 6], [0, 1, 2, 3, 0, 1], [100000, 100000, 500, 1, 900000, 100000],
                                            [0, 2, 3], [0, 1, 0])
    off, g_t, g_p = ctx.group_by_member([0, 4, 6], out_p, out_m, 2)
    lists = [[(int(g_t[j]), int(g_p[j])) for j in range(off[r], off[r + 1])] for r in range(2)]
    assert lists[0] == [(0, 0), (0, 2), (1, 0), (1, 1)]          # consumer-1: topic1-0, topic1-2, topic2-0, topic2-1
    assert lists[1] == [(0, 1), (0, 3)]                          # consumer-2: topic1-1, topic1-3


def test_group_last_by_member_equals_group_by_member(ctx):
    # results kept on the device by the assign call == the same grouping of downloaded results
    for w in (synth.ragged(31, 400, 300, 40, negative=True), synth.config("block_b", 0.1), synth.config("cfg3", 0.2)):
        n_members = int(w.cons_rank.max()) + 1 if w.cons_rank.size else 0
        exp_p, exp_m, exp_t = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
        want = ctx.group_by_member(w.part_off, exp_p, exp_m, n_members)
        p, m, t = ctx.assign_batch_lags(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank, keep_on_device=True)
        assert p is None and m is None
        np.testing.assert_array_equal(t, exp_t)
        got = ctx.group_last_by_member(w.n_partitions, n_members)
        for g, e, what in zip(got, want, ("member_off", "grouped_topic", "grouped_partition")):
            np.testing.assert_array_equal(g, e, err_msg=what)
        got2 = ctx.group_last_by_member(w.n_partitions, n_members)           # the results are still there
        np.testing.assert_array_equal(got2[2], want[2])
    # a host-buffer grouping call reuses the scratch: the kept results are gone afterwards
    ctx.group_by_member(w.part_off, exp_p, exp_m, n_members)
    with pytest.raises(N.LagAssignError) as ei:
        ctx.group_last_by_member(w.n_partitions, n_members)
    assert ei.value.code == N.LA_EINVAL
    # one output array without the other is refused
    import ctypes
    one = np.zeros(3, dtype=np.int32)
    keep = (np.array([0, 3], dtype=np.int64), np.arange(3, dtype=np.int32), np.array([5, 6, 7], dtype=np.int64),
            np.array([0, 1], dtype=np.int64), np.zeros(1, dtype=np.int32))        # (N._p64 is an address: the arrays must live)
    rc = ctx._lib.la_assign_batch_lags(ctx._h, 1, N._p64(keep[0]), N._p32(keep[1]), N._p64(keep[2]), N._p64(keep[3]),
                                       N._p32(keep[4]), N._p32(one), None, None)
    assert rc == N.LA_EINVAL


@pytest.mark.parametrize("n,m", [(1, 1), (3, 2), (63, 2), (64, 64), (65, 1), (100, 3), (777, 40), (1024, 1), (1023, 900), (1024, 4094),
                                 (1024, 64), (1025, 5), (1000, 4095), (2000, 5), (2560, 8190), (2560, 3), (2561, 7), (2500, 8191), (8192, 3),
                                 (8192, 8190), (16384, 1), (12345, 200)])
def test_group_by_member_small_form(ctx, n, m):
    """Up to 2 560 entries and 8 190 members ONE workgroup groups the entries by a stable counting sort that is linear in n
    (round 4: ranks among a chunk's peers by ballots, the chunks take their places by a wavefront-ordered hand-over of the
    group cursors; round 3's form walked the entries before each entry and stopped at 1 024); past either limit the radix passes
    do.  Same answer on both sides of the limits, entries of topics without consumers (rank -1) in front, empty members, empty
    topics, one crowded member (every lane of a chunk the same group), every entry its own group."""
    rng = np.random.default_rng(n * 31 + m)
    out_m = rng.integers(-1, m, n).astype(np.int32)
    if n > 10:
        out_m[rng.integers(0, n, n // 7)] = m - 1              # a crowded last member
        out_m[out_m == m // 2] = -1                             # an empty member in the middle (when m > 1)
    out_p = rng.integers(0, 1 << 20, n).astype(np.int32)
    cuts = np.sort(rng.integers(0, n + 1, 6))
    part_off = np.concatenate([[0], cuts, [n]]).astype(np.int64)   # 7 topics, some of them empty
    off, g_t, g_p = ctx.group_by_member(part_off, out_p, out_m, m)
    e_off, e_t, e_p = _expected_groups(part_off, out_p, out_m, m)
    np.testing.assert_array_equal(off, e_off)
    np.testing.assert_array_equal(g_t, e_t)
    np.testing.assert_array_equal(g_p, e_p)


@pytest.mark.parametrize("shape", [(1, 3, 2), (10, 10, 3), (60, 64, 8), (300, 256, 32), (3, 5000, 300), (4000, 200, 20)])
def test_assign_batch_grouped_equals_the_two_calls(ctx, shape):
    """la_assign_batch_grouped = la_assign_batch with the result left on the device + la_group_last_by_member: in one
    download for a call that fits the small path's staging buffer, one step after the other past it (the last shape)."""
    t, max_p, max_c = shape
    w = synth.ragged(11 * t + max_p, t, max_p, max_c, dist="small")
    n_members = int(w.cons_rank.max()) + 1 if w.cons_rank.size else 0
    for mode in (N.LA_RESET_LATEST, N.LA_RESET_EARLIEST):
        a = (w.part_off, w.partition_id, None if mode == N.LA_RESET_LATEST else w.begin, w.end, w.committed, mode,
             w.cons_off, w.cons_rank)
        _, _, tot = ctx.assign_batch(*a, keep_on_device=True)
        want = ctx.group_last_by_member(w.n_partitions, n_members)
        off, g_t, g_p, tot1 = ctx.assign_batch_grouped(*a, n_members)
        np.testing.assert_array_equal(off, want[0])
        np.testing.assert_array_equal(g_t, want[1])
        np.testing.assert_array_equal(g_p, want[2])
        np.testing.assert_array_equal(tot1, tot)
        again = ctx.group_last_by_member(w.n_partitions, n_members)          # the results are still held
        np.testing.assert_array_equal(again[2], want[2])
        off2, none_t, g_p2, none_tot = ctx.assign_batch_grouped(*a, n_members, want_totals=False, want_topic=False)
        assert none_t is None and none_tot is None
        np.testing.assert_array_equal(off2, want[0])
        np.testing.assert_array_equal(g_p2, want[2])
    # the lists against the oracle's assignment, grouped by a stable sort
    lag = oracle.compute_lags(w.begin, w.end, w.committed, False)
    e_p, e_m, _ = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
    e_off, e_t, e_gp = _expected_groups(np.asarray(w.part_off), e_p, e_m, n_members)
    np.testing.assert_array_equal(off, e_off)
    np.testing.assert_array_equal(g_t, e_t)
    np.testing.assert_array_equal(g_p, e_gp)


def test_assign_batch_grouped_argument_errors(ctx):
    w = synth.ragged(5, 4, 10, 3)
    a = (w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
    with pytest.raises(N.LagAssignError) as ei:
        ctx.assign_batch_grouped(*a, -1)
    assert ei.value.code == N.LA_EINVAL
    bad = w.cons_rank.copy()
    k0 = next(int(w.cons_off[t]) for t in range(w.n_topics) if w.cons_off[t + 1] - w.cons_off[t] >= 2)
    bad[k0], bad[k0 + 1] = bad[k0 + 1], bad[k0]                              # one topic's ranks out of order
    with pytest.raises(N.LagAssignError) as ei:
        ctx.assign_batch_grouped(w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST,
                                 w.cons_off, bad, int(bad.max()) + 1)
    assert ei.value.code == N.LA_EINVAL


def test_group_by_member_many_members(ctx):
    rng = np.random.default_rng(5)
    n, m = 200000, 70000
    out_m = rng.integers(-1, m, n).astype(np.int32)
    out_p = rng.integers(0, 1 << 20, n).astype(np.int32)
    part_off = np.array([0, 1000, 1000, n], dtype=np.int64)
    off, g_t, g_p = ctx.group_by_member(part_off, out_p, out_m, m)
    e_off, e_t, e_p = _expected_groups(part_off, out_p, out_m, m)
    np.testing.assert_array_equal(off, e_off)
    np.testing.assert_array_equal(g_t, e_t)
    np.testing.assert_array_equal(g_p, e_p)


def test_device_entry_batch_without_partitions(ctx):
    """Topics with consumers and no partition metadata at all (Main.java:182 hands assignTopic an empty list): nothing
    to write, so the zero-length result arrays -- a null data pointer in torch -- are not an argument error; every
    consumer still reports a total of 0."""
    w = synth.ragged(3, 5, 0, 4)
    assert w.n_partitions == 0 and w.cons_rank.size > 0
    p, m, t = _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=True)
    assert p.size == 0 and m.size == 0 and t.tolist() == [0] * w.cons_rank.size


# ---- randomized sweep: shapes x distributions x entry points ----------------------------------------------
@pytest.mark.parametrize("seed", range(24))
def test_fuzz_tile_batches(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    max_p = int(rng.choice([1, 2, 7, 8, 9, 31, 64, 65, 100, 128, 255, 256, 257, 511, 512, 1000, 1024]))
    max_c = int(rng.choice([1, 2, 3, 7, 8, 9, 16, 17, 31, 32, 33, 63, 64]))
    t = int(rng.integers(1, 400))
    dist = str(rng.choice(["mixed", "zero", "ties", "small", "u40", "u63", "full"]))
    w = synth.ragged(seed * 7919 + 13, t, max_p, max_c, dist=dist, negative=bool(rng.integers(0, 2)))
    if rng.integers(0, 2):
        _check_lags(ctx, w, "fuzz %d" % seed)
    mode = N.LA_RESET_LATEST if rng.integers(0, 2) else N.LA_RESET_EARLIEST
    _check_offsets(ctx, w, mode, "fuzz %d" % seed)
    lag = oracle.compute_lags(w.begin, w.end, w.committed, mode == N.LA_RESET_LATEST)
    exp = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
    got = _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=False, latest=(mode == N.LA_RESET_LATEST))
    for g, e in zip(got, exp):
        np.testing.assert_array_equal(g, e)


@pytest.mark.parametrize("seed", range(12))
def test_fuzz_large_topics(ctx, seed):
    rng = np.random.default_rng(5000 + seed)
    p = int(rng.choice([1025, 1500, 4096, 4097, 9000, 20000, 66000]))
    c = int(rng.choice([1, 2, 63, 64, 65, 100, 127, 128, 129, 500, 1024, 1025, 3000]))
    kind = str(rng.choice(["u40", "ties", "zero", "u63", "full"]))
    po, pid, lag, co, ranks = _single_topic(seed + 77, p, c, kind, shuffled=bool(rng.integers(0, 2)),
                                            negative=(kind == "full"))
    exp = oracle.assign_flat(po, pid, lag, co, ranks)
    got = ctx.assign_batch_lags(po, pid, lag, co, ranks)
    for g, e, what in zip(got, exp, ("partition order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg="%s p=%d c=%d %s" % (what, p, c, kind))


# ---- large path, > 1 024 consumers: every greedy round is a sample sort of the bins (la_large.hip) --------------
def _pareto_topic(seed, p, c):
    rng = np.random.default_rng(seed)
    u = 1.0 - rng.random(p)
    lag = np.floor(np.minimum(float(1 << 40), 1000.0 * u ** (-1.0 / 1.5))).astype(np.int64)
    w = synth.Workload("pareto", 1, np.array([0, p], np.int64), rng.permutation(p).astype(np.int32),
                       np.zeros(p, np.int64), lag.copy(), np.zeros(p, np.int64), lag,
                       np.array([0, c], np.int64), np.arange(c, dtype=np.int32), p, c)
    return w


@pytest.mark.parametrize("p,c,kind", [
    (20000, 1500, "u40"), (30000, 2049, "u40"), (40000, 3000, "pareto"), (70000, 4096, "ties"),
    (100000, 8192, "zero"), (50000, 5000, "u40"), (262144, 8192, "pareto"), (9000, 8192, "u63"),
    (60000, 4097, "pareto"), (33000, 1025, "ties"),
])
def test_large_sample_sort_rounds(ctx, p, c, kind):
    """sample-sorted rounds == rounds sorted by the full network == rounds where both interleave == the oracle"""
    if kind == "pareto":
        w = _pareto_topic(p + c, p, c)
    else:
        po, pid, lag, co, ranks = _single_topic(p ^ c, p, c, kind)
        w = synth.Workload(kind, 1, np.asarray(po, np.int64), pid, np.zeros(p, np.int64), lag.copy(),
                           np.zeros(p, np.int64), lag, np.asarray(co, np.int64), ranks, p, c)
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    for flags, what in ((0, "sample sort"), (N.LA_FLAG_NO_SAMPLE_SORT, "full network"),
                        (N.LA_FLAG_SAMPLE_TIGHT, "interleaved"), (N.LA_FLAG_SORT_MULTIKERNEL, "four-kernel radix passes"),
                        (N.LA_FLAG_NO_RUN_MERGE, "no run merge"), (N.LA_FLAG_NO_RUN_MERGE | N.LA_FLAG_SAMPLE_TIGHT, "no run merge, interleaved")):
        got = _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=True, flags=flags)
        for g, e, name in zip(got, exp, ("partition order", "member", "totals")):
            np.testing.assert_array_equal(g, e, err_msg="%s, %s" % (name, what))


@pytest.mark.parametrize("p,c,kind,shuffled", [
    (4096 * 300 + 17, 5, "u40", True), (3_000_000, 7, "pareto", True), (4096 * 129, 3, "ties", True),
    (2_500_000, 64, "u63", True), (1_000_000, 2, "zero", False), (5_000_001, 1, "u40", True),
])
def test_large_radix_pass_forms_agree(ctx, p, c, kind, shuffled):
    """The device radix sort as single-kernel passes (decoupled look-back: a tile's digit offsets come from the
    granules the tiles before it publish) and as four-kernel passes (tile counts, scans, scatter): the same result, the
    oracle's, on topics of hundreds to thousands of tiles -- ragged last tile, one consumer, ties, ids already ascending."""
    if kind == "pareto":
        w = _pareto_topic(p + c, p, c)
    else:
        po, pid, lag, co, ranks = _single_topic(p ^ c, p, c, kind, shuffled=shuffled)
        w = synth.Workload(kind, 1, np.asarray(po, np.int64), pid, np.zeros(p, np.int64), lag.copy(),
                           np.zeros(p, np.int64), lag, np.asarray(co, np.int64), ranks, p, c)
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    for rep in range(3):                                      # the look-back is a race by design: more than one run
        got = _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=True)
        for g, e, name in zip(got, exp, ("partition order", "member", "totals")):
            np.testing.assert_array_equal(g, e, err_msg="%s, single-kernel passes, run %d" % (name, rep))
    got = _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=True, flags=N.LA_FLAG_SORT_MULTIKERNEL)
    for g, e, name in zip(got, exp, ("partition order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg="%s, four-kernel passes" % name)


@pytest.mark.parametrize("p,c,kind", [
    (400_000, 8192, "pareto"), (524_288, 4096, "pareto"), (300_000, 2048, "pareto"), (200_000, 8192, "ties"),
    (150_000, 5000, "zero"), (100_000, 3000, "few"), (250_000, 8000, "steps"), (90_000, 2049, "steps"),
])
def test_large_rounds_merge_ascending_runs(ctx, p, c, kind):
    """Greedy rounds whose bins form a few ascending runs (a flat tail of lags, ties, zero lags: one run) merge the runs
    instead of sorting: the same result as with LA_FLAG_NO_RUN_MERGE, the oracle's."""
    rng = np.random.default_rng(p + c)
    if kind == "pareto":
        w = _pareto_topic(p + c, p, c)
    else:
        if kind == "ties":
            lag = rng.integers(0, 5, p) * 1000
        elif kind == "zero":
            lag = np.zeros(p, dtype=np.int64)
        elif kind == "few":
            lag = rng.choice(np.array([7, 7, 7, 8, 1_000_000]), p)           # rounds with two or three distinct lags
        else:
            lag = (np.arange(p) // 997)[::-1] + rng.integers(0, 2, p)        # a staircase: a run per step and then some
        lag = np.ascontiguousarray(lag, dtype=np.int64)
        w = synth.Workload(kind, 1, np.array([0, p], np.int64), rng.permutation(p).astype(np.int32), np.zeros(p, np.int64),
                           lag.copy(), np.zeros(p, np.int64), lag, np.array([0, c], np.int64),
                           np.sort(rng.choice(3 * c + 1, c, replace=False)).astype(np.int32), p, c)
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    for flags, what in ((0, "runs merged"), (N.LA_FLAG_NO_RUN_MERGE, "runs sorted")):
        got = _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=True, flags=flags)
        for g, e, name in zip(got, exp, ("partition order", "member", "totals")):
            np.testing.assert_array_equal(g, e, err_msg="%s, %s" % (name, what))


@pytest.mark.parametrize("p,c,kind", [
    (400_000, 8192, "pareto"), (300_000, 4096, "pareto"), (120_000, 2048, "pareto"), (150_000, 5000, "pareto"),
    (200_000, 8192, "bulk16"), (200_000, 8192, "bulk5"), (200_000, 8192, "bulk3"), (100_000, 4000, "bulk16"),
    (60_000, 2048, "bulk5"), (160_000, 8192, "apart"), (90_000, 3000, "apart"), (200_000, 8192, "pairs"),
    # the wide form of the small sort (more bins move than the narrow form holds, at most twice as many): a power-law topic whose
    # rounds move 2 050 - 3 100 of 8 192 bins; dense bulks of 2 730 and of exactly 4 096 (the capacity) of 8 192 bins, of 1 365 and
    # of exactly 2 048 of 4 096 (four bins per thread); consumers that do not fill the workgroup's slots
    (131_072, 8192, "pareto"), (200_000, 8192, "bulk2"), (100_000, 4096, "bulk3"), (60_000, 4096, "bulk2"), (120_000, 8190, "bulk2"),
])
def test_large_rounds_sort_only_the_bins_that_move(ctx, p, c, kind):
    """Greedy rounds in which few bins change places (moved_sort_bins, la_large.hip) sort only those: the same result as
    with LA_FLAG_NO_MOVED_SORT, as with the run merge off too, as with tight bucket limits (rounds of every form mixed),
    the oracle's.  bulkN: 1 - 1/N of the consumers stand 10^9 apart after the first round, the rest is a dense bulk that
    reshuffles in every round (N = 3: more than the narrow form of the small sort holds -- the wide form's case since round 6; N = 2:
    exactly what the wide form holds); apart: nothing ever moves
    after round 1; pairs: neighbours swap."""
    rng = np.random.default_rng(7 * p + c)
    if kind == "pareto":
        w = _pareto_topic(p + c, p, c)
    else:
        first = np.empty(c, dtype=np.int64)
        if kind.startswith("bulk"):
            far = c - c // int(kind[4:])
            first[:far] = 10**13 - np.arange(far, dtype=np.int64) * 10**9
            first[far:] = 10**6 + rng.integers(0, 1000, c - far)
            rest = rng.integers(0, 100_000, p - c)
        elif kind == "apart":
            first[:] = 10**13 - np.arange(c, dtype=np.int64) * 10**9
            rest = rng.integers(0, 1_000_000, p - c)
        else:
            first[:] = 10**13 - np.arange(c, dtype=np.int64) * 10**9
            rest = rng.integers(0, 3 * 10**9, p - c)                         # a bin passes its neighbour, sometimes two
        lag = np.ascontiguousarray(rng.permutation(np.concatenate([first, rest])), dtype=np.int64)
        w = synth.Workload(kind, 1, np.array([0, p], np.int64), rng.permutation(p).astype(np.int32), np.zeros(p, np.int64),
                           lag.copy(), np.zeros(p, np.int64), lag, np.array([0, c], np.int64),
                           np.sort(rng.choice(3 * c + 1, c, replace=False)).astype(np.int32), p, c)
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    for flags, what in ((0, "default"), (N.LA_FLAG_NO_MOVED_SORT, "moved sort off"),
                        (N.LA_FLAG_NO_MOVED_SORT | N.LA_FLAG_NO_RUN_MERGE, "sample sort only"),
                        (N.LA_FLAG_SAMPLE_TIGHT, "tight limits"), (N.LA_FLAG_NO_RUN_MERGE, "moved sort, no run merge")):
        got = _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=True, flags=flags)
        for g, e, name in zip(got, exp, ("partition order", "member", "totals")):
            np.testing.assert_array_equal(g, e, err_msg="%s, %s" % (name, what))


def test_large_path_rank_forms_in_fresh_processes():
    """The radix sort ranks equal digits with returning LDS atomics when the device passes the lane-order self-test of
    la_create (LA_FEATURE_ATOMIC_RANK) and with wave-match ballots otherwise; LA_SORT_RANK=match forces the second form.
    The choice is made once per process, so each form gets its own: same topics, the oracle's result, both pass forms."""
    import subprocess
    import sys
    code = r"""
import numpy as np, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from kafka_lag_based_assignor_amd import _native as N
from oracle import oracle
import test_gpu_parity as t
ctx = N.Context(0)
want_atomic = %d
if want_atomic == 0:      # (whether a device passes the self-test is the device's business; forcing the match form is ours)
    assert not (ctx.device_features(0) & N.LA_FEATURE_ATOMIC_RANK)
for p, c, kind in ((300_000, 5, "u40"), (70_000, 2048, "ties"), (4096 * 40 + 7, 3, "u63"), (20_000, 3000, "zero")):
    po, pid, lag, co, ranks = t._single_topic(p + c, p, c, kind)
    exp = oracle.assign_flat(po, pid, lag, co, ranks)
    got = ctx.assign_batch_lags(po, pid, lag, co, ranks)
    assert all(np.array_equal(g, e) for g, e in zip(got, exp)), (p, c, kind)
    off, g_t, g_p = ctx.group_by_member(po, exp[0], exp[1], int(ranks.max()) + 1)
    assert np.array_equal(g_p, exp[0][np.argsort(exp[1], kind="stable")]), "lists"
print("ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for env_rank, want_atomic in (("match", 0), ("atomic", 1)):
        env = dict(os.environ, LA_SORT_RANK=env_rank)
        out = subprocess.run([sys.executable, "-c", code % (root, os.path.join(root, "tests"), want_atomic)],
                             env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0 and "ok" in out.stdout, (env_rank, out.stdout[-1500:], out.stderr[-1500:])


def test_large_topic_with_more_tiles_than_resident_workgroups(ctx):
    """5 M partitions = 306 tiles of 16 384 elements against the 256 workgroups of 1 024 threads a MI355X keeps resident:
    the single-kernel radix passes then run in more than one wave of workgroups (arrival tickets decide which tile a
    workgroup takes, so the look-back never waits for a tile that has not started)."""
    p, c = 5_000_000, 3
    po, pid, lag, co, ranks = _single_topic(99, p, c, "u40")
    exp = oracle.assign_flat(po, pid, lag, co, ranks)
    got = ctx.assign_batch_lags(po, pid, lag, co, ranks)
    for g, e, name in zip(got, exp, ("partition order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg=name)


def test_large_phase_times(ctx):
    fresh = N.Context(0)
    with pytest.raises(N.LagAssignError) as ei:                                            # nothing profiled yet
        fresh.last_phase_times()
    assert ei.value.code == N.LA_EINVAL
    fresh.close()
    w = _pareto_topic(3, 300000, 4096)
    _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=True, flags=N.LA_FLAG_PROFILE)
    t = ctx.last_phase_times()
    assert t.n_partitions == 300000 and t.id_passes == 3 and 1 <= t.key_passes <= 8      # ids < 2^24 shuffled
    assert t.keys_ms > 0 and t.sort_ms > 0 and t.greedy_ms > 0
    _run_device(ctx, synth.config("cfg3", 0.01), N.LA_ALGO_AUTO)                          # no large topic, no flag
    assert ctx.last_phase_times().n_partitions == 300000                                  # the record stays


# ---- block path: one workgroup per topic, 1 024 < P <= 8 192 or 64 < C <= 2 048 ------------------------------
@pytest.mark.parametrize("p,c,kind", [
    (1025, 1, "u40"), (8192, 2048, "u40"), (100, 65, "ties"), (1, 65, "u40"), (0, 100, "zero"), (5000, 0, "u40"),
    (8192, 3, "full"), (4097, 1025, "zero"), (2048, 256, "u63"), (2049, 257, "full"), (200, 100, "u40"),
    (127, 128, "ties"), (129, 128, "u40"), (3000, 129, "ties"), (8191, 2047, "u63"), (64, 2048, "u40"),
    # the 16-records-per-thread class (8 192 < P <= 16 384, C <= 1 024) and its borders with the large path
    (8193, 1, "u40"), (16384, 1024, "u40"), (16384, 1024, "full"), (10000, 128, "ties"), (12000, 300, "u63"),
    (16385, 7, "u40"), (9000, 1025, "ties"), (16000, 3, "zero"),
])
def test_block_single_topic(ctx, p, c, kind):
    po, pid, lag, co, ranks = _single_topic(31 * p + c, p, c, kind, negative=(kind == "full"))
    exp = oracle.assign_flat(po, pid, lag, co, ranks)
    got = ctx.assign_batch_lags(po, pid, lag, co, ranks)
    for g, e, what in zip(got, exp, ("partition order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg=what)


# up to 64 consumers and more than 1 024 partitions: the one-wavefront greedy whose round is one asm statement
# (la_sort64.h (d): widths 2 / 4 / 8 / 16) or the generic form (1 / 32 / 64); every consumer count around the widths,
# partition counts that end a round early, lags that pack ("u40", "ties", "zero", "u22" = near the 2^62 total limit for
# few rounds) and lags that do not ("u63", "full" with negatives: the 96-bit path)
@pytest.mark.parametrize("c", [1, 2, 3, 4, 5, 7, 8, 9, 12, 15, 16, 17, 24, 31, 32, 33, 48, 63, 64])
def test_block_one_wave_rounds_every_width(ctx, c):
    rng = np.random.default_rng(4200 + c)
    for p, kind in [(1025, "u40"), (1024 + c, "ties"), (1023 + 2 * c, "u40"), (2048, "zero"), (3000 + c, "u40"),
                    (8192, "ties"), (8191, "u40"), (8192 - c + 1, "u63"), (5000, "full"), (int(rng.integers(1025, 8193)), "u40"),
                    (16384 if c <= 32 else 8192, "u40")]:
        po, pid, lag, co, ranks = _single_topic(97 * p + c, p, c, kind, shuffled=bool(rng.integers(0, 2)),
                                                negative=(kind == "full"))
        exp = oracle.assign_flat(po, pid, lag, co, ranks)
        got = ctx.assign_batch_lags(po, pid, lag, co, ranks)
        for g, e, what in zip(got, exp, ("partition order", "member", "totals")):
            np.testing.assert_array_equal(g, e, err_msg="%s p=%d c=%d %s" % (what, p, c, kind))


def test_block_one_wave_rounds_at_the_packing_limit(ctx):
    # lag bits + round bits + index bits = 62 packs, 63 does not: both sides of the switch give the oracle's answer
    for c, p in [(16, 4096), (4, 2000), (64, 8192), (3, 1500)]:
        n_c = 1 << (c - 1).bit_length()
        idx_bits = max(n_c.bit_length() - 1, 0)
        round_bits = (-(-p // c)).bit_length()
        for lag_bits in (62 - round_bits - idx_bits, 63 - round_bits - idx_bits):
            rng = np.random.default_rng(lag_bits * 131 + c)
            lag = rng.integers(0, 1 << lag_bits, p).astype(np.int64)
            lag[0] = (1 << lag_bits) - 1
            pid = rng.permutation(p).astype(np.int32)
            ranks = np.arange(c, dtype=np.int32)
            exp = oracle.assign_flat([0, p], pid, lag, [0, c], ranks)
            got = ctx.assign_batch_lags([0, p], pid, lag, [0, c], ranks)
            for g, e, what in zip(got, exp, ("partition order", "member", "totals")):
                np.testing.assert_array_equal(g, e, err_msg="%s p=%d c=%d lag_bits=%d" % (what, p, c, lag_bits))


def test_block_one_wave_rounds_many_topics(ctx):
    # 300 topics of different widths side by side (more workgroups than CUs), partitions not a multiple of the consumers
    rng = np.random.default_rng(99)
    t = 300
    ps = rng.integers(1025, 3000, t)
    cs = rng.integers(1, 65, t)
    part_off = np.concatenate([[0], np.cumsum(ps)]).astype(np.int64)
    cons_off = np.concatenate([[0], np.cumsum(cs)]).astype(np.int64)
    pid = np.concatenate([rng.permutation(int(x)) for x in ps]).astype(np.int32)
    lag = rng.integers(0, 1 << 30, int(part_off[-1])).astype(np.int64)
    lag[rng.random(lag.size) < 0.3] = 0                                   # caught-up partitions: ties at the tail
    ranks = np.concatenate([np.sort(rng.choice(200, int(x), replace=False)) for x in cs]).astype(np.int32)
    exp = oracle.assign_flat(part_off, pid, lag, cons_off, ranks)
    got = ctx.assign_batch_lags(part_off, pid, lag, cons_off, ranks)
    for g, e, what in zip(got, exp, ("partition order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg=what)


@pytest.mark.parametrize("seed,max_p,max_c", [(1, 3000, 300), (2, 9000, 70), (3, 1500, 2500), (4, 600, 100),
                                              (5, 8192, 2048)])
def test_block_batches_mixed_with_tile_and_large_topics(ctx, seed, max_p, max_c):
    # ragged batches whose topics fall in all three classes (wave tile, block, large), offsets entry point in
    # both reset modes and the lag entry point with negative lags
    w = synth.ragged(9000 + seed, 120 if max_p < 5000 else 40, max_p, max_c, negative=True)
    for mode in (N.LA_RESET_LATEST, N.LA_RESET_EARLIEST):
        _check_offsets(ctx, w, mode, "ragged block batch %d" % seed)
    _check_lags(ctx, w, "ragged block batch %d, lags" % seed)


def test_block_many_topics_same_shape(ctx):
    # more topics than resident workgroups: 1 500 topics x 300 partitions x 100 consumers
    rng = np.random.default_rng(77)
    t, p, c = 1500, 300, 100
    part_off = np.arange(t + 1, dtype=np.int64) * p
    cons_off = np.arange(t + 1, dtype=np.int64) * c
    pid = np.concatenate([rng.permutation(p) for _ in range(t)]).astype(np.int32)
    lag = rng.integers(0, 1 << 34, t * p).astype(np.int64)
    ranks = np.tile(np.arange(c, dtype=np.int32), t)
    exp = oracle.assign_flat(part_off, pid, lag, cons_off, ranks)
    for rep in range(2):                                  # twice: the host's list slots rotate
        got = ctx.assign_batch_lags(part_off, pid, lag, cons_off, ranks)
        for g, e, what in zip(got, exp, ("partition order", "member", "totals")):
            np.testing.assert_array_equal(g, e, err_msg=what)


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_block_topics(ctx, seed):
    rng = np.random.default_rng(7000 + seed)
    p = int(rng.choice([0, 1, 63, 64, 65, 127, 128, 129, 1000, 1024, 1025, 2047, 2048, 2049, 4096, 4097, 8191, 8192, 8193,
                        12000, 16383, 16384]))
    c = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1023, 1024, 1025, 2047, 2048]))
    kind = str(rng.choice(["u40", "ties", "zero", "u63", "full"]))
    po, pid, lag, co, ranks = _single_topic(seed + 177, p, c, kind, shuffled=bool(rng.integers(0, 2)),
                                            negative=(kind == "full"))
    exp = oracle.assign_flat(po, pid, lag, co, ranks)
    got = ctx.assign_batch_lags(po, pid, lag, co, ranks)
    for g, e, what in zip(got, exp, ("partition order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg="%s p=%d c=%d %s" % (what, p, c, kind))


# ---- ragged batches: tile-sized topics grouped by shape class (one launch per class over a topic list) ------
def _ragged_skewed(seed, t, big_every):
    """Mostly tiny topics, every `big_every`-th one near the tile limit: the shape mix that makes classes pay."""
    rng = np.random.default_rng(seed)
    ps = rng.integers(0, 40, t)
    cs = rng.integers(0, 7, t)
    mid = rng.random(t) < 0.2
    ps[mid] = rng.integers(65, 256, int(mid.sum()))
    cs[mid] = rng.integers(1, 33, int(mid.sum()))
    ps[::big_every] = rng.integers(600, 1025, ps[::big_every].size)
    cs[::big_every] = rng.integers(1, 65, cs[::big_every].size)
    part_off = np.concatenate([[0], np.cumsum(ps)]).astype(np.int64)
    cons_off = np.concatenate([[0], np.cumsum(cs)]).astype(np.int64)
    n = int(part_off[-1])
    pid = np.concatenate([rng.permutation(int(p)) for p in ps]).astype(np.int32)
    lag = rng.integers(0, 1 << 36, n).astype(np.int64)
    lag[rng.integers(0, n, 50)] = -5                                  # a few tiles take the wide-record kernel
    ranks = np.concatenate([np.sort(rng.choice(200, int(c), replace=False)) for c in cs]).astype(np.int32)
    zeros = np.zeros(n, dtype=np.int64)
    return synth.Workload("skewed", t, part_off, pid, zeros, lag.copy(), zeros, lag, cons_off, ranks,
                          int(ps.max()), int(cs.max()))


@pytest.mark.parametrize("t,big_every", [(6000, 3), (20000, 4), (5000, 1000), (300, 2)])
def test_ragged_tile_batch_by_shape_class(ctx, t, big_every):
    w = _ragged_skewed(t + big_every, t, big_every)
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    got = ctx.assign_batch_lags(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)   # host entry: classes itself
    for g, e, what in zip(got, exp, ("partition order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg=what)
    forced = N.LA_FLAG_RAGGED | N.LA_FLAG_SHAPE_CLASSES
    for flags in (0, N.LA_FLAG_RAGGED, forced, forced | N.LA_FLAG_INDEX64 | N.LA_FLAG_DEFER_WIDE):
        got = _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=True, flags=flags)
        for g, e, what in zip(got, exp, ("partition order", "member", "totals")):
            np.testing.assert_array_equal(g, e, err_msg="%s flags=%d" % (what, flags))


def test_ragged_batch_all_three_paths(ctx):
    # shape classes for the tile-sized majority + block topics + one large topic in the same call
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
    got = ctx.assign_batch_lags(part_off, pid, lag, cons_off, ranks)
    for g, e, what in zip(got, exp, ("partition order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg=what)


# ---- committed fixtures: the HIP path against tests/golden/oracle_frozen.json -------------------------------
def test_hip_path_matches_frozen_digests(ctx):
    import importlib.util
    import json
    import os
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(gdir, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    frozen = json.load(open(os.path.join(gdir, "oracle_frozen.json")))
    for name, scale, mode in mg.CASES:
        w = synth.config(name, scale)
        if mode == "lags":
            p, m, t = ctx.assign_batch_lags(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
        else:
            latest = mode == "latest"
            p, m, t = ctx.assign_batch(w.part_off, w.partition_id, None if latest else w.begin, w.end, w.committed,
                                       N.LA_RESET_LATEST if latest else N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
        got = mg.digest(p.astype(np.int32), m.astype(np.int32), t.astype(np.int64))
        assert got == frozen[mg.case_key(name, scale, mode)]["sha256"], "%s %s" % (name, mode)


def test_hip_path_matches_the_full_size_frozen_digests(ctx):
    """tests/golden/oracle_frozen_full.json (make_golden.py --full): every BASELINE config at FULL size, the headline batch
    and its form without any committed offset -- committed fixtures of the literal oracle; the HIP path must reproduce the
    digests through the host entry point (and the inputs are the generator's: their digest is in the fixture too)."""
    import importlib.util
    import json
    import os
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(gdir, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    frozen = json.load(open(os.path.join(gdir, "oracle_frozen_full.json")))
    for name, scale, mode in mg.FULL_CASES:
        none_frac = float(mode.split("none=")[1]) if "none=" in mode else None
        w = synth.config(name, scale, none_frac=none_frac) if none_frac is not None else synth.config(name, scale)
        fz = frozen[mg.case_key(name, scale, mode)]
        assert mg.digest(w.part_off, w.partition_id, w.begin, w.end, w.committed, w.cons_off, w.cons_rank) == fz["inputs_sha256"]
        p, m, t = ctx.assign_batch(w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
        assert mg.digest(p.astype(np.int32), m.astype(np.int32), t.astype(np.int64)) == fz["sha256"], "%s %s" % (name, mode)


def test_deferred_wide_tiles_in_a_batch_larger_than_the_resident_grid(ctx):
    # Small batches run one kernel with the wide-record code inline; beyond one round of resident workgroups
    # the packed kernel defers tiles it cannot pack to a list that the wide kernel walks.  100 000 tiny topics
    # (8 per wavefront) with a sprinkling of negative / huge lags exercise that list.
    w = _deferring_workload()
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    for rep in range(2):                                             # twice: the counter pair alternates
        got = _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=True)
        for g, e, what in zip(got, exp, ("partition order", "member", "totals")):
            np.testing.assert_array_equal(g, e, err_msg="%s rep %d" % (what, rep))


@pytest.mark.parametrize("flags", [N.LA_FLAG_INDEX64, N.LA_FLAG_DEFER_WIDE, N.LA_FLAG_INDEX64 | N.LA_FLAG_DEFER_WIDE])
@pytest.mark.parametrize("max_p,max_c", [(8, 8), (100, 16), (256, 32), (1024, 64)])
def test_kernel_variants_selected_by_size_agree(ctx, flags, max_p, max_c):
    # 64-bit indexing is otherwise only taken beyond 2^29 partitions, the deferred-tile list only beyond one
    # round of resident workgroups: force both on small mixed batches (packed and wide tiles in one launch)
    w = synth.ragged(max_p * 3 + max_c + flags, 300, max_p, max_c, negative=True)
    for latest in (True, False):
        lag = oracle.compute_lags(w.begin, w.end, w.committed, latest)
        exp = oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)
        got = _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=False, latest=latest, flags=flags)
        for g, e in zip(got, exp):
            np.testing.assert_array_equal(g, e)
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    got = _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=True, flags=flags)
    for g, e in zip(got, exp):
        np.testing.assert_array_equal(g, e)
