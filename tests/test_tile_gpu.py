"""Wave-tile path (csrc/la_wave_tile_impl.h) through the C ABI, against the oracle.

computePartitionLag's fall-back (Main.java:384-396) at every share of partitions without a committed offset: the
second-stage `begin` loads of the tile kernel are 16 bytes per lane since round 6 (one load per pair of partitions
whenever either needs its beginning offset), and a brand-new consumer group -- NO committed offset anywhere -- is the one
workload in which the whole 36 B/partition of SURVEY 8d really moves.
"""
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
from test_gpu_parity import _run_device

pytestmark = pytest.mark.gpu


def _expect(w, latest):
    lag = oracle.compute_lags(w.begin, w.end, w.committed, latest)
    return oracle.assign_flat(w.part_off, w.partition_id, lag, w.cons_off, w.cons_rank)


@pytest.mark.parametrize("none_frac", [0.0, 0.01, 0.3, 0.5, 0.97, 1.0])
@pytest.mark.parametrize("shape", [(300, 256, 32), (513, 64, 8), (77, 100, 5), (41, 1000, 64), (1200, 7, 3), (9, 1, 1)])
def test_every_share_of_partitions_without_a_committed_offset(ctx, none_frac, shape):
    t, p, c = shape
    w = synth.make_uniform("none", 31 + p, t, p, c, "uniform40", none_frac=none_frac)
    # begin offsets that matter: where a committed offset exists they must be ignored, so make them large there
    rng = np.random.default_rng(p)
    w.begin = np.where(w.committed < 0, w.begin, rng.integers(0, 1 << 41, w.begin.size)).astype(np.int64)
    for latest in (False, True):
        exp = _expect(w, latest)
        got = _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=False, latest=latest)
        for g, e, what in zip(got, exp, ("partition order", "member", "totals")):
            np.testing.assert_array_equal(g, e, err_msg="none_frac %s %s latest=%s: %s" % (none_frac, shape, latest, what))
        got = ctx.assign_batch(w.part_off, w.partition_id, w.begin, w.end, w.committed,
                               N.LA_RESET_LATEST if latest else N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
        for g, e, what in zip(got, exp, ("partition order", "member", "totals")):
            np.testing.assert_array_equal(g, e, err_msg="host call, none_frac %s %s latest=%s: %s" % (none_frac, shape, latest, what))


def test_ragged_batch_without_committed_offsets(ctx):
    w = synth.ragged(77, 900, 300, 40)
    w.committed = np.full_like(w.committed, -1)
    w.begin = np.minimum(w.begin, w.end)
    for latest in (False, True):
        exp = _expect(w, latest)
        got = _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=False, latest=latest, flags=N.LA_FLAG_RAGGED)
        for g, e in zip(got, exp):
            np.testing.assert_array_equal(g, e)


def test_target_full_size_without_committed_offsets(ctx):
    """100 000 x 256 x 32 with NO committed offset (earliest): begin = what committed would have been, so every lag is the
    drawn one (in the 1 % workload the partitions without a committed offset have lag = end - 0 instead); against the literal
    oracle on all 25.6 M partitions, through the device and the host entry points."""
    w = synth.config("target", none_frac=1.0)
    assert (w.committed < 0).all()
    assert np.array_equal(oracle.compute_lags(w.begin, w.end, w.committed, False), w.lag)
    exp = _expect(w, False)
    got = _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=False, latest=False)
    for g, e, what in zip(got, exp, ("partition order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg="device call: " + what)
    got = ctx.assign_batch(w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
    for g, e, what in zip(got, exp, ("partition order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg="host call: " + what)
    # latest: every lag is 0 (Main.java:391-392: next offset = end), the order is by partition id alone
    got = _run_device(ctx, w, N.LA_ALGO_AUTO, use_lag=False, latest=True)
    assert not got[2].any()
    for g, e in zip(got, _expect(w, True)):
        np.testing.assert_array_equal(g, e)


# ---- wire format ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("max_id,n_members,n,shift", [
    (255, 32, 100003, 0), (63, 8, 8 * 1024, 0), (255, 32, 1000, 3), (70000, 3, 50001, 0), (70000, 3, 777, 1),
    (-1, 100, 40000, 0), (2 ** 20 - 1, 8192, 12345, 2), (0, 0, 17, 0), (255, 255, 4096, 0), (255, 256, 4096, 5)])
def test_wire_pack_unpack_equal_the_numpy_restatement(ctx, torch_dev, max_id, n_members, n, shift):
    torch, dev = torch_dev
    rng = np.random.default_rng(n + shift)
    fmt = N.wire_format_for(max_id, n_members)
    assert (fmt.elem_bytes, fmt.id_bits) == sharding.wire_format_numpy(max_id, n_members)
    if max_id < 0:
        pid = rng.integers(-2 ** 31, 2 ** 31 - 1, n).astype(np.int32)
    else:
        pid = rng.integers(0, max_id + 1, n).astype(np.int32)
        pid[: min(n, 4)] = max_id
    rank = rng.integers(-1, max(n_members, 1), n).astype(np.int32) if n_members else np.full(n, -1, np.int32)
    if n_members:
        rank[-1] = n_members - 1
    # `shift` elements of offset: pointers that are NOT 16-byte aligned take the scalar kernels
    d_pid = torch.zeros(n + 8, dtype=torch.int32, device=dev)
    d_rank = torch.zeros(n + 8, dtype=torch.int32, device=dev)
    d_pid[shift:shift + n] = torch.from_numpy(pid).to(dev)
    d_rank[shift:shift + n] = torch.from_numpy(rank).to(dev)
    eb = fmt.elem_bytes
    d_wire = torch.zeros((n + 8) * eb, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    ctx.pack_results(n, d_pid.data_ptr() + 4 * shift, d_rank.data_ptr() + 4 * shift, fmt, d_wire.data_ptr() + eb * shift, stream)
    ctx.sync(stream)
    got = d_wire.cpu().numpy()[eb * shift: eb * (shift + n)].view(fmt.dtype)
    exp = sharding.pack_results_numpy(pid, rank, eb, fmt.id_bits)
    np.testing.assert_array_equal(got, exp)
    o_pid = torch.full((n + 8,), 7, dtype=torch.int32, device=dev)
    o_rank = torch.full((n + 8,), 7, dtype=torch.int32, device=dev)
    ctx.unpack_results(n, d_wire.data_ptr() + eb * shift, fmt, o_pid.data_ptr() + 4 * shift, o_rank.data_ptr() + 4 * shift, stream)
    ctx.sync(stream)
    np.testing.assert_array_equal(o_pid.cpu().numpy()[shift:shift + n], pid)
    np.testing.assert_array_equal(o_rank.cpu().numpy()[shift:shift + n], rank)
    assert int(o_pid[shift + n]) == 7 and (shift == 0 or int(o_pid[shift - 1]) == 7)      # nothing outside the run is touched


def test_wire_misfit_is_an_error_not_a_truncation(ctx, torch_dev):
    torch, dev = torch_dev
    stream = torch.cuda.current_stream().cuda_stream
    fmt = N.wire_format_for(255, 32)                                                      # 2 bytes, 8 id bits
    for pid, rank in ((np.array([1, 256, 3], np.int32), np.array([0, 1, 2], np.int32)),   # an id beyond the format
                      (np.array([1, 2, 3], np.int32), np.array([0, 255, 2], np.int32)),   # a rank beyond it
                      (np.array([1, -2, 3], np.int32), np.array([0, 1, 2], np.int32))):   # a negative id
        d_pid, d_rank = torch.from_numpy(pid).to(dev), torch.from_numpy(rank).to(dev)
        d_wire = torch.zeros(16, dtype=torch.uint8, device=dev)
        ctx.pack_results(3, d_pid.data_ptr(), d_rank.data_ptr(), fmt, d_wire.data_ptr(), stream)
        with pytest.raises(N.LagAssignError) as e:
            ctx.sync(stream)
        assert e.value.code == N.LA_EINVAL and "wire format" in str(e.value)
    ctx.sync(stream)                                                                      # the status word was cleared
    bad = N.WireFormat()
    bad.elem_bytes, bad.id_bits = 3, 8
    with pytest.raises(N.LagAssignError):
        ctx.pack_results(3, 1, 1, bad, 1, stream)


def test_wire_round_trip_of_real_results_cfg4(ctx, torch_dev):
    """A shard's results -> wire -> back: exactly what a rank sends and what every rank rebuilds."""
    torch, dev = torch_dev
    w = synth.config("cfg4", 0.05)
    p, m, _ = ctx.assign_batch(w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
    fmt = N.wire_format_for(int(w.partition_id.max()), int(w.cons_rank.max()) + 1)
    assert fmt.elem_bytes == 2
    n = p.size
    stream = torch.cuda.current_stream().cuda_stream
    d_p, d_m = torch.from_numpy(p).to(dev), torch.from_numpy(m).to(dev)
    d_w = torch.empty(n * 2, dtype=torch.uint8, device=dev)
    o = torch.empty(2 * n, dtype=torch.int32, device=dev)
    ctx.pack_results(n, d_p.data_ptr(), d_m.data_ptr(), fmt, d_w.data_ptr(), stream)
    ctx.unpack_results(n, d_w.data_ptr(), fmt, o.data_ptr(), o.data_ptr() + 4 * n, stream)
    ctx.sync(stream)
    np.testing.assert_array_equal(o[:n].cpu().numpy(), p)
    np.testing.assert_array_equal(o[n:].cpu().numpy(), m)


# ---- LA_FLAG_BOUNDS: one launch when the caller's bounds prove that every tile packs ---------------------------------------------
def test_bounds_hint_single_launch_and_violations(ctx):
    """With bounds on lags and ids that prove the packed format for every tile the tile path skips its second launch; results are
    the same; a tile that does not pack after all (the bounds were wrong) is LA_EINVAL, not a different result; launches with
    and without the hint, deferring and not, alternate on one context without disturbing the deferred-tile counters."""
    import ctypes
    import torch
    dev = torch.device("cuda", 0)

    def run(w, bounds=None, flags=0):
        d = {k: torch.from_numpy(np.ascontiguousarray(getattr(w, k))).to(dev) for k in ("part_off", "partition_id", "lag", "cons_off", "cons_rank")}
        out_p = torch.full((w.n_partitions,), -7, device=dev, dtype=torch.int32)
        out_m = torch.full((w.n_partitions,), -7, device=dev, dtype=torch.int32)
        out_t = torch.full((max(w.cons_rank.size, 1),), -7, device=dev, dtype=torch.int64)
        b = N.DeviceBatch()
        b.n_topics, b.reset_mode, b.algo, b.flags = w.n_topics, N.LA_RESET_LATEST, N.LA_ALGO_AUTO, flags
        b.n_partitions, b.n_consumers = w.n_partitions, w.cons_rank.size
        b.max_partitions_per_topic, b.max_consumers_per_topic = w.max_partitions, w.max_consumers
        b.d_part_off, b.d_partition_id, b.d_lag = d["part_off"].data_ptr(), d["partition_id"].data_ptr(), d["lag"].data_ptr()
        b.d_cons_off, b.d_cons_rank = d["cons_off"].data_ptr(), d["cons_rank"].data_ptr()
        b.d_out_partition, b.d_out_member_rank, b.d_out_total_lag = out_p.data_ptr(), out_m.data_ptr(), out_t.data_ptr()
        if bounds is not None:
            b.flags |= N.LA_FLAG_BOUNDS
            b.max_lag_hint, b.max_partition_id_hint = bounds
        stream = torch.cuda.current_stream().cuda_stream
        ctx.assign_batch_device(b, stream)
        ctx.sync(stream)
        return out_p.cpu().numpy(), out_m.cpu().numpy(), out_t.cpu().numpy()[: w.cons_rank.size]

    small = synth.make_uniform("b", 41, 3000, 256, 32, "zipf", offsets=False)          # lags <= 1e9, ids < 256: packs
    wide = synth.make_uniform("b", 42, 3000, 64, 8, "uniform63", offsets=False)        # 63-bit lags: every tile defers
    e_small = oracle.assign_flat(small.part_off, small.partition_id, small.lag, small.cons_off, small.cons_rank)
    e_wide = oracle.assign_flat(wide.part_off, wide.partition_id, wide.lag, wide.cons_off, wide.cons_rank)
    tight = (int(small.lag.max()), int(small.partition_id.max()))
    for step in range(3):
        _same3(run(wide, flags=N.LA_FLAG_DEFER_WIDE), e_wide, "deferring launch %d" % step)
        _same3(run(small, tight, flags=N.LA_FLAG_DEFER_WIDE), e_small, "bounded launch %d" % step)
        _same3(run(small, flags=N.LA_FLAG_DEFER_WIDE), e_small, "unbounded launch %d" % step)
        _same3(run(small, tight), e_small, "bounded, single-launch form")
    # bounds that prove nothing (too wide) are simply not used
    _same3(run(wide, ((1 << 62), 63), flags=N.LA_FLAG_DEFER_WIDE), e_wide, "bounds too wide to prove anything")
    # bounds that are wrong: a tile that cannot pack after all is an error
    with pytest.raises(N.LagAssignError) as e:
        run(wide, (1000, 63), flags=N.LA_FLAG_DEFER_WIDE)
    assert e.value.code == N.LA_EINVAL and "LA_FLAG_BOUNDS" in str(e.value)
    _same3(run(wide, flags=N.LA_FLAG_DEFER_WIDE), e_wide, "after the error")
    _same3(run(small, tight, flags=N.LA_FLAG_DEFER_WIDE), e_small, "bounded, after the error")


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
    import gpu_helpers as t4
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
    import gpu_helpers as t4
    for (t, p, c) in [(1000, 256, 32), (300, 64, 8), (50, 1000, 9), (2000, 100, 5), (7, 1024, 64), (400, 8, 8), (3000, 30, 3)]:
        w = synth.make_uniform("widen", t + p, t, p, c, "zipf")
        exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)     # (_device_call hands over w.lag)
        _same3(t4._device_call(ctx, w), exp, "%d x %d x %d" % (t, p, c))
