"""Round-4 additions, through the C ABI, bit-exact against the oracle:

* the narrow wire format of the all-gather (la_pack_results_on / la_unpack_results_on, include/lagassign.h): equal to the numpy
  restatement in sharding.py at every width, any alignment, a misfit is an error, round trip over real results;
* la_assign_batch_sparse / la_assign_batch_grouped_sparse: `begin` only where there is no committed offset (Main.java:384-396)
  -- 0 %, 1 %, 100 % uncommitted, `latest`, the one-copy, lanes and three-stream pipelines, several shards, bad lists;
* the block path with P == np_cap (explicit zero slot), group paths report device status, the caller's current
  device survives every call, la_version follows the header.
"""
import os
import re

import numpy as np
import pytest

from kafka_lag_based_assignor_amd import _native as N
from kafka_lag_based_assignor_amd import sharding, synth
from oracle import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ctx():
    c = N.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def torch_dev():
    import torch
    return torch, torch.device("cuda", 0)


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


def _pinned(c, a):
    a = np.ascontiguousarray(a)
    p = c.host_alloc(a.shape, a.dtype)
    p[...] = a
    return p


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


# ---- ADVICE r3 ----------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("c", [1, 7, 64])
def test_block_path_with_exactly_np_cap_partitions(ctx, c):
    """16 384 partitions x <= 64 consumers: the E = 16 class filled to the last slot; the one-wavefront slots greedy reads
    s_key[P] as its zero slot, which has its own 16 bytes now (it used to alias s_tot[0])."""
    rng = np.random.default_rng(c)
    p = 16384
    part_off = np.array([0, p, 2 * p], np.int64)
    cons_off = np.array([0, c, 2 * c], np.int64)
    pid = np.concatenate([rng.permutation(p), rng.permutation(p)]).astype(np.int32)
    lag = rng.integers(0, 1 << 30, 2 * p).astype(np.int64)
    lag[p:] = rng.integers(0, 3, p)                                                # heavy ties in the second topic
    ranks = np.concatenate([np.sort(rng.choice(500, c, replace=False)) for _ in range(2)]).astype(np.int32)
    got = ctx.assign_batch_lags(part_off, pid, lag, cons_off, ranks)
    exp = oracle.assign_flat(part_off, pid, lag, cons_off, ranks)
    for g, e, what in zip(got, exp, ("order", "member", "totals")):
        np.testing.assert_array_equal(g, e, err_msg=what)


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


# ---- large topics side by side and more than 8 192 consumers (#6) ------------------------------------------
from oracle.round_form import round_form  # noqa: E402


@pytest.mark.parametrize("topics,p,c", [
    (1, 513, 3), (3, 2049, 70), (2, 4097, 200), (1, 8193, 128), (2, 10000, 128), (1, 12289, 256), (40, 700, 128), (40, 1500, 256),
    (600, 300, 128), (600, 300, 200), (5, 9000, 65), (3, 5000, 129),
])
def test_block_sort_skips_sentinel_halves_and_small_multi_wave_greedy(ctx, topics, p, c):
    """Round 4, block path: (i) the records of a topic go to as few wavefronts as hold them and the packed sort skips every
    merge whose upper half is all sentinels -- partition counts just above a class border (513, 2 049, 4 097, 8 193) and in
    the middle of one; (ii) 65 .. 256 consumers with packed bins: one bin per lane on 2 / 4 wavefronts, the others leave the
    workgroup (128 bins only in launches of up to 512 topics: 600 topics take the one-wavefront form).  Mixed lag kinds (the
    "full" topics do not pack: they take the 96-bit forms of both steps), against the oracle."""
    w = _batch_of([(p + (i % 3), c - (i % 2)) for i in range(topics)], 1000 * topics + p + c)
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    _same3(_device_call(ctx, w), exp)


def test_block_sort_forms_in_fresh_processes():
    """The block path sorts packed records by digits inside the workgroup (block_sort_radix: ranks from returning LDS
    atomics) and with the bitonic network where the device lacks LA_FEATURE_ATOMIC_RANK or LA_BLOCK_RADIX says so; the
    choice is made once per process.  Every form, the same topics (all five size classes, partition counts at and between
    the class borders, packed and 96-bit records), the oracle's result."""
    import subprocess
    import sys
    code = r"""
import numpy as np, sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
from kafka_lag_based_assignor_amd import _native as N
from oracle import oracle
import test_round4_gpu as t
ctx = N.Context(0)
for topics, p, c in ((3, 100, 65), (2, 512, 3), (2, 513, 70), (2, 2048, 256), (1, 2049, 300), (2, 4096, 1000), (1, 4097, 17),
                     (1, 8192, 2048), (1, 8193, 5), (1, 10000, 128), (1, 16384, 1024), (300, 130, 70), (7, 1, 65)):
    w = t._batch_of([(p - (i %% 2), c) for i in range(topics)], 77 * topics + p + c)
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    t._same3(t._device_call(ctx, w), exp, what=str((topics, p, c)))
print("ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("0", "1", "2", "3"):                              # 3: topics below the size threshold too
        env = dict(os.environ, LA_BLOCK_RADIX=mode)
        out = subprocess.run([sys.executable, "-c", code % (root, os.path.join(root, "tests"))], env=env, capture_output=True,
                             text=True, timeout=900)
        assert out.returncode == 0 and "ok" in out.stdout, (mode, out.stdout[-1500:], out.stderr[-1500:])


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


def test_large_topics_side_by_side_mixed_batch(ctx):
    """Large topics of every tile class and rounds class, with and without consumers, between tile- and block-sized topics and
    topics without partitions: the shared launches give what the serial form gives, which is what the oracle gives."""
    shapes = [(100, 5), (20000, 50), (0, 3), (70000, 300), (3000, 200), (18000, 3000), (17000, 0), (150000, 8192),
              (40000, 1), (0, 0), (16385, 64), (300000, 700), (20000, 2049), (64, 8), (3300000, 5)]
    w = _batch_of(shapes, 11)
    exp = round_form(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    got = _device_call(ctx, w)
    _same3(got, exp, "side by side")
    serial = _device_call(ctx, w, flags=N.LA_FLAG_SERIAL_LARGE)
    _same3(serial, exp, "serial hook")
    host = ctx.assign_batch_lags(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)       # the host entry, chunks and all
    _same3(host, exp, "host entry")
    # and twice in a row on the same context (the staging slots alternate; scratch is reused)
    _same3(_device_call(ctx, w), exp, "second call")
    # test hooks of the sort / the greedy still apply to every topic of the launch
    for fl in (N.LA_FLAG_NO_SAMPLE_SORT, N.LA_FLAG_SAMPLE_TIGHT, N.LA_FLAG_NO_RUN_MERGE, N.LA_FLAG_SORT_MULTIKERNEL):
        _same3(_device_call(ctx, w, flags=fl), exp, "flag %d" % fl)


def test_many_equal_large_topics_literal_oracle(ctx):
    """24 topics x 20 000 partitions x 2 100 consumers with wide, negative and tied lags, against the LITERAL oracle."""
    w = _batch_of([(20000, 2100)] * 24, 5, kinds=["full", "u40", "ties"], negative=True)
    exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    _same3(_device_call(ctx, w), exp)


@pytest.mark.parametrize("p,c,kind", [
    (200000, 20000, "pareto"), (8193 * 3, 8193, "u40"), (5000, 9000, "u40"), (18000, 9000, "ties"), (30000, 10000, "zero"),
    (27000, 9000, "full"), (100000, 70000, "u20")])
def test_more_consumers_than_one_workgroup_holds(ctx, p, c, kind):
    """> 8 192 consumers: bins in HBM, a device sort per round.  P < C (one partial round), P = k * C exactly,
    ties, zero lags, negative lags with wrapping totals."""
    w = _batch_of([(p, c)], p + c, kinds=[kind], negative=True)
    exp = round_form(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    _same3(_device_call(ctx, w), exp, "%d x %d %s" % (p, c, kind))
    if p * c <= 200_000_000:                                             # the literal per-step min where it is affordable
        _same3(exp, oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank), "round form vs literal")
    got = _device_call(ctx, w, want_totals=False)
    np.testing.assert_array_equal(got[1], exp[1])


def test_huge_consumer_topic_inside_a_batch(ctx):
    w = _batch_of([(256, 32), (30000, 8300), (20000, 100), (0, 9000), (60000, 20000), (5000, 300)], 21)
    exp = round_form(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    _same3(_device_call(ctx, w), exp)
    _same3(ctx.assign_batch_lags(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank), exp, "host entry")


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


@pytest.mark.parametrize("n,kind,shuffled,dup,expect_first,expect_redo", [
    (70000, "wide", True, False, 1, 0), (200000, "runs", True, False, 1, 0), (150000, "run4096", True, False, 1, 0),
    (150000, "run9000", True, False, 1, 1), (300000, "run20000_top", True, False, 1, 1), (50000, "equal", True, False, 0, 0),
    (90000, "full", True, False, 1, 1), (120000, "runs", True, True, 1, 0), (80000, "runs", False, False, 0, 0),
    (20000, "runs", True, False, 1, 0), (16385, "wide", True, False, 1, 0), (40000, "pairs", True, False, 1, 0)])
def test_keys_first_sort_forced_on_small_topics(n, kind, shuffled, dup, expect_first, expect_redo):
    """LA_SORT_KEYS_FIRST=2 (test hook): every large-path sort with shuffled ids skips its id passes and repairs the runs of
    equal lags afterwards, whatever the sample says.  Same order as the comparator's (lag desc, id asc): no ties, thousands of
    short runs, a run that just fits a workgroup, runs that do not (the redo slots), negative lags, duplicate ids; ids already
    ascending and all-equal lags never go keys first."""
    os.environ["LA_SORT_KEYS_FIRST"] = "2"
    try:
        with N.Context(0) as c:
            w = _sort_topic(n, kind, n + len(kind), shuffled, dup)
            got = _device_call(c, w, flags=N.LA_FLAG_PROFILE)
            t = c.last_phase_times()
            assert (t.keys_first, t.redone) == (expect_first, expect_redo), (t.keys_first, t.redone)
            order = np.lexsort((w.partition_id, ~w.lag))
            np.testing.assert_array_equal(got[0], w.partition_id[order])
            assert (got[1] == -1).all()
            # with consumers: the greedy reads the repaired order
            w2 = _batch_of([(n, 37)], n, kinds=["ties"])
            _same3(_device_call(c, w2), round_form(w2.part_off, w2.partition_id, w2.lag, w2.cons_off, w2.cons_rank), "with consumers")
            # several topics side by side, each with its own decision
            w3 = _batch_of([(30000, 5), (50000, 0), (20000, 100), (65536, 3)], n + 1, kinds=["ties", "u40", "zero", "pareto"])
            _same3(_device_call(c, w3), round_form(w3.part_off, w3.partition_id, w3.lag, w3.cons_off, w3.cons_rank), "side by side")
    finally:
        os.environ.pop("LA_SORT_KEYS_FIRST", None)


@pytest.mark.parametrize("kind,expect_first", [("wide", 1), ("runs", 0), ("run20000_top", 0), ("equal", 0)])
def test_keys_first_sort_by_the_sample_at_five_million(ctx, kind, expect_first):
    """The default rule: from 4 M partitions on, with shuffled ids, keys first unless the sample of the lags shows a frequent one."""
    n = 5_000_000
    w = _sort_topic(n, kind, 77)
    got = _device_call(ctx, w, flags=N.LA_FLAG_PROFILE)
    t = ctx.last_phase_times()
    assert t.keys_first == expect_first and t.redone == 0, (t.keys_first, t.redone, t.id_passes, t.key_passes)
    order = np.lexsort((w.partition_id, ~w.lag))
    np.testing.assert_array_equal(got[0], w.partition_id[order])
    os.environ["LA_SORT_KEYS_FIRST"] = "0"                  # never: the round-3 order of passes, same answer
    try:
        got0 = _device_call(ctx, w, flags=N.LA_FLAG_PROFILE)
        assert ctx.last_phase_times().keys_first == 0
        np.testing.assert_array_equal(got0[0], got[0])
    finally:
        os.environ.pop("LA_SORT_KEYS_FIRST", None)


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
