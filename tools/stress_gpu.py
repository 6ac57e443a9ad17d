#!/usr/bin/env python3
"""Long randomized parity sweep on the GPU (developer tool; the committed tests run a short version).
    [STRESS_SEED0=<first seed>] python tools/stress_gpu.py [tile] [large] [block] [sample-sort] [multi-shard] [round-4] [round-5 seeds]
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib.util
spec = importlib.util.spec_from_file_location("tgp", os.path.join(ROOT, "tests", "test_gpu_parity.py"))
tgp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tgp)
from kafka_lag_based_assignor_amd import _native as N

S0 = int(os.environ.get("STRESS_SEED0", "100"))      # first seed of every section (fresh seeds: STRESS_SEED0=5000 ...)


def main():
    nt = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    nl = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 300
    ctx = N.Context(0)
    bad = 0
    for seed in range(S0, S0 + nt):
        try:
            tgp.test_fuzz_tile_batches.__wrapped__(ctx, seed) if hasattr(tgp.test_fuzz_tile_batches, "__wrapped__") else tgp.test_fuzz_tile_batches(ctx, seed)
        except AssertionError as e:
            bad += 1; print("TILE seed", seed, "FAILED:", str(e)[:200])
    # full tiles (every topic P == lanes x records of some tile shape): the tile kernel's form without clamps / sentinels,
    # alone and mixed with topics one partition short, every kind of ids and lags, lags in and offsets in
    nf = 0
    for seed in range(S0, S0 + nt):
        rng = np.random.default_rng(seed)
        p = int(rng.choice([8, 16, 32, 64, 128, 256, 512, 1024]))
        c = int(rng.integers(1, min(64, p) + 1))
        t = int(rng.integers(1, 120))
        kind = str(rng.choice(["u40", "mixed", "sparse", "dup", "ties", "zero", "tiny", "u63", "negative"]))
        w = tgp._full_tile_batch(seed, t, p, c, kind)
        try:
            tgp._check_lags(ctx, w, "full %d x %d x %d %s" % (t, p, c, kind))
            if kind in ("u40", "mixed", "ties", "zero", "tiny"):
                w.lag = None
                tgp._check_offsets(ctx, w, N.LA_RESET_LATEST if seed & 1 else N.LA_RESET_EARLIEST, "full, offsets")
            nf += 1
        except AssertionError as e:
            bad += 1; print("FULL seed", seed, "FAILED:", str(e)[:200])
    print("full-tile batches checked:", nf)
    for seed in range(S0, S0 + nl):
        try:
            tgp.test_fuzz_large_topics(ctx, seed)
        except AssertionError as e:
            bad += 1; print("LARGE seed", seed, "FAILED:", str(e)[:200])
    for seed in range(S0, S0 + nb):
        try:
            tgp.test_fuzz_block_topics(ctx, seed)
        except AssertionError as e:
            bad += 1; print("BLOCK seed", seed, "FAILED:", str(e)[:200])
    for seed in range(S0, S0 + nb // 10):
        rng = np.random.default_rng(seed)
        try:
            tgp.test_block_batches_mixed_with_tile_and_large_topics(ctx, seed, int(rng.choice([600, 1500, 3000, 9000])),
                                                                    int(rng.choice([70, 100, 300, 2500])))
            tgp.test_ragged_tile_batch_by_shape_class(ctx, int(rng.integers(300, 9000)), int(rng.choice([2, 3, 50])))
        except AssertionError as e:
            bad += 1; print("MIXED seed", seed, "FAILED:", str(e)[:200])
    # large path with more than 1 024 consumers: sample-sorted greedy rounds, full network, both interleaved
    ns = int(sys.argv[4]) if len(sys.argv) > 4 else 60
    for seed in range(S0, S0 + ns):
        rng = np.random.default_rng(seed)
        c = int(rng.integers(1025, 8193))
        p = int(rng.integers(c, min(40 * c, int(4e8) // c) + 1))
        kind = str(rng.choice(["u40", "ties", "zero", "u63", "pareto"]))
        try:
            tgp.test_large_sample_sort_rounds(ctx, p, c, kind)
        except AssertionError as e:
            bad += 1; print("SAMPLE seed", seed, "p", p, "c", c, kind, "FAILED:", str(e)[:200])
    # the host-buffer pipeline: random ragged batches through shards x lanes x chunks (4 logical shards on device 0,
    # LA_CREATE_SPLIT_ALWAYS), results straight into the caller's arrays, and the grouped lists across shards
    nm = int(sys.argv[5]) if len(sys.argv) > 5 else 60
    from kafka_lag_based_assignor_amd import synth
    from oracle import oracle
    multi = [N.Context([0, 0, 0, 0], flags=N.LA_CREATE_SPLIT_ALWAYS), N.Context([0, 0], flags=N.LA_CREATE_SPLIT_ALWAYS | 2),
             N.Context(0, flags=N.LA_CREATE_SPLIT_ALWAYS | 3)]
    for seed in range(S0, S0 + nm):
        rng = np.random.default_rng(seed)
        w = synth.ragged(seed, int(rng.integers(1, 400)), int(rng.choice([5, 60, 300, 1500])), int(rng.choice([1, 8, 40, 90])),
                         negative=bool(rng.integers(0, 2)))
        exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
        n_members = int(w.cons_rank.max()) + 1 if w.cons_rank.size else 0
        want = ctx.group_by_member(w.part_off, exp[0], exp[1], n_members)
        for c in multi:
            try:
                got = c.assign_batch_lags(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
                assert all(np.array_equal(g, e) for g, e in zip(got, exp)), "assignment"
                c.assign_batch_lags(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank, keep_on_device=True)
                gl = c.group_last_by_member(w.n_partitions, n_members)
                assert all(np.array_equal(g, e) for g, e in zip(gl, want)), "grouped lists"
                # the same through pinned arrays: the three-stream form (no worker threads)
                def pin(a):
                    a = np.ascontiguousarray(a)
                    o = c.host_alloc(a.shape, a.dtype)
                    o[...] = a
                    return o
                pw = [pin(x) for x in (w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)]
                po = (c.host_alloc((w.n_partitions,), np.int32), c.host_alloc((w.n_partitions,), np.int32),
                      c.host_alloc((w.cons_rank.size,), np.int64))
                got = c.assign_batch_lags(*pw, out=po)
                assert all(np.array_equal(g, e) for g, e in zip(got, exp)), "assignment, pinned arrays"
                assert w.n_partitions == 0 or c.last_pipeline() in (N.LA_PIPELINE_STREAMS, N.LA_PIPELINE_MAPPED), "pipeline %d" % c.last_pipeline()
            except (AssertionError, N.LagAssignError) as e:
                bad += 1; print("MULTI seed", seed, "shards", c.shard_count, "FAILED:", str(e)[:200])
    for c in multi:
        c.close()
    # round 4: random tie structures through the keys-first sort (forced on small topics: LA_SORT_KEYS_FIRST=2) -- runs of every
    # length around the scan's four, the repair's window (4 096) and capacity (8 192), at every offset; several large topics side
    # by side; topics beyond 8 192 consumers; the sparse-begin entry, dense and sparse, pageable and pinned (in place / three streams)
    n4 = int(sys.argv[6]) if len(sys.argv) > 6 else 40
    from oracle.round_form import round_form
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    spec4 = importlib.util.spec_from_file_location("tr4", os.path.join(ROOT, "tests", "gpu_helpers.py"))     # (the tests' helpers)
    tr4 = importlib.util.module_from_spec(spec4); spec4.loader.exec_module(tr4)
    os.environ["LA_SORT_KEYS_FIRST"] = "2"
    try:
        for seed in range(S0, S0 + n4):
            rng = np.random.default_rng(seed)
            n = int(rng.integers(17000, 90000))
            lens = []
            while sum(lens) < n:
                kind = rng.integers(0, 6)
                lens.append(int({0: 1, 1: rng.integers(2, 6), 2: rng.integers(5, 70), 3: rng.integers(3000, 4200),
                                 4: rng.integers(8000, 8400), 5: rng.integers(1, 3)}[int(kind)]) if rng.random() < 0.9 else int(rng.integers(1, 12000)))
            lag = np.repeat(rng.permutation(len(lens)).astype(np.int64) * 3 + int(rng.integers(0, 1 << 30)), lens)[:n]
            lag = lag[rng.permutation(n)] if rng.random() < 0.7 else lag
            pid = rng.permutation(n).astype(np.int32)
            c_ = int(rng.choice([0, 3, 100]))
            w = synth.Workload("ties", 1, np.array([0, n], np.int64), pid, np.zeros(n, np.int64), lag.copy(), np.zeros(n, np.int64), lag,
                               np.array([0, c_], np.int64), np.arange(c_, dtype=np.int32), n, c_)
            try:
                got = tr4._device_call(ctx, w)
                exp = round_form(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
                assert all(np.array_equal(g, e) for g, e in zip(got, exp)), "keys-first order"
            except (AssertionError, N.LagAssignError) as e:
                bad += 1; print("TIES seed", seed, "n", n, "FAILED:", str(e)[:200])
    finally:
        os.environ.pop("LA_SORT_KEYS_FIRST", None)
    # round 5: (a) the block path's 65 .. 256-consumer greedy through 32-bit keys (every kind of lags, ids dense or not); (b) host
    # calls of every size up to the lanes through la_assign_batch_grouped on pageable and on pinned arrays, hinted or not:
    # zero-copy staging with the fused end, the two-launch grouping, mapped arrays read in place -- against the oracle + a stable sort
    n5 = int(sys.argv[7]) if len(sys.argv) > 7 else 40
    tr5 = tr4
    for seed in range(S0, S0 + n5):
        rng = np.random.default_rng(seed)
        C = int(rng.integers(65, 257))
        P = int(rng.integers(C, 16385))
        kind = str(rng.choice(["u40", "bigties", "nearties", "zero", "u20", "pareto", "u55"]))
        lag = {"u40": lambda: rng.integers(0, 1 << 40, P), "bigties": lambda: (1 << 39) + rng.integers(0, 3, P) * (1 << 20),
               "nearties": lambda: (1 << 41) + rng.integers(0, 64, P), "zero": lambda: np.zeros(P, np.int64),
               "u20": lambda: rng.integers(0, 1 << 20, P), "u55": lambda: rng.integers(0, 1 << 55, P),
               "pareto": lambda: np.floor(np.minimum(float(1 << 40), 1000.0 * (1.0 - rng.random(P)) ** (-1.0 / 1.5))).astype(np.int64)}[kind]()
        w = tr5._one_topic(P, C, lag, seed)
        if rng.random() < 0.4:
            w.partition_id = (w.partition_id.astype(np.int64) * 3 + int(rng.integers(0, 1000))).astype(np.int32)   # not 0 .. P-1
        try:
            exp = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
            got = tr4._device_call(ctx, w)
            assert all(np.array_equal(g, e) for g, e in zip(got, exp)), "block greedy"
        except (AssertionError, N.LagAssignError) as e:
            bad += 1; print("KEY32 seed", seed, P, C, kind, "FAILED:", str(e)[:200])
    for seed in range(S0, S0 + n5):
        rng = np.random.default_rng(seed + 77)
        shape = int(rng.integers(0, 5))
        t, mp, mc = [(int(rng.integers(1, 60)), 300, 40), (int(rng.integers(50, 2000)), 64, 8), (int(rng.integers(1, 12)), 9000, 300),
                     (int(rng.integers(200, 1500)), 400, 64), (int(rng.integers(1, 4)), 60000, 90)][shape]
        w = synth.ragged(seed, t, mp, mc, dist=str(rng.choice(["mixed", "u40", "small"])))
        if w.n_partitions == 0:
            continue
        w.begin = np.zeros_like(w.lag); w.committed = np.zeros_like(w.lag); w.end = w.lag.copy()
        n_members = (int(w.cons_rank.max()) + 1 if w.cons_rank.size else 0) + int(rng.integers(0, 3))
        a = (w.part_off, w.partition_id, w.begin, w.end, w.committed, N.LA_RESET_EARLIEST, w.cons_off, w.cons_rank)
        try:
            e_pid, e_rank, e_tot = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
            order = np.argsort(e_rank, kind="stable")
            first = np.searchsorted(e_rank[order], np.arange(n_members + 1))
            topic = (np.searchsorted(w.part_off, order, side="right") - 1).astype(np.int32)
            hb = N.offset_bounds(w.begin, w.end, w.committed, w.partition_id)
            for pinned in (False, True):
                args = tr5._pinned_copy(ctx, a) if pinned else a
                if hb is not None and rng.random() < 0.5:
                    ctx.hint_next_call(hb)
                g_off, g_t, g_p, g_tot = ctx.assign_batch_grouped(*args, n_members, want_totals=not pinned)
                assert np.array_equal(g_off, first) and np.array_equal(g_p, e_pid[order]) and np.array_equal(g_t, topic), \
                    "lists (pipeline %d, pinned %s)" % (ctx.last_pipeline(), pinned)
                assert pinned or np.array_equal(g_tot, e_tot), "totals"
                got = ctx.assign_batch(*a)
                assert all(np.array_equal(g, e) for g, e in zip(got, (e_pid, e_rank, e_tot))), "assignment (pipeline %d)" % ctx.last_pipeline()
        except (AssertionError, N.LagAssignError) as e:
            bad += 1; print("HOST seed", seed, "shape", shape, w.n_topics, w.n_partitions, "FAILED:", str(e)[:200])
    for seed in range(S0, S0 + n4 // 2):
        rng = np.random.default_rng(seed)
        shapes = [(int(rng.integers(16385, 120000)), int(rng.choice([0, 1, 50, 1500, 5000, 8192]))) for _ in range(int(rng.integers(2, 7)))]
        shapes += [(int(rng.integers(1, 3000)), int(rng.integers(0, 200))) for _ in range(int(rng.integers(0, 5)))]
        if seed % 4 == 0:
            shapes.append((int(rng.integers(9000, 40000)), int(rng.integers(8193, 12000))))
        order = rng.permutation(len(shapes))
        w = tr4._batch_of([shapes[i] for i in order], seed, negative=bool(rng.integers(0, 2)))
        try:
            exp = round_form(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
            got = tr4._device_call(ctx, w)
            assert all(np.array_equal(g, e) for g, e in zip(got, exp)), "side by side"
            got = ctx.assign_batch_lags(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
            assert all(np.array_equal(g, e) for g, e in zip(got, exp)), "host entry"
        except (AssertionError, N.LagAssignError) as e:
            bad += 1; print("MANY seed", seed, shapes, "FAILED:", str(e)[:200])
    for seed in range(S0, S0 + n4):
        rng = np.random.default_rng(seed)
        w = tr4._workload(seed, float(rng.choice([0.0, 0.01, 0.5, 1.0])), topics=int(rng.integers(1, 500)), big=bool(rng.integers(0, 2)))
        idx, val = N.sparse_begin(w.begin, w.committed)
        exp = tr4._expected(w, False)
        try:
            got = ctx.assign_batch_sparse(w.part_off, w.partition_id, w.end, w.committed, N.LA_RESET_EARLIEST, idx, val, w.cons_off, w.cons_rank)
            assert all(np.array_equal(g, e) for g, e in zip(got, exp)), "sparse, pageable (pipeline %d)" % ctx.last_pipeline()
            pin = lambda a: tr4._pinned(ctx, a)
            out = (ctx.host_alloc((w.n_partitions,), np.int32), ctx.host_alloc((w.n_partitions,), np.int32), ctx.host_alloc((w.cons_rank.size,), np.int64))
            for env in (None, "1"):
                if env: os.environ["LA_NO_MAPPED_PIPELINE"] = env
                try:
                    got = ctx.assign_batch_sparse(pin(w.part_off), pin(w.partition_id), pin(w.end), pin(w.committed), N.LA_RESET_EARLIEST,
                                                  pin(idx), pin(val), pin(w.cons_off), pin(w.cons_rank), out=out)
                    assert all(np.array_equal(g, e) for g, e in zip(got, exp)), "sparse, pinned (pipeline %d)" % ctx.last_pipeline()
                    got = ctx.assign_batch(pin(w.part_off), pin(w.partition_id), pin(w.begin), pin(w.end), pin(w.committed), N.LA_RESET_EARLIEST,
                                           pin(w.cons_off), pin(w.cons_rank), out=out)
                    assert all(np.array_equal(g, e) for g, e in zip(got, exp)), "dense, pinned (pipeline %d)" % ctx.last_pipeline()
                finally:
                    os.environ.pop("LA_NO_MAPPED_PIPELINE", None)
        except (AssertionError, N.LagAssignError) as e:
            bad += 1; print("SPARSE seed", seed, "FAILED:", str(e)[:200])
    print("stress done (first seed %d): %d tile + %d large + %d block + %d sample-sort + %d multi-shard + %d round-4 + %d round-5 seeds, %d failures"
          % (S0, nt, nl, nb, ns, nm, n4, n5, bad))
    ctx.close()

if __name__ == "__main__":
    main()
