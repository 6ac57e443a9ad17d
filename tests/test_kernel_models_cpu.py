"""numpy models of the three round-4 device algorithms, checked against plain sorts on the CPU: the invariants the kernels rest
on, pinned where no GPU is needed.

* moved_sort_bins (la_large.hip): a bin that is larger than every bin before it and smaller than every bin behind it separates
  the two sides, so sorting the OTHER bins among themselves and putting them back, in order, into the places they came from
  sorts the whole array -- also when more bins than necessary are handed in.
* block_sort_packed's skipped merges (la_block.hip): with sentinels in every slot at or beyond `live`, a bitonic merge whose
  upper half starts there changes nothing.
* block_sort_radix / rank_in_wave: ranks "old value of a returning counter, lanes served in lane order" -- and the variant
  where the lanes that share lane 0's digit take ONE counter step plus their place among themselves -- are the stable ranks
  of an LSD pass.
"""
import numpy as np
import pytest


def _moved_sort_model(v, extra=None):
    v = np.asarray(v, dtype=np.uint64)
    n = v.size
    pm = np.maximum.accumulate(np.concatenate(([np.uint64(0)], v[:-1])))                     # max of everything before
    sm = np.minimum.accumulate(np.concatenate((v[1:], [np.uint64(2**64 - 1)]))[::-1])[::-1]   # min of everything behind
    stays = (np.arange(n) == 0) | (pm < v)
    stays &= (np.arange(n) == n - 1) | (v < sm)
    moves = ~stays
    if extra is not None:                                   # hand in more than necessary (thread granularity did)
        moves |= extra
    out = v.copy()
    out[np.nonzero(moves)[0]] = np.sort(v[moves])           # places ascending, values ascending
    return out, int(moves.sum())


@pytest.mark.parametrize("kind", ["runs", "bulk", "pairs", "sorted", "reversed", "random"])
def test_moved_bins_model_sorts(kind):
    rng = np.random.default_rng(len(kind))
    n = 4096
    if kind == "runs":                                      # ascending totals + descending lags: a few ascending runs
        base = np.sort(rng.integers(0, 1 << 40, n))
        v = base + np.sort(rng.integers(0, 1 << 30, n))[::-1]
    elif kind == "bulk":                                    # far-apart bins and a dense bulk that reshuffles
        v = np.arange(n, dtype=np.int64) * (1 << 30)
        v[100:900] = (100 << 30) + rng.permutation(800)
    elif kind == "pairs":
        v = np.arange(n, dtype=np.int64) * 10
        for i in rng.choice(n - 1, 50, replace=False):
            v[i], v[i + 1] = v[i + 1], v[i]
    elif kind == "sorted":
        v = np.arange(n, dtype=np.int64) * 3
    elif kind == "reversed":
        v = np.arange(n, dtype=np.int64)[::-1].copy()
    else:
        v = rng.permutation(n).astype(np.int64)
    v = (v.astype(np.uint64) << np.uint64(13)) | np.arange(n, dtype=np.uint64) % np.uint64(8192)   # distinct, like packed bins
    got, moved = _moved_sort_model(v)
    assert np.array_equal(got, np.sort(v))
    if kind == "sorted":
        assert moved == 0
    if kind == "bulk":
        assert moved <= 800
    extra = rng.random(n) < 0.05                            # a superset of the bins that move is harmless
    got2, moved2 = _moved_sort_model(v, extra)
    assert np.array_equal(got2, np.sort(v)) and moved2 >= moved


def _bitonic_merge_blocks(a, K):
    """One merge level of the direction-free bitonic sorter: blocks of K slots, mirror step then i <-> i ^ j."""
    a = a.copy()
    n = a.size
    idx = np.arange(n)
    part = idx ^ (K - 1)
    lo = np.minimum(a, a[part])
    hi = np.maximum(a, a[part])
    a = np.where(idx < part, lo, hi)
    j = K >> 2
    while j >= 1:
        part = idx ^ j
        lo = np.minimum(a, a[part])
        hi = np.maximum(a, a[part])
        a = np.where((idx & j) == 0, lo, hi)
        j >>= 1
    return a


@pytest.mark.parametrize("n,live", [(1024, 1024), (1024, 640), (1024, 64), (4096, 2560), (4096, 2048), (2048, 1984)])
def test_bitonic_merges_over_sentinel_halves_change_nothing(n, live):
    """The sort as block_sort_packed runs it -- sorted spans of `span` slots, then merge levels, a level skipped for every
    block whose upper half starts at or beyond `live` -- equals the full network's result when slots >= live hold sentinels."""
    rng = np.random.default_rng(n + live)
    span = 64
    sentinel = np.uint64(2**64 - 1)
    a = rng.integers(0, 1 << 62, n).astype(np.uint64)
    a[live:] = sentinel
    a[rng.integers(0, live, live // 10)] = sentinel         # sentinels inside the live part too (a topic's tail)
    full = a.copy().reshape(-1, span)
    full.sort(axis=1)
    full = full.reshape(-1)
    skip = full.copy()
    K = 2 * span
    while K <= n:
        merged = _bitonic_merge_blocks(full, K)
        full = merged
        m2 = _bitonic_merge_blocks(skip, K)
        base = (np.arange(n) // K) * K
        take = base + K // 2 < live                          # blocks that are not skipped
        skip = np.where(take, m2, skip)
        K <<= 1
    assert np.array_equal(full, np.sort(a))
    assert np.array_equal(skip, full)


def _lsd_pass_ranks(digits, lanes=64, group=False):
    """Ranks of one wavefront-sized chunk at a time, counters carried across the chunks of one wave: what the returning atomics
    hand out when lanes are served in lane order; with `group`, the lanes sharing lane 0's digit take one step together."""
    counters = np.zeros(256, dtype=np.int64)
    ranks = np.empty(digits.size, dtype=np.int64)
    for s in range(0, digits.size, lanes):
        d = digits[s:s + lanes]
        r = np.empty(d.size, dtype=np.int64)
        done = np.zeros(d.size, dtype=bool)
        if group and (d == d[0]).sum() >= 8:
            g = d == d[0]
            r[g] = counters[d[0]] + np.arange(int(g.sum()))
            counters[d[0]] += int(g.sum())
            done = g
        for i in np.nonzero(~done)[0]:                       # lane order
            r[i] = counters[d[i]]
            counters[d[i]] += 1
        ranks[s:s + lanes] = r
    return ranks, counters


@pytest.mark.parametrize("group", [False, True])
@pytest.mark.parametrize("kind", ["uniform", "skewed", "constant"])
def test_lsd_passes_with_counter_ranks_are_a_stable_sort(kind, group):
    rng = np.random.default_rng(7 + (kind == "skewed"))
    n = 64 * 40
    if kind == "uniform":
        keys = rng.integers(0, 1 << 47, n).astype(np.uint64)
    elif kind == "skewed":                                   # most high digits equal, a few outliers
        keys = rng.zipf(1.1, n).astype(np.uint64) % np.uint64(1 << 47)
    else:
        keys = np.full(n, 12345, dtype=np.uint64)
    keys[-37:] = np.uint64(2**64 - 1)                         # sentinels: digit 255 in every pass
    tag = np.arange(n)                                       # to observe stability
    cur_k, cur_t = keys.copy(), tag.copy()
    for shift in range(0, 48, 8):
        d = ((cur_k >> np.uint64(shift)) & np.uint64(255)).astype(np.int64)
        ranks, totals = _lsd_pass_ranks(d, group=group)
        first = np.concatenate(([0], np.cumsum(totals)[:-1]))
        pos = first[d] + ranks
        assert np.array_equal(np.sort(pos), np.arange(n))    # a permutation
        nk, nt = np.empty_like(cur_k), np.empty_like(cur_t)
        nk[pos], nt[pos] = cur_k, cur_t
        cur_k, cur_t = nk, nt
    low = keys & np.uint64((1 << 48) - 1)
    order = np.argsort(low, kind="stable")
    assert np.array_equal(cur_t, tag[order])


# ---- round 5: the block path's greedy through 32-bit keys (la_block.hip, greedy_one_wave_key32) -----------------------------------
def _key32_greedy_model(lags_sorted_desc, C, n_c):
    """Numpy restatement of greedy_one_wave_key32: bins stay at home, one 32-bit key per bin
    ((total - floor) >> drop) << idx_bits | e with floor = the sum of the smallest lag of every full round so far; a round whose
    sorted keys hold two neighbours with the same truncated total (and drop > 0) is ordered again exactly.  Returns (winner per
    sorted partition, totals, number of exact re-orderings, largest key field seen)."""
    P = len(lags_sorted_desc)
    idx_bits = int(np.log2(n_c))                                # n_c: the power of two the bins' network spans
    lmax = int(lags_sorted_desc[0]) if P else 0
    lag_bits = lmax.bit_length()
    keep = 31 - idx_bits
    drop = max(0, lag_bits - keep)
    tot = [0] * C
    won = [0] * P
    redone, field_max = 0, 0
    keys = list(range(C))                                       # round 0: totals 0, positions ascending
    floor = 0
    for q in range((P + C - 1) // C):
        sk = sorted(keys)
        order = [k & ((1 << idx_bits) - 1) for k in sk]
        if q > 0 and drop > 0 and any((a >> idx_bits) == (b >> idx_bits) for a, b in zip(sk[:-1], sk[1:])):
            order = sorted(range(C), key=lambda e: (tot[e], e))
            redone += 1
        for s, e in enumerate(order):
            g = q * C + s
            if g < P:
                tot[e] += int(lags_sorted_desc[g])
                won[g] = e
        if (q + 1) * C - 1 < P:                                 # a full round: every bin took at least its smallest lag
            floor += int(lags_sorted_desc[(q + 1) * C - 1])
        keys = []
        for e in range(C):
            rel = tot[e] - floor
            assert 0 <= rel < (1 << max(lag_bits, 1)) or (lag_bits == 0 and rel == 0), (rel, lag_bits)
            field = rel >> drop
            assert 0 <= field < (1 << keep), (field, keep)
            field_max = max(field_max, field)
            keys.append((field << idx_bits) | e)
    return won, tot, redone, field_max


@pytest.mark.parametrize("P,C,kind,seed", [(10000, 128, "u40", 1), (3000, 100, "u40", 2), (5000, 256, "pareto", 3), (2000, 70, "zero", 4),
                                          (4000, 130, "ties", 5), (9000, 200, "u20", 6), (700, 256, "u40", 7), (100, 128, "u40", 8),
                                          (6000, 128, "u55", 9), (1000, 65, "small", 10)])
def test_key32_greedy_model_equals_the_literal_oracle(P, C, kind, seed):
    """The order the 32-bit keys give (with the exact re-ordering of tied rounds) is the reference's (total lag, memberId) order:
    same winners, same totals as the literal per-step min (oracle/lag_oracle.c) -- and the key field never leaves its bits, the
    bound the kernel's `drop` rests on (spread of the totals <= the largest lag under the round form)."""
    from oracle import oracle
    rng = np.random.default_rng(seed)
    if kind == "u40":
        lag = rng.integers(0, 1 << 40, P)
    elif kind == "u55":
        lag = rng.integers(0, 1 << 48, P)
    elif kind == "u20":
        lag = rng.integers(0, 1 << 20, P)
    elif kind == "small":
        lag = rng.integers(0, 50, P)
    elif kind == "zero":
        lag = np.zeros(P, np.int64)
    elif kind == "ties":
        lag = rng.integers(0, 7, P) * (1 << 30)
    else:
        lag = np.floor(np.minimum(float(1 << 40), 1000.0 * (1.0 - rng.random(P)) ** (-1.0 / 1.5))).astype(np.int64)
    lag = np.asarray(lag, np.int64)
    pid = rng.permutation(P).astype(np.int32)
    part_off, cons_off = np.array([0, P], np.int64), np.array([0, C], np.int64)
    ranks = np.arange(C, dtype=np.int32)
    e_pid, e_rank, e_tot = oracle.assign_flat(part_off, pid, lag, cons_off, ranks)
    lag_of = dict(zip(pid.tolist(), lag.tolist()))
    sorted_lags = np.array([lag_of[p] for p in e_pid.tolist()], np.int64)      # the oracle's own sort: (lag desc, id asc)
    n_c = 1 << max(0, (C - 1).bit_length())
    won, tot, redone, field_max = _key32_greedy_model(sorted_lags, C, n_c)
    np.testing.assert_array_equal(np.array(won, np.int32), e_rank)
    np.testing.assert_array_equal(np.array(tot, np.int64), e_tot)
    if kind in ("zero", "small"):
        assert redone == 0                                                      # drop == 0: the key is exact, ties included


def test_generated_sort_networks_sort_in_the_lane_level_simulator_and_are_current():
    """la_sort32_net.h is generated (tools/gen_sort32_net.py): the committed header is what the generator emits today, and the
    instruction streams -- DPP moves, v_permlane*_swap, v_med3_u32 with data-driven direction -- sort 64 / 128 / 256 keys in a
    lane-level simulator of exactly those instructions (random keys, heavy duplicates, all-ones sentinels, reversed input)."""
    import importlib.util
    import os
    import random
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_sort32_net", os.path.join(root, "tools", "gen_sort32_net.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    header = os.path.join(root, "kafka_lag_based_assignor_amd", "csrc", "la_sort32_net.h")
    before = open(header).read()
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "gen_sort32_net.py")], stdout=subprocess.DEVNULL)
    assert open(header).read() == before, "la_sort32_net.h is stale: run tools/gen_sort32_net.py"
    random.seed(7)
    for E in (1, 2, 4):
        ins = g.gen(E)
        n = 64 * E
        for trial in range(24):
            kind = trial % 4
            if kind == 0:
                keys = [random.getrandbits(32) for _ in range(n)]
            elif kind == 1:
                keys = [random.randrange(5) for _ in range(n)]
            elif kind == 2:
                live = random.randrange(n + 1)
                keys = [random.getrandbits(31) for _ in range(live)] + [0xFFFFFFFF] * (n - live)
                random.shuffle(keys)
            else:
                keys = list(range(n))[::-1]
            assert g.simulate(E, keys, ins) == sorted(keys), (E, trial)
        # the hazard rule the scheduler keeps: >= 2 issue slots between a VALU write of a register and a DPP / permlane read of it
        last = {}
        for i, text in enumerate(ins):
            ops = [t.rstrip(",") for t in text.split() if t.startswith("%[")]
            if text.startswith(("v_mov_b32_dpp", "v_permlane")):
                reads = ops[1:] if text.startswith("v_mov_b32_dpp") else ops
                for r in reads:
                    assert r not in last or i - last[r] - 1 + sum(
                        int(x.split()[1]) for x in ins[last[r] + 1:i] if x.startswith("s_nop")) >= 2, (E, i, text)
            if not text.startswith("s_nop"):
                for r in (ops[:2] if text.startswith("v_permlane") else ops[:1]):
                    last[r] = i

