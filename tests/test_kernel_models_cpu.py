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
