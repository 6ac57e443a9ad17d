"""Synthetic workloads of BASELINE.json / SURVEY.md section 8(d), as SoA arrays.

One generator feeds the HIP path, the oracle and the bench, so no two implementations of
a PRNG or of pow() can diverge.  PRNG: counter-based SplitMix64,
seed = 0x9E3779B97F4A7C15 ^ config_id.  Partition ids are 0..P-1 SHUFFLED per topic (input
order must never be relied on), member ranks are 0..C-1 per topic (the host has already
ranked "consumer-<i>" ids with String.compareTo), offsets are built from the drawn lag:
begin = 0, committed = c uniform on [0, 2^20), end = c + lag, and 1 % of partitions have no
committed offset (-1) to exercise the auto.offset.reset fallback.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M64 = (1 << 64) - 1


def splitmix64(seed: int, n: int, stream: int = 0) -> np.ndarray:
    """n SplitMix64 outputs of the generator seeded with `seed` (stream = independent offset)."""
    with np.errstate(over="ignore"):
        base = np.uint64((seed + stream * 0xD1342543DE82EF95) & _M64)
        z = base + _GOLDEN * (np.arange(1, n + 1, dtype=np.uint64))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


@dataclass
class Workload:
    name: str
    n_topics: int
    part_off: np.ndarray        # int64 [T+1]
    partition_id: np.ndarray    # int32 [N]
    begin: np.ndarray           # int64 [N]
    end: np.ndarray             # int64 [N]
    committed: np.ndarray       # int64 [N]  (-1 = none)
    lag: np.ndarray             # int64 [N]  the drawn lag (what LATEST-mode offsets encode)
    cons_off: np.ndarray        # int64 [T+1]
    cons_rank: np.ndarray       # int32 [K]
    max_partitions: int
    max_consumers: int

    @property
    def n_partitions(self) -> int:
        return int(self.part_off[-1])


def zipf_lags(p: int) -> np.ndarray:
    """lag of popularity rank k = floor(1e9 * k^-1.1), k = 1..P (cfg3 / target)."""
    k = np.arange(1, p + 1, dtype=np.float64)
    return np.floor(1e9 * k ** -1.1).astype(np.int64)


def _draw_lags(dist: str, seed: int, t: int, p: int) -> np.ndarray:
    n = t * p
    if dist == "uniform63":
        return (splitmix64(seed, n, 1) >> np.uint64(1)).astype(np.int64)
    if dist == "uniform40":
        return (splitmix64(seed, n, 1) >> np.uint64(24)).astype(np.int64)
    if dist == "zipf":
        base = zipf_lags(p)
        order = np.argsort(splitmix64(seed, n, 1).reshape(t, p), axis=1, kind="stable")
        return base[order].reshape(-1)
    if dist == "pareto":
        u = ((splitmix64(seed, n, 1) >> np.uint64(11)).astype(np.float64) + 1.0) / float(1 << 53)   # (0,1]
        return np.floor(np.minimum(float(1 << 40), 1000.0 * u ** (-1.0 / 1.5))).astype(np.int64)
    if dist == "zero":
        return np.zeros(n, dtype=np.int64)
    raise ValueError(dist)


def make_uniform(name: str, config_id: int, n_topics: int, partitions: int, consumers: int, dist: str,
                 offsets: bool = True, none_frac: Optional[float] = None) -> Workload:
    """T topics, each with `partitions` partitions and `consumers` consumers.  `none_frac`: the share of partitions
    without a committed offset (default: the fixed ~1 % of SURVEY 8d; 1.0 = a brand-new consumer group, where
    computePartitionLag falls back for EVERY partition, Main.java:393-396).  With a fall-back to `earliest` such a
    partition's lag is end - begin, so `begin` is drawn too (uniform on [0, 2^20), end = begin + lag): the whole
    36 B/partition of SURVEY 8d then really moves and the lags stay the drawn ones."""
    seed = 0x9E3779B97F4A7C15 ^ config_id
    t, p, c = n_topics, partitions, consumers
    n = t * p
    lag = _draw_lags(dist, seed, t, p)
    pid = np.argsort(splitmix64(seed, n, 2).reshape(t, p), axis=1, kind="stable").astype(np.int32).reshape(-1)
    if offsets:
        r = splitmix64(seed, n, 3)
        com = (r >> np.uint64(44)).astype(np.int64)                    # [0, 2^20)
        end = com + lag                                                # wraps only for uniform63
        if none_frac is None:
            none = (r & np.uint64(0xFFFF)) < np.uint64(655)            # ~1 %
            begin = np.zeros(n, dtype=np.int64)
        else:
            none = (r & np.uint64(0xFFFF)) < np.uint64(int(round(min(max(none_frac, 0.0), 1.0) * 65536)))
            # where there is no committed offset the drawn lag is end - begin: begin = what committed would have been
            begin = np.where(none, com, np.int64(0))
        com = np.where(none, np.int64(-1), com)
    else:
        com = np.zeros(n, dtype=np.int64)
        end = lag.copy()
        begin = np.zeros(n, dtype=np.int64)
    return Workload(
        name=name, n_topics=t,
        part_off=np.arange(t + 1, dtype=np.int64) * p,
        partition_id=pid, begin=begin, end=end, committed=com, lag=lag,
        cons_off=np.arange(t + 1, dtype=np.int64) * c,
        cons_rank=np.tile(np.arange(c, dtype=np.int32), t),
        max_partitions=p, max_consumers=c)


# The BASELINE.json configs.  `scale` shrinks the topic count
# (or, for single-topic configs, the partition count) for CPU-sized parity runs.
def config(name: str, scale: float = 1.0, none_frac: Optional[float] = None) -> Workload:
    def s(x: int) -> int:
        return max(1, int(round(x * scale)))
    if none_frac is not None:
        shape = {"cfg3": (3, 1000, 256, 32, "zipf"), "cfg4": (4, 100000, 64, 8, "uniform40"),
                 "target": (6, 100000, 256, 32, "zipf")}[name]
        return make_uniform(name, shape[0], s(shape[1]), shape[2], shape[3], shape[4], none_frac=none_frac)
    if name == "cfg1":      # README example, README.md:42-52
        w = make_uniform("cfg1", 1, 1, 3, 2, "zero", offsets=False)
        w.partition_id = np.array([0, 1, 2], dtype=np.int32)
        w.lag = np.array([100000, 50000, 60000], dtype=np.int64)
        w.end = w.lag.copy()
        return w
    if name == "cfg2a":
        return make_uniform("cfg2a", 2, 1, s(10000), 128, "uniform63", offsets=False)
    if name == "cfg2b":
        return make_uniform("cfg2b", 2, 1, s(10000), 128, "uniform40")
    if name == "cfg3":
        return make_uniform("cfg3", 3, s(1000), 256, 32, "zipf")
    if name == "cfg4":
        return make_uniform("cfg4", 4, s(100000), 64, 8, "uniform40")
    if name == "cfg5":
        return make_uniform("cfg5", 5, 1, s(1048576), 8192, "pareto")
    if name == "target":
        return make_uniform("target", 6, s(100000), 256, 32, "zipf")
    # not in BASELINE.json: shapes between a wave tile and the large path (the block path of DESIGN.md 4.2b)
    if name == "block_a":
        return make_uniform("block_a", 7, s(200), 2000, 100, "uniform40")
    if name == "block_b":
        return make_uniform("block_b", 8, s(500), 200, 100, "zipf")
    if name == "block_c":
        return make_uniform("block_c", 9, s(20), 8000, 700, "pareto")
    raise ValueError(name)


def lag_ratio(out_total: np.ndarray, cons_off: np.ndarray) -> np.ndarray:
    """max/min per-consumer total lag per topic (min clamped to 1), BASELINE.md section 2."""
    t = cons_off.size - 1
    ratios = np.empty(t, dtype=np.float64)
    uniform = t > 0 and np.all(np.diff(cons_off) == cons_off[1] - cons_off[0]) and cons_off[1] > 0
    if uniform:
        m = out_total.reshape(t, -1).astype(np.float64)
        return m.max(axis=1) / np.maximum(m.min(axis=1), 1.0)
    for i in range(t):
        seg = out_total[cons_off[i]:cons_off[i + 1]].astype(np.float64)
        ratios[i] = seg.max() / max(seg.min(), 1.0) if seg.size else 1.0
    return ratios


def ragged(seed: int, n_topics: int, max_partitions: int, max_consumers: int, dist: str = "mixed",
           negative: bool = False) -> Workload:
    """Ragged batch for parity tests: every topic its own P in [0, max], C in [0, max]."""
    rng = np.random.default_rng(seed)
    ps = rng.integers(0, max_partitions + 1, n_topics)
    cs = rng.integers(0, max_consumers + 1, n_topics)
    part_off = np.concatenate([[0], np.cumsum(ps)]).astype(np.int64)
    cons_off = np.concatenate([[0], np.cumsum(cs)]).astype(np.int64)
    n = int(part_off[-1])
    pid = np.empty(n, dtype=np.int32)
    lag = np.empty(n, dtype=np.int64)
    for t in range(n_topics):
        p = int(ps[t])
        ids = rng.permutation(p).astype(np.int32)
        if rng.random() < 0.2 and p:
            ids = (ids * 7 + rng.integers(-5, 100000)).astype(np.int32)      # sparse / negative-ish ids
        pid[part_off[t]:part_off[t + 1]] = ids
        kind = dist if dist != "mixed" else rng.choice(["zero", "ties", "small", "u40", "u63", "full"])
        if kind == "zero":
            l = np.zeros(p, dtype=np.int64)
        elif kind == "ties":
            l = rng.integers(0, 4, p).astype(np.int64) * 1000
        elif kind == "small":
            l = rng.integers(0, 1 << 20, p).astype(np.int64)
        elif kind == "u40":
            l = rng.integers(0, 1 << 40, p).astype(np.int64)
        elif kind == "u63":
            l = rng.integers(0, (1 << 63) - 1, p).astype(np.int64)
        else:
            l = rng.integers(-(1 << 63), (1 << 63) - 1, p).astype(np.int64)
        if not negative:
            l = np.where(l < 0, ~l, l)
        lag[part_off[t]:part_off[t + 1]] = l
    # member ranks: ascending subset of a global rank space
    ranks = []
    for t in range(n_topics):
        ranks.append(np.sort(rng.choice(max(max_consumers * 3, 1), int(cs[t]), replace=False)).astype(np.int32))
    cons_rank = np.concatenate(ranks) if ranks else np.empty(0, dtype=np.int32)
    com = rng.integers(0, 1 << 20, n).astype(np.int64)
    with np.errstate(over="ignore"):
        end = com + np.maximum(lag, 0)
    none = rng.random(n) < 0.05
    com = np.where(none, np.int64(-1), com)
    begin = rng.integers(0, 1 << 10, n).astype(np.int64)
    return Workload("ragged", n_topics, part_off, pid, begin, end, com, lag, cons_off,
                    cons_rank.astype(np.int32), int(ps.max()) if n_topics else 0,
                    int(cs.max()) if n_topics else 0)
