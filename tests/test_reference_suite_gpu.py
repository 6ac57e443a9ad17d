"""The reference's own seven JUnit tests (LagBasedPartitionAssignorTest.java:21-228), restated
against the host mirror.  Every number below is computed on the MI355X behind the C ABI."""
import random

import pytest

from kafka_lag_based_assignor_amd import (LagBasedPartitionAssignor, OffsetAndMetadata, TopicPartition,
                                          TopicPartitionLag)
from oracle import oracle

pytestmark = pytest.mark.gpu


def test_compute_partition_lag():                                     # Test.java:21-33
    lag = LagBasedPartitionAssignor.compute_partition_lag(OffsetAndMetadata(5555), 1111, 9999, "none")
    assert lag == 4444


def test_compute_partition_lag_no_end_offset():                       # Test.java:38-50
    assert LagBasedPartitionAssignor.compute_partition_lag(OffsetAndMetadata(5555), 0, 0, "none") == 0


def test_compute_partition_lag_no_committed_offset_reset_mode_latest():   # Test.java:52-64
    assert LagBasedPartitionAssignor.compute_partition_lag(None, 1111, 9999, "latest") == 0


def test_compute_partition_lag_no_committed_offset_reset_mode_earliest():  # Test.java:66-80
    begin, end = 1111, 9999
    assert LagBasedPartitionAssignor.compute_partition_lag(None, begin, end, "earliest") == end - begin


def test_assign():                                                    # Test.java:82-132
    partition_lag_per_topic = {
        "topic1": [TopicPartitionLag("topic1", 0, 100000), TopicPartitionLag("topic1", 1, 100000),
                   TopicPartitionLag("topic1", 2, 500), TopicPartitionLag("topic1", 3, 1)],
        "topic2": [TopicPartitionLag("topic2", 0, 900000), TopicPartitionLag("topic2", 1, 100000)],
    }
    subscriptions = {"consumer-1": ["topic1", "topic2"], "consumer-2": ["topic1"]}
    expected = {
        "consumer-1": [TopicPartition("topic1", 0), TopicPartition("topic1", 2),
                       TopicPartition("topic2", 0), TopicPartition("topic2", 1)],
        "consumer-2": [TopicPartition("topic1", 1), TopicPartition("topic1", 3)],
    }
    assert LagBasedPartitionAssignor.assign_lags(partition_lag_per_topic, subscriptions) == expected


def test_assign_with_zero_lags():                                     # Test.java:134-175
    lags = {"topic1": [TopicPartitionLag("topic1", p, 0) for p in range(7)]}
    subs = {"consumer-1": ["topic1"], "consumer-2": ["topic1"]}
    actual = LagBasedPartitionAssignor.assign_lags(lags, subs)
    sizes = [len(v) for v in actual.values()]
    assert max(sizes) <= min(sizes) + 1


def test_assign_with_heavily_skewed_lags():                           # Test.java:177-228
    values = [360, 359, 230, 118, 444, 122, 65, 111, 455000, 424000]
    lags = {"topic1": [TopicPartitionLag("topic1", p, v) for p, v in enumerate(values)]}
    subs = {"consumer-%d" % i: ["topic1"] for i in (1, 2, 3)}
    actual = LagBasedPartitionAssignor.assign_lags(lags, subs)
    sizes = [len(v) for v in actual.values()]
    assert max(sizes) <= min(sizes) + 1


# ---- beyond the reference's tests: same API, checked against the oracle's container model ----
def test_readme_example():                                            # README.md:42-57
    lags = {"t0": [TopicPartitionLag("t0", 0, 100000), TopicPartitionLag("t0", 1, 50000),
                   TopicPartitionLag("t0", 2, 60000)]}
    got = LagBasedPartitionAssignor.assign_lags(lags, {"C0": ["t0"], "C1": ["t0"]})
    assert got == {"C0": [("t0", 0)], "C1": [("t0", 2), ("t0", 1)]}


def test_static_assign_fuzz_list_order_matches_oracle():
    rng = random.Random(11)
    for trial in range(25):
        topics = ["topic-%d" % i for i in range(rng.randint(1, 40))]
        members = ["consumer-%d" % i for i in range(rng.randint(1, 15))]
        rng.shuffle(members)
        lags = {}
        for t in topics:
            if rng.random() < 0.9:
                ids = list(range(rng.randint(0, 50)))
                rng.shuffle(ids)
                lags[t] = [TopicPartitionLag(t, p, rng.choice([0, 0, 5, rng.randint(0, 1 << 40)])) for p in ids]
        subs = {m: rng.sample(topics + ["ghost"], rng.randint(0, len(topics))) for m in members}
        if rng.random() < 0.3 and subs[members[0]]:
            subs[members[0]] = subs[members[0]] + [subs[members[0]][0]]          # duplicate topic
        got = LagBasedPartitionAssignor.assign_lags(lags, subs)
        exp = oracle.assign_named({t: [tuple(e) for e in v] for t, v in lags.items()}, subs)
        assert got == exp, trial


class FakeOffsets:
    """Plays the side KafkaConsumer: one call per kind for ALL partitions."""

    def __init__(self, begin, end, committed):
        self.begin, self.end, self.com = begin, end, committed
        self.calls = []

    def beginning_offsets(self, tps):
        self.calls.append(("begin", len(tps)))
        return {tp: self.begin[tp] for tp in tps if tp in self.begin}

    def end_offsets(self, tps):
        self.calls.append(("end", len(tps)))
        return {tp: self.end[tp] for tp in tps if tp in self.end}

    def committed(self, tps):
        self.calls.append(("committed", len(tps)))
        return {tp: self.com.get(tp) for tp in tps}


@pytest.mark.parametrize("mode", ["latest", "earliest", None, "none"])
def test_plugin_level_assign_with_offsets(mode):
    rng = random.Random(5)
    metadata = {"orders": list(range(12)), "payments": list(range(5)), "empty": []}
    begin, end, com = {}, {}, {}
    for t, ps in metadata.items():
        for p in ps:
            b = rng.randint(0, 100)
            e = b + rng.randint(0, 10000)
            begin[(t, p)], end[(t, p)] = b, e
            if rng.random() < 0.7:
                com[(t, p)] = rng.randint(b, e)
    del end[("orders", 3)]                                             # failed lookup -> 0 (Main.java:351)
    subs = {"app-2": ["orders", "payments", "nometa"], "app-10": ["orders"], "app-1": ["payments", "empty"]}
    a = LagBasedPartitionAssignor()
    cfg = {"group.id": "g"}
    if mode is not None:
        cfg["auto.offset.reset"] = mode
    a.configure(cfg)
    warnings = []
    a.set_warn(warnings.append)
    src = FakeOffsets(begin, end, com)
    got = a.assign(metadata, subs, src)
    assert sorted(c[0] for c in src.calls) == ["begin", "committed", "end"]     # batched: 3 calls total
    assert len(warnings) == 2 and all("no metadata" in w for w in warnings)     # "nometa" and "empty"

    # expectation through the oracle: lag per partition, then the container model.  The plugin
    # level walks members in HashMap order (Main.java:141-146).
    eff_mode = "latest" if mode is None else mode
    lags = {}
    for t, ps in metadata.items():
        if ps:
            lags[t] = [(t, p, oracle.compute_partition_lag(com.get((t, p)), begin.get((t, p), 0),
                                                           end.get((t, p), 0), eff_mode)) for p in ps]
    from oracle.java_collections import JavaHashMap
    hm = JavaHashMap()
    for m, ts in subs.items():
        hm.put(m, ts)
    exp = oracle.assign_named(lags, dict(hm.items()))
    assert got == exp
    totals = a.last_topic_totals()
    for t in ("orders", "payments"):
        assigned = {m: sum(l for (tt, p, l) in lags[t] if (tt, p) in got[m]) for m in totals[t]}
        assert totals[t] == assigned
    # round 5: the marshalling loop vouches for what it saw -- the largest end offset (a failed lookup counts as 0) and the
    # largest partition id -- through la_hint_next_call; a rebalance this small is one launch for the tiles + one for the lists
    st = LagBasedPartitionAssignor.last_native_call()
    assert st["hinted"] and st["max_partition_id"] == 11
    assert st["max_lag"] == max(end.get((t, p), 0) for t, ps in metadata.items() for p in ps)
    assert 1 <= st["launches"] <= 2


def test_debug_summary_matches_reference_format():
    """LOGGER.debug of Main.java:279-306: one message per topic, consumers in consumerTotalLags' HashMap
    order, each followed by every partition it holds so far (cumulative map, Main.java:296)."""
    from oracle.java_collections import JavaHashMap
    metadata = {"topic1": [0, 1, 2, 3], "topic2": [0, 1]}
    lag = {("topic1", 0): 100000, ("topic1", 1): 100000, ("topic1", 2): 500, ("topic1", 3): 1,
           ("topic2", 0): 900000, ("topic2", 1): 100000}
    begin = {k: 0 for k in lag}
    end = dict(lag)
    com = {k: 0 for k in lag}
    subs = {"consumer-1": ["topic1", "topic2"], "consumer-2": ["topic1"]}
    a = LagBasedPartitionAssignor()
    a.configure({"group.id": "g", "auto.offset.reset": "earliest"})
    messages = []
    a.set_debug(messages.append)
    got = a.assign(metadata, subs, FakeOffsets(begin, end, com))
    assert got == {"consumer-1": [("topic1", 0), ("topic1", 2), ("topic2", 0), ("topic2", 1)],
                   "consumer-2": [("topic1", 1), ("topic1", 3)]}                     # Test.java:112-125
    assert len(messages) == 2 and messages[0].startswith("Assignment for topic1:\n")
    # expected text from the container model: per topic, a HashMap of its consumers filled by put()
    cumulative = {"consumer-1": [], "consumer-2": []}
    totals = a.last_topic_totals()
    for msg, topic in zip(messages, ["topic1", "topic2"]):
        for m in cumulative:
            cumulative[m] += [tp for tp in got[m] if tp[0] == topic]
        consumers = [m for m, ts in subs.items() if topic in ts]
        hm = JavaHashMap(len(consumers))                         # new HashMap<>(consumers.size()), Main.java:216
        for m in consumers:
            hm.put(m, 0)
        exp = "Assignment for %s:\n" % topic
        for m in hm.keys():
            exp += "\t%s (total_lag=%d)\n" % (m, totals[topic][m])
            for (t, p) in cumulative[m]:
                exp += "\t\t%s-%d\n" % (t, p)
        assert msg == exp
    assert "\tconsumer-1 (total_lag=100500)\n\t\ttopic1-0\n\t\ttopic1-2\n" in messages[0]


def test_order_exact_is_surfaced_when_a_bucket_would_treeify():
    """>= 9 topic names in one bucket of a >= 64-slot consumersPerTopic table.  The C++ host cannot
    reproduce a tree bin's iteration order: it must say so (flag + warn) and still assign every partition exactly."""
    from oracle.java_collections import java_string_hash

    def bucket(k, mask):
        x = java_string_hash(k) & 0xFFFFFFFF
        return (x ^ (x >> 16)) & mask

    colliding, i = [], 0
    while len(colliding) < 9:
        k = "topic-%d" % i
        if bucket(k, 255) == 0:
            colliding.append(k)
        i += 1
    filler = [k for k in ("f%d" % j for j in range(300)) if bucket(k, 15) != 0][:60]
    topics = filler + colliding
    rng = random.Random(11)
    lags = {t: [TopicPartitionLag(t, p, rng.randint(0, 10 ** 6)) for p in range(5)] for t in topics}
    subs = {"consumer-%d" % c: list(topics) for c in range(3)}
    got = LagBasedPartitionAssignor.assign_lags(lags, subs)
    assert LagBasedPartitionAssignor.last_static_order_exact() is False
    # who gets what: topic by topic against the oracle (the per-topic problem does not depend on any map order)
    for t in topics:
        exp = oracle.assign_named({t: [tuple(e) for e in lags[t]]}, {m: [t] for m in subs})
        for m in subs:
            assert [tp for tp in got[m] if tp[0] == t] == exp[m]
    # the flag is per call: an ordinary call resets it
    LagBasedPartitionAssignor.assign_lags({"t0": [TopicPartitionLag("t0", 0, 1)]}, {"C0": ["t0"]})
    assert LagBasedPartitionAssignor.last_static_order_exact() is True
    # instance level: flag + warn
    a = LagBasedPartitionAssignor()
    a.configure({"group.id": "g"})
    warnings = []
    a.set_warn(warnings.append)
    metadata = {t: [0, 1] for t in topics}
    zeros = {(t, p): 0 for t in topics for p in (0, 1)}
    a.assign(metadata, subs, FakeOffsets(zeros, {k: 7 for k in zeros}, {}))
    assert a.last_order_exact() is False and any("tree-bin" in w for w in warnings)


# ---- round 6: the GPU path against the SECOND reading of the reference (oracle/literal_py.py), on what the reference's own
# tests leave unpinned: memberIds whose string order is not their numeric order (and UTF-16 order), shuffled partition ids with
# tied lags, negative lags, totals that wrap a long, a topic listed twice, topics without a lag list -------------
@pytest.mark.parametrize("kind", ["numbered", "utf16", "prefixes"])
@pytest.mark.parametrize("lags", ["ties", "negative", "wrapping", "uniform"])
def test_static_assign_against_the_second_reading(kind, lags):
    from oracle import literal_py as lit
    rng = random.Random(hash((kind, lags)) % 100000)
    pools = {"numbered": lambda i: "consumer-%d" % i,
             "utf16": lambda i: "%s-%d" % (["z", "é", "￮", "\U0001f600", "퟿", "", "Z", "中"][i % 8], i // 8),
             "prefixes": lambda i: ["a", "ab", "abc", "b", "B", "aB", "a0", "a-", "", "a-1", "a-10", "a-2"][i % 12] + ("" if i < 12 else str(i))}
    for trial in range(10):
        members = [pools[kind](i) for i in rng.sample(range(40), rng.randint(1, 14))]
        topics = ["topic-%d" % i for i in range(rng.randint(1, 6))]
        subs = {}
        for m in members:
            mine = [t for t in topics if rng.random() < 0.7]
            if mine and rng.random() < 0.3:
                mine.append(mine[0])
            subs[m] = mine
        pl = {}
        for t in topics:
            if rng.random() < 0.15:
                continue
            p = rng.randint(0, 60)
            ids = rng.sample(range(3 * p + 1), p)
            if lags == "ties":
                ls = [rng.choice([0, 1000, 2000]) for _ in range(p)]
            elif lags == "negative":
                ls = [rng.randint(-(1 << 40), 1 << 40) for _ in range(p)]
            elif lags == "wrapping":
                ls = [rng.randint(1 << 62, (1 << 63) - 1) for _ in range(p)]
            else:
                ls = [rng.randint(0, 1 << 40) for _ in range(p)]
            pl[t] = [TopicPartitionLag(t, i, l) for i, l in zip(ids, ls)]
        got = LagBasedPartitionAssignor.assign_lags(pl, subs)
        exp = lit.assign({t: [tuple(e) for e in v] for t, v in pl.items()}, subs)
        assert set(got) == set(exp) == set(subs)
        for m in subs:                                       # list order inside every topic; across topics it is the HashMap's
            for t in topics:
                assert [tuple(tp) for tp in got[m] if tp[0] == t] == [tp for tp in exp[m] if tp[0] == t], (trial, m, t)
