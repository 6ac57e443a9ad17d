"""Pins the CPU oracle (oracle/) on every known-answer vector the reference holds for
the hot path: LagBasedPartitionAssignorTest.java:21-228 and README.md:42-57.
The oracle is the checker for the HIP path, so it is checked first."""
import numpy as np
import pytest

from oracle import oracle
from oracle.java_collections import JavaHashMap, java_string_compare, java_string_hash


# ---- computePartitionLag: Test.java:21-80 ----------------------------------
def test_compute_partition_lag():                       # Test.java:21-33
    assert oracle.compute_partition_lag(5555, 1111, 9999, "none") == 4444


def test_compute_partition_lag_no_end_offset():         # Test.java:38-50
    assert oracle.compute_partition_lag(5555, 0, 0, "none") == 0


def test_compute_partition_lag_no_committed_latest():   # Test.java:52-64
    assert oracle.compute_partition_lag(None, 1111, 9999, "latest") == 0


def test_compute_partition_lag_no_committed_earliest():  # Test.java:66-80
    assert oracle.compute_partition_lag(None, 1111, 9999, "earliest") == 9999 - 1111


@pytest.mark.parametrize("mode,expect", [
    ("LATEST", 0), ("Latest", 0), ("lateſt", 0),      # equalsIgnoreCase, Main.java:391
    ("none", 8888), ("", 8888), ("latest ", 8888), ("earliest", 8888),
])
def test_reset_mode_matching(mode, expect):
    assert oracle.compute_partition_lag(None, 1111, 9999, mode) == expect


def test_lag_wraps_like_java_long():
    # end - next overflows to a negative long -> clamped to 0 (Main.java:402)
    assert oracle.compute_partition_lag(0, 0, -(2**63), "latest") == 0
    assert oracle.compute_partition_lag(2**63 - 1, 0, -(2**63), "latest") == 1


def test_vector_lag_matches_scalar():
    rng = np.random.default_rng(1)
    n = 500
    begin = rng.integers(0, 1000, n)
    end = rng.integers(0, 5000, n)
    com = rng.integers(-1, 5000, n)
    for latest in (True, False):
        got = oracle.compute_lags(begin, end, com, latest)
        for i in range(n):
            c = None if com[i] < 0 else int(com[i])
            assert got[i] == oracle.compute_partition_lag(
                c, int(begin[i]), int(end[i]), "latest" if latest else "earliest")


# ---- assign: Test.java:82-228, README.md:42-57 ------------------------------
def _tpl(topic, lags):
    return [(topic, p, l) for p, l in enumerate(lags)]


def test_assign_reference_vector():                     # Test.java:82-132
    lags = {"topic1": _tpl("topic1", [100000, 100000, 500, 1]),
            "topic2": _tpl("topic2", [900000, 100000])}
    subs = {"consumer-1": ["topic1", "topic2"], "consumer-2": ["topic1"]}
    expected = {
        "consumer-1": [("topic1", 0), ("topic1", 2), ("topic2", 0), ("topic2", 1)],
        "consumer-2": [("topic1", 1), ("topic1", 3)],
    }
    assert oracle.assign_named(lags, subs) == expected


def test_assign_zero_lags():                            # Test.java:134-175
    lags = {"topic1": _tpl("topic1", [0] * 7)}
    subs = {"consumer-1": ["topic1"], "consumer-2": ["topic1"]}
    got = oracle.assign_named(lags, subs)
    sizes = [len(v) for v in got.values()]
    assert max(sizes) <= min(sizes) + 1
    # derived: even partitions -> consumer-1, odd -> consumer-2
    assert [p for _, p in got["consumer-1"]] == [0, 2, 4, 6]
    assert [p for _, p in got["consumer-2"]] == [1, 3, 5]


def test_assign_heavily_skewed():                       # Test.java:177-228
    lags = {"topic1": _tpl("topic1", [360, 359, 230, 118, 444, 122, 65, 111, 455000, 424000])}
    subs = {"consumer-%d" % i: ["topic1"] for i in (1, 2, 3)}
    got = oracle.assign_named(lags, subs)
    sizes = [len(v) for v in got.values()]
    assert max(sizes) <= min(sizes) + 1
    # a lag-first comparator would give sizes 1/1/8 and fail the reference's property
    assert sorted(sizes) == [3, 3, 4]
    assert [p for _, p in got["consumer-1"]] == [8, 2, 7]
    assert [p for _, p in got["consumer-2"]] == [9, 1, 3]
    assert [p for _, p in got["consumer-3"]] == [4, 0, 5, 6]


def test_readme_example():                              # README.md:42-57
    lags = {"t0": _tpl("t0", [100000, 50000, 60000])}
    subs = {"C0": ["t0"], "C1": ["t0"]}
    got = oracle.assign_named(lags, subs)
    assert set(got["C0"]) == {("t0", 0)}
    assert set(got["C1"]) == {("t0", 1), ("t0", 2)}
    out_p, out_m, tot = oracle.assign_flat([0, 3], [0, 1, 2], [100000, 50000, 60000],
                                           [0, 2], [0, 1], ["C0", "C1"])
    assert out_p.tolist() == [0, 2, 1] and out_m.tolist() == [0, 1, 1]
    assert tot.tolist() == [100000, 110000]


# ---- semantics the reference code implies but its tests do not pin ----------
def test_member_id_order_is_string_order():
    # "consumer-10" < "consumer-2" under String.compareTo (Main.java:259)
    out_p, out_m, _ = oracle.assign_flat([0, 1], [0], [5], [0, 2], [0, 1],
                                         ["consumer-2", "consumer-10"])
    assert out_m.tolist() == [1]


def test_partition_tiebreak_on_shuffled_input():
    out_p, _, _ = oracle.assign_flat([0, 4], [3, 1, 2, 0], [7, 7, 9, 7], [0, 1], [0], ["a"])
    assert out_p.tolist() == [2, 0, 1, 3]


def test_total_lag_overflow_wraps_negative_and_wins():
    big = 2**63 - 1
    # round 1: c0<-p0(big), c1<-p1(big); round 2: tie -> c0<-p2 wraps negative;
    # c1<-p3 (lag 0, total stays big).  round 3: c0's wrapped-negative total is smallest.
    out_p, out_m, tot = oracle.assign_flat([0, 5], [0, 1, 2, 3, 4], [big, big, 5, 0, 0],
                                           [0, 2], [0, 1])
    assert out_m.tolist() == [0, 1, 0, 1, 0]
    assert tot.tolist()[0] == ((big + 5) & (2**64 - 1)) - 2**64


def test_duplicate_consumer_entries_are_one_consumer():
    a = oracle.assign_flat([0, 3], [0, 1, 2], [3, 2, 1], [0, 3], [0, 1, 0], ["x", "y"])
    b = oracle.assign_flat([0, 3], [0, 1, 2], [3, 2, 1], [0, 2], [0, 1], ["x", "y"])
    assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()


def test_topic_without_consumers_or_lags():
    out_p, out_m, _ = oracle.assign_flat([0, 2, 2], [4, 5], [1, 2], [0, 0, 1], [0])
    assert out_m.tolist() == [-1, -1]
    got = oracle.assign_named({}, {"m": ["ghost"]})      # Main.java:182 getOrDefault
    assert got == {"m": []}


# ---- JDK behaviours ----------------------------------------------------------
def test_java_string_hash_known_answers():
    assert java_string_hash("hello") == 99162322
    assert java_string_hash("") == 0
    assert oracle.java_string_hash("hello") == 99162322
    for s in ["topic1", "topic2", "consumer-10", "é\U0001F600x"]:
        assert oracle.java_string_hash(s) == java_string_hash(s)


def test_java_string_compare_utf16_order():
    # U+FF5E (BMP) vs U+1F600 (surrogates D83D DE00): UTF-16 order puts the emoji FIRST
    assert java_string_compare("\U0001F600", "～") < 0
    assert oracle.java_string_compare("\U0001F600", "～") < 0
    assert oracle.java_string_compare("consumer-10", "consumer-2") < 0
    assert oracle.java_string_compare("ab", "abc") < 0
    assert oracle.java_string_compare("abc", "abc") == 0


def test_hashmap_iteration_order_topic1_before_topic2():
    m = JavaHashMap()
    for k in ["topic2", "topic1"]:
        m.compute_if_absent(k, list)
    assert list(m.keys()) == ["topic1", "topic2"]


def test_hashmap_resize_keeps_all_keys():
    m = JavaHashMap()
    keys = ["k%d" % i for i in range(200)]
    for k in keys:
        m.compute_if_absent(k, list)
    assert sorted(m.keys()) == sorted(keys) and len(m) == 200


# ---- committed fixtures (tests/golden/) ------------------------------------------------------------------
import json
import os

_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_reference_vectors_fixture():
    """tests/golden/reference_vectors.json: the reference's own known answers, transcribed with citations."""
    ref = json.load(open(os.path.join(_GOLDEN, "reference_vectors.json")))
    for v in ref["compute_partition_lag"]:
        assert oracle.compute_partition_lag(v["committed"], v["begin"], v["end"], v["mode"]) == v["lag"], v["cite"]
    for v in ref["assign_exact"]:
        lags = {t: _tpl(t, l) for t, l in v["lags"].items()}
        got = oracle.assign_named(lags, v["subscriptions"])
        assert got == {m: [tuple(tp) for tp in tps] for m, tps in v["expected"].items()}, v["cite"]
    for v in ref["assign_sets"]:
        lags = {t: _tpl(t, l) for t, l in v["lags"].items()}
        got = oracle.assign_named(lags, v["subscriptions"])
        assert {m: set(tps) for m, tps in got.items()} == \
            {m: {tuple(tp) for tp in tps} for m, tps in v["expected"].items()}, v["cite"]
    for v in ref["assign_property"]:
        lags = {t: _tpl(t, l) for t, l in v["lags"].items()}
        sizes = [len(x) for x in oracle.assign_named(lags, v["subscriptions"]).values()]
        assert max(sizes) <= min(sizes) + 1, v["cite"]


def test_oracle_reproduces_frozen_digests():
    """tests/golden/oracle_frozen.json (made by tests/golden/make_golden.py): the oracle cannot drift."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(_GOLDEN, "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    from kafka_lag_based_assignor_amd import synth
    frozen = json.load(open(os.path.join(_GOLDEN, "oracle_frozen.json")))
    assert set(frozen) == {mg.case_key(*c) for c in mg.CASES}
    for name, scale, mode in mg.CASES:
        w = synth.config(name, scale)
        f = frozen[mg.case_key(name, scale, mode)]
        assert mg.digest(w.part_off, w.partition_id, w.begin, w.end, w.committed, w.cons_off, w.cons_rank) == \
            f["inputs_sha256"], "generator drifted: " + name
        p, m, t = mg.run_oracle(w, mode)
        assert mg.digest(p.astype(np.int32), m.astype(np.int32), t.astype(np.int64)) == f["sha256"], name


def test_round_form_equals_the_literal_per_step_min():
    """SURVEY.md section 8a note 5: the comparator's first key is the assigned count, so the literal per-partition
    Collections.min is the same as rounds of C partitions handed to the consumers in (total, rank) order as of the
    round start.  Every device kernel relies on that; here the literal oracle and an independent numpy round form
    are compared on random shapes, ties, zero lags, negative lags and totals that wrap."""
    from round_form import round_form
    from kafka_lag_based_assignor_amd import synth
    cases = 0
    for seed in range(60):
        for dist in ("mixed", "ties", "zero", "u63", "full"):
            w = synth.ragged(1000 * seed + len(dist), 12, 90, 17, dist=dist, negative=(dist in ("mixed", "full")))
            e = oracle.assign_flat(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
            r = round_form(w.part_off, w.partition_id, w.lag, w.cons_off, w.cons_rank)
            for x, y, what in zip(e, r, ("partition order", "member", "totals")):
                np.testing.assert_array_equal(x, y, err_msg="%s seed %d %s" % (what, seed, dist))
            cases += w.n_topics
    assert cases == 3600
