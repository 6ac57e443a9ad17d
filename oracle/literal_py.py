"""TEST INFRASTRUCTURE ONLY -- a SECOND, independent reading of the reference's hot path, in plain Python.

oracle/lag_oracle.c is one reading of LagBasedPartitionAssignor.java ("Main.java"); the kernels were written by the same
hands, and the reference's own tests pin neither memberId order where string order differs from numeric order, nor negative
lags, nor wrapping totals, nor the partition-id tie-break on shuffled input.  This module
restates the same lines a second time with none of the first restatement's code: dictionaries keyed by memberId as the
reference's HashMaps are, Python's own sort and min driven by comparators written from the Java text, Java's `long`
arithmetic made explicit.  tests/test_oracle_golden.py holds the two readings against each other on exactly those unpinned
behaviours (and both against the reference's known answers).  Small cases only: every step is a Python call.

May be imported by tests/ only.
"""
from __future__ import annotations

from functools import cmp_to_key
from typing import Dict, List, Optional, Sequence, Tuple

_M64 = (1 << 64) - 1


def as_long(x: int) -> int:
    """a Java `long`: two's complement, 64 bits"""
    x &= _M64
    return x - (1 << 64) if x >> 63 else x


def string_compare_to(a: str, b: str) -> int:
    """java.lang.String.compareTo: the first differing UTF-16 code unit decides, else the lengths (Main.java:259 calls it)."""
    ua = a.encode("utf-16-be", "surrogatepass")
    ub = b.encode("utf-16-be", "surrogatepass")
    for i in range(0, min(len(ua), len(ub)), 2):
        ca, cb = (ua[i] << 8) | ua[i + 1], (ub[i] << 8) | ub[i + 1]
        if ca != cb:
            return ca - cb
    return (len(ua) - len(ub)) // 2


def compute_partition_lag(committed: Optional[int], begin: int, end: int, mode: str) -> int:
    """Main.java:376-404: the next offset is the committed one if there is one; otherwise the end offset when the mode is
    "latest" in any letter case, the beginning offset for every other string; lag = max(end - next, 0) in `long` arithmetic."""
    if committed is not None:
        nxt = committed
    elif mode.lower() == "latest" and len(mode) == 6:      # equalsIgnoreCase("latest")
        nxt = end
    else:
        nxt = begin
    return max(as_long(end - nxt), 0)


def _compare(a: int, b: int) -> int:                      # Long.compare / Integer.compare
    return -1 if a < b else (1 if a > b else 0)


def assign_topic(assignment: Dict[str, List[Tuple[str, int]]], topic: str, consumers: Sequence[str],
                 partition_lags: List[Tuple[str, int, int]]) -> Dict[str, int]:
    """Main.java:204-266 for one topic.  partition_lags: (topic, partition, lag) triples, sorted IN PLACE as the reference
    sorts its caller's list (:228).  Returns the consumers' final total lags."""
    if not consumers:                                      # :211-213
        return {}
    total_lag = {m: 0 for m in consumers}                  # :216-219  (keyed by memberId: a duplicate member is one bin)
    total_parts = {m: 0 for m in consumers}                # :222-225

    def by_lag_then_partition(p1, p2):                     # :228-235
        if p1[2] == p2[2]:
            return _compare(p1[1], p2[1])
        return _compare(p2[2], p1[2])                      # the larger lag first
    partition_lags.sort(key=cmp_to_key(by_lag_then_partition))      # (list.sort is stable, as TimSort is)

    def consumer_order(c1, c2):                            # :243-260, over the (memberId, total lag) entries
        by_count = _compare(total_parts[c1[0]], total_parts[c2[0]])
        if by_count != 0:
            return by_count
        by_lag = _compare(c1[1], c2[1])
        if by_lag != 0:
            return by_lag
        return string_compare_to(c1[0], c2[0])

    for (tp_topic, partition, lag) in partition_lags:      # :237
        member = min(total_lag.items(), key=cmp_to_key(consumer_order))[0]
        assignment[member].append((tp_topic, partition))   # :264 (the element's own topic field)
        total_lag[member] = as_long(total_lag[member] + lag)       # :265, a wrapping long
        total_parts[member] += 1                           # :266
    return total_lag


def assign(partition_lag_per_topic: Dict[str, List[Tuple[str, int, int]]], subscriptions: Dict[str, List[str]],
           topic_order: Optional[Sequence[str]] = None) -> Dict[str, List[Tuple[str, int]]]:
    """Main.java:166-188.  Who gets what does not depend on the order the topics are walked in (SURVEY 8a note 3); the order
    of a member's list does, so the caller passes `topic_order` (the HashMap iteration order of consumersPerTopic, modelled
    elsewhere) when it wants to compare lists, and gets first-seen order otherwise."""
    assignment = {m: [] for m in subscriptions}            # :171-174: a list for EVERY member
    consumers_per_topic: Dict[str, List[str]] = {}         # :410-426: a member that lists a topic twice appears twice
    for member, topics in subscriptions.items():
        for t in topics:
            consumers_per_topic.setdefault(t, []).append(member)
    for t in (topic_order if topic_order is not None else list(consumers_per_topic)):
        assign_topic(assignment, t, consumers_per_topic[t], list(partition_lag_per_topic.get(t, [])))    # :182: missing -> empty
    return assignment
