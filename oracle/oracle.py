"""TEST INFRASTRUCTURE ONLY -- ctypes front end of oracle/lag_oracle.c.

May be imported by tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline``
leg, and by nothing under ``kafka_lag_based_assignor_amd/``.

Parity status: "port" (the reference is Java; no JVM exists in this image, so there is
no oracle/_ref).  Pinned against the reference's own known-answer vectors in
tests/test_oracle_golden.py.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from .java_collections import JavaHashMap

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liblagoracle.so")
_lib: Optional[ctypes.CDLL] = None

_i64p = ctypes.c_void_p      # array parameters are passed as addresses (see _addr)
_i32p = ctypes.c_void_p


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "lag_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-B", "-C", _HERE, "liblagoracle.so"])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        L.lao_java_string_compare.restype = ctypes.c_int
        L.lao_java_string_compare.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        L.lao_java_string_hash.restype = ctypes.c_int32
        L.lao_java_string_hash.argtypes = [ctypes.c_char_p]
        L.lao_compute_partition_lag.restype = ctypes.c_int64
        L.lao_compute_partition_lag.argtypes = [ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                                ctypes.c_int64, ctypes.c_char_p]
        L.lao_compute_lags.restype = None
        L.lao_compute_lags.argtypes = [ctypes.c_int64, _i64p, _i64p, _i64p, ctypes.c_int, _i64p]
        L.lao_assign_flat.restype = ctypes.c_int
        L.lao_assign_flat.argtypes = [ctypes.c_int32, _i64p, _i32p, _i64p, _i64p, _i32p,
                                      ctypes.POINTER(ctypes.c_char_p), _i32p, _i32p, _i64p]
        _lib = L
    return _lib


def _addr(a: Optional[np.ndarray]):
    """Address of a contiguous array for a c_void_p parameter through the buffer protocol (0.3 us; `a.ctypes.data_as` costs
    1-2.5 us, which is most of a small call).  The same shortcut as the product's binding takes, so that bench.py's
    small_call rows compare the two libraries and not two ways of making a pointer.  The callers keep `a` alive."""
    if a is None:
        return None
    try:
        return ctypes.addressof(ctypes.c_char.from_buffer(a))
    except (TypeError, ValueError, BufferError):       # read-only or empty
        return a.ctypes.data


_p64 = _addr
_p32 = _addr


def java_string_compare(a: str, b: str) -> int:
    return lib().lao_java_string_compare(a.encode("utf-8"), b.encode("utf-8"))


def java_string_hash(s: str) -> int:
    return lib().lao_java_string_hash(s.encode("utf-8"))


def compute_partition_lag(committed: Optional[int], begin: int, end: int, mode: str) -> int:
    """computePartitionLag(OffsetAndMetadata|null, begin, end, mode), Main.java:376-404."""
    return lib().lao_compute_partition_lag(0 if committed is None else 1,
                                           0 if committed is None else committed,
                                           begin, end, mode.encode("utf-8"))


def compute_lags(begin: Optional[np.ndarray], end: np.ndarray, committed: np.ndarray,
                 reset_latest: bool) -> np.ndarray:
    end = np.ascontiguousarray(end, dtype=np.int64)
    committed = np.ascontiguousarray(committed, dtype=np.int64)
    if begin is not None:
        begin = np.ascontiguousarray(begin, dtype=np.int64)
    out = np.empty_like(end)
    lib().lao_compute_lags(end.size, _p64(begin), _p64(end), _p64(committed),
                           1 if reset_latest else 0, _p64(out))
    return out


def assign_flat(part_off: np.ndarray, partition: np.ndarray, lag: np.ndarray,
                cons_off: np.ndarray, cons_member: np.ndarray,
                member_ids: Optional[Sequence[str]] = None,
                want_totals: bool = True) -> Tuple[np.ndarray, np.ndarray, Optional[np.ndarray]]:
    """Per-topic loop of assign(Map,Map) over SoA input (see lao_assign_flat).

    ``cons_member`` holds indices into ``member_ids``; with ``member_ids=None`` it holds
    String.compareTo ranks and the third comparator level compares them numerically.
    Returns (partition ids in assignment order, chosen member handle, per-consumer
    final total lag)."""
    part_off = np.ascontiguousarray(part_off, dtype=np.int64)
    cons_off = np.ascontiguousarray(cons_off, dtype=np.int64)
    partition = np.ascontiguousarray(partition, dtype=np.int32)
    lag = np.ascontiguousarray(lag, dtype=np.int64)
    cons_member = np.ascontiguousarray(cons_member, dtype=np.int32)
    n_topics = part_off.size - 1
    out_p = np.empty(partition.size, dtype=np.int32)
    out_m = np.empty(partition.size, dtype=np.int32)
    out_t = np.zeros(cons_member.size, dtype=np.int64) if want_totals else None
    ids = None
    if member_ids is not None:
        ids = (ctypes.c_char_p * len(member_ids))(*[m.encode("utf-8") for m in member_ids])
    rc = lib().lao_assign_flat(n_topics, _p64(part_off), _p32(partition), _p64(lag),
                               _p64(cons_off), _p32(cons_member), ids,
                               _p32(out_p), _p32(out_m), _p64(out_t))
    if rc != 0:
        raise MemoryError("lao_assign_flat failed")
    return out_p, out_m, out_t


# --------------------------------------------------------------------------- #
# The static assign(Map,Map) with the reference's container semantics.         #
# --------------------------------------------------------------------------- #
TopicPartitionLag = Tuple[str, int, int]      # (topic, partition, lag), Main.java:431-455
TopicPartition = Tuple[str, int]


def assign_named(partition_lag_per_topic: Dict[str, List[TopicPartitionLag]],
                 subscriptions: Dict[str, List[str]]) -> Dict[str, List[TopicPartition]]:
    """assign(Map<String,List<TopicPartitionLag>>, Map<String,List<String>>),
    Main.java:166-188, including list order.

    ``subscriptions`` is walked in the caller's iteration order (the reference test
    passes an insertion-ordered ImmutableMap, Test.java:100-110)."""
    assignment: Dict[str, List[TopicPartition]] = {m: [] for m in subscriptions}   # :171-174

    # consumersPerTopic, Main.java:410-426 (HashMap + computeIfAbsent)
    consumers_per_topic: JavaHashMap[List[str]] = JavaHashMap()
    for member, topics in subscriptions.items():
        for topic in topics:
            consumers_per_topic.compute_if_absent(topic, list).append(member)

    for topic, consumers in consumers_per_topic.items():                            # :177
        lags = partition_lag_per_topic.get(topic, [])                               # :182
        if not consumers:
            continue
        members = list(dict.fromkeys(consumers))
        index = {m: i for i, m in enumerate(members)}
        part_off = np.array([0, len(lags)], dtype=np.int64)
        cons_off = np.array([0, len(consumers)], dtype=np.int64)
        pid = np.array([p for (_, p, _) in lags], dtype=np.int32)
        lag = np.array([l for (_, _, l) in lags], dtype=np.int64)
        cons = np.array([index[m] for m in consumers], dtype=np.int32)
        out_p, out_m, _ = assign_flat(part_off, pid, lag, cons_off, cons, members)
        # TopicPartition(partition.getTopic(), ...): the element's own topic field, :264
        topic_of = {}
        for (tp_topic, p, _) in lags:
            topic_of.setdefault(p, tp_topic)
        for p, m in zip(out_p.tolist(), out_m.tolist()):
            assignment[members[m]].append((topic_of[p], p))
    return assignment
