"""TEST INFRASTRUCTURE ONLY (part of the oracle; never imported by the product).

Restates the two JDK behaviours the reference's *list order* depends on:

* ``String.hashCode``                         (UTF-16 units, 31-polynomial, int wrap)
* ``java.util.HashMap`` iteration order       (JDK 8+: power-of-two table, hash spread
  ``h ^ (h >>> 16)``, load factor 0.75, order-preserving split on resize, ``put``
  appends at a bin's tail, ``computeIfAbsent`` inserts at a bin's HEAD)

Why it matters: ``assign(Map,Map)`` walks ``consumersPerTopic.entrySet()`` -- a
``HashMap<String, List<String>>`` filled with ``computeIfAbsent`` -- and appends each
topic's partitions to the members' lists in that order (Main.java:176-184, :410-426,
:264).  The reference's ``testAssign`` compares those lists with ``List.equals``
(Test.java:112-130), so list order is observable.

[upstream-knowledge] The JDK source is not on this box; this is a restatement of
OpenJDK 8's ``HashMap`` from memory and is pinned only by Test.java:112-130 (topic1
before topic2) and by ``"hello".hashCode() == 99162322``.  Tree bins (>= 9 keys in one
bucket of a >= 64-slot table) reorder a bin via ``moveRootToFront``; that case raises
``NotImplementedError`` rather than guessing.
"""
from __future__ import annotations

from typing import Dict, Generic, Iterator, List, Optional, Tuple, TypeVar

V = TypeVar("V")

_MIN_TREEIFY_CAPACITY = 64
_TREEIFY_THRESHOLD = 8


def utf16_units(s: str) -> List[int]:
    b = s.encode("utf-16-be", "surrogatepass")
    return [(b[i] << 8) | b[i + 1] for i in range(0, len(b), 2)]


def java_string_hash(s: str) -> int:
    """java.lang.String.hashCode as a signed 32-bit int."""
    h = 0
    for u in utf16_units(s):
        h = (31 * h + u) & 0xFFFFFFFF
    return h - (1 << 32) if h & 0x80000000 else h


def java_string_compare(a: str, b: str) -> int:
    """java.lang.String.compareTo (UTF-16 code-unit order, then length)."""
    ua, ub = utf16_units(a), utf16_units(b)
    for x, y in zip(ua, ub):
        if x != y:
            return x - y
    return len(ua) - len(ub)


def _spread(h: int) -> int:
    h &= 0xFFFFFFFF
    return (h ^ (h >> 16)) & 0xFFFFFFFF


class JavaHashMap(Generic[V]):
    """Order-faithful model of ``new HashMap<String, V>()`` and ``new HashMap<String, V>(initialCapacity)``."""

    def __init__(self, initial_capacity: Optional[int] = None) -> None:
        self._table: Optional[List[List[Tuple[int, str]]]] = None
        self._threshold = 0
        if initial_capacity is not None:            # HashMap(int): threshold = tableSizeFor(initialCapacity)
            cap = 1
            while cap < initial_capacity:
                cap <<= 1
            self._threshold = cap
        self._size = 0
        self._values: Dict[str, V] = {}

    # -- internals -----------------------------------------------------------
    def _resize(self) -> None:
        """HashMap.resize(): double (the threshold doubles only from 16 slots up, else int(0.75f * newCap)); an empty
        table takes the capacity the constructor left in `threshold`, or 16 / 12."""
        old = self._table or []
        old_cap = len(old)
        new_thr = 0
        if old_cap > 0:
            new_cap = old_cap * 2
            if old_cap >= 16:
                new_thr = self._threshold * 2
        elif self._threshold > 0:
            new_cap = self._threshold
        else:
            new_cap, new_thr = 16, 12
        if new_thr == 0:
            new_thr = int(new_cap * 0.75)
        new: List[List[Tuple[int, str]]] = [[] for _ in range(new_cap)]
        for j, chain in enumerate(old):
            for node in chain:                      # lo/hi split keeps relative order
                new[j + old_cap if (node[0] & old_cap) else j].append(node)
        self._table = new
        self._threshold = new_thr

    def _treeify_bin(self) -> None:
        assert self._table is not None
        if len(self._table) < _MIN_TREEIFY_CAPACITY:
            self._resize()
        else:
            raise NotImplementedError("tree bins reorder iteration; not modelled")

    # -- HashMap API subset --------------------------------------------------
    def contains(self, key: str) -> bool:
        return key in self._values

    def get(self, key: str) -> V:
        return self._values[key]

    def put(self, key: str, value: V) -> None:
        """HashMap.putVal: append at the bin's tail; resize AFTER insertion."""
        if key in self._values:
            self._values[key] = value
            return
        if self._table is None:
            self._resize()
        assert self._table is not None
        h = _spread(java_string_hash(key))
        chain = self._table[(len(self._table) - 1) & h]
        chain.append((h, key))
        self._values[key] = value
        if len(chain) >= _TREEIFY_THRESHOLD + 1:    # binCount >= 7 when appending the 9th
            self._treeify_bin()
        self._size += 1
        if self._size > self._threshold:
            self._resize()

    def compute_if_absent(self, key: str, make) -> V:
        """HashMap.computeIfAbsent: resize BEFORE (if size > threshold); new node
        becomes the bin's HEAD."""
        if key in self._values:
            return self._values[key]
        if self._table is None or self._size > self._threshold:
            self._resize()
        assert self._table is not None
        h = _spread(java_string_hash(key))
        chain = self._table[(len(self._table) - 1) & h]
        bin_count = len(chain)
        chain.insert(0, (h, key))
        value = make()
        self._values[key] = value
        if bin_count >= _TREEIFY_THRESHOLD - 1:
            self._treeify_bin()
        self._size += 1
        return value

    def keys(self) -> Iterator[str]:
        if self._table is None:
            return
        for chain in self._table:
            for _, key in chain:
                yield key

    def items(self) -> Iterator[Tuple[str, V]]:
        for k in self.keys():
            yield k, self._values[k]

    def __len__(self) -> int:
        return self._size
