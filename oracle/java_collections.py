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
before topic2) and by ``"hello".hashCode() == 99162322``.

Tree bins (round 5).  A bucket of a >= 64-slot table that reaches 9 keys through ``put`` (8 through
``computeIfAbsent``) becomes a red-black tree of ``TreeNode``s that KEEP their ``next`` links: iteration still walks
``next``, but ``treeify`` / ``putTreeVal`` call ``moveRootToFront`` (the tree's root is unlinked and put first) and
``putTreeVal`` links a new node right behind its tree PARENT instead of at the tail -- so the order depends on the
tree's shape.  Restated below: ``treeifyBin`` / ``treeify`` / ``putTreeVal`` / ``balanceInsertion`` / the rotations /
``moveRootToFront`` / ``TreeNode.split`` with ``untreeify`` at <= 6 nodes; ordering inside the tree by the spread hash as a
signed int, then ``String.compareTo``.  UNVERIFIED AGAINST A JVM: no JDK exists in this image; the C++ host carries an
independent restatement of the same code (csrc/host/java_compat.hpp) and the two are compared on colliding keys.
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


def _signed32(h: int) -> int:
    h &= 0xFFFFFFFF
    return h - (1 << 32) if h & 0x80000000 else h


class _TreeNode:
    """HashMap.TreeNode: a red-black tree node that is also a node of the bin's doubly linked ``next`` list."""
    __slots__ = ("hash", "key", "parent", "left", "right", "red", "prev", "next")

    def __init__(self, h: int, key: str) -> None:
        self.hash, self.key = h, key                # hash: the spread hash, unsigned here; compared as a Java int
        self.parent = self.left = self.right = self.prev = self.next = None
        self.red = False


class _TreeBin:
    """One treeified bucket: ``first`` is table[index] (the tree's root after every structural change)."""

    def __init__(self, nodes: List[Tuple[int, str]]) -> None:
        prev = None
        self.first: Optional[_TreeNode] = None
        for h, key in nodes:                        # treeifyBin: replacementTreeNode for every node, same order
            x = _TreeNode(h, key)
            x.prev = prev
            if prev is None:
                self.first = x
            else:
                prev.next = x
            prev = x
        self.treeify()

    # ---- order ----
    def nodes(self) -> List[Tuple[int, str]]:
        out, x = [], self.first
        while x is not None:
            out.append((x.hash, x.key))
            x = x.next
        return out

    def __len__(self) -> int:
        return len(self.nodes())

    # ---- TreeNode.treeify: insert the nodes in `next` order, then moveRootToFront ----
    @staticmethod
    def _dir(h: int, key: str, p: "_TreeNode") -> int:
        ph, hh = _signed32(p.hash), _signed32(h)
        if ph > hh:
            return -1
        if ph < hh:
            return 1
        d = java_string_compare(key, p.key)         # compareComparables(String.class, k, pk)
        return -1 if d < 0 else 1                   # (0 cannot happen: equal keys never meet here)

    def treeify(self) -> None:
        root = None
        x = self.first
        while x is not None:
            nxt = x.next
            x.left = x.right = None
            if root is None:
                x.parent = None
                x.red = False
                root = x
            else:
                p = root
                while True:
                    d = self._dir(x.hash, x.key, p)
                    xp = p
                    p = p.left if d <= 0 else p.right
                    if p is None:
                        x.parent = xp
                        if d <= 0:
                            xp.left = x
                        else:
                            xp.right = x
                        root = self._balance_insertion(root, x)
                        break
            x = nxt
        self._move_root_to_front(root)

    def put(self, h: int, key: str) -> None:
        """TreeNode.putTreeVal for a NEW key: a leaf below its tree parent xp, linked right BEHIND xp in the next list."""
        root = self.first
        while root.parent is not None:              # root(): tab[index] is the root after moveRootToFront, but be literal
            root = root.parent
        p = root
        while True:
            d = self._dir(h, key, p)
            xp = p
            p = p.left if d <= 0 else p.right
            if p is None:
                xpn = xp.next
                x = _TreeNode(h, key)
                x.next = xpn
                if d <= 0:
                    xp.left = x
                else:
                    xp.right = x
                xp.next = x
                x.parent = x.prev = xp
                if xpn is not None:
                    xpn.prev = x
                self._move_root_to_front(self._balance_insertion(root, x))
                return

    def _move_root_to_front(self, root: Optional["_TreeNode"]) -> None:
        if root is None or root is self.first:
            return
        first = self.first
        rn, rp = root.next, root.prev
        if rn is not None:
            rn.prev = rp
        if rp is not None:
            rp.next = rn
        if first is not None:
            first.prev = root
        root.next = first
        root.prev = None
        self.first = root

    @staticmethod
    def _rotate_left(root, p):
        r = p.right if p is not None else None
        if p is not None and r is not None:
            rl = p.right = r.left
            if rl is not None:
                rl.parent = p
            pp = r.parent = p.parent
            if pp is None:
                root = r
                r.red = False
            elif pp.left is p:
                pp.left = r
            else:
                pp.right = r
            r.left = p
            p.parent = r
        return root

    @staticmethod
    def _rotate_right(root, p):
        l = p.left if p is not None else None
        if p is not None and l is not None:
            lr = p.left = l.right
            if lr is not None:
                lr.parent = p
            pp = l.parent = p.parent
            if pp is None:
                root = l
                l.red = False
            elif pp.right is p:
                pp.right = l
            else:
                pp.left = l
            l.right = p
            p.parent = l
        return root

    @classmethod
    def _balance_insertion(cls, root, x):
        x.red = True
        while True:
            xp = x.parent
            if xp is None:
                x.red = False
                return x
            if not xp.red or xp.parent is None:
                return root
            xpp = xp.parent
            xppl = xpp.left
            if xp is xppl:
                xppr = xpp.right
                if xppr is not None and xppr.red:
                    xppr.red = False
                    xp.red = False
                    xpp.red = True
                    x = xpp
                else:
                    if x is xp.right:
                        x = xp
                        root = cls._rotate_left(root, x)
                        xp = x.parent
                        xpp = None if xp is None else xp.parent
                    if xp is not None:
                        xp.red = False
                        if xpp is not None:
                            xpp.red = True
                            root = cls._rotate_right(root, xpp)
            else:
                if xppl is not None and xppl.red:
                    xppl.red = False
                    xp.red = False
                    xpp.red = True
                    x = xpp
                else:
                    if x is xp.left:
                        x = xp
                        root = cls._rotate_right(root, x)
                        xp = x.parent
                        xpp = None if xp is None else xp.parent
                    if xp is not None:
                        xp.red = False
                        if xpp is not None:
                            xpp.red = True
                            root = cls._rotate_left(root, xpp)


_UNTREEIFY_THRESHOLD = 6


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
        self.treeified = False                      # some bucket became a tree bin at some point (order: see the module notes)

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
        new: list = [[] for _ in range(new_cap)]
        for j, chain in enumerate(old):
            if isinstance(chain, _TreeBin):
                # TreeNode.split: lo / hi lists in `next` order; a half of <= 6 nodes is untreeified (plain nodes, same order),
                # a larger one is treeified AGAIN (from its list, root to the front) -- unless the other half is empty: then the
                # tree is "already treeified" and stays exactly as it is
                nodes = chain.nodes()
                lo = [nd for nd in nodes if not (nd[0] & old_cap)]
                hi = [nd for nd in nodes if nd[0] & old_cap]
                for part, other, at in ((lo, hi, j), (hi, lo, j + old_cap)):
                    if not part:
                        continue
                    if len(part) <= _UNTREEIFY_THRESHOLD:
                        new[at] = list(part)
                    elif other:
                        new[at] = _TreeBin(part)
                    else:
                        new[at] = chain
                continue
            for node in chain:                      # lo/hi split keeps relative order
                new[j + old_cap if (node[0] & old_cap) else j].append(node)
        self._table = new
        self._threshold = new_thr

    def _treeify_bin(self, index: int) -> None:
        assert self._table is not None
        if len(self._table) < _MIN_TREEIFY_CAPACITY:
            self._resize()
        else:
            self._table[index] = _TreeBin(self._table[index])
            self.treeified = True

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
        index = (len(self._table) - 1) & h
        chain = self._table[index]
        self._values[key] = value
        if isinstance(chain, _TreeBin):
            chain.put(h, key)                       # putTreeVal
        else:
            chain.append((h, key))
            if len(chain) >= _TREEIFY_THRESHOLD + 1:    # binCount >= 7 when appending the 9th
                self._treeify_bin(index)
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
        index = (len(self._table) - 1) & h
        chain = self._table[index]
        value = make()
        self._values[key] = value
        if isinstance(chain, _TreeBin):
            chain.put(h, key)                       # t.putTreeVal(this, tab, hash, key, v)
        else:
            bin_count = len(chain)
            chain.insert(0, (h, key))
            if bin_count >= _TREEIFY_THRESHOLD - 1:
                self._treeify_bin(index)
        self._size += 1
        return value

    def keys(self) -> Iterator[str]:
        if self._table is None:
            return
        for chain in self._table:
            for _, key in (chain.nodes() if isinstance(chain, _TreeBin) else chain):
                yield key

    def items(self) -> Iterator[Tuple[str, V]]:
        for k in self.keys():
            yield k, self._values[k]

    def __len__(self) -> int:
        return self._size
