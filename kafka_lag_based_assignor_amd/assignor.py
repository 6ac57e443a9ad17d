"""Python face of the host mirror (csrc/host/) -- same names and argument meaning as the
reference's ``LagBasedPartitionAssignor`` (Main.java:83-457), so parity tests read like
``LagBasedPartitionAssignorTest.java``.

All container/string logic lives in the C++ host; all arithmetic runs in the HIP kernels
behind ``include/lagassign.h``.  Nothing here computes a lag or an assignment on the CPU.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, List, Mapping, NamedTuple, Optional, Sequence, Tuple


class TopicPartition(NamedTuple):
    topic: str
    partition: int


class TopicPartitionLag(NamedTuple):
    """Main.java:431-455."""
    topic: str
    partition: int
    lag: int


class OffsetAndMetadata(NamedTuple):
    offset: int


def _host():
    from . import _native
    _native.load()                      # liblagassign.so first (and torch before it, see _native.load)
    try:
        from . import _host as h
    except ImportError as e:            # no fallback: say how to build it
        raise ImportError("kafka_lag_based_assignor_amd._host is not built: run "
                          "`python -m kafka_lag_based_assignor_amd.build`") from e
    return h


class LagBasedPartitionAssignor:
    """Drop-in surface: ``configure`` / ``name`` / ``assign`` (Main.java:97-157) plus the two
    package-private statics the reference's tests call (Main.java:166, :376)."""

    def __init__(self) -> None:
        self._impl = _host().LagBasedPartitionAssignor()

    # -- plugin surface -------------------------------------------------------------
    def configure(self, configs: Mapping[str, object]) -> None:
        """Main.java:97-130.  Raises ValueError (IllegalArgumentException) without group.id."""
        self._impl.configure({str(k): str(v) for k, v in configs.items()})

    def name(self) -> str:
        return self._impl.name()

    def assign(self, metadata: Mapping[str, Sequence[int]], subscriptions: Mapping[str, Sequence[str]],
               offsets) -> Dict[str, List[TopicPartition]]:
        """assign(Cluster, GroupSubscription), Main.java:137-157.

        ``metadata``: topic -> partition ids (Cluster.partitionsForTopic).
        ``offsets``: object with ``beginning_offsets(tps)``, ``end_offsets(tps)``,
        ``committed(tps)`` -> dict keyed by (topic, partition); it plays the side
        KafkaConsumer and is called ONCE per kind for all topics."""
        out = self._impl.assign({t: list(p) for t, p in metadata.items()},
                                [(m, list(ts)) for m, ts in subscriptions.items()], offsets)
        return {m: [TopicPartition(*tp) for tp in tps] for m, tps in out.items()}

    def metadata_consumer_props(self) -> Dict[str, str]:
        return dict(self._impl.metadata_consumer_props())

    def last_topic_totals(self) -> Dict[str, Dict[str, int]]:
        """Per-topic per-member total lag of the last assign() (the reference's debug summary)."""
        return {t: dict(v) for t, v in self._impl.last_topic_totals().items()}

    def last_order_exact(self) -> bool:
        """False when the last assign() met a HashMap bucket a JVM turns into a tree bin: the ORDER of topics inside the
        members' lists then follows the C++ host's restatement of HashMap's TreeNode handling (definite, equal to the oracle's
        own restatement, unverified against a JVM); who gets what is unaffected.  Also reported through the warn hook."""
        return bool(self._impl.last_order_exact())

    @staticmethod
    def last_static_order_exact() -> bool:
        """The same for the last assign_lags() on this thread."""
        return bool(_host().LagBasedPartitionAssignor.last_static_order_exact())

    @staticmethod
    def last_native_call() -> Dict[str, object]:
        """What the last native assign call on this thread was given and did: `hinted` (the marshalling loop vouched for
        bounds: la_hint_next_call), `max_lag` / `max_partition_id`, the library's `pipeline` and kernel `launches`."""
        return dict(_host().LagBasedPartitionAssignor.last_native_call())

    def set_warn(self, fn: Callable[[str], None]) -> None:
        self._impl.set_warn(fn)

    def set_debug(self, fn: Callable[[str], None]) -> None:
        """LOGGER.debug of Main.java:279-306: one "Assignment for <topic>:\\n<summary>" message per topic."""
        self._impl.set_debug(fn)

    # -- the seams the reference's tests use ------------------------------------------------
    @staticmethod
    def assign_lags(partition_lag_per_topic: Mapping[str, Iterable[TopicPartitionLag]],
                    subscriptions: Mapping[str, Sequence[str]]) -> Dict[str, List[TopicPartition]]:
        """static assign(Map, Map), Main.java:166-188.  Maps are walked in their own
        iteration order, as Java would walk the caller's maps."""
        lags = [(t, [(e[0], int(e[1]), int(e[2])) for e in v]) for t, v in partition_lag_per_topic.items()]
        subs = [(m, list(ts)) for m, ts in subscriptions.items()]
        out = _host().LagBasedPartitionAssignor.assign_static(lags, subs)
        return {m: [TopicPartition(*tp) for tp in tps] for m, tps in out.items()}

    @staticmethod
    def compute_partition_lag(partition_metadata: Optional[OffsetAndMetadata], begin_offset: int,
                              end_offset: int, auto_offset_reset_mode: str) -> int:
        """static computePartitionLag, Main.java:376-404."""
        committed = None if partition_metadata is None else int(partition_metadata.offset)
        return _host().LagBasedPartitionAssignor.compute_partition_lag(committed, begin_offset, end_offset,
                                                                       auto_offset_reset_mode)
