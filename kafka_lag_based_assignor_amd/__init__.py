"""MI355X-native lag-based Kafka partition assignor (hot path of
grantneale/kafka-lag-based-assignor behind its own plugin surface).

    include/lagassign.h            the C ABI (drop-in boundary; JNI / ctypes bind this)
    csrc/*.hip                     HIP kernels for gfx950 + the C ABI implementation
    csrc/host/                     C++ host mirror of the reference's plugin class
    _native.py                     ctypes binding of the C ABI
    assignor.py                    Python face of the host mirror
    synth.py                       BASELINE.json workloads
    sharding.py                    topic sharding across ranks (one process per GPU)

Importing this package does not load any native code; the first use does, and fails loudly
if the libraries are not built (there is no CPU fallback).
"""
from .assignor import (LagBasedPartitionAssignor, OffsetAndMetadata, TopicPartition,  # noqa: F401
                       TopicPartitionLag)

__all__ = ["LagBasedPartitionAssignor", "TopicPartition", "TopicPartitionLag", "OffsetAndMetadata"]
