#!/usr/bin/env python3
"""Writes the first K topics of a synthetic workload as a LAV1 file for java/src/bench/.../ReferenceBaseline.java, with the
checksum of the ORACLE's assignment in the header, so that the reference Java class -- once a JDK exists -- is timed on the
very vectors bench.py uses and checked against the oracle in one go (one generator, files for everyone else).

    python tools/export_vectors.py --workload target --topics 2000 --out /tmp/target_2000.lav1

Layout (little endian): int32 magic "LAV1", int32 topics, int32 partitions per topic, int32 consumers per topic,
int64 checksum, int32 partition_id[T*P], int64 lag[T*P].  Lags are what auto.offset.reset=earliest gives on the workload's
offsets (the static assign(Map,Map) seam takes lags, Main.java:166).  checksum = sum over partitions of
mix(topic * P + partition id, member rank + 1) mod 2^64, mix(a, b) = (a * 0x9E3779B97F4A7C15) ^ (b * 0xBF58476D1CE4E5B9).
"""
from __future__ import annotations

import argparse
import os
import struct
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def checksum(topic_of, pid, rank, parts):
    with np.errstate(over="ignore"):
        idx = topic_of.astype(np.uint64) * np.uint64(parts) + pid.astype(np.int64).astype(np.uint64)
        z = (idx * np.uint64(0x9E3779B97F4A7C15)) ^ ((rank.astype(np.int64) + 1).astype(np.uint64) * np.uint64(0xBF58476D1CE4E5B9))
        return int(z.sum(dtype=np.uint64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="target", choices=["target", "cfg3", "cfg4"])
    ap.add_argument("--topics", type=int, default=2000)
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    from kafka_lag_based_assignor_amd import synth
    from oracle import oracle
    w = synth.config(a.workload)
    T = min(a.topics, w.n_topics)
    P, C = w.max_partitions, w.max_consumers
    n = T * P
    assert int(w.part_off[T]) == n and int(w.cons_off[T]) == T * C, "uniform workloads only"
    lag = oracle.compute_lags(w.begin[:n], w.end[:n], w.committed[:n], False)
    o_pid, o_rank, _ = oracle.assign_flat(w.part_off[:T + 1], w.partition_id[:n], lag, w.cons_off[:T + 1], w.cons_rank[:T * C])
    topic_of = np.repeat(np.arange(T, dtype=np.int64), P)
    cs = checksum(topic_of, o_pid, o_rank, P)
    with open(a.out, "wb") as fh:
        fh.write(struct.pack("<iiiiQ", 0x3156414C, T, P, C, cs))
        fh.write(np.ascontiguousarray(w.partition_id[:n], dtype="<i4").tobytes())
        fh.write(np.ascontiguousarray(lag, dtype="<i8").tobytes())
    print("%s: %d topics x %d partitions x %d consumers, checksum %d" % (a.out, T, P, C, cs))


if __name__ == "__main__":
    main()
