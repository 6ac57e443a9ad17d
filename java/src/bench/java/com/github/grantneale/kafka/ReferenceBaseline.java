package com.github.grantneale.kafka;

import java.io.DataInputStream;
import java.io.BufferedInputStream;
import java.io.FileInputStream;
import java.io.IOException;
import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.util.ArrayList;
import java.util.Arrays;
import java.util.HashMap;
import java.util.List;
import java.util.Map;

import org.apache.kafka.common.TopicPartition;

/**
 * Times the REFERENCE's own static assign(Map,Map) (LagBasedPartitionAssignor.java:166-188 of
 * grantneale/kafka-lag-based-assignor, compiled from its own source by java/run_reference_tests.sh), single-threaded, on
 * vectors written by tools/export_vectors.py -- the same SplitMix64 vectors bench.py runs on the GPU -- and prints the
 * {@code cpu_baseline} object of the bench line with {@code "kind": "reference"}.
 *
 * It lives in the reference's package because assign(Map,Map) and TopicPartitionLag are package-private there.  It is
 * measurement scaffolding: nothing in the product loads it.
 *
 * File format (little endian): int32 magic 0x3156414C ("LAV1"), int32 topics, int32 partitions per topic, int32 consumers
 * per topic, int64 expected checksum (of the oracle's assignment, 0 = none), then int32 partition id [T*P], int64 lag [T*P].
 */
public final class ReferenceBaseline {
    private ReferenceBaseline() { }

    static long mix(long index, long rankPlusOne) {
        return (index * 0x9E3779B97F4A7C15L) ^ (rankPlusOne * 0xBF58476D1CE4E5B9L);
    }

    public static void main(String[] args) throws IOException {
        if (args.length < 1) {
            System.err.println("usage: ReferenceBaseline <vectors.bin> [min seconds, default 10]");
            System.exit(2);
        }
        final double minSeconds = args.length > 1 ? Double.parseDouble(args[1]) : 10.0;
        final int topics, parts, cons;
        final long expected;
        final int[] pid;
        final long[] lag;
        try (DataInputStream in = new DataInputStream(new BufferedInputStream(new FileInputStream(args[0]), 1 << 20))) {
            byte[] head = new byte[24];
            in.readFully(head);
            ByteBuffer hb = ByteBuffer.wrap(head).order(ByteOrder.LITTLE_ENDIAN);
            if (hb.getInt() != 0x3156414C) {
                throw new IOException("not a LAV1 vectors file: " + args[0]);
            }
            topics = hb.getInt();
            parts = hb.getInt();
            cons = hb.getInt();
            expected = hb.getLong();
            final int n = Math.multiplyExact(topics, parts);
            byte[] raw = new byte[n * 4];
            in.readFully(raw);
            pid = new int[n];
            ByteBuffer.wrap(raw).order(ByteOrder.LITTLE_ENDIAN).asIntBuffer().get(pid);
            raw = new byte[n * 8];
            in.readFully(raw);
            lag = new long[n];
            ByteBuffer.wrap(raw).order(ByteOrder.LITTLE_ENDIAN).asLongBuffer().get(lag);
        }

        // memberIds "consumer-<i>": rank r is the r-th id under String.compareTo (what the GPU path's host computes)
        final String[] members = new String[cons];
        for (int i = 0; i < cons; ++i) {
            members[i] = "consumer-" + i;
        }
        Arrays.sort(members);
        final Map<String, Integer> rankOf = new HashMap<>();
        for (int r = 0; r < cons; ++r) {
            rankOf.put(members[r], r);
        }
        final String[] topicName = new String[topics];
        final Map<String, Integer> topicIndex = new HashMap<>();
        for (int t = 0; t < topics; ++t) {
            topicName[t] = "topic-" + t;
            topicIndex.put(topicName[t], t);
        }

        long checksum = 0;
        long calls = 0;
        double seconds = 0;
        boolean first = true;
        // one warm-up pass (JIT), then whole passes until the time budget is spent; inputs are rebuilt per pass because
        // assign sorts the caller's lists in place (LagBasedPartitionAssignor.java:228)
        while (first || seconds < minSeconds) {
            final Map<String, List<LagBasedPartitionAssignor.TopicPartitionLag>> lags = new HashMap<>();
            for (int t = 0; t < topics; ++t) {
                final List<LagBasedPartitionAssignor.TopicPartitionLag> l = new ArrayList<>(parts);
                for (int i = 0; i < parts; ++i) {
                    l.add(new LagBasedPartitionAssignor.TopicPartitionLag(topicName[t], pid[t * parts + i], lag[t * parts + i]));
                }
                lags.put(topicName[t], l);
            }
            final Map<String, List<String>> subscriptions = new HashMap<>();
            for (String m : members) {
                subscriptions.put(m, new ArrayList<>(Arrays.asList(topicName)));
            }
            final long t0 = System.nanoTime();
            final Map<String, List<TopicPartition>> out = LagBasedPartitionAssignor.assign(lags, subscriptions);
            final long t1 = System.nanoTime();
            if (first) {
                for (Map.Entry<String, List<TopicPartition>> e : out.entrySet()) {
                    final long r1 = rankOf.get(e.getKey()) + 1L;
                    for (TopicPartition tp : e.getValue()) {
                        checksum += mix((long) topicIndex.get(tp.topic()) * parts + tp.partition(), r1);
                    }
                }
                first = false;
            } else {
                seconds += (t1 - t0) * 1e-9;
                ++calls;
            }
        }
        final long n = (long) topics * parts;
        final String match = expected == 0 ? "null" : Boolean.toString(expected == checksum);
        System.out.println("{\"cpu_baseline\": {\"value\": " + String.format("%.1f", n * calls / seconds)
                + ", \"unit\": \"partition-assignments/sec\", \"cores\": 1, \"kind\": \"reference\", \"sample\": \""
                + topics + " topics x " + parts + " partitions x " + cons + " consumers of the bench workload ("
                + args[0] + "), the reference's own static assign(Map,Map), single thread, " + calls + " passes, "
                + String.format("%.1f", seconds) + " s after one JIT warm-up pass\", \"host_cpus\": "
                + Runtime.getRuntime().availableProcessors() + ", \"java\": \"" + System.getProperty("java.version")
                + "\", \"assignment_checksum_matches_oracle\": " + match + "}}");
        if (expected != 0 && expected != checksum) {
            System.exit(1);
        }
    }
}
