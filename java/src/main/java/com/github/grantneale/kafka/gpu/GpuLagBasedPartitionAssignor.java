package com.github.grantneale.kafka.gpu;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.IntBuffer;
import java.nio.LongBuffer;
import java.util.ArrayList;
import java.util.Arrays;
import java.util.HashMap;
import java.util.HashSet;
import java.util.List;
import java.util.Map;
import java.util.Properties;
import java.util.Set;

import org.apache.kafka.clients.consumer.ConsumerConfig;
import org.apache.kafka.clients.consumer.ConsumerPartitionAssignor;
import org.apache.kafka.clients.consumer.KafkaConsumer;
import org.apache.kafka.clients.consumer.OffsetAndMetadata;
import org.apache.kafka.common.Cluster;
import org.apache.kafka.common.Configurable;
import org.apache.kafka.common.PartitionInfo;
import org.apache.kafka.common.TopicPartition;

/**
 * Drop-in for the reference LagBasedPartitionAssignor: same plugin surface
 * (configure / name() == "lag" / assign(Cluster, GroupSubscription)), selected with
 * {@code partition.assignment.strategy=com.github.grantneale.kafka.gpu.GpuLagBasedPartitionAssignor}.
 * The lag arithmetic, the per-topic sort and the greedy assignment run on an MI355X through
 * {@link LagAssignNative}; this class only does what is string- or container-shaped.
 *
 * SOURCE ONLY: no JDK / kafka-clients jar exists in the build image, so this file has not been
 * compiled.  The C++ class csrc/host/lag_based_partition_assignor.cpp is the tested twin of
 * this logic (same steps, same order).
 *
 * Differences from the reference that do not change results:
 *  - offsets for ALL topics are fetched with one beginningOffsets / endOffsets / committed call
 *    each, instead of three calls per topic;
 *  - there is no pure-Java arithmetic path: if the native library or the GPU is unavailable the
 *    assignor fails loudly (IllegalStateException) rather than silently computing elsewhere.
 */
public class GpuLagBasedPartitionAssignor implements ConsumerPartitionAssignor, Configurable, AutoCloseable {

    private Properties groupProps;
    private Properties sideConsumerProps;
    private KafkaConsumer<byte[], byte[]> sideConsumer;
    private long nativeCtx;

    @Override
    public void configure(Map<String, ?> configs) {
        groupProps = new Properties();
        for (Map.Entry<String, ?> e : configs.entrySet()) {
            if (e.getValue() != null) {
                groupProps.put(e.getKey(), e.getValue());
            }
        }
        String groupId = groupProps.getProperty(ConsumerConfig.GROUP_ID_CONFIG);
        if (groupId == null) {
            throw new IllegalArgumentException(ConsumerConfig.GROUP_ID_CONFIG + " is required by "
                + getClass().getName());
        }
        sideConsumerProps = new Properties();
        sideConsumerProps.putAll(groupProps);
        sideConsumerProps.put(ConsumerConfig.ENABLE_AUTO_COMMIT_CONFIG, "false");
        sideConsumerProps.put(ConsumerConfig.CLIENT_ID_CONFIG, groupId + ".assignor");
    }

    @Override
    public String name() {
        return "lag";
    }

    @Override
    public GroupAssignment assign(Cluster metadata, GroupSubscription groupSubscription) {
        // memberId -> topics, walked exactly like the reference walks its HashMap copy
        Map<String, List<String>> memberTopics = new HashMap<>();
        Set<String> allTopics = new HashSet<>();
        for (Map.Entry<String, Subscription> e : groupSubscription.groupSubscription().entrySet()) {
            memberTopics.put(e.getKey(), e.getValue().topics());
            allTopics.addAll(e.getValue().topics());
        }

        // ranks under String.compareTo: the device compares ranks, never strings
        String[] byRank = memberTopics.keySet().toArray(new String[0]);
        Arrays.sort(byRank);
        Map<String, Integer> rankOf = new HashMap<>();
        for (int r = 0; r < byRank.length; r++) {
            rankOf.put(byRank[r], r);
        }

        // topic -> subscribers, same container and fill order as the reference, so the topic
        // iteration order (and with it every member's list order) is identical
        Map<String, List<String>> consumersPerTopic = new HashMap<>();
        for (Map.Entry<String, List<String>> e : memberTopics.entrySet()) {
            for (String topic : e.getValue()) {
                consumersPerTopic.computeIfAbsent(topic, k -> new ArrayList<>()).add(e.getKey());
            }
        }

        // one offset request per kind for every partition of every topic that has metadata
        List<String> topicOrder = new ArrayList<>(consumersPerTopic.keySet());
        List<TopicPartition> all = new ArrayList<>();
        int[] partCount = new int[topicOrder.size()];
        for (int t = 0; t < topicOrder.size(); t++) {
            List<PartitionInfo> infos = metadata.partitionsForTopic(topicOrder.get(t));
            if (infos != null) {
                for (PartitionInfo p : infos) {
                    all.add(new TopicPartition(p.topic(), p.partition()));
                }
                partCount[t] = infos.size();
            }
        }
        if (sideConsumer == null) {
            sideConsumer = new KafkaConsumer<>(sideConsumerProps);
        }
        Map<TopicPartition, Long> begin = sideConsumer.beginningOffsets(all);
        Map<TopicPartition, Long> end = sideConsumer.endOffsets(all);
        Map<TopicPartition, OffsetAndMetadata> committed = sideConsumer.committed(new HashSet<>(all));

        // marshal into direct buffers (SoA, see include/lagassign.h)
        int nTopics = topicOrder.size();
        int n = all.size();
        LongBuffer partOff = longs(nTopics + 1);
        LongBuffer consOff = longs(nTopics + 1);
        IntBuffer partitionId = ints(n);
        LongBuffer beginOff = longs(n);
        LongBuffer endOff = longs(n);
        LongBuffer committedOff = longs(n);
        List<Integer> consRankList = new ArrayList<>();
        int cursor = 0;
        for (int t = 0; t < nTopics; t++) {
            partOff.put(t, cursor);
            consOff.put(t, consRankList.size());
            for (int i = 0; i < partCount[t]; i++, cursor++) {
                TopicPartition tp = all.get(cursor);
                OffsetAndMetadata md = committed.get(tp);
                partitionId.put(cursor, tp.partition());
                beginOff.put(cursor, begin.getOrDefault(tp, 0L));
                endOff.put(cursor, end.getOrDefault(tp, 0L));
                committedOff.put(cursor, md == null ? LagAssignNative.NO_COMMITTED : md.offset());
            }
            int[] ranks = consumersPerTopic.get(topicOrder.get(t)).stream().mapToInt(rankOf::get)
                .distinct().sorted().toArray();
            for (int r : ranks) {
                consRankList.add(r);
            }
        }
        partOff.put(nTopics, cursor);
        consOff.put(nTopics, consRankList.size());
        IntBuffer consRank = ints(consRankList.size());
        for (int i = 0; i < consRankList.size(); i++) {
            consRank.put(i, consRankList.get(i));
        }

        String resetMode = groupProps.getProperty(ConsumerConfig.AUTO_OFFSET_RESET_CONFIG, "latest");
        int reset = resetMode.equalsIgnoreCase("latest") ? LagAssignNative.RESET_LATEST
                                                         : LagAssignNative.RESET_EARLIEST;
        if (nativeCtx == 0) {
            nativeCtx = LagAssignNative.create(0);
        }
        int rc = LagAssignNative.assignBatch(nativeCtx, nTopics, bytes(partOff), bytes(partitionId),
            bytes(beginOff), bytes(endOff), bytes(committedOff), reset, bytes(consOff), bytes(consRank),
            null, null, null);              // the ungrouped result stays on the device
        if (rc != 0) {
            throw new IllegalStateException("liblagassign error " + rc + ": " + LagAssignNative.lastError(nativeCtx));
        }

        // member -> list: the device groups the entries it still holds by member (stable, so every list keeps the
        // reference's order: topic by topic in container order, inside a topic in assignment order,
        // Main.java:171-174 and :264); the host only wraps its own slice per member
        int nMembers = byRank.length;
        LongBuffer memberOff = longs(nMembers + 1);
        IntBuffer groupedTopic = ints(n);
        IntBuffer groupedPartition = ints(n);
        rc = LagAssignNative.groupLastByMember(nativeCtx, nMembers, bytes(memberOff), bytes(groupedTopic),
            bytes(groupedPartition));
        if (rc != 0) {
            throw new IllegalStateException("liblagassign error " + rc + ": " + LagAssignNative.lastError(nativeCtx));
        }
        Map<String, List<TopicPartition>> lists = new HashMap<>();
        for (int r = 0; r < nMembers; r++) {
            int from = (int) memberOff.get(r);
            int to = (int) memberOff.get(r + 1);
            List<TopicPartition> list = new ArrayList<>(to - from);
            for (int j = from; j < to; j++) {
                list.add(new TopicPartition(topicOrder.get(groupedTopic.get(j)), groupedPartition.get(j)));
            }
            lists.put(byRank[r], list);
        }
        Map<String, Assignment> out = new HashMap<>();
        for (Map.Entry<String, List<TopicPartition>> e : lists.entrySet()) {
            out.put(e.getKey(), new Assignment(e.getValue()));
        }
        return new GroupAssignment(out);
    }

    @Override
    public void close() {
        if (sideConsumer != null) {
            sideConsumer.close();
            sideConsumer = null;
        }
        if (nativeCtx != 0) {
            LagAssignNative.destroy(nativeCtx);
            nativeCtx = 0;
        }
    }

    // ---- direct-buffer helpers; the views keep a reference to their backing ByteBuffer ----
    private final Map<Object, ByteBuffer> backing = new java.util.IdentityHashMap<>();

    private LongBuffer longs(int n) {
        ByteBuffer b = ByteBuffer.allocateDirect(Math.max(1, n) * 8).order(ByteOrder.nativeOrder());
        LongBuffer v = b.asLongBuffer();
        backing.put(v, b);
        return v;
    }

    private IntBuffer ints(int n) {
        ByteBuffer b = ByteBuffer.allocateDirect(Math.max(1, n) * 4).order(ByteOrder.nativeOrder());
        IntBuffer v = b.asIntBuffer();
        backing.put(v, b);
        return v;
    }

    private ByteBuffer bytes(Object view) {
        return backing.get(view);
    }
}
