package com.github.grantneale.kafka.gpu;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.IntBuffer;
import java.nio.LongBuffer;
import java.util.ArrayList;
import java.util.Arrays;
import java.util.Collections;
import java.util.HashMap;
import java.util.HashSet;
import java.util.List;
import java.util.Map;
import java.util.Properties;

import org.apache.kafka.clients.consumer.ConsumerConfig;
import org.apache.kafka.clients.consumer.ConsumerPartitionAssignor;
import org.apache.kafka.clients.consumer.KafkaConsumer;
import org.apache.kafka.clients.consumer.OffsetAndMetadata;
import org.apache.kafka.common.Cluster;
import org.apache.kafka.common.Configurable;
import org.apache.kafka.common.PartitionInfo;
import org.apache.kafka.common.TopicPartition;
import org.slf4j.Logger;
import org.slf4j.LoggerFactory;

/**
 * Drop-in for the reference LagBasedPartitionAssignor: same plugin surface
 * (configure / name() == "lag" / assign(Cluster, GroupSubscription)), selected with
 * {@code partition.assignment.strategy=com.github.grantneale.kafka.gpu.GpuLagBasedPartitionAssignor}.
 * The lag arithmetic, the per-topic sort and the greedy assignment run on the MI355Xs of the node through
 * {@link LagAssignNative}; this class only does what is string- or container-shaped.
 *
 * <p>NOT COMPILED in the image this repository is built in (no JDK, no kafka-clients jar there).
 * {@code java/run_reference_tests.sh} compiles it and runs the reference's own JUnit class against it wherever a JDK
 * exists; until then the C++ class csrc/host/lag_based_partition_assignor.cpp is the tested twin of this logic (same
 * steps, same order, same log text).
 *
 * <p>Differences from the reference that do not change results:
 * <ul>
 *  <li>offsets for ALL topics are fetched with one beginningOffsets / endOffsets / committed call each, instead of
 *      three calls per topic (LagBasedPartitionAssignor.java:339-342);</li>
 *  <li>the native context spans every GPU of the node (la_create_multi): topics are split into contiguous ranges
 *      balanced by partition count, one per device, and every shard's results are copied straight to their offset in
 *      this class's buffers -- that is the reassembly of the global assignment; no RCCL collective is involved because
 *      nothing on a device ever needs another device's topics (DESIGN.md section 5);</li>
 *  <li>there is no pure-Java arithmetic path.  On a native failure (no GPU, a HIP error) the assignor throws
 *      IllegalStateException out of assign() -- which propagates out of KafkaConsumer.poll() on the group leader --
 *      unless a fallback takes that rebalance over: the ConsumerPartitionAssignor {@code lag.assignor.fallback.class}
 *      names, or -- when the property is unset -- the reference class itself if it is on the class path (looked up by
 *      name; nothing of it is linked).  See INTEGRATION.md section 3.</li>
 * </ul>
 */
public class GpuLagBasedPartitionAssignor implements ConsumerPartitionAssignor, Configurable, AutoCloseable {

    private static final Logger LOGGER = LoggerFactory.getLogger(GpuLagBasedPartitionAssignor.class);

    /** Consumer property: FQCN of the ConsumerPartitionAssignor that takes over a rebalance the GPU path failed. */
    public static final String FALLBACK_CLASS_CONFIG = "lag.assignor.fallback.class";
    /** What takes over when {@link #FALLBACK_CLASS_CONFIG} is unset and this class is on the class path: the reference. */
    public static final String DEFAULT_FALLBACK_CLASS = "com.github.grantneale.kafka.LagBasedPartitionAssignor";
    /** Consumer property: comma-separated HIP device ids (default: every device of the node). */
    public static final String DEVICES_CONFIG = "lag.assignor.devices";

    /** (topic, partition, lag): what the reference calls TopicPartitionLag (LagBasedPartitionAssignor.java:431-455). */
    public static final class PartitionLag {
        final String topic;
        final int partition;
        final long lag;

        public PartitionLag(String topic, int partition, long lag) {
            this.topic = topic;
            this.partition = partition;
            this.lag = lag;
        }
    }

    private Properties groupProps;
    private Properties sideConsumerProps;
    private KafkaConsumer<byte[], byte[]> sideConsumer;
    private ConsumerPartitionAssignor fallback;
    private boolean defaultFallbackTried;
    private Map<String, ?> rawConfigs;
    private final Engine engine = new Engine();

    // ------------------------------------------------------------------------------------------------------------
    // plugin surface
    // ------------------------------------------------------------------------------------------------------------

    @Override
    public void configure(Map<String, ?> configs) {
        rawConfigs = configs;
        groupProps = new Properties();
        for (Map.Entry<String, ?> e : configs.entrySet()) {
            if (e.getValue() != null) {          // Properties rejects nulls; the reference would throw NPE here
                groupProps.put(e.getKey(), e.getValue());
            }
        }
        final String groupId = groupProps.getProperty(ConsumerConfig.GROUP_ID_CONFIG);
        if (groupId == null) {
            throw new IllegalArgumentException(
                ConsumerConfig.GROUP_ID_CONFIG + " cannot be null when using "
                    + ConsumerConfig.PARTITION_ASSIGNMENT_STRATEGY_CONFIG + "=" + getClass().getName());
        }
        sideConsumerProps = new Properties();
        sideConsumerProps.putAll(groupProps);
        sideConsumerProps.put(ConsumerConfig.ENABLE_AUTO_COMMIT_CONFIG, "false");
        final String clientId = groupId + ".assignor";
        sideConsumerProps.put(ConsumerConfig.CLIENT_ID_CONFIG, clientId);
        // the two plugin-only keys mean nothing to a KafkaConsumer
        sideConsumerProps.remove(FALLBACK_CLASS_CONFIG);
        sideConsumerProps.remove(DEVICES_CONFIG);
        engine.devices = parseDevices(groupProps.getProperty(DEVICES_CONFIG));

        LOGGER.debug(
            "Configured LagBasedPartitionAssignor with values:\n"
                + "\tgroup.id = {}\n"
                + "\tclient.id = {}\n",
            groupId,
            clientId
        );
    }

    @Override
    public String name() {
        return "lag";
    }

    @Override
    public GroupAssignment assign(Cluster metadata, GroupSubscription groupSubscription) {
        try {
            return assignOnGpu(metadata, groupSubscription);
        } catch (NativeAssignException | LinkageError nativeFailure) {     // la_* error, or the library itself is missing;
            // anything the side KafkaConsumer throws (closed, fenced, timed out) is NOT a GPU failure and propagates
            final ConsumerPartitionAssignor other = fallback();
            if (other == null) {
                throw nativeFailure;
            }
            LOGGER.warn("GPU assignment failed ({}); this rebalance is delegated to {}", nativeFailure.toString(),
                other.getClass().getName());
            return other.assign(metadata, groupSubscription);
        }
    }

    private GroupAssignment assignOnGpu(Cluster metadata, GroupSubscription groupSubscription) {
        // memberId -> topics, in the same container the reference copies them into (its iteration order is what the
        // static assign walks)
        final Map<String, List<String>> topicSubscriptions = new HashMap<>();
        for (Map.Entry<String, Subscription> e : groupSubscription.groupSubscription().entrySet()) {
            topicSubscriptions.put(e.getKey(), e.getValue().topics());
        }
        final Plan plan = new Plan(topicSubscriptions);

        // readTopicPartitionLags, batched: ONE request per kind for every partition of every topic with metadata
        final int nTopics = plan.topics.size();
        final List<TopicPartition> all = new ArrayList<>();
        final int[] partCount = new int[nTopics];
        for (int t = 0; t < nTopics; t++) {
            final String topic = plan.topics.get(t);
            final List<PartitionInfo> infos = metadata.partitionsForTopic(topic);
            if (infos != null && !infos.isEmpty()) {
                for (PartitionInfo p : infos) {
                    all.add(new TopicPartition(p.topic(), p.partition()));
                }
                partCount[t] = infos.size();
            } else {
                LOGGER.warn("Skipping assignment for topic {} since no metadata is available", topic);
            }
        }
        if (sideConsumer == null) {
            sideConsumer = new KafkaConsumer<>(sideConsumerProps);
        }
        final Map<TopicPartition, Long> begin = sideConsumer.beginningOffsets(all);
        final Map<TopicPartition, Long> end = sideConsumer.endOffsets(all);
        final Map<TopicPartition, OffsetAndMetadata> committed = sideConsumer.committed(new HashSet<>(all));

        if (nTopics == 0) {                        // nobody subscribes to anything: every member gets an empty list
            final Map<String, Assignment> none = new HashMap<>();
            for (String memberId : plan.byRank) {
                none.put(memberId, new Assignment(new ArrayList<TopicPartition>()));
            }
            return new GroupAssignment(none);
        }

        // marshal (SoA, include/lagassign.h) into the engine's grow-only pinned buffers
        final int n = all.size();
        engine.open();
        engine.reserve(nTopics, n, plan.totalConsumers);
        final LongBuffer partOff = engine.partOff.longs;
        final LongBuffer consOff = engine.consOff.longs;
        final IntBuffer partitionId = engine.partitionId.ints;
        final LongBuffer beginOff = engine.begin.longs;
        final LongBuffer endOff = engine.end.longs;
        final LongBuffer committedOff = engine.committed.longs;
        final IntBuffer consRank = engine.consRank.ints;
        // The beginning offset is read only where there is no committed offset (Main.java:384-396): those partitions are
        // listed as (position, begin) pairs and the dense array stays on this side of PCIe (la_assign_batch_grouped_sparse).
        final LongBuffer noneIndex = engine.noneIndex.longs;
        final LongBuffer noneBegin = engine.noneBegin.longs;
        long nNone = 0;
        int cursor = 0;
        int k = 0;
        // what this loop sees on its way (the reference's own loop walks the same offsets, LagBasedPartitionAssignor.java:
        // 344-356): the largest end offset and partition id, and whether anything is negative -- la_hint_next_call's bounds
        long maxEnd = 0;
        int maxPartition = 0;
        boolean anyNegative = false;
        final boolean trace = LOGGER.isTraceEnabled();        // only the trace path reads the dense begin array
        for (int t = 0; t < nTopics; t++) {
            partOff.put(t, cursor);
            consOff.put(t, k);
            for (int i = 0; i < partCount[t]; i++, cursor++) {
                final TopicPartition tp = all.get(cursor);
                final OffsetAndMetadata md = committed.get(tp);
                final long b = begin.getOrDefault(tp, 0L);
                final long e = end.getOrDefault(tp, 0L);
                partitionId.put(cursor, tp.partition());
                if (trace) {
                    beginOff.put(cursor, b);
                }
                endOff.put(cursor, e);
                committedOff.put(cursor, md == null ? LagAssignNative.NO_COMMITTED : md.offset());
                maxEnd = Math.max(maxEnd, e);
                maxPartition = Math.max(maxPartition, tp.partition());
                anyNegative |= e < 0 || tp.partition() < 0;
                if (md == null) {
                    noneIndex.put((int) nNone, cursor);
                    noneBegin.put((int) nNone, b);
                    nNone++;
                    anyNegative |= b < 0;
                }
            }
            final int[] ranks = plan.topicRanks.get(t);
            for (int r : ranks) {
                consRank.put(k++, r);
            }
        }
        partOff.put(nTopics, cursor);
        consOff.put(nTopics, k);

        final String resetMode = groupProps.getProperty(ConsumerConfig.AUTO_OFFSET_RESET_CONFIG, "latest");
        final int reset = resetMode.equalsIgnoreCase("latest") ? LagAssignNative.RESET_LATEST
                                                               : LagAssignNative.RESET_EARLIEST;
        final Map<String, List<TopicPartition>> lists;
        if (!anyNegative && n > 0 && engine.hasHints) {
            // with no negative offset a lag never exceeds its end offset: the bounds prove, for the usual offsets and ids, that
            // every tile's records pack into 64 bits, and the tile path is one launch per chunk instead of two.  (A refused hint
            // is no hint; a violated one would fail the call with LA_EINVAL -- it cannot be: the loop above saw every value.)
            LagAssignNative.hintNextCallBounds(engine.ctx, maxEnd, maxPartition);
        }
        if (trace) {
            // the trace lines need the ungrouped arrays as well: two native calls
            engine.check(LagAssignNative.assignBatch(engine.ctx, nTopics, engine.partOff.bytes, engine.partitionId.bytes,
                engine.begin.bytes, engine.end.bytes, engine.committed.bytes, reset, engine.consOff.bytes,
                engine.consRank.bytes, engine.outPartition.bytes, engine.outMemberRank.bytes, engine.outTotal.bytes));
            lists = engine.memberLists(plan, n);
        } else {
            // normally the ungrouped result never leaves the device, and assignment + every member's list are ONE native
            // call: for a rebalance of ordinary size no copy at all and ONE kernel launch (la_assign_batch_grouped_sparse)
            final int nMembers = plan.byRank.length;
            engine.memberOff.ensure(engine, 8L * (nMembers + 1));
            engine.check(LagAssignNative.assignBatchGroupedSparse(engine.ctx, nTopics, engine.partOff.bytes,
                engine.partitionId.bytes, engine.end.bytes, engine.committed.bytes, reset, nNone,
                engine.noneIndex.bytes, engine.noneBegin.bytes, engine.consOff.bytes, engine.consRank.bytes, nMembers,
                engine.memberOff.bytes, engine.groupedTopic.bytes, engine.groupedPartition.bytes, engine.outTotal.bytes));
            lists = engine.wrapLists(plan);
        }
        if (trace) {
            // per-partition lags for the trace lines, from the device as well (la_compute_lag on the same buffers)
            engine.check(LagAssignNative.computeLag(engine.ctx, n, engine.begin.bytes, engine.end.bytes,
                engine.committed.bytes, reset, engine.lag.bytes));
            engine.logTrace(plan, nTopics);
        }
        if (LOGGER.isDebugEnabled()) {
            engine.logDebugSummaries(plan, nTopics);
        }

        final Map<String, Assignment> assignments = new HashMap<>();
        for (Map.Entry<String, List<TopicPartition>> e : lists.entrySet()) {
            assignments.put(e.getKey(), new Assignment(e.getValue()));
        }
        return new GroupAssignment(assignments);
    }

    @Override
    public void close() {
        if (sideConsumer != null) {
            sideConsumer.close();
            sideConsumer = null;
        }
        engine.close();
        if (fallback instanceof AutoCloseable) {
            try {
                ((AutoCloseable) fallback).close();
            } catch (Exception e) {
                LOGGER.debug("closing the fallback assignor", e);
            }
        }
    }

    // ------------------------------------------------------------------------------------------------------------
    // the seams the reference's own tests use (package-private statics there: LagBasedPartitionAssignor.java:166, :376)
    // ------------------------------------------------------------------------------------------------------------

    private static final Engine SHARED = new Engine();      // the statics carry no state in the reference either

    /** static computePartitionLag (LagBasedPartitionAssignor.java:376-404), computed on the device. */
    public static synchronized long computePartitionLag(OffsetAndMetadata partitionMetadata, long beginOffset,
                                                        long endOffset, String autoOffsetResetMode) {
        final Engine e = SHARED;
        e.open();
        e.reserve(0, 1, 0);
        e.begin.longs.put(0, beginOffset);
        e.end.longs.put(0, endOffset);
        e.committed.longs.put(0, partitionMetadata == null ? LagAssignNative.NO_COMMITTED : partitionMetadata.offset());
        final int reset = autoOffsetResetMode.equalsIgnoreCase("latest") ? LagAssignNative.RESET_LATEST
                                                                         : LagAssignNative.RESET_EARLIEST;
        e.check(LagAssignNative.computeLag(e.ctx, 1, e.begin.bytes, e.end.bytes, e.committed.bytes, reset, e.lag.bytes));
        return e.lag.longs.get(0);
    }

    /**
     * static assign(Map, Map) (LagBasedPartitionAssignor.java:166-188): every member gets a list; topics are walked in
     * consumersPerTopic order; a topic without a lag entry contributes nothing.
     */
    public static synchronized Map<String, List<TopicPartition>> assignLags(
        Map<String, List<PartitionLag>> partitionLagPerTopic, Map<String, List<String>> subscriptions) {
        final Engine e = SHARED;
        final Plan plan = new Plan(subscriptions);
        final int nTopics = plan.topics.size();
        if (nTopics == 0) {
            final Map<String, List<TopicPartition>> none = new HashMap<>();
            for (String memberId : plan.byRank) {
                none.put(memberId, new ArrayList<TopicPartition>());
            }
            return none;
        }
        int n = 0;
        for (String topic : plan.topics) {
            n += partitionLagPerTopic.getOrDefault(topic, Collections.<PartitionLag>emptyList()).size();
        }
        e.open();
        e.reserve(nTopics, n, plan.totalConsumers);
        // the element's own topic string ends up in the TopicPartition (:264): remember it per entry
        final String[] elementTopic = new String[n];
        int cursor = 0;
        int k = 0;
        for (int t = 0; t < nTopics; t++) {
            e.partOff.longs.put(t, cursor);
            e.consOff.longs.put(t, k);
            for (PartitionLag pl : partitionLagPerTopic.getOrDefault(plan.topics.get(t),
                                                                     Collections.<PartitionLag>emptyList())) {
                e.partitionId.ints.put(cursor, pl.partition);
                e.lag.longs.put(cursor, pl.lag);
                elementTopic[cursor++] = pl.topic;
            }
            for (int r : plan.topicRanks.get(t)) {
                e.consRank.ints.put(k++, r);
            }
        }
        e.partOff.longs.put(nTopics, cursor);
        e.consOff.longs.put(nTopics, k);
        e.check(LagAssignNative.assignBatchLags(e.ctx, nTopics, e.partOff.bytes, e.partitionId.bytes, e.lag.bytes,
            e.consOff.bytes, e.consRank.bytes, null, null, e.outTotal.bytes));
        e.elementTopic = elementTopic;
        try {
            return e.memberLists(plan, n);
        } finally {
            e.elementTopic = null;
        }
    }

    // ------------------------------------------------------------------------------------------------------------
    // subscriptions -> ranks and topic order (strings and containers only)
    // ------------------------------------------------------------------------------------------------------------

    static final class Plan {
        final String[] byRank;                      // memberIds under String.compareTo
        final List<String> topics;                  // consumersPerTopic.keySet() iteration order
        final List<List<String>> topicConsumers;    // per topic: the consumers list as the reference builds it
        final List<int[]> topicRanks;               // per topic: ascending unique ranks of its subscribers
        final int totalConsumers;

        Plan(Map<String, List<String>> subscriptions) {
            byRank = subscriptions.keySet().toArray(new String[0]);
            Arrays.sort(byRank);                    // the device compares ranks, never strings
            final Map<String, Integer> rankOf = new HashMap<>(2 * byRank.length);
            for (int r = 0; r < byRank.length; r++) {
                rankOf.put(byRank[r], r);
            }
            // same container and fill order as the reference (LagBasedPartitionAssignor.java:410-426), so the topic
            // iteration order -- and with it every member's list order -- is the JVM's own
            final Map<String, List<String>> consumersPerTopic = new HashMap<>();
            for (Map.Entry<String, List<String>> e : subscriptions.entrySet()) {
                for (String topic : e.getValue()) {
                    consumersPerTopic.computeIfAbsent(topic, key -> new ArrayList<>()).add(e.getKey());
                }
            }
            topics = new ArrayList<>(consumersPerTopic.size());
            topicConsumers = new ArrayList<>(consumersPerTopic.size());
            topicRanks = new ArrayList<>(consumersPerTopic.size());
            int total = 0;
            for (Map.Entry<String, List<String>> e : consumersPerTopic.entrySet()) {
                final List<String> consumers = e.getValue();
                final int[] ranks = new int[consumers.size()];
                for (int i = 0; i < ranks.length; i++) {
                    ranks[i] = rankOf.get(consumers.get(i));
                }
                Arrays.sort(ranks);
                int unique = 0;                     // a member that lists a topic twice appears twice (:418); the
                for (int i = 0; i < ranks.length; i++) {   // reference's keyed bins de-duplicate (:216-225)
                    if (i == 0 || ranks[i] != ranks[i - 1]) {
                        ranks[unique++] = ranks[i];
                    }
                }
                topics.add(e.getKey());
                topicConsumers.add(consumers);
                topicRanks.add(unique == ranks.length ? ranks : Arrays.copyOf(ranks, unique));
                total += unique;
            }
            totalConsumers = total;
        }
    }

    // ------------------------------------------------------------------------------------------------------------
    // native context + buffers: created once, grow-only, reused by every rebalance
    // ------------------------------------------------------------------------------------------------------------

    /**
     * A failure of the native library (an la_* error code, no usable device, a marshalling array too large).  It is an
     * IllegalStateException, so that without a fallback class it leaves assign() exactly as before; assign() delegates to
     * the fallback ONLY for this type and for LinkageError -- never for what the side KafkaConsumer throws.
     */
    public static final class NativeAssignException extends IllegalStateException {
        private static final long serialVersionUID = 1L;

        NativeAssignException(String message) {
            super(message);
        }
    }

    /** One pinned (la_host_alloc) direct buffer with its typed views; grows, never shrinks. */
    private static final class Buf {
        ByteBuffer bytes;
        LongBuffer longs;
        IntBuffer ints;
        boolean pinned;

        void ensure(Engine owner, long wantBytes) {
            if (bytes != null && bytes.capacity() >= wantBytes) {
                return;
            }
            if (wantBytes > Integer.MAX_VALUE) {    // a ByteBuffer's capacity is an int: fail before marshalling, not in it
                throw new NativeAssignException("a marshalling array of " + wantBytes + " bytes exceeds the 2^31-1 bytes a "
                    + "direct ByteBuffer can hold (" + (wantBytes / 8) + " partitions or consumer entries in one rebalance)");
            }
            release(owner);
            final long cap = Math.max(64, Math.min((long) Integer.MAX_VALUE, wantBytes + wantBytes / 4));
            ByteBuffer b = LagAssignNative.hostAlloc(owner.ctx, cap);
            pinned = b != null;
            if (b == null) {                        // pinned memory exhausted: pageable works too, only slower
                b = ByteBuffer.allocateDirect((int) cap);
            }
            bytes = b.order(ByteOrder.nativeOrder());
            longs = bytes.asLongBuffer();
            ints = bytes.asIntBuffer();
        }

        void release(Engine owner) {
            if (bytes != null && pinned) {
                LagAssignNative.hostFree(owner.ctx, bytes);
            }
            bytes = null;
            longs = null;
            ints = null;
        }
    }

    private static final class Engine {
        long ctx;
        int[] devices;                              // null: every device of the node
        String[] elementTopic;                      // assignLags only: topic string of every input entry
        final Buf partOff = new Buf();
        final Buf consOff = new Buf();
        final Buf partitionId = new Buf();
        final Buf begin = new Buf();
        final Buf noneIndex = new Buf();
        final Buf noneBegin = new Buf();
        final Buf end = new Buf();
        final Buf committed = new Buf();
        final Buf lag = new Buf();
        final Buf consRank = new Buf();
        final Buf outPartition = new Buf();
        final Buf outMemberRank = new Buf();
        final Buf outTotal = new Buf();
        final Buf memberOff = new Buf();
        final Buf groupedTopic = new Buf();
        final Buf groupedPartition = new Buf();

        boolean hasHints;

        void open() {
            if (ctx == 0) {
                // the sparse-begin entry point exists since ABI 0.3.0 (include/lagassign.h: LA_VERSION)
                final int version = LagAssignNative.version();
                if (version < 300) {
                    throw new NativeAssignException("liblagassign ABI " + version + " is older than the 300 this host binds");
                }
                hasHints = version >= 400;                       // la_hint_next_call exists since ABI 0.4.0
                try {
                    ctx = LagAssignNative.createMulti(devices);  // the shim throws IllegalStateException(la_last_error)
                } catch (IllegalStateException e) {
                    throw new NativeAssignException(e.getMessage());
                }
                LOGGER.debug("liblagassign context over {} device(s) of {}", LagAssignNative.shardCount(ctx),
                    LagAssignNative.deviceCount());
            }
        }

        void reserve(int nTopics, int n, int k) {
            partOff.ensure(this, 8L * (nTopics + 1));
            consOff.ensure(this, 8L * (nTopics + 1));
            partitionId.ensure(this, 4L * n);
            begin.ensure(this, 8L * n);
            noneIndex.ensure(this, 8L * n);
            noneBegin.ensure(this, 8L * n);
            end.ensure(this, 8L * n);
            committed.ensure(this, 8L * n);
            lag.ensure(this, 8L * n);
            consRank.ensure(this, 4L * k);
            outPartition.ensure(this, 4L * n);
            outMemberRank.ensure(this, 4L * n);
            outTotal.ensure(this, 8L * k);
            groupedTopic.ensure(this, 4L * n);
            groupedPartition.ensure(this, 4L * n);
        }

        void check(int rc) {
            if (rc != 0) {
                throw new NativeAssignException("liblagassign error " + rc + ": " + LagAssignNative.lastError(ctx));
            }
        }

        /**
         * member -> list.  The device groups the entries it still holds by member (stable, so every list keeps the
         * reference's order: topic by topic in container order, inside a topic in assignment order,
         * LagBasedPartitionAssignor.java:171-174 and :264); the host only wraps its own slice per member.
         */
        Map<String, List<TopicPartition>> memberLists(Plan plan, int n) {
            final int nMembers = plan.byRank.length;
            memberOff.ensure(this, 8L * (nMembers + 1));
            check(LagAssignNative.groupLastByMember(ctx, nMembers, memberOff.bytes, groupedTopic.bytes,
                groupedPartition.bytes));
            return wrapLists(plan);
        }

        /** The grouped arrays (memberOff / groupedTopic / groupedPartition, filled by the native side) as member -> list. */
        Map<String, List<TopicPartition>> wrapLists(Plan plan) {
            final int nMembers = plan.byRank.length;
            // topic -> (partition -> the entry's own topic string) is only needed by the static seam, where an entry
            // may carry a topic string that differs from its map key
            final Map<String, List<TopicPartition>> lists = new HashMap<>();
            for (int r = 0; r < nMembers; r++) {
                final int from = (int) memberOff.longs.get(r);
                final int to = (int) memberOff.longs.get(r + 1);
                final List<TopicPartition> list = new ArrayList<>(to - from);
                for (int j = from; j < to; j++) {
                    final int t = groupedTopic.ints.get(j);
                    final int p = groupedPartition.ints.get(j);
                    list.add(new TopicPartition(topicString(plan, t, p), p));
                }
                lists.put(plan.byRank[r], list);    // a list for EVERY member (:171-174)
            }
            return lists;
        }

        private String topicString(Plan plan, int t, int partition) {
            if (elementTopic == null) {
                return plan.topics.get(t);
            }
            final int from = (int) partOff.longs.get(t);
            final int to = (int) partOff.longs.get(t + 1);
            for (int i = from; i < to; i++) {       // the static seam is the tests' seam: topics are small there
                if (partitionId.ints.get(i) == partition) {
                    return elementTopic[i];
                }
            }
            return plan.topics.get(t);
        }

        /**
         * LOGGER.debug of LagBasedPartitionAssignor.java:279-306: per topic, the consumers in consumerTotalLags'
         * iteration order (a HashMap with initial capacity consumers.size(), filled in list order, :216-219), each with
         * its total lag for this topic and every partition it holds SO FAR (the reference prints the cumulative map).
         */
        void logDebugSummaries(Plan plan, int nTopics) {
            final int nMembers = plan.byRank.length;
            final Map<String, Integer> rankOf = new HashMap<>(2 * nMembers);
            for (int r = 0; r < nMembers; r++) {
                rankOf.put(plan.byRank[r], r);
            }
            for (int t = 0; t < nTopics; t++) {
                final List<String> consumers = plan.topicConsumers.get(t);
                final int[] ranks = plan.topicRanks.get(t);
                final int k0 = (int) consOff.longs.get(t);
                final Map<String, Long> consumerTotalLags = new HashMap<>(consumers.size());
                for (String memberId : consumers) {
                    consumerTotalLags.put(memberId, outTotal.longs.get(k0 + Arrays.binarySearch(ranks, rankOf.get(memberId))));
                }
                final StringBuilder topicSummary = new StringBuilder();
                for (Map.Entry<String, Long> entry : consumerTotalLags.entrySet()) {
                    topicSummary.append(String.format("\t%s (total_lag=%d)\n", entry.getKey(), entry.getValue()));
                    final int r = rankOf.get(entry.getKey());
                    final int to = (int) memberOff.longs.get(r + 1);
                    for (int j = (int) memberOff.longs.get(r); j < to && groupedTopic.ints.get(j) <= t; j++) {
                        topicSummary.append(String.format("\t\t%s\n",
                            new TopicPartition(plan.topics.get(groupedTopic.ints.get(j)), groupedPartition.ints.get(j))));
                    }
                }
                LOGGER.debug("Assignment for {}:\n{}", plan.topics.get(t), topicSummary);
            }
        }

        /**
         * LOGGER.trace of LagBasedPartitionAssignor.java:268-275, one line per assignment in assignment order.  The
         * lags come from the device (la_compute_lag); the running per-consumer totals are re-accumulated here, with
         * Java's wrapping long addition, only to be printed.
         */
        void logTrace(Plan plan, int nTopics) {
            final long[] running = new long[plan.byRank.length];
            for (int t = 0; t < nTopics; t++) {
                Arrays.fill(running, 0L);
                final int from = (int) partOff.longs.get(t);
                final int to = (int) partOff.longs.get(t + 1);
                // lag of a partition id of this topic: the inputs are in metadata order, the outputs in assignment order
                final Map<Integer, Long> lagOf = new HashMap<>(2 * (to - from));
                for (int i = from; i < to; i++) {
                    lagOf.put(partitionId.ints.get(i), lag.longs.get(i));
                }
                for (int i = from; i < to; i++) {
                    final int r = outMemberRank.ints.get(i);
                    if (r < 0) {
                        continue;                   // the topic has no consumers (:211-213)
                    }
                    final int p = outPartition.ints.get(i);
                    final long partitionLag = lagOf.get(p);
                    running[r] += partitionLag;
                    LOGGER.trace(
                        "Assigned partition {}-{} to consumer {}.  partition_lag={}, consumer_current_total_lag={}",
                        plan.topics.get(t), p, plan.byRank[r], partitionLag, running[r]);
                }
            }
        }

        void close() {
            if (ctx != 0) {
                for (Buf b : new Buf[] {partOff, consOff, partitionId, begin, noneIndex, noneBegin, end, committed, lag, consRank,
                                        outPartition, outMemberRank, outTotal, memberOff, groupedTopic,
                                        groupedPartition}) {
                    b.release(this);
                }
                LagAssignNative.destroy(ctx);
                ctx = 0;
            }
        }
    }

    // ------------------------------------------------------------------------------------------------------------

    private static int[] parseDevices(String spec) {
        if (spec == null || spec.trim().isEmpty()) {
            return null;
        }
        final String[] parts = spec.split(",");
        final int[] ids = new int[parts.length];
        for (int i = 0; i < parts.length; i++) {
            ids[i] = Integer.parseInt(parts[i].trim());
        }
        return ids;
    }

    private ConsumerPartitionAssignor fallback() {
        if (fallback == null) {
            String cls = groupProps == null ? null : groupProps.getProperty(FALLBACK_CLASS_CONFIG);
            if (cls != null && cls.trim().equalsIgnoreCase("none")) {
                return null;            // explicit opt-out: a native failure propagates out of assign() (ADVICE r5)
            }
            if (cls == null || cls.trim().isEmpty()) {
                // A drop-in must not break a rebalance on a GPU fault (SURVEY 8b, "errors"): with no class configured, the
                // reference itself takes over when it is on the class path -- found by name, nothing of it is linked here.
                if (defaultFallbackTried) {
                    return null;
                }
                defaultFallbackTried = true;
                try {
                    final Class<?> found =
                        Class.forName(DEFAULT_FALLBACK_CLASS, false, GpuLagBasedPartitionAssignor.class.getClassLoader());
                    if (GpuLagBasedPartitionAssignor.class.isAssignableFrom(found)) {
                        // the test adapter (java/src/adapter) carries the reference's name and IS this host: no way out there
                        return null;
                    }
                } catch (ClassNotFoundException | LinkageError absent) {
                    LOGGER.info("{} is unset and {} is not on the class path: a native failure will propagate out of assign()",
                        FALLBACK_CLASS_CONFIG, DEFAULT_FALLBACK_CLASS);
                    return null;
                }
                LOGGER.info("{} is unset: {} (found on the class path) takes over a rebalance the GPU path fails",
                    FALLBACK_CLASS_CONFIG, DEFAULT_FALLBACK_CLASS);
                cls = DEFAULT_FALLBACK_CLASS;
            }
            try {
                final Object o = Class.forName(cls.trim()).getDeclaredConstructor().newInstance();
                if (o instanceof Configurable) {
                    ((Configurable) o).configure(rawConfigs);
                }
                fallback = (ConsumerPartitionAssignor) o;
            } catch (ReflectiveOperationException | ClassCastException e) {
                throw new IllegalStateException(FALLBACK_CLASS_CONFIG + "=" + cls + " cannot be used", e);
            }
        }
        return fallback;
    }
}
