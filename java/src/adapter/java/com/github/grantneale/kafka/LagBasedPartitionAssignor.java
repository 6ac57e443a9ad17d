package com.github.grantneale.kafka;

import java.util.ArrayList;
import java.util.HashMap;
import java.util.List;
import java.util.Map;

import org.apache.kafka.common.TopicPartition;

import com.github.grantneale.kafka.gpu.GpuLagBasedPartitionAssignor;

/**
 * TEST ADAPTER, not product: gives the GPU host the name and the package-private surface the reference's own JUnit
 * class (src/test/java/com/github/grantneale/kafka/LagBasedPartitionAssignorTest.java) compiles against -- the two
 * statics and the nested TopicPartitionLag -- so that file can run UNCHANGED against the native path.
 * java/run_reference_tests.sh puts this directory and the reference's test directory on the test source path; the
 * reference's main class is NOT on it (this class takes its name).
 */
public class LagBasedPartitionAssignor extends GpuLagBasedPartitionAssignor {

    static Map<String, List<TopicPartition>> assign(Map<String, List<TopicPartitionLag>> partitionLagPerTopic,
                                                    Map<String, List<String>> subscriptions) {
        final Map<String, List<PartitionLag>> lags = new HashMap<>();
        for (Map.Entry<String, List<TopicPartitionLag>> e : partitionLagPerTopic.entrySet()) {
            final List<PartitionLag> list = new ArrayList<>(e.getValue().size());
            for (TopicPartitionLag tpl : e.getValue()) {
                list.add(new PartitionLag(tpl.getTopic(), tpl.getPartition(), tpl.getLag()));
            }
            lags.put(e.getKey(), list);
        }
        return assignLags(lags, subscriptions);
    }

    // computePartitionLag(OffsetAndMetadata, long, long, String): the reference's test calls it through this class's
    // name and resolves to the PUBLIC static inherited from GpuLagBasedPartitionAssignor.  (Re-declaring it here
    // package-private, as the reference does, would not compile: a hiding static may not reduce access, JLS 8.4.8.3.)

    static class TopicPartitionLag {
        private final String topic;
        private final int partition;
        private final long lag;

        TopicPartitionLag(String topic, int partition, long lag) {
            this.topic = topic;
            this.partition = partition;
            this.lag = lag;
        }

        String getTopic() {
            return topic;
        }

        int getPartition() {
            return partition;
        }

        long getLag() {
            return lag;
        }
    }
}
