// lag_based_partition_assignor.cpp -- see the header.  Host-side container/string work only;
// all arithmetic goes through the C ABI (include/lagassign.h) to the HIP kernels.
#include "lag_based_partition_assignor.hpp"

#include <algorithm>
#include <functional>
#include <mutex>
#include <numeric>
#include <unordered_map>

#include "../../../include/lagassign.h"
#include "java_compat.hpp"

namespace kafka_lag {

namespace {

// One process-wide native context for the static entry points (the reference's statics carry
// no state either); guarded because la_ctx is single-threaded.
std::mutex g_ctx_mutex;
la_ctx* g_ctx = nullptr;

la_ctx* shared_ctx_locked() {
    if (!g_ctx) {
        int rc = la_create_multi(&g_ctx, 0, nullptr, 0);      // every GPU of the node, like the Java host
        if (rc != LA_OK) {
            g_ctx = nullptr;
            throw std::runtime_error(std::string("liblagassign: ") + la_last_error(nullptr));
        }
    }
    return g_ctx;
}

// Whether the list order of the last static assign() on this thread is the modelled HashMap's exact order
// (false: a bucket of consumersPerTopic reached tree-bin size: the order is java_compat.hpp's restatement of TreeNode's list
// handling, unverified against a JVM).
thread_local bool t_last_order_exact = true;
thread_local LagBasedPartitionAssignor::NativeCallStats t_last_native;

void check(la_ctx* ctx, int rc) {
    if (rc == LA_OK) return;
    const std::string msg = std::string("liblagassign error ") + std::to_string(rc) + ": " + la_last_error(ctx);
    if (rc == LA_EINVAL) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}

// The subscription side of one rebalance, resolved to ranks and topic order.
struct Plan {
    std::vector<std::string> members;              // unique memberIds, caller's iteration order
    std::vector<int32_t> rank_of_member;           // String.compareTo rank of members[i]
    std::vector<int> member_of_rank;
    std::vector<std::string> topics;               // consumersPerTopic.entrySet() order
    std::vector<std::vector<int32_t>> topic_ranks; // per topic: sorted unique subscriber ranks
    std::vector<std::vector<int>> topic_members;   // per topic: subscriber member indices in consumers-list order
    bool order_exact = true;
};

Plan make_plan(const GroupSubscription& subscriptions) {
    Plan p;
    std::unordered_map<std::string, int> member_index;
    std::vector<const std::vector<std::string>*> member_topics;
    for (const auto& kv : subscriptions) {                      // a Map has unique keys: last wins
        auto it = member_index.find(kv.first);
        if (it == member_index.end()) {
            member_index.emplace(kv.first, (int)p.members.size());
            p.members.push_back(kv.first);
            member_topics.push_back(&kv.second);
        } else {
            member_topics[it->second] = &kv.second;
        }
    }
    p.rank_of_member = rankMembers(p.members);
    p.member_of_rank.assign(p.members.size(), 0);
    for (size_t i = 0; i < p.members.size(); ++i) p.member_of_rank[p.rank_of_member[i]] = (int)i;

    // consumersPerTopic, Main.java:410-426: HashMap + computeIfAbsent, walked in entry order
    JavaHashMapOrder order;
    std::unordered_map<std::string, int> topic_index;
    std::vector<std::string> names;
    std::vector<std::vector<int32_t>> ranks;
    std::vector<std::vector<int>> members_of;
    for (size_t m = 0; m < p.members.size(); ++m) {
        for (const std::string& topic : *member_topics[m]) {
            auto it = topic_index.find(topic);
            int ti;
            if (it == topic_index.end()) {
                ti = (int)names.size();
                topic_index.emplace(topic, ti);
                names.push_back(topic);
                ranks.emplace_back();
                members_of.emplace_back();
                order.compute_if_absent_new(ti, topic);
            } else {
                ti = it->second;
            }
            ranks[ti].push_back(p.rank_of_member[m]);          // duplicates allowed here ...
            members_of[ti].push_back((int)m);
        }
    }
    for (int ti : order.order()) {
        std::vector<int32_t> r = ranks[ti];
        std::sort(r.begin(), r.end());
        r.erase(std::unique(r.begin(), r.end()), r.end());     // ... the keyed bins de-dup, Main.java:216-225
        p.topics.push_back(names[ti]);
        p.topic_ranks.push_back(std::move(r));
        p.topic_members.push_back(members_of[ti]);
    }
    p.order_exact = order.order_exact();
    return p;
}

// Per-topic partition data in SoA form, ready for the C ABI.
struct TopicData {
    std::vector<int32_t> partition;
    std::vector<std::string> element_topic;   // TopicPartitionLag.topic of each element (Main.java:264)
    std::vector<int64_t> lag;                 // lag mode
    std::vector<int64_t> begin, end, committed;   // offset mode
};

struct Flat {
    std::vector<int64_t> part_off{0}, cons_off{0};
    std::vector<int32_t> pid, cons_rank;
    std::vector<int64_t> lag, end, committed;
    // the beginning offset is read only where a partition has no committed offset (Main.java:384-396): it crosses the
    // boundary as (position, begin) pairs for those partitions alone (la_assign_batch_grouped_sparse)
    std::vector<int64_t> none_index, none_begin;
    // what the marshalling loop sees on its way (Main.java:344-356 walks every partition's offsets): the largest end offset /
    // lag and partition id, and whether any offset, lag or id is negative -- la_hint_next_call's bounds
    int64_t max_value = 0, max_id = 0;
    bool any_negative = false;
    void see(int64_t value, int32_t id) {
        if (value > max_value) max_value = value;
        if (id > max_id) max_id = id;
        any_negative |= (value < 0) | (id < 0);
    }
};

// The bounds of the next call, when the marshaller can vouch for them: with no negative offset a lag never exceeds its end
// offset (computePartitionLag, Main.java:376-404: end - committed / end - begin / 0, clamped at 0).  One launch per chunk of
// tile-sized topics instead of two; a caller that saw a negative value promises nothing.
void hint_bounds(la_ctx* ctx, const Flat& f) {
    t_last_native = LagBasedPartitionAssignor::NativeCallStats{};
    if (f.any_negative || f.pid.empty()) return;
    la_call_hints h{};
    h.struct_size = (int32_t)sizeof h;
    h.flags = LA_HINT_BOUNDS;
    h.max_lag = f.max_value;
    h.max_partition_id = f.max_id;
    if (la_hint_next_call(ctx, &h) != LA_OK) return;          // a refused hint is no hint: the call itself decides
    t_last_native.hinted = true;
    t_last_native.max_lag = h.max_lag;
    t_last_native.max_partition_id = h.max_partition_id;
}

Assignment run_native(const Plan& plan, const std::vector<const TopicData*>& data, bool offsets_mode,
                      int32_t reset_mode, std::map<std::string, std::map<std::string, int64_t>>* totals_out,
                      const std::function<void(const std::string&)>* debug = nullptr) {
    Flat f;
    for (size_t t = 0; t < plan.topics.size(); ++t) {
        const TopicData* d = data[t];
        if (d) {
            f.pid.insert(f.pid.end(), d->partition.begin(), d->partition.end());
            if (offsets_mode) {
                const int64_t base = (int64_t)f.end.size();
                for (size_t i = 0; i < d->committed.size(); ++i) {
                    f.see(d->end[i], d->partition[i]);
                    if (d->committed[i] < 0) {                        // partitionMetadata == null, Main.java:384
                        f.none_index.push_back(base + (int64_t)i);
                        f.none_begin.push_back(d->begin[i]);
                        f.any_negative |= d->begin[i] < 0;
                    }
                }
                f.end.insert(f.end.end(), d->end.begin(), d->end.end());
                f.committed.insert(f.committed.end(), d->committed.begin(), d->committed.end());
            } else {
                for (size_t i = 0; i < d->lag.size(); ++i) f.see(d->lag[i], d->partition[i]);
                f.lag.insert(f.lag.end(), d->lag.begin(), d->lag.end());
            }
        }
        f.part_off.push_back((int64_t)f.pid.size());
        f.cons_rank.insert(f.cons_rank.end(), plan.topic_ranks[t].begin(), plan.topic_ranks[t].end());
        f.cons_off.push_back((int64_t)f.cons_rank.size());
    }
    const size_t n = f.pid.size(), k = f.cons_rank.size();
    // The assignment stays on the device (out_partition = out_member_rank = NULL) and crosses PCIe once, already
    // grouped: every member's list in the reference's order (topic by topic as consumersPerTopic iterates, inside a
    // topic in assignment order, Main.java:171-174 and :264); the host only wraps it.
    std::vector<int64_t> out_total(k);
    const int32_t n_members = (int32_t)plan.members.size();
    std::vector<int64_t> member_off((size_t)n_members + 1, 0);
    std::vector<int32_t> grouped_topic(n), grouped_pid(n);
    if (!plan.topics.empty()) {
        std::lock_guard<std::mutex> lock(g_ctx_mutex);       // one lock over both calls: the second reads the first's results
        la_ctx* ctx = shared_ctx_locked();
        hint_bounds(ctx, f);
        if (offsets_mode) {
            // assign(Cluster, GroupSubscription): both steps in ONE native call -- for a rebalance of ordinary size one
            // upload, one download, one wait (la_assign_batch_grouped_sparse: `begin` only where it is read)
            check(ctx, la_assign_batch_grouped_sparse(ctx, (int32_t)plan.topics.size(), f.part_off.data(), f.pid.data(),
                                                      f.end.data(), f.committed.data(), reset_mode,
                                                      (int64_t)f.none_index.size(), f.none_index.data(), f.none_begin.data(),
                                                      f.cons_off.data(), f.cons_rank.data(), n_members, member_off.data(),
                                                      grouped_topic.data(), grouped_pid.data(), out_total.data()));
        } else {
            check(ctx, la_assign_batch_lags(ctx, (int32_t)plan.topics.size(), f.part_off.data(), f.pid.data(), f.lag.data(),
                                            f.cons_off.data(), f.cons_rank.data(), nullptr, nullptr, out_total.data()));
            t_last_native.pipeline = la_last_pipeline(ctx);
            t_last_native.launches = la_last_launches(ctx);
            check(ctx, la_group_last_by_member(ctx, n_members, member_off.data(), grouped_topic.data(), grouped_pid.data()));
            t_last_native.launches += la_last_launches(ctx);
        }
        if (offsets_mode) {
            t_last_native.pipeline = la_last_pipeline(ctx);
            t_last_native.launches = la_last_launches(ctx);
        }
    }
    // partition id -> the element's own topic string (normally the map key), per topic
    std::vector<std::unordered_map<int32_t, const std::string*>> topic_of(plan.topics.size());
    for (size_t t = 0; t < plan.topics.size(); ++t)
        if (const TopicData* d = data[t])
            for (size_t i = 0; i < d->partition.size(); ++i) topic_of[t].emplace(d->partition[i], &d->element_topic[i]);
    Assignment assignment;
    for (int32_t r = 0; r < n_members; ++r) {
        auto& list = assignment[plan.members[plan.member_of_rank[r]]];     // a list for EVERY member, :171-174
        for (int64_t j = member_off[r]; j < member_off[r + 1]; ++j)
            list.push_back(TopicPartition{*topic_of[grouped_topic[j]].at(grouped_pid[j]), grouped_pid[j]});   // :264
    }
    if (debug && *debug) {
        // Main.java:279-306.  consumerTotalLags is a HashMap filled with put() in consumers-list order (:216-225).
        for (size_t t = 0; t < plan.topics.size(); ++t) {
            JavaHashMapOrder order(plan.topic_members[t].size());          // new HashMap<>(consumers.size()), :216
            std::vector<int> uniq;
            for (int m : plan.topic_members[t])
                if (std::find(uniq.begin(), uniq.end(), m) == uniq.end()) {
                    order.put_new((int)uniq.size(), plan.members[m]);
                    uniq.push_back(m);
                }
            std::string summary;
            for (int u : order.order()) {
                const int m = uniq[u];
                const int32_t r = plan.rank_of_member[m];
                int64_t total = 0;
                for (int64_t c = f.cons_off[t]; c < f.cons_off[t + 1]; ++c)
                    if (f.cons_rank[c] == r) total = out_total[c];
                summary += "\t" + plan.members[m] + " (total_lag=" + std::to_string(total) + ")\n";
                for (int64_t j = member_off[r]; j < member_off[r + 1]; ++j) {
                    if ((size_t)grouped_topic[j] > t) break;               // the map as it stood after this topic
                    summary += "\t\t" + *topic_of[grouped_topic[j]].at(grouped_pid[j]) + "-" +
                               std::to_string(grouped_pid[j]) + "\n";    // TopicPartition.toString()
                }
            }
            (*debug)("Assignment for " + plan.topics[t] + ":\n" + summary);
        }
    }
    if (totals_out) {
        for (size_t t = 0; t < plan.topics.size(); ++t) {
            auto& per = (*totals_out)[plan.topics[t]];
            for (int64_t c = f.cons_off[t]; c < f.cons_off[t + 1]; ++c)
                per[plan.members[plan.member_of_rank[f.cons_rank[c]]]] = out_total[c];
        }
    }
    return assignment;
}

}  // namespace

// String.compareTo ranks (0 = smallest).  Equal ids share a rank slot order by position, but
// callers pass unique ids.
std::vector<int32_t> rankMembers(const std::vector<std::string>& memberIds) {
    std::vector<std::u16string> u;
    u.reserve(memberIds.size());
    for (const auto& m : memberIds) u.push_back(utf8_to_utf16(m));
    std::vector<int> idx(memberIds.size());
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return java_compare(u[a], u[b]) < 0; });
    std::vector<int32_t> rank(memberIds.size());
    for (size_t r = 0; r < idx.size(); ++r) rank[idx[r]] = (int32_t)r;
    return rank;
}

std::vector<std::string> consumersPerTopicOrder(const GroupSubscription& subscriptions) {
    return make_plan(subscriptions).topics;
}

bool LagBasedPartitionAssignor::lastStaticOrderExact() { return t_last_order_exact; }
LagBasedPartitionAssignor::NativeCallStats LagBasedPartitionAssignor::lastNativeCall() { return t_last_native; }

LagBasedPartitionAssignor::LagBasedPartitionAssignor() = default;
LagBasedPartitionAssignor::~LagBasedPartitionAssignor() = default;

void LagBasedPartitionAssignor::configure(const std::map<std::string, std::string>& configs) {
    consumer_group_props_ = configs;                                             // Main.java:101-104
    auto gid = consumer_group_props_.find("group.id");
    if (gid == consumer_group_props_.end())                                      // :107-113
        throw std::invalid_argument("group.id cannot be null when using partition.assignment.strategy=" +
                                    std::string("LagBasedPartitionAssignor"));
    metadata_consumer_props_ = consumer_group_props_;                            // :116-120
    metadata_consumer_props_["enable.auto.commit"] = "false";
    metadata_consumer_props_["client.id"] = gid->second + ".assignor";
}

int64_t LagBasedPartitionAssignor::computePartitionLag(const std::optional<OffsetAndMetadata>& partitionMetadata,
                                                       int64_t beginOffset, int64_t endOffset,
                                                       const std::string& autoOffsetResetMode) {
    const int64_t committed = partitionMetadata ? partitionMetadata->offset : LA_NO_COMMITTED;
    const int32_t mode = equals_ignore_case_latest(autoOffsetResetMode) ? LA_RESET_LATEST : LA_RESET_EARLIEST;
    int64_t lag = 0;
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    la_ctx* ctx = shared_ctx_locked();
    check(ctx, la_compute_lag(ctx, 1, &beginOffset, &endOffset, &committed, mode, &lag));
    return lag;
}

Assignment LagBasedPartitionAssignor::assign(const OrderedMap<std::vector<TopicPartitionLag>>& partitionLagPerTopic,
                                             const GroupSubscription& subscriptions) {
    const Plan plan = make_plan(subscriptions);
    std::unordered_map<std::string, TopicData> by_topic;
    for (const auto& kv : partitionLagPerTopic) {
        TopicData d;
        for (const TopicPartitionLag& e : kv.second) {
            d.partition.push_back(e.partition);
            d.element_topic.push_back(e.topic);
            d.lag.push_back(e.lag);
        }
        by_topic[kv.first] = std::move(d);                                       // later duplicate key wins
    }
    std::vector<const TopicData*> data;
    for (const std::string& t : plan.topics) {
        auto it = by_topic.find(t);
        data.push_back(it == by_topic.end() ? nullptr : &it->second);            // getOrDefault(..., emptyList()), :182
    }
    t_last_order_exact = plan.order_exact;
    return run_native(plan, data, false, LA_RESET_LATEST, nullptr);
}

Assignment LagBasedPartitionAssignor::assign(const Cluster& metadata, const GroupSubscription& subscriptions,
                                             OffsetSource& offsets) {
    // Everything is cold when a rebalance comes (Kafka calls the leader's assign() minutes apart): la_wake runs a one-partition
    // rebalance through the real path NOW (~95 us, cold) -- the three broker round trips below (Main.java:147 -> :317-365)
    // take milliseconds before there is an offset to hand over, and the call that matters then runs warm (at the C ABI a
    // 100-partition call 34 us instead of 85, profiles/r06_p_cold_c.txt).  Best effort: a failure here is the assign call's to report.
    {
        std::lock_guard<std::mutex> lock(g_ctx_mutex);
        try { (void)la_wake(shared_ctx_locked()); } catch (const std::exception&) {}
    }
    // topicSubscriptions is a HashMap<memberId, topics> filled with put (Main.java:141-146);
    // the static assign then walks it in HashMap order.
    GroupSubscription hashed;
    {
        JavaHashMapOrder order;
        std::unordered_map<std::string, int> seen;
        std::vector<const std::pair<std::string, std::vector<std::string>>*> entries;
        for (const auto& kv : subscriptions) {
            auto it = seen.find(kv.first);
            if (it == seen.end()) {
                seen.emplace(kv.first, (int)entries.size());
                order.put_new((int)entries.size(), kv.first);
                entries.push_back(&kv);
            } else {
                entries[it->second] = &kv;
            }
        }
        for (int i : order.order()) hashed.push_back(*entries[i]);
        last_order_exact_ = order.order_exact();
    }
    const Plan plan = make_plan(hashed);
    last_order_exact_ = last_order_exact_ && plan.order_exact;
    if (!last_order_exact_)
        warn("a HashMap bucket reached tree-bin size (>= 9 colliding keys in a >= 64-slot table): the list order follows the "
             "C++ host's restatement of java.util.HashMap's tree bins, which no JVM has confirmed here; the partition -> "
             "member map is unaffected");

    // readTopicPartitionLags, Main.java:317-365 -- batched: one request per kind for all topics
    std::vector<TopicPartition> all;
    std::unordered_map<std::string, TopicData> by_topic;
    for (const std::string& topic : plan.topics) {
        auto it = metadata.find(topic);
        if (it == metadata.end() || it->second.empty()) {
            warn("Skipping assignment for topic " + topic + " since no metadata is available");   // :359
            continue;
        }
        TopicData d;
        for (int32_t p : it->second) {
            d.partition.push_back(p);
            d.element_topic.push_back(topic);
            all.push_back(TopicPartition{topic, p});
        }
        by_topic[topic] = std::move(d);
    }
    const auto begin = offsets.beginningOffsets(all);                            // :339
    const auto end = offsets.endOffsets(all);                                    // :340
    const auto committed = offsets.committed(all);                               // :342
    for (auto& kv : by_topic) {
        TopicData& d = kv.second;
        for (int32_t p : d.partition) {
            const TopicPartition tp{kv.first, p};
            auto b = begin.find(tp);
            auto e = end.find(tp);
            auto c = committed.find(tp);
            d.begin.push_back(b == begin.end() ? 0 : b->second);                 // getOrDefault(.., 0L), :350
            d.end.push_back(e == end.end() ? 0 : e->second);                     // :351
            d.committed.push_back(c == committed.end() || !c->second ? LA_NO_COMMITTED : c->second->offset);
        }
    }
    auto mode_it = consumer_group_props_.find("auto.offset.reset");              // default "latest", :346-347
    const std::string mode = mode_it == consumer_group_props_.end() ? "latest" : mode_it->second;
    const int32_t reset = equals_ignore_case_latest(mode) ? LA_RESET_LATEST : LA_RESET_EARLIEST;

    std::vector<const TopicData*> data;
    for (const std::string& t : plan.topics) {
        auto it = by_topic.find(t);
        data.push_back(it == by_topic.end() ? nullptr : &it->second);
    }
    last_totals_.clear();
    return run_native(plan, data, true, reset, &last_totals_, &debug);           // wrap: :152-156
}

}  // namespace kafka_lag
