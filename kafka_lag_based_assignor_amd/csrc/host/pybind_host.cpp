// pybind_host.cpp -- Python binding of the C++ host mirror (tests read like the reference's
// JUnit tests).  Conversions only; no logic.
#include <pybind11/functional.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "java_compat.hpp"
#include "lag_based_partition_assignor.hpp"

namespace py = pybind11;
using namespace kafka_lag;

namespace {

using PyTpl = std::tuple<std::string, int32_t, int64_t>;         // (topic, partition, lag)
using PyTp = std::pair<std::string, int32_t>;                    // (topic, partition)

py::dict to_py(const Assignment& a) {
    py::dict out;
    for (const auto& kv : a) {
        py::list l;
        for (const auto& tp : kv.second) l.append(py::make_tuple(tp.topic, tp.partition));
        out[py::str(kv.first)] = l;
    }
    return out;
}

// OffsetSource backed by a Python object with beginning_offsets / end_offsets / committed,
// each taking a list of (topic, partition) and returning a dict keyed by (topic, partition).
struct PyOffsetSource : OffsetSource {
    py::object obj;
    explicit PyOffsetSource(py::object o) : obj(std::move(o)) {}
    static py::list keys(const std::vector<TopicPartition>& tps) {
        py::list l;
        for (const auto& tp : tps) l.append(py::make_tuple(tp.topic, tp.partition));
        return l;
    }
    std::map<TopicPartition, int64_t> longs(const char* method, const std::vector<TopicPartition>& tps) {
        std::map<TopicPartition, int64_t> out;
        py::dict d = obj.attr(method)(keys(tps));
        for (auto item : d) {
            auto k = item.first.cast<PyTp>();
            if (!item.second.is_none()) out[TopicPartition{k.first, k.second}] = item.second.cast<int64_t>();
        }
        return out;
    }
    std::map<TopicPartition, int64_t> beginningOffsets(const std::vector<TopicPartition>& tps) override {
        return longs("beginning_offsets", tps);
    }
    std::map<TopicPartition, int64_t> endOffsets(const std::vector<TopicPartition>& tps) override {
        return longs("end_offsets", tps);
    }
    std::map<TopicPartition, std::optional<OffsetAndMetadata>> committed(const std::vector<TopicPartition>& tps) override {
        std::map<TopicPartition, std::optional<OffsetAndMetadata>> out;
        py::dict d = obj.attr("committed")(keys(tps));
        for (auto item : d) {
            auto k = item.first.cast<PyTp>();
            if (item.second.is_none()) out[TopicPartition{k.first, k.second}] = std::nullopt;
            else out[TopicPartition{k.first, k.second}] = OffsetAndMetadata(item.second.cast<int64_t>());
        }
        return out;
    }
};

OrderedMap<std::vector<TopicPartitionLag>> lags_from_py(const std::vector<std::pair<std::string, std::vector<PyTpl>>>& in) {
    OrderedMap<std::vector<TopicPartitionLag>> out;
    for (const auto& kv : in) {
        std::vector<TopicPartitionLag> v;
        for (const auto& e : kv.second) v.push_back(TopicPartitionLag{std::get<0>(e), std::get<1>(e), std::get<2>(e)});
        out.emplace_back(kv.first, std::move(v));
    }
    return out;
}

}  // namespace

PYBIND11_MODULE(_host, m) {
    m.doc() = "C++ host mirror of LagBasedPartitionAssignor over liblagassign (HIP)";
    m.def("java_string_hash", &java_string_hash);
    m.def("java_string_compare", &java_string_compare);
    m.def("equals_ignore_case_latest", &equals_ignore_case_latest);
    m.def("rank_members", &rankMembers);
    m.def("consumers_per_topic_order", &consumersPerTopicOrder);
    m.def("hashmap_put_order", [](const std::vector<std::string>& keys, py::object initial_capacity) {
        JavaHashMapOrder o = initial_capacity.is_none() ? JavaHashMapOrder()
                                                        : JavaHashMapOrder(initial_capacity.cast<size_t>());
        for (size_t i = 0; i < keys.size(); ++i) o.put_new((int)i, keys[i]);
        std::vector<std::string> out;
        for (int i : o.order()) out.push_back(keys[i]);
        return py::make_tuple(out, o.order_exact());
    }, py::arg("keys"), py::arg("initial_capacity") = py::none());
    m.def("hashmap_compute_if_absent_order", [](const std::vector<std::string>& keys) {
        JavaHashMapOrder o;
        for (size_t i = 0; i < keys.size(); ++i) o.compute_if_absent_new((int)i, keys[i]);
        std::vector<std::string> out;
        for (int i : o.order()) out.push_back(keys[i]);
        return py::make_tuple(out, o.order_exact());
    }, py::arg("keys"));

    py::class_<LagBasedPartitionAssignor>(m, "LagBasedPartitionAssignor")
        .def(py::init<>())
        .def("configure", &LagBasedPartitionAssignor::configure)
        .def("name", &LagBasedPartitionAssignor::name)
        .def("metadata_consumer_props", &LagBasedPartitionAssignor::metadataConsumerProps)
        .def("last_topic_totals", &LagBasedPartitionAssignor::lastTopicTotals)
        .def("last_order_exact", &LagBasedPartitionAssignor::lastOrderExact)
        .def_static("last_static_order_exact", &LagBasedPartitionAssignor::lastStaticOrderExact)
        .def_static("last_native_call", [] {
            const auto st = LagBasedPartitionAssignor::lastNativeCall();
            py::dict d;
            d["hinted"] = st.hinted;
            d["max_lag"] = st.max_lag;
            d["max_partition_id"] = st.max_partition_id;
            d["pipeline"] = st.pipeline;
            d["launches"] = st.launches;
            return d;
        })
        .def("set_warn", [](LagBasedPartitionAssignor& self, std::function<void(const std::string&)> f) { self.warn = std::move(f); })
        .def("set_debug", [](LagBasedPartitionAssignor& self, std::function<void(const std::string&)> f) { self.debug = std::move(f); })
        .def("assign",
             [](LagBasedPartitionAssignor& self, const Cluster& metadata, const GroupSubscription& subs, py::object offsets) {
                 PyOffsetSource src(std::move(offsets));
                 return to_py(self.assign(metadata, subs, src));
             })
        .def_static("assign_static",
                    [](const std::vector<std::pair<std::string, std::vector<PyTpl>>>& lags, const GroupSubscription& subs) {
                        return to_py(LagBasedPartitionAssignor::assign(lags_from_py(lags), subs));
                    })
        .def_static("compute_partition_lag",
                    [](py::object committed, int64_t begin, int64_t end, const std::string& mode) {
                        std::optional<OffsetAndMetadata> md;
                        if (!committed.is_none()) md = OffsetAndMetadata(committed.cast<int64_t>());
                        return LagBasedPartitionAssignor::computePartitionLag(md, begin, end, mode);
                    });
}
