// java_compat.hpp -- the JDK behaviours the reference's observable results depend on,
// re-implemented for the C++ host (the stand-in for the Java host where no JVM exists).
//
//   * String.compareTo  : UTF-16 code-unit order, then length.  Decides the greedy's last
//                         tie-break (Main.java:259) -> member ranks handed to the device.
//   * String.hashCode + HashMap iteration order: decide the ORDER in which topics are
//                         appended to each member's list (Main.java:176-184, :410-426).
//
// [upstream-knowledge] OpenJDK 8+ java.util.HashMap: power-of-two table (16, load factor
// 0.75), hash spread h ^ (h >>> 16), order-preserving lo/hi split on resize, put() appends at
// a bin's tail and resizes after insertion, computeIfAbsent() resizes before insertion and
// links the new node at the bin's HEAD.  Tree bins (>= 9 keys in one bucket of a >= 64 slot
// table) are not modelled: order_exact() turns false and callers may report it.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>

namespace kafka_lag {

inline std::u16string utf8_to_utf16(const std::string& s) {
    std::u16string out;
    out.reserve(s.size());
    const unsigned char* p = reinterpret_cast<const unsigned char*>(s.data());
    const unsigned char* e = p + s.size();
    auto cont = [&](const unsigned char* q) { return q < e && (*q & 0xC0) == 0x80; };
    while (p < e) {
        uint32_t cp;
        if (p[0] < 0x80) { cp = p[0]; p += 1; }
        else if ((p[0] & 0xE0) == 0xC0 && cont(p + 1)) { cp = ((p[0] & 0x1Fu) << 6) | (p[1] & 0x3Fu); p += 2; }
        else if ((p[0] & 0xF0) == 0xE0 && cont(p + 1) && cont(p + 2)) {
            cp = ((p[0] & 0x0Fu) << 12) | ((p[1] & 0x3Fu) << 6) | (p[2] & 0x3Fu); p += 3;
        } else if ((p[0] & 0xF8) == 0xF0 && cont(p + 1) && cont(p + 2) && cont(p + 3)) {
            cp = ((p[0] & 0x07u) << 18) | ((p[1] & 0x3Fu) << 12) | ((p[2] & 0x3Fu) << 6) | (p[3] & 0x3Fu); p += 4;
        } else { cp = 0xFFFD; p += 1; }
        if (cp >= 0x10000) {
            cp -= 0x10000;
            out.push_back(static_cast<char16_t>(0xD800 + (cp >> 10)));
            out.push_back(static_cast<char16_t>(0xDC00 + (cp & 0x3FF)));
        } else {
            out.push_back(static_cast<char16_t>(cp));
        }
    }
    return out;
}

// java.lang.String.compareTo
inline int java_compare(const std::u16string& a, const std::u16string& b) {
    const size_t lim = a.size() < b.size() ? a.size() : b.size();
    for (size_t k = 0; k < lim; ++k)
        if (a[k] != b[k]) return static_cast<int>(a[k]) - static_cast<int>(b[k]);
    return static_cast<int>(a.size()) - static_cast<int>(b.size());
}
inline int java_string_compare(const std::string& a, const std::string& b) {
    return java_compare(utf8_to_utf16(a), utf8_to_utf16(b));
}

// java.lang.String.hashCode
inline int32_t java_string_hash(const std::string& s) {
    uint32_t h = 0;
    for (char16_t u : utf8_to_utf16(s)) h = 31u * h + u;
    return static_cast<int32_t>(h);
}

// String.equalsIgnoreCase(mode, "latest"), Main.java:391.  Java folds per UTF-16 unit with
// toUpperCase then toLowerCase; besides ASCII case the only unit that folds onto a letter of
// "latest" is U+017F (long s).
inline bool equals_ignore_case_latest(const std::string& mode) {
    static const char16_t want[6] = {u'l', u'a', u't', u'e', u's', u't'};
    const std::u16string u = utf8_to_utf16(mode);
    if (u.size() != 6) return false;
    for (int k = 0; k < 6; ++k) {
        char16_t c = u[k];
        if (c >= u'A' && c <= u'Z') c = static_cast<char16_t>(c - u'A' + u'a');
        if (c == 0x017F) c = u's';
        if (c != want[k]) return false;
    }
    return true;
}

// Iteration-order model of `new HashMap<String, V>()`.  Keys are identified by a caller
// chosen int id (index into the caller's own storage); the map only tracks order.
class JavaHashMapOrder {
 public:
    JavaHashMapOrder() = default;                               // new HashMap<>()
    // new HashMap<>(initialCapacity): threshold = tableSizeFor(initialCapacity); the first resize() allocates a table
    // of that size (Main.java:216 builds consumerTotalLags with consumers.size(), duplicates included)
    explicit JavaHashMapOrder(size_t initial_capacity) {
        size_t cap = 1;
        while (cap < initial_capacity) cap <<= 1;
        threshold_ = cap;
    }
    // HashMap.put of a NEW key (caller guarantees absence)
    void put_new(int id, int32_t hash_code) {
        if (table_.empty()) resize();
        const uint32_t h = spread(hash_code);
        auto& chain = table_[(table_.size() - 1) & h];
        chain.push_back({h, id});
        if (chain.size() >= 9) treeify_bin();
        if (++size_ > threshold_) resize();
    }
    // HashMap.computeIfAbsent of a NEW key
    void compute_if_absent_new(int id, int32_t hash_code) {
        if (table_.empty() || size_ > threshold_) resize();
        const uint32_t h = spread(hash_code);
        auto& chain = table_[(table_.size() - 1) & h];
        const size_t bin_count = chain.size();
        chain.insert(chain.begin(), {h, id});
        if (bin_count >= 7) treeify_bin();
        ++size_;
    }
    // ids in entrySet() iteration order
    std::vector<int> order() const {
        std::vector<int> out;
        out.reserve(size_);
        for (const auto& chain : table_)
            for (const auto& n : chain) out.push_back(n.second);
        return out;
    }
    bool order_exact() const { return exact_; }
    size_t size() const { return size_; }

 private:
    static uint32_t spread(int32_t hc) {
        const uint32_t h = static_cast<uint32_t>(hc);
        return h ^ (h >> 16);
    }
    // HashMap.resize(): oldCap > 0 -> double (threshold doubles only from 16 slots up, else 0.75 * newCap, truncated);
    // oldCap == 0 with a threshold set by the capacity constructor -> that many slots; else 16 / 12.
    void resize() {
        const size_t old_cap = table_.size();
        size_t new_cap, new_thr = 0;
        if (old_cap > 0) {
            new_cap = old_cap * 2;
            if (old_cap >= 16) new_thr = threshold_ * 2;
        } else if (threshold_ > 0) {
            new_cap = threshold_;
        } else {
            new_cap = 16;
            new_thr = 12;
        }
        if (new_thr == 0) new_thr = (size_t)((float)new_cap * 0.75f);
        std::vector<std::vector<std::pair<uint32_t, int>>> fresh(new_cap);
        for (size_t j = 0; j < old_cap; ++j)
            for (const auto& n : table_[j]) fresh[(n.first & old_cap) ? j + old_cap : j].push_back(n);
        table_.swap(fresh);
        threshold_ = new_thr;
    }
    void treeify_bin() {
        if (table_.size() < 64) resize();
        else exact_ = false;           // a real HashMap would build a tree bin and move its root to the front
    }
    std::vector<std::vector<std::pair<uint32_t, int>>> table_;
    size_t threshold_ = 0, size_ = 0;
    bool exact_ = true;
};

}  // namespace kafka_lag
