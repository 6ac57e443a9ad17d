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
// table) ARE modelled since round 5 -- treeifyBin / putTreeVal / moveRootToFront / split with
// untreeify and re-treeify, restated from the published OpenJDK 8 algorithm -- and agree with the
// independent restatement in oracle/java_collections.py, but no JVM has confirmed either (there is
// none in the image): order_exact() now means "no tree bin occurred"; it turns false the moment a
// bucket treeifies, i.e. "the order below is this model's, unverified against a JVM", and callers
// may report that.
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
//
// Tree bins (round 5).  A bucket of a >= 64-slot table that reaches 9 keys through put (8 through computeIfAbsent) becomes a
// red-black tree of TreeNodes that KEEP their `next` links: iteration still walks `next`, but treeify / putTreeVal call
// moveRootToFront (the tree's root is unlinked and put first) and putTreeVal links a new node right behind its tree PARENT
// instead of at the tail -- the order depends on the tree's shape.  Restated here from OpenJDK 8's HashMap: treeifyBin,
// TreeNode.treeify / putTreeVal / balanceInsertion / rotateLeft / rotateRight / moveRootToFront / split (with untreeify at
// <= 6 nodes); inside a tree keys order by the spread hash as a signed int, then String.compareTo.  UNVERIFIED AGAINST A JVM
// (none in the image): the oracle carries an independent restatement (oracle/java_collections.py) and the tests compare the two.
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
    // HashMap.put of a NEW key (caller guarantees absence): a plain bin appends at its tail
    void put_new(int id, const std::string& key) {
        if (table_.empty()) resize();
        const int n = new_node(id, key);
        const size_t index = (table_.size() - 1) & nodes_[n].h;
        Bin& bin = table_[index];
        if (bin.tree) {
            put_tree_val(bin, n);
        } else {
            bin.list.push_back(n);
            if (bin.list.size() >= 9) treeify_bin(index);           // binCount >= TREEIFY_THRESHOLD - 1 when the 9th is appended
        }
        if (++size_ > threshold_) resize();
    }
    // HashMap.computeIfAbsent of a NEW key: resize BEFORE the insertion when size > threshold; a plain bin takes the new node
    // at its HEAD
    void compute_if_absent_new(int id, const std::string& key) {
        if (table_.empty() || size_ > threshold_) resize();
        const int n = new_node(id, key);
        const size_t index = (table_.size() - 1) & nodes_[n].h;
        Bin& bin = table_[index];
        if (bin.tree) {
            put_tree_val(bin, n);
        } else {
            const size_t bin_count = bin.list.size();
            bin.list.insert(bin.list.begin(), n);
            if (bin_count >= 7) treeify_bin(index);
        }
        ++size_;
    }
    // ids in entrySet() iteration order
    std::vector<int> order() const {
        std::vector<int> out;
        out.reserve(size_);
        for (const Bin& bin : table_) {
            if (bin.tree) {
                for (int x = bin.first; x >= 0; x = nodes_[x].next) out.push_back(nodes_[x].id);
            } else {
                for (int x : bin.list) out.push_back(nodes_[x].id);
            }
        }
        return out;
    }
    // false once some bucket became a tree bin: the order is then this file's restatement of TreeNode's list handling, which
    // no JVM has confirmed here (who-gets-what never depends on it)
    bool order_exact() const { return !treeified_; }
    bool treeified() const { return treeified_; }
    size_t size() const { return size_; }

 private:
    struct Node {
        uint32_t h = 0;                    // spread hash (compared as a Java int inside a tree)
        int id = 0;
        std::u16string key;
        int parent = -1, left = -1, right = -1, prev = -1, next = -1;
        bool red = false;
    };
    struct Bin {
        bool tree = false;
        std::vector<int> list;             // plain bin: node indices in `next` order
        int first = -1;                    // tree bin: table[index]
    };

    static uint32_t spread(int32_t hc) {
        const uint32_t h = static_cast<uint32_t>(hc);
        return h ^ (h >> 16);
    }
    int new_node(int id, const std::string& key) {
        Node n;
        n.id = id;
        n.key = utf8_to_utf16(key);
        uint32_t hc = 0;
        for (char16_t u : n.key) hc = 31u * hc + u;
        n.h = spread(static_cast<int32_t>(hc));
        nodes_.push_back(std::move(n));
        return (int)nodes_.size() - 1;
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
        std::vector<Bin> fresh(new_cap);
        for (size_t j = 0; j < old_cap; ++j) {
            Bin& bin = table_[j];
            if (!bin.tree) {
                for (int x : bin.list) fresh[(nodes_[x].h & old_cap) ? j + old_cap : j].list.push_back(x);
                continue;
            }
            // TreeNode.split: lo / hi lists in `next` order.  A half of <= 6 nodes is untreeified (plain nodes, same order); a
            // larger one is treeified AGAIN from its list -- unless the other half is empty: the tree then stays exactly as it is.
            std::vector<int> lo, hi;
            for (int x = bin.first; x >= 0; x = nodes_[x].next) ((nodes_[x].h & old_cap) ? hi : lo).push_back(x);
            auto place = [&](const std::vector<int>& part, const std::vector<int>& other, size_t at) {
                if (part.empty()) return;
                if (part.size() <= 6) {
                    fresh[at].list = part;
                } else if (!other.empty()) {
                    fresh[at].tree = true;
                    link_and_treeify(fresh[at], part);
                } else {
                    fresh[at].tree = true;
                    fresh[at].first = bin.first;
                }
            };
            place(lo, hi, j);
            place(hi, lo, j + old_cap);
        }
        table_.swap(fresh);
        threshold_ = new_thr;
    }
    void treeify_bin(size_t index) {
        if (table_.size() < 64) { resize(); return; }              // MIN_TREEIFY_CAPACITY: grow instead
        Bin& bin = table_[index];
        const std::vector<int> nodes = bin.list;
        bin.list.clear();
        bin.tree = true;
        link_and_treeify(bin, nodes);
        treeified_ = true;
    }
    // treeifyBin's relinking (same order, prev / next) + TreeNode.treeify
    void link_and_treeify(Bin& bin, const std::vector<int>& order) {
        for (size_t k = 0; k < order.size(); ++k) {
            Node& x = nodes_[order[k]];
            x.prev = k ? order[k - 1] : -1;
            x.next = k + 1 < order.size() ? order[k + 1] : -1;
        }
        bin.first = order.front();
        int root = -1;
        for (int x = bin.first, next; x >= 0; x = next) {
            next = nodes_[x].next;
            nodes_[x].left = nodes_[x].right = -1;
            if (root < 0) {
                nodes_[x].parent = -1;
                nodes_[x].red = false;
                root = x;
                continue;
            }
            for (int p = root;;) {
                const int dir = tree_dir(x, p);
                const int xp = p;
                p = dir <= 0 ? nodes_[p].left : nodes_[p].right;
                if (p < 0) {
                    nodes_[x].parent = xp;
                    (dir <= 0 ? nodes_[xp].left : nodes_[xp].right) = x;
                    root = balance_insertion(root, x);
                    break;
                }
            }
        }
        move_root_to_front(bin, root);
    }
    // which way a key goes below p: the spread hashes as Java ints, then compareComparables (String.compareTo)
    int tree_dir(int x, int p) const {
        const int32_t ph = static_cast<int32_t>(nodes_[p].h), h = static_cast<int32_t>(nodes_[x].h);
        if (ph > h) return -1;
        if (ph < h) return 1;
        return java_compare(nodes_[x].key, nodes_[p].key) < 0 ? -1 : 1;      // (equal keys never meet here)
    }
    // TreeNode.putTreeVal of a NEW key: a leaf below its tree parent xp, linked right BEHIND xp in the next list
    void put_tree_val(Bin& bin, int x) {
        int root = bin.first;
        while (nodes_[root].parent >= 0) root = nodes_[root].parent;
        for (int p = root;;) {
            const int dir = tree_dir(x, p);
            const int xp = p;
            p = dir <= 0 ? nodes_[p].left : nodes_[p].right;
            if (p < 0) {
                const int xpn = nodes_[xp].next;
                nodes_[x].next = xpn;
                (dir <= 0 ? nodes_[xp].left : nodes_[xp].right) = x;
                nodes_[xp].next = x;
                nodes_[x].parent = nodes_[x].prev = xp;
                if (xpn >= 0) nodes_[xpn].prev = x;
                move_root_to_front(bin, balance_insertion(root, x));
                return;
            }
        }
    }
    void move_root_to_front(Bin& bin, int root) {
        if (root < 0 || root == bin.first) return;
        const int first = bin.first, rn = nodes_[root].next, rp = nodes_[root].prev;
        if (rn >= 0) nodes_[rn].prev = rp;
        if (rp >= 0) nodes_[rp].next = rn;
        if (first >= 0) nodes_[first].prev = root;
        nodes_[root].next = first;
        nodes_[root].prev = -1;
        bin.first = root;
    }
    int rotate_left(int root, int p) {
        const int r = p >= 0 ? nodes_[p].right : -1;
        if (p >= 0 && r >= 0) {
            const int rl = nodes_[p].right = nodes_[r].left;
            if (rl >= 0) nodes_[rl].parent = p;
            const int pp = nodes_[r].parent = nodes_[p].parent;
            if (pp < 0) { root = r; nodes_[r].red = false; }
            else if (nodes_[pp].left == p) nodes_[pp].left = r;
            else nodes_[pp].right = r;
            nodes_[r].left = p;
            nodes_[p].parent = r;
        }
        return root;
    }
    int rotate_right(int root, int p) {
        const int l = p >= 0 ? nodes_[p].left : -1;
        if (p >= 0 && l >= 0) {
            const int lr = nodes_[p].left = nodes_[l].right;
            if (lr >= 0) nodes_[lr].parent = p;
            const int pp = nodes_[l].parent = nodes_[p].parent;
            if (pp < 0) { root = l; nodes_[l].red = false; }
            else if (nodes_[pp].right == p) nodes_[pp].right = l;
            else nodes_[pp].left = l;
            nodes_[l].right = p;
            nodes_[p].parent = l;
        }
        return root;
    }
    int balance_insertion(int root, int x) {
        nodes_[x].red = true;
        for (;;) {
            int xp = nodes_[x].parent;
            if (xp < 0) { nodes_[x].red = false; return x; }
            if (!nodes_[xp].red || nodes_[xp].parent < 0) return root;
            int xpp = nodes_[xp].parent;
            const int xppl = nodes_[xpp].left;
            if (xp == xppl) {
                const int xppr = nodes_[xpp].right;
                if (xppr >= 0 && nodes_[xppr].red) {
                    nodes_[xppr].red = false; nodes_[xp].red = false; nodes_[xpp].red = true; x = xpp;
                } else {
                    if (x == nodes_[xp].right) {
                        x = xp;
                        root = rotate_left(root, x);
                        xp = nodes_[x].parent;
                        xpp = xp < 0 ? -1 : nodes_[xp].parent;
                    }
                    if (xp >= 0) {
                        nodes_[xp].red = false;
                        if (xpp >= 0) { nodes_[xpp].red = true; root = rotate_right(root, xpp); }
                    }
                }
            } else {
                if (xppl >= 0 && nodes_[xppl].red) {
                    nodes_[xppl].red = false; nodes_[xp].red = false; nodes_[xpp].red = true; x = xpp;
                } else {
                    if (x == nodes_[xp].left) {
                        x = xp;
                        root = rotate_right(root, x);
                        xp = nodes_[x].parent;
                        xpp = xp < 0 ? -1 : nodes_[xp].parent;
                    }
                    if (xp >= 0) {
                        nodes_[xp].red = false;
                        if (xpp >= 0) { nodes_[xpp].red = true; root = rotate_left(root, xpp); }
                    }
                }
            }
        }
    }

    std::vector<Node> nodes_;
    std::vector<Bin> table_;
    size_t threshold_ = 0, size_ = 0;
    bool treeified_ = false;
};

}  // namespace kafka_lag
