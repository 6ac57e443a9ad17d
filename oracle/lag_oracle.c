/*
 * lag_oracle.c -- TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * A literal, single-threaded CPU restatement of the hot path of
 * grantneale/kafka-lag-based-assignor v2.0.0, used as the parity checker for the
 * HIP kernels.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library.  The product path (kafka_lag_based_assignor_amd/)
 * never links, loads or calls anything in this directory.
 *
 * "Main.java:N" below =
 *   /root/reference/src/main/java/com/github/grantneale/kafka/LagBasedPartitionAssignor.java:N
 *
 * Parity pinning: the reference itself cannot be run here (no JVM in the image),
 * so this restatement is pinned against every known-answer vector in the
 * reference's own tests (LagBasedPartitionAssignorTest.java:21-228) and the README
 * worked example (README.md:42-57); see tests/test_oracle_golden.py.
 *
 * Java semantics that matter and how they are restated:
 *   - `long` arithmetic wraps (Main.java:265, :402)  -> uint64_t add/sub, cast back.
 *   - Long.compare / Integer.compare are signed        -> plain signed compares.
 *   - List.sort is a stable merge sort (TimSort)       -> stable top-down merge sort.
 *   - String.compareTo compares UTF-16 code units      -> lao_java_string_compare().
 *   - HashMap<String,..> keyed by memberId de-duplicates a consumer that appears
 *     twice in a topic's consumer list (Main.java:216-225) -> explicit de-dup.
 *   - Collections.min returns the first minimal element; the comparator is a total
 *     order over distinct memberIds, so iteration order cannot change the answer.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define LAO_EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* String.compareTo (UTF-16 code-unit order) on UTF-8 input.                  */

/* Decode UTF-8 into UTF-16 code units.  Returns the number of units written.
 * Malformed bytes decode to U+FFFD like Java's decoder does. */
static int utf8_to_utf16(const char *s, uint16_t *out, int cap) {
    const unsigned char *p = (const unsigned char *)s;
    int n = 0;
    while (*p && n + 2 <= cap) {
        uint32_t cp;
        if (p[0] < 0x80) { cp = p[0]; p += 1; }
        else if ((p[0] & 0xE0) == 0xC0 && (p[1] & 0xC0) == 0x80) {
            cp = ((uint32_t)(p[0] & 0x1F) << 6) | (p[1] & 0x3F); p += 2;
        } else if ((p[0] & 0xF0) == 0xE0 && (p[1] & 0xC0) == 0x80 && (p[2] & 0xC0) == 0x80) {
            cp = ((uint32_t)(p[0] & 0x0F) << 12) | ((uint32_t)(p[1] & 0x3F) << 6) | (p[2] & 0x3F); p += 3;
        } else if ((p[0] & 0xF8) == 0xF0 && (p[1] & 0xC0) == 0x80 && (p[2] & 0xC0) == 0x80 &&
                   (p[3] & 0xC0) == 0x80) {
            cp = ((uint32_t)(p[0] & 0x07) << 18) | ((uint32_t)(p[1] & 0x3F) << 12) |
                 ((uint32_t)(p[2] & 0x3F) << 6) | (p[3] & 0x3F); p += 4;
        } else { cp = 0xFFFD; p += 1; }
        if (cp >= 0x10000) {
            cp -= 0x10000;
            out[n++] = (uint16_t)(0xD800 + (cp >> 10));
            out[n++] = (uint16_t)(0xDC00 + (cp & 0x3FF));
        } else {
            out[n++] = (uint16_t)cp;
        }
    }
    return n;
}

/* java.lang.String.compareTo: first differing UTF-16 unit decides, else length.
 * Used by the greedy comparator's third level, Main.java:259. */
LAO_EXPORT int lao_java_string_compare(const char *a_utf8, const char *b_utf8) {
    size_t ca = 2 * strlen(a_utf8) + 2, cb = 2 * strlen(b_utf8) + 2;
    uint16_t *a = (uint16_t *)malloc(ca * sizeof(uint16_t));
    uint16_t *b = (uint16_t *)malloc(cb * sizeof(uint16_t));
    int la = utf8_to_utf16(a_utf8, a, (int)ca), lb = utf8_to_utf16(b_utf8, b, (int)cb);
    int lim = la < lb ? la : lb, r = la - lb;
    for (int k = 0; k < lim; ++k)
        if (a[k] != b[k]) { r = (int)a[k] - (int)b[k]; break; }
    free(a); free(b);
    return r;
}

/* java.lang.String.hashCode: s[0]*31^(n-1) + ... over UTF-16 units, int wrap. */
LAO_EXPORT int32_t lao_java_string_hash(const char *s_utf8) {
    size_t cap = 2 * strlen(s_utf8) + 2;
    uint16_t *u = (uint16_t *)malloc(cap * sizeof(uint16_t));
    int n = utf8_to_utf16(s_utf8, u, (int)cap);
    uint32_t h = 0;
    for (int k = 0; k < n; ++k) h = 31u * h + u[k];
    free(u);
    return (int32_t)h;
}

/* ------------------------------------------------------------------------- */
/* computePartitionLag, Main.java:376-404.                                    */

/* String.equalsIgnoreCase(mode, "latest") (Main.java:391), restated for the only
 * right-hand side that matters.  Java folds each char pair with toUpperCase then
 * toLowerCase; the only non-ASCII code unit that folds onto a letter of "latest"
 * is U+017F LATIN SMALL LETTER LONG S (upper-cases to 'S'). */
static int equals_ignore_case_latest(const char *mode_utf8) {
    static const uint16_t want[6] = {'l', 'a', 't', 'e', 's', 't'};
    uint16_t u[64];
    if (strlen(mode_utf8) > 24) return 0;
    int n = utf8_to_utf16(mode_utf8, u, 64);
    if (n != 6) return 0;
    for (int k = 0; k < 6; ++k) {
        uint16_t c = u[k];
        if (c >= 'A' && c <= 'Z') c = (uint16_t)(c - 'A' + 'a');
        if (c == 0x017F) c = 's';
        if (c != want[k]) return 0;
    }
    return 1;
}

/* has_committed == 0 restates `partitionMetadata == null` (Main.java:384). */
LAO_EXPORT int64_t lao_compute_partition_lag(int has_committed, int64_t committed_offset,
                                             int64_t begin_offset, int64_t end_offset,
                                             const char *auto_offset_reset_mode) {
    int64_t next_offset;
    if (has_committed) {
        next_offset = committed_offset;                               /* Main.java:386 */
    } else if (equals_ignore_case_latest(auto_offset_reset_mode)) {
        next_offset = end_offset;                                     /* Main.java:391-392 */
    } else {
        next_offset = begin_offset;                                   /* Main.java:393-396 */
    }
    int64_t d = (int64_t)((uint64_t)end_offset - (uint64_t)next_offset); /* wrapping long */
    return d > 0 ? d : 0;                                             /* Main.java:402 */
}

/* Vector form used for large parity runs: committed < 0 stands for "no committed
 * offset" (OffsetAndMetadata rejects negative offsets, so -1 is free), reset_latest
 * is the already-evaluated equalsIgnoreCase("latest"). */
LAO_EXPORT void lao_compute_lags(int64_t n, const int64_t *begin, const int64_t *end,
                                 const int64_t *committed, int reset_latest, int64_t *out_lag) {
    for (int64_t i = 0; i < n; ++i) {
        int64_t next = committed[i] >= 0 ? committed[i]
                       : (reset_latest ? end[i] : (begin ? begin[i] : 0));
        int64_t d = (int64_t)((uint64_t)end[i] - (uint64_t)next);
        out_lag[i] = d > 0 ? d : 0;
    }
}

/* ------------------------------------------------------------------------- */
/* assignTopic, Main.java:204-266.                                            */

typedef struct { int32_t partition; int64_t lag; } tpl_t;   /* TopicPartitionLag, Main.java:431-455 */

/* The sort comparator, Main.java:228-235. */
static int cmp_partition(const tpl_t *p1, const tpl_t *p2) {
    if (p1->lag == p2->lag)
        return (p1->partition > p2->partition) - (p1->partition < p2->partition);
    return (p2->lag > p1->lag) - (p2->lag < p1->lag);                 /* Long.compare(p2, p1) */
}

/* Stable merge sort == List.sort's contract (TimSort is stable). */
static void merge_sort(tpl_t *a, tpl_t *tmp, int64_t n) {
    if (n < 2) return;
    int64_t h = n / 2;
    merge_sort(a, tmp, h);
    merge_sort(a + h, tmp, n - h);
    int64_t i = 0, j = h, k = 0;
    while (i < h && j < n) tmp[k++] = (cmp_partition(&a[j], &a[i]) < 0) ? a[j++] : a[i++];
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, (size_t)n * sizeof(tpl_t));
}

/* Third comparator level: either real memberId strings or precomputed ranks. */
typedef struct {
    const char *const *member_ids;   /* NULL => compare ranks numerically */
} member_cmp_t;

static int cmp_member(const member_cmp_t *mc, int32_t m1, int32_t m2) {
    if (mc->member_ids) return lao_java_string_compare(mc->member_ids[m1], mc->member_ids[m2]);
    return (m1 > m2) - (m1 < m2);
}

/* One topic.  consumers[0..nc) are member handles (index into member_ids, or a
 * String.compareTo rank when member_ids == NULL); duplicates allowed.
 * out_partition / out_member [np]: the (partition, chosen member) pairs in the order
 * the reference appends them (Main.java:264).  out_total [nc]: final
 * consumerTotalLags value of consumers[i] (what the debug summary prints,
 * Main.java:283-291).  Returns 0, or -1 on allocation failure. */
static int assign_topic(const member_cmp_t *mc, int64_t np, const int32_t *partition,
                        const int64_t *lag, int64_t nc, const int32_t *consumers,
                        int32_t *out_partition, int32_t *out_member, int64_t *out_total) {
    if (nc == 0) {                                                    /* Main.java:211-213 */
        /* The reference returns before sorting and assigns nothing.  The flat ABI still
         * has to fill the segment: member = -1, partitions in (lag desc, id asc) order. */
        tpl_t *s0 = (tpl_t *)malloc((size_t)(np ? np : 1) * sizeof(tpl_t));
        tpl_t *t0 = (tpl_t *)malloc((size_t)(np ? np : 1) * sizeof(tpl_t));
        if (!s0 || !t0) { free(s0); free(t0); return -1; }
        for (int64_t i = 0; i < np; ++i) { s0[i].partition = partition[i]; s0[i].lag = lag[i]; }
        merge_sort(s0, t0, np);
        for (int64_t i = 0; i < np; ++i) { out_partition[i] = s0[i].partition; out_member[i] = -1; }
        free(s0); free(t0);
        return 0;
    }
    /* consumerTotalLags / consumerTotalPartitions, Main.java:216-225; keyed maps de-dup */
    int32_t *uniq = (int32_t *)malloc((size_t)nc * sizeof(int32_t));
    int64_t *total = (int64_t *)calloc((size_t)nc, sizeof(int64_t));
    int32_t *count = (int32_t *)calloc((size_t)nc, sizeof(int32_t));
    tpl_t *sorted = (tpl_t *)malloc((size_t)(np ? np : 1) * sizeof(tpl_t));
    tpl_t *tmp = (tpl_t *)malloc((size_t)(np ? np : 1) * sizeof(tpl_t));
    if (!uniq || !total || !count || !sorted || !tmp) {
        free(uniq); free(total); free(count); free(sorted); free(tmp);
        return -1;
    }
    int64_t nu = 0;
    for (int64_t i = 0; i < nc; ++i) {
        int64_t j = 0;
        while (j < nu && uniq[j] != consumers[i]) ++j;
        if (j == nu) uniq[nu++] = consumers[i];
    }
    for (int64_t i = 0; i < np; ++i) { sorted[i].partition = partition[i]; sorted[i].lag = lag[i]; }
    merge_sort(sorted, tmp, np);                                      /* Main.java:228-235 */

    for (int64_t i = 0; i < np; ++i) {                                /* Main.java:237 */
        int64_t best = 0;                                             /* Collections.min, :240-263 */
        for (int64_t c = 1; c < nu; ++c) {
            int r = (count[c] > count[best]) - (count[c] < count[best]);          /* :246-250 */
            if (r == 0) r = (total[c] > total[best]) - (total[c] < total[best]);  /* :253-256 */
            if (r == 0) r = cmp_member(mc, uniq[c], uniq[best]);                  /* :259 */
            if (r < 0) best = c;
        }
        out_partition[i] = sorted[i].partition;                       /* :264 */
        out_member[i] = uniq[best];
        total[best] = (int64_t)((uint64_t)total[best] + (uint64_t)sorted[i].lag); /* :265, wraps */
        count[best] += 1;                                             /* :266 */
    }
    if (out_total)
        for (int64_t i = 0; i < nc; ++i) {
            int64_t j = 0;
            while (uniq[j] != consumers[i]) ++j;
            out_total[i] = total[j];
        }
    free(uniq); free(total); free(count); free(sorted); free(tmp);
    return 0;
}

/* Batched flat form: the per-topic loop of assign(Map,Map), Main.java:177-184, over
 * SoA input.  Topic t owns partitions [part_off[t], part_off[t+1]) and consumer
 * handles [cons_off[t], cons_off[t+1]).  member_ids == NULL: handles are ranks. */
LAO_EXPORT int lao_assign_flat(int32_t n_topics, const int64_t *part_off, const int32_t *partition,
                               const int64_t *lag, const int64_t *cons_off,
                               const int32_t *cons_member, const char *const *member_ids,
                               int32_t *out_partition, int32_t *out_member, int64_t *out_total) {
    member_cmp_t mc = { member_ids };
    for (int32_t t = 0; t < n_topics; ++t) {
        int64_t p0 = part_off[t], c0 = cons_off[t];
        int rc = assign_topic(&mc, part_off[t + 1] - p0, partition + p0, lag + p0,
                              cons_off[t + 1] - c0, cons_member + c0,
                              out_partition + p0, out_member + p0,
                              out_total ? out_total + c0 : NULL);
        if (rc) return rc;
    }
    return 0;
}
