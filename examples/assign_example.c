/* assign_example.c -- the README example of the reference (README.md:42-57) through the C ABI, from plain C.
 *
 *   gcc -Iinclude examples/assign_example.c -Lkafka_lag_based_assignor_amd -llagassign \
 *       -Wl,-rpath,$PWD/kafka_lag_based_assignor_amd -o assign_example && ./assign_example
 *
 * One topic t0, partitions 0..2 with lags 100 000 / 50 000 / 60 000, consumers C0 and C1:
 *   expected  C0 = {t0p0}            total 100 000
 *             C1 = {t0p2, t0p1}      total 110 000     (60 000 is handed out before 50 000)
 */
#include <stdint.h>
#include <stdio.h>

#include "lagassign.h"

int main(void) {
    la_ctx *ctx = NULL;
    int rc = la_create(&ctx, 0, 0);
    if (rc != LA_OK) {
        fprintf(stderr, "la_create: %d %s\n", rc, la_last_error(NULL));
        return 1;
    }
    /* SoA marshalling: see INTEGRATION.md section 3 */
    const int64_t part_off[2] = {0, 3};
    const int32_t partition_id[3] = {0, 1, 2};
    const int64_t begin_off[3] = {0, 0, 0};
    const int64_t end_off[3] = {100000, 50000, 60000};
    const int64_t committed_off[3] = {0, 0, 0};                 /* lag = end - committed */
    const int64_t cons_off[2] = {0, 2};
    const int32_t cons_rank[2] = {0, 1};                        /* "C0" < "C1" under String.compareTo */

    /* a real host calls this when it ENTERS assign(), before it fetches the offsets: a one-partition rebalance through the real
     * path, so that the call that matters runs warm (ABI 0.5.0; optional) */
    if (la_version() >= 500) (void)la_wake(ctx);
    /* what the marshalling loop knows for free (ABI 0.4.0): no lag exceeds the largest end offset (no offset here is negative)
     * and no partition id exceeds 2.  One-shot: it applies to the next assign call.  Optional -- a call without it is the same
     * call, with one more (empty) kernel launch per chunk of a large batch. */
    if (la_version() >= 400) {
        la_call_hints hints;
        hints.struct_size = (int32_t)sizeof hints;
        hints.flags = LA_HINT_BOUNDS;
        hints.max_lag = 100000;
        hints.max_partition_id = 2;
        rc = la_hint_next_call(ctx, &hints);
        if (rc != LA_OK) {
            fprintf(stderr, "la_hint_next_call: %d %s\n", rc, la_last_error(ctx));
            return 1;
        }
    }
    /* results stay on the device; every member's list comes back grouped */
    int64_t total[2];
    rc = la_assign_batch(ctx, 1, part_off, partition_id, begin_off, end_off, committed_off, LA_RESET_EARLIEST,
                         cons_off, cons_rank, NULL, NULL, total);
    if (rc != LA_OK) {
        fprintf(stderr, "la_assign_batch: %d %s\n", rc, la_last_error(ctx));
        return 1;
    }
    int64_t member_off[3];
    int32_t grouped_topic[3], grouped_partition[3];
    rc = la_group_last_by_member(ctx, 2, member_off, grouped_topic, grouped_partition);
    if (rc != LA_OK) {
        fprintf(stderr, "la_group_last_by_member: %d %s\n", rc, la_last_error(ctx));
        return 1;
    }
    int ok = 1;
    for (int m = 0; m < 2; ++m) {
        printf("C%d (total lag %lld):", m, (long long)total[m]);
        for (int64_t j = member_off[m]; j < member_off[m + 1]; ++j)
            printf(" t%dp%d", grouped_topic[j], grouped_partition[j]);
        printf("\n");
    }
    ok &= member_off[0] == 0 && member_off[1] == 1 && member_off[2] == 3;
    ok &= grouped_partition[0] == 0 && grouped_partition[1] == 2 && grouped_partition[2] == 1;
    ok &= total[0] == 100000 && total[1] == 110000;

    /* The same rebalance in ONE call, with the beginning offsets handed over only where they are read (ABI 0.3.0): partition 1
     * has no committed offset (-1), so under auto.offset.reset=earliest its lag is end - begin = 50 000 - 0 (Main.java:384-396). */
    if (la_version() >= 300) {
        const int64_t committed_sparse[3] = {0, LA_NO_COMMITTED, 0};
        const int64_t none_index[1] = {1}, none_begin[1] = {0};
        int64_t off2[3], total2[2];
        int32_t topic2[3], part2[3];
        rc = la_assign_batch_grouped_sparse(ctx, 1, part_off, partition_id, end_off, committed_sparse, LA_RESET_EARLIEST, 1,
                                            none_index, none_begin, cons_off, cons_rank, 2, off2, topic2, part2, total2);
        if (rc != LA_OK) {
            fprintf(stderr, "la_assign_batch_grouped_sparse: %d %s\n", rc, la_last_error(ctx));
            return 1;
        }
        ok &= off2[0] == 0 && off2[1] == 1 && off2[2] == 3 && part2[0] == 0 && part2[1] == 2 && part2[2] == 1;
        ok &= total2[0] == 100000 && total2[1] == 110000;
        /* what one assigned partition costs on the wire of the multi-GPU all-gather: ids < 3, 2 members -> 2 bytes */
        la_wire_format fmt;
        ok &= la_wire_format_for(2, 2, &fmt) == LA_OK && fmt.elem_bytes == 2 && fmt.id_bits == 2;
    }
    la_destroy(ctx);
    printf(ok ? "matches the reference's README example\n" : "MISMATCH\n");
    return ok ? 0 : 2;
}
