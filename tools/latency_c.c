/* latency_c.c -- what ONE small rebalance costs at the C ABI, without an interpreter around it.
 *   gcc -O2 -std=c99 -Iinclude tools/latency_c.c -Lkafka_lag_based_assignor_amd -llagassign -Wl,-rpath,$PWD/kafka_lag_based_assignor_amd -o /tmp/latency_c
 * Prints the median and the minimum wall time of la_assign_batch_grouped (assignment + every member's list) and of
 * la_assign_batch over a few batch shapes, and the kernel launches of a call (la_last_launches). */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "lagassign.h"

static double now_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

static int cmp_double(const void *a, const void *b) {
    const double x = *(const double *)a, y = *(const double *)b;
    return x < y ? -1 : x > y;
}

int main(void) {
    la_ctx *ctx = NULL;
    if (la_create(&ctx, 0, 0) != LA_OK) {
        fprintf(stderr, "la_create: %s\n", la_last_error(NULL));
        return 1;
    }
    static const int shapes[][3] = {{1, 3, 2}, {10, 10, 3}, {40, 50, 5}, {100, 20, 4}, {100, 100, 8}};
    for (unsigned s = 0; s < sizeof shapes / sizeof shapes[0]; ++s) {
        const int T = shapes[s][0], P = shapes[s][1], C = shapes[s][2];
        const int64_t n = (int64_t)T * P, k = (int64_t)T * C;
        int64_t *part_off = malloc((T + 1) * 8), *cons_off = malloc((T + 1) * 8);
        int32_t *pid = malloc(n * 4), *cons_rank = malloc(k * 4);
        int64_t *begin = calloc(n, 8), *end = malloc(n * 8), *committed = malloc(n * 8);
        int64_t *member_off = malloc((C + 1) * 8), *total = malloc(k * 8);
        int32_t *g_topic = malloc(n * 4), *g_part = malloc(n * 4), *o_pid = malloc(n * 4), *o_rank = malloc(n * 4);
        uint64_t x = 88172645463325252ull;
        for (int t = 0; t <= T; ++t) { part_off[t] = (int64_t)t * P; cons_off[t] = (int64_t)t * C; }
        for (int64_t i = 0; i < n; ++i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            pid[i] = (int32_t)(i % P);
            committed[i] = (x & 127) == 0 ? LA_NO_COMMITTED : (int64_t)(x >> 44);
            end[i] = (int64_t)(x >> 44) + (int64_t)((x >> 8) & 0xFFFFFFFFFFull);
        }
        for (int64_t i = 0; i < k; ++i) cons_rank[i] = (int32_t)(i % C);
        enum { REPS = 2000 };
        static double tg[REPS], ta[REPS];
        int64_t launches_g = 0, launches_a = 0;
        for (int r = -50; r < REPS; ++r) {
            const double t0 = now_us();
            int rc = la_assign_batch_grouped(ctx, T, part_off, pid, begin, end, committed, LA_RESET_EARLIEST, cons_off, cons_rank, C,
                                             member_off, g_topic, g_part, total);
            const double t1 = now_us();
            if (rc != LA_OK) { fprintf(stderr, "grouped: %d %s\n", rc, la_last_error(ctx)); return 1; }
            launches_g = la_last_launches(ctx);
            rc = la_assign_batch(ctx, T, part_off, pid, begin, end, committed, LA_RESET_EARLIEST, cons_off, cons_rank, o_pid, o_rank, total);
            const double t2 = now_us();
            if (rc != LA_OK) { fprintf(stderr, "assign: %d %s\n", rc, la_last_error(ctx)); return 1; }
            launches_a = la_last_launches(ctx);
            if (r >= 0) { tg[r] = t1 - t0; ta[r] = t2 - t1; }
        }
        qsort(tg, REPS, sizeof(double), cmp_double);
        qsort(ta, REPS, sizeof(double), cmp_double);
        printf("%4d topics x %4d partitions x %2d consumers (%6lld partitions) pipeline %d: la_assign_batch_grouped median %.1f us, min %.1f us "
               "(%lld launch(es)); la_assign_batch median %.1f us, min %.1f us (%lld launch(es))\n",
               T, P, C, (long long)n, la_last_pipeline(ctx), tg[REPS / 2], tg[0], (long long)launches_g, ta[REPS / 2], ta[0],
               (long long)launches_a);
        free(part_off); free(cons_off); free(pid); free(cons_rank); free(begin); free(end); free(committed);
        free(member_off); free(total); free(g_topic); free(g_part); free(o_pid); free(o_rank);
    }
    la_destroy(ctx);
    return 0;
}
