/* latency_c.c -- what ONE small rebalance costs at the C ABI, without an interpreter around it.
 *   gcc -O2 -std=c99 -Iinclude tools/latency_c.c -Lkafka_lag_based_assignor_amd -llagassign -Wl,-rpath,$PWD/kafka_lag_based_assignor_amd -o /tmp/latency_c
 * Prints the median and the minimum wall time of la_assign_batch_grouped (assignment + every member's list) and of
 * la_assign_batch over a few batch shapes, and the kernel launches of a call (la_last_launches).
 *   latency_c [--json] [path/to/oracle/liblagoracle.so]
 * With the oracle's shared object (TEST INFRASTRUCTURE: the checker, dlopen'ed here only to be timed beside the product and to
 * compare results): the same call on one host core -- lao_compute_lags + lao_assign_flat + a stable counting sort by member --
 * so that the crossover between the CPU and the GPU path is on the record without an interpreter on either side. */
#define _POSIX_C_SOURCE 199309L
#include <dlfcn.h>
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

typedef void (*lao_lags_fn)(int64_t, const int64_t *, const int64_t *, const int64_t *, int, int64_t *);
typedef int (*lao_assign_fn)(int32_t, const int64_t *, const int32_t *, const int64_t *, const int64_t *, const int32_t *,
                             const char *const *, int32_t *, int32_t *, int64_t *);

int main(int argc, char **argv) {
    int json = 0;
    lao_lags_fn o_lags = NULL;
    lao_assign_fn o_assign = NULL;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--json")) { json = 1; continue; }
        void *h = dlopen(argv[i], RTLD_NOW | RTLD_LOCAL);
        if (!h) { fprintf(stderr, "dlopen %s: %s\n", argv[i], dlerror()); return 1; }
        o_lags = (lao_lags_fn)dlsym(h, "lao_compute_lags");
        o_assign = (lao_assign_fn)dlsym(h, "lao_assign_flat");
        if (!o_lags || !o_assign) { fprintf(stderr, "%s lacks lao_compute_lags / lao_assign_flat\n", argv[i]); return 1; }
    }
    la_ctx *ctx = NULL;
    if (la_create(&ctx, 0, 0) != LA_OK) {
        fprintf(stderr, "la_create: %s\n", la_last_error(NULL));
        return 1;
    }
    static const int shapes[][3] = {{1, 3, 2}, {10, 10, 3}, {20, 25, 4}, {40, 50, 5}, {100, 20, 4}, {100, 100, 8}};
    if (json) printf("[");
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
        double cpu_med = -1;
        int same = -1;
        if (o_assign) {
            /* the oracle on the same call: lags, assignment, every member's list (a stable counting sort by member) */
            int64_t *lag = malloc(n * 8), *c_off = malloc((C + 2) * 8);
            int32_t *c_pid = malloc(n * 4), *c_rank = malloc(n * 4), *c_gp = malloc(n * 4), *c_gt = malloc(n * 4);
            static double tc[400];
            for (int r = -20; r < 400; ++r) {
                const double t0 = now_us();
                o_lags(n, begin, end, committed, 0, lag);
                if (o_assign(T, part_off, pid, lag, cons_off, cons_rank, NULL, c_pid, c_rank, total) != 0) return 3;
                memset(c_off, 0, (C + 2) * 8);
                for (int64_t i = 0; i < n; ++i) c_off[c_rank[i] + 2]++;
                for (int m = 1; m <= C + 1; ++m) c_off[m] += c_off[m - 1];
                for (int t = 0; t < T; ++t)
                    for (int64_t i = part_off[t]; i < part_off[t + 1]; ++i) {
                        const int64_t at = c_off[c_rank[i] + 1]++;
                        c_gp[at] = c_pid[i];
                        c_gt[at] = t;
                    }
                const double t1 = now_us();
                if (r >= 0) tc[r] = t1 - t0;
            }
            qsort(tc, 400, sizeof(double), cmp_double);
            cpu_med = tc[200];
            /* (one more grouped GPU call, so that g_part holds its lists) */
            if (la_assign_batch_grouped(ctx, T, part_off, pid, begin, end, committed, LA_RESET_EARLIEST, cons_off, cons_rank, C,
                                        member_off, g_topic, g_part, total) != LA_OK) return 4;
            same = memcmp(c_gp, g_part, n * 4) == 0 && memcmp(c_gt, g_topic, n * 4) == 0;
            free(lag); free(c_off); free(c_pid); free(c_rank); free(c_gp); free(c_gt);
        }
        if (json) {
            printf("%s{\"topics\": %d, \"partitions_per_topic\": %d, \"consumers\": %d, \"partitions\": %lld, \"pipeline\": %d, "
                   "\"grouped_us\": %.1f, \"grouped_min_us\": %.1f, \"grouped_launches\": %lld, \"assign_us\": %.1f, \"assign_launches\": %lld, "
                   "\"cpu_oracle_us\": %s%.1f, \"same_lists\": %s}",
                   s ? ", " : "", T, P, C, (long long)n, la_last_pipeline(ctx), tg[REPS / 2], tg[0], (long long)launches_g, ta[REPS / 2],
                   (long long)launches_a, cpu_med < 0 ? "-" : "", cpu_med < 0 ? 1.0 : cpu_med, same < 0 ? "null" : (same ? "true" : "false"));
        } else
        printf("%4d topics x %4d partitions x %2d consumers (%6lld partitions) pipeline %d: la_assign_batch_grouped median %.1f us, min %.1f us "
               "(%lld launch(es)); la_assign_batch median %.1f us, min %.1f us (%lld launch(es))\n",
               T, P, C, (long long)n, la_last_pipeline(ctx), tg[REPS / 2], tg[0], (long long)launches_g, ta[REPS / 2], ta[0],
               (long long)launches_a);
        if (!json && o_assign) printf("      the C oracle + lists on one host core: median %.1f us; same lists: %s\n", cpu_med, same ? "yes" : "NO");
        free(part_off); free(cons_off); free(pid); free(cons_rank); free(begin); free(end); free(committed);
        free(member_off); free(total); free(g_topic); free(g_part); free(o_pid); free(o_rank);
    }
    if (json) printf("]\n");
    la_destroy(ctx);
    return 0;
}
