/* cold_c.c -- what a REAL rebalance pays at the C ABI: one la_assign_batch_grouped after the process and the device idled, with
 * and without la_wake before it, no interpreter anywhere.
 *   gcc -O2 -std=c99 -Iinclude tools/cold_c.c -Lkafka_lag_based_assignor_amd -llagassign -Wl,-rpath,$PWD/kafka_lag_based_assignor_amd -o /tmp/cold_c
 *   cold_c [idle_ms ...]        (default 50 1000)
 *   cold_c --json T P C         one shape, 1 s of idle, 5 tries, la_wake 5 ms before: one JSON object (bench.py's small_call.cold)
 * Per batch shape: the back-to-back median; then per idle time, medians over 7 tries of (a) the call right after the idle
 * (nanosleep), (b) la_wake after the idle, LEAD ms of "broker round trips" (nanosleep: the host really is away), the call. */
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
static void sleep_ms(double ms) {
    struct timespec ts;
    ts.tv_sec = (time_t)(ms / 1000.0);
    ts.tv_nsec = (long)((ms - 1000.0 * (double)ts.tv_sec) * 1e6);
    nanosleep(&ts, NULL);
}
static int cmp_double(const void *a, const void *b) {
    const double x = *(const double *)a, y = *(const double *)b;
    return x < y ? -1 : x > y;
}
static double median(double *v, int n) { qsort(v, n, sizeof(double), cmp_double); return v[n / 2]; }

static int json_mode(int T, int P, int C);

int main(int argc, char **argv) {
    double idles[8];
    int n_idle = 0;
    if (argc == 5 && !strcmp(argv[1], "--json")) return json_mode(atoi(argv[2]), atoi(argv[3]), atoi(argv[4]));
    for (int i = 1; i < argc && n_idle < 8; ++i) idles[n_idle++] = atof(argv[i]);
    if (!n_idle) { idles[0] = 50; idles[1] = 1000; n_idle = 2; }
    static const double leads[] = {1.0, 20.0, 200.0};
    la_ctx *ctx = NULL;
    if (la_create(&ctx, 0, 0) != LA_OK) { fprintf(stderr, "la_create: %s\n", la_last_error(NULL)); return 1; }
    static const int shapes[][3] = {{10, 10, 3}, {100, 20, 4}, {100, 100, 8}, {1000, 256, 32}};
    for (unsigned s = 0; s < sizeof shapes / sizeof shapes[0]; ++s) {
        const int T = shapes[s][0], P = shapes[s][1], C = shapes[s][2];
        const int64_t n = (int64_t)T * P, k = (int64_t)T * C;
        int64_t *part_off = malloc((T + 1) * 8), *cons_off = malloc((T + 1) * 8);
        int32_t *pid = malloc(n * 4), *cons_rank = malloc(k * 4);
        int64_t *begin = calloc(n, 8), *end = malloc(n * 8), *committed = malloc(n * 8);
        int64_t *member_off = malloc((C + 1) * 8), *total = malloc(k * 8);
        int32_t *g_topic = malloc(n * 4), *g_part = malloc(n * 4);
        uint64_t x = 88172645463325252ull;
        for (int t = 0; t <= T; ++t) { part_off[t] = (int64_t)t * P; cons_off[t] = (int64_t)t * C; }
        for (int64_t i = 0; i < n; ++i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            pid[i] = (int32_t)(i % P);
            committed[i] = (x & 127) == 0 ? LA_NO_COMMITTED : (int64_t)(x >> 44);
            end[i] = (int64_t)(x >> 44) + (int64_t)((x >> 8) & 0xFFFFFFFFFFull);
        }
        for (int64_t i = 0; i < k; ++i) cons_rank[i] = (int32_t)(i % C);
#define CALL()                                                                                                              \
    do {                                                                                                                    \
        int rc_ = la_assign_batch_grouped(ctx, T, part_off, pid, begin, end, committed, LA_RESET_EARLIEST, cons_off,        \
                                          cons_rank, C, member_off, g_topic, g_part, total);                                \
        if (rc_ != LA_OK) { fprintf(stderr, "grouped: %d %s\n", rc_, la_last_error(ctx)); return 1; }                       \
    } while (0)
        double w[200];
        for (int r = -30; r < 200; ++r) { const double t0 = now_us(); CALL(); if (r >= 0) w[r] = now_us() - t0; }
        printf("%7lld partitions (%d x %d x %d): back to back %6.1f us", (long long)n, T, P, C, median(w, 200));
        for (int d = 0; d < n_idle; ++d) {
            double c[7];
            for (int r = 0; r < 7; ++r) { sleep_ms(idles[d]); const double t0 = now_us(); CALL(); c[r] = now_us() - t0; }
            printf(" | after %.0f ms idle %6.1f", idles[d], median(c, 7));
        }
        const double idle = idles[n_idle - 1];
        for (unsigned l = 0; l < sizeof leads / sizeof leads[0]; ++l) {
            double c[7], wk[7];
            for (int r = 0; r < 7; ++r) {
                sleep_ms(idle);
                double t0 = now_us();
                if (la_wake(ctx) != LA_OK) { fprintf(stderr, "la_wake: %s\n", la_last_error(ctx)); return 1; }
                wk[r] = now_us() - t0;
                sleep_ms(leads[l]);
                t0 = now_us();
                CALL();
                c[r] = now_us() - t0;
            }
            printf(" | la_wake (%5.1f us) %.0f ms before: %6.1f", median(wk, 7), leads[l], median(c, 7));
        }
        printf("\n");
        fflush(stdout);
    }
    la_destroy(ctx);
    return 0;
}

static int json_mode(int T, int P, int C) {
    la_ctx *ctx = NULL;
    if (la_create(&ctx, 0, 0) != LA_OK) { fprintf(stderr, "la_create: %s\n", la_last_error(NULL)); return 1; }
    const int64_t n = (int64_t)T * P, k = (int64_t)T * C;
    int64_t *part_off = malloc((T + 1) * 8), *cons_off = malloc((T + 1) * 8);
    int32_t *pid = malloc(n * 4), *cons_rank = malloc(k * 4);
    int64_t *begin = calloc(n, 8), *end = malloc(n * 8), *committed = malloc(n * 8);
    int64_t *member_off = malloc((C + 1) * 8), *total = malloc(k * 8);
    int32_t *g_topic = malloc(n * 4), *g_part = malloc(n * 4);
    uint64_t x = 88172645463325252ull;
    for (int t = 0; t <= T; ++t) { part_off[t] = (int64_t)t * P; cons_off[t] = (int64_t)t * C; }
    for (int64_t i = 0; i < n; ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        pid[i] = (int32_t)(i % P);
        committed[i] = (x & 127) == 0 ? LA_NO_COMMITTED : (int64_t)(x >> 44);
        end[i] = (int64_t)(x >> 44) + (int64_t)((x >> 8) & 0xFFFFFFFFFFull);
    }
    for (int64_t i = 0; i < k; ++i) cons_rank[i] = (int32_t)(i % C);
    double w[200], c[5], v[5], wk[5];
    for (int r = -30; r < 200; ++r) { const double t0 = now_us(); CALL(); if (r >= 0) w[r] = now_us() - t0; }
    for (int r = 0; r < 5; ++r) { sleep_ms(1000); const double t0 = now_us(); CALL(); c[r] = now_us() - t0; }
    for (int r = 0; r < 5; ++r) {
        sleep_ms(1000);
        double t0 = now_us();
        if (la_wake(ctx) != LA_OK) { fprintf(stderr, "la_wake: %s\n", la_last_error(ctx)); return 1; }
        wk[r] = now_us() - t0;
        sleep_ms(5);
        t0 = now_us();
        CALL();
        v[r] = now_us() - t0;
    }
    printf("{\"topics\": %d, \"partitions_per_topic\": %d, \"consumers\": %d, \"partitions\": %lld, \"back_to_back_us\": %.1f, "
           "\"after_1s_idle_us\": %.1f, \"la_wake_us\": %.1f, \"after_1s_idle_and_la_wake_5ms_before_us\": %.1f}\n",
           T, P, C, (long long)n, median(w, 200), median(c, 5), median(wk, 5), median(v, 5));
    la_destroy(ctx);
    return 0;
}
