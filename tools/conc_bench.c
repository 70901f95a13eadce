/* Native concurrency benchmark of the reference ABI (no Python, no GIL): T pthreads each issue R calls of one
 * small precompile input through libeip2537_hip.so and check every output against the first one.
 *   gcc -O2 tools/conc_bench.c -ldl -lpthread -o /tmp/conc_bench
 *   /tmp/conc_bench <lib.so> <g1msm|g2msm|pairing> <units> <threads> <calls-per-thread>
 * Used for profiles/r02_concurrent_callers_native.txt. */
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int (*call_fn)(unsigned char *, unsigned char *, size_t);
typedef int (*gen_msm_fn)(uint8_t *, size_t, const uint8_t *, const uint8_t *, uint64_t, uint64_t);
typedef int (*gen_pair_fn)(uint8_t *, size_t, const uint8_t *, const uint8_t *, const uint8_t *, const uint8_t *, uint64_t);
typedef void (*stats_fn)(uint64_t *, uint64_t *, uint64_t *);

static call_fn g_call;
static unsigned char *g_in, g_want[256];
static size_t g_len, g_out;
static int g_calls;
static pthread_barrier_t g_bar;

static void *worker(void *arg) {
    long bad = 0;
    unsigned char out[256];
    pthread_barrier_wait(&g_bar);
    for (int i = 0; i < g_calls; i++) {
        if (g_call(out, g_in, g_len) != 0 || memcmp(out, g_want, g_out)) bad++;
    }
    *(long *)arg = bad;
    return NULL;
}
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s lib op units threads calls\n", argv[0]); return 2; }
    void *h = dlopen(argv[1], RTLD_NOW);
    if (!h) { fprintf(stderr, "%s\n", dlerror()); return 2; }
    const char *op = argv[2];
    size_t units = (size_t)atol(argv[3]);
    int threads = atoi(argv[4]);
    g_calls = atoi(argv[5]);
    uint8_t a[32] = {3}, b[32] = {5}, c[32] = {7}, d[32] = {11};
    if (!strcmp(op, "pairing")) {
        g_len = units * 384; g_out = 32; g_in = malloc(g_len);
        ((gen_pair_fn)dlsym(h, "eip2537_hip_gen_pairing_input"))(g_in, units, a, b, c, d, 0);
        g_call = (call_fn)dlsym(h, "bls12_pairing");
    } else {
        int g2 = !strcmp(op, "g2msm");
        size_t rec = g2 ? 288 : 160;
        g_len = units * rec; g_out = g2 ? 256 : 128; g_in = malloc(g_len);
        ((gen_msm_fn)dlsym(h, g2 ? "eip2537_hip_gen_g2_msm_input" : "eip2537_hip_gen_g1_msm_input"))(g_in, units, a, b, 7, 0);
        g_call = (call_fn)dlsym(h, g2 ? "bls12_g2multiexp" : "bls12_g1multiexp");
    }
    if (g_call(g_want, g_in, g_len) != 0) { fprintf(stderr, "reference call failed\n"); return 1; }
    for (int i = 0; i < 8; i++) { unsigned char o[256]; g_call(o, g_in, g_len); }
    pthread_t *t = malloc(sizeof(pthread_t) * (size_t)threads);
    long *bad = calloc((size_t)threads, sizeof(long));
    /* untimed round: slots touched for the first time create their streams and grow their workspaces */
    int calls = g_calls; g_calls = 4;
    pthread_barrier_init(&g_bar, NULL, (unsigned)threads);
    for (int i = 0; i < threads; i++) pthread_create(&t[i], NULL, worker, &bad[i]);
    for (int i = 0; i < threads; i++) pthread_join(t[i], NULL);
    g_calls = calls;
    pthread_barrier_destroy(&g_bar);
    pthread_barrier_init(&g_bar, NULL, (unsigned)threads + 1);
    for (int i = 0; i < threads; i++) pthread_create(&t[i], NULL, worker, &bad[i]);
    pthread_barrier_wait(&g_bar);
    double t0 = now();
    long total_bad = 0;
    for (int i = 0; i < threads; i++) { pthread_join(t[i], NULL); total_bad += bad[i]; }
    double dt = now() - t0;
    uint64_t s0 = 0, s1 = 0, s2 = 0;
    stats_fn st = (stats_fn)dlsym(h, "eip2537_hip_coalesce_stats");
    if (st) st(&s0, &s1, &s2);
    printf("%-8s units=%-5zu T=%-3d %9.0f calls/s  (%d calls per thread, %ld mismatches; coalescing: %llu pipelines for %llu calls, largest batch %llu)\n",
           op, units, threads, threads * (double)g_calls / dt, g_calls, total_bad, (unsigned long long)s0, (unsigned long long)s1, (unsigned long long)s2);
    return total_bad ? 1 : 0;
}
