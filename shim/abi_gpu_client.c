/* A C caller shaped like the reference's FFI users (rust/src/lib.rs:18-96) for the GPU precompiles:
 * links ONLY the static shim + libc, reads (input, expected output or error code) cases from files,
 * and calls bls12_g1multiexp / bls12_g2multiexp / bls12_pairing -- and the _naive / _bc names -- through
 * the archive from several threads at once.  Used by tests/test_gpu_shim.py.
 *   usage: abi_gpu_client <threads> <reps> <case-file>...
 *   case file: line 1 = op name, line 2 = expected error code, line 3 = expected output hex ("-" if an
 *   error is expected), line 4 = input hex */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../include/eip2537.h"

typedef EIP2537_ERROR (*fn_t)(byte *, byte *, size_t);
struct kase { fn_t fn; int want_code; size_t in_len, out_len; byte *in, *want; char op[64]; };
static struct kase cases[64];
static int ncases, reps;

static unsigned nib(char c) { return c <= '9' ? (unsigned)(c - '0') : (unsigned)((c | 32) - 'a' + 10); }
static size_t unhex(const char *s, byte **out) {
    size_t n = strlen(s);
    while (n && (s[n - 1] == '\n' || s[n - 1] == '\r')) n--;
    if (n == 1 && s[0] == '-') { *out = NULL; return 0; }
    *out = malloc(n / 2 + 1);
    for (size_t i = 0; i < n / 2; i++) (*out)[i] = (byte)((nib(s[2 * i]) << 4) | nib(s[2 * i + 1]));
    return n / 2;
}
static fn_t lookup(const char *op) {
    if (!strcmp(op, "bls12_g1multiexp")) return bls12_g1multiexp;
    if (!strcmp(op, "bls12_g1multiexp_naive")) return bls12_g1multiexp_naive;
    if (!strcmp(op, "bls12_g1multiexp_bc")) return bls12_g1multiexp_bc;
    if (!strcmp(op, "bls12_g2multiexp")) return bls12_g2multiexp;
    if (!strcmp(op, "bls12_g2multiexp_naive")) return bls12_g2multiexp_naive;
    if (!strcmp(op, "bls12_g2multiexp_bc")) return bls12_g2multiexp_bc;
    if (!strcmp(op, "bls12_pairing")) return bls12_pairing;
    return NULL;
}
static void *worker(void *arg) {
    long bad = 0, id = *(long *)arg;
    byte out[256];
    for (int r = 0; r < reps; r++)
        for (int c = 0; c < ncases; c++) {
            struct kase *k = &cases[(c + id) % ncases];            /* threads start at different cases */
            memset(out, 0xA5, sizeof out);
            int rc = (int)k->fn(out, k->in, k->in_len);
            if (rc != k->want_code) { bad++; fprintf(stderr, "%s: code %d, want %d\n", k->op, rc, k->want_code); continue; }
            if (rc == 0 && memcmp(out, k->want, k->out_len)) { bad++; fprintf(stderr, "%s: output differs\n", k->op); }
        }
    *(long *)arg = bad;
    return NULL;
}
int main(int argc, char **argv) {
    if (argc < 4) return 2;
    int threads = atoi(argv[1]);
    reps = atoi(argv[2]);
    static char line[(1 << 25) + 16];      /* a 2^16-record input is 21 MB of hex */
    for (int a = 3; a < argc && ncases < 64; a++) {
        FILE *f = fopen(argv[a], "r");
        if (!f) { perror(argv[a]); return 2; }
        struct kase *k = &cases[ncases++];
        if (!fgets(k->op, sizeof k->op, f)) return 2;
        k->op[strcspn(k->op, "\r\n")] = 0;
        k->fn = lookup(k->op);
        if (!k->fn) { fprintf(stderr, "unknown op %s\n", k->op); return 2; }
        if (!fgets(line, sizeof line, f)) return 2;
        k->want_code = atoi(line);
        if (!fgets(line, sizeof line, f)) return 2;
        k->out_len = unhex(line, &k->want);
        if (!fgets(line, sizeof line, f)) return 2;
        k->in_len = unhex(line, &k->in);
        fclose(f);
    }
    pthread_t t[16];
    long bad[16];
    if (threads > 16) threads = 16;
    for (long i = 0; i < threads; i++) { bad[i] = i; pthread_create(&t[i], NULL, worker, &bad[i]); }
    long total = 0;
    for (int i = 0; i < threads; i++) { pthread_join(t[i], NULL); total += bad[i]; }
    printf("gpu shim client: %d cases x %d reps x %d threads, %ld failures\n", ncases, reps, threads, total);
    return total ? 1 : 0;
}
