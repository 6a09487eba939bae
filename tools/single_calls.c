/* Single-document TextToIds through the drop-in C-ABI from native threads (no interpreter lock in the way):
   tools/single_calls <library.so> <model.bin> <documents file: one per line> [seconds per point]
   prints calls/s from 1, 4, 16 and 64 threads on ONE handle.  Works on any library with the reference's entry points
   (LoadModel / TextToIds / FreeModel), so the compiled reference can be timed beside the GPU library. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef void *(*load_fn)(const char *);
typedef int (*t2i_fn)(void *, const char *, int, int32_t *, int, int);
typedef int (*free_fn)(void *);

static t2i_fn g_t2i; static void *g_h; static char **g_docs; static int *g_len; static int g_nd;
static uint64_t *g_expect;       /* per document: a hash of (count, ids) from the first single-threaded pass; every later call is checked against it */
static volatile int g_stop; static int g_fail; static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

struct Arg { int t, T; long calls; long ids; long bad; };
static uint64_t hash_ids(const int32_t *b, int n) { uint64_t h = 1469598103934665603ull ^ (uint64_t)n; for (int i = 0; i < n && i < 512; ++i) h = (h ^ (uint32_t)b[i]) * 1099511628211ull; return h; }
static void *work(void *vp)
{
    struct Arg *a = (struct Arg *)vp; int32_t buf[512]; long k = a->t;
    while (!g_stop) {
        const int d = (int)(k % g_nd); const int n = g_t2i(g_h, g_docs[d], g_len[d], buf, 512, 100);
        if (hash_ids(buf, n) != g_expect[d]) ++a->bad;
        a->ids += n; ++a->calls; k += a->T;
    }
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s lib.so model.bin docs.txt [seconds]\n", argv[0]); return 2; }
    const double secs = argc > 4 ? atof(argv[4]) : 2.0;
    void *L = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
    if (!L) { fprintf(stderr, "%s\n", dlerror()); return 1; }
    load_fn load = (load_fn)dlsym(L, "LoadModel"); g_t2i = (t2i_fn)dlsym(L, "TextToIds"); free_fn fr = (free_fn)dlsym(L, "FreeModel");
    if (!load || !g_t2i || !fr) { fprintf(stderr, "entry points missing\n"); return 1; }
    FILE *f = fopen(argv[3], "rb"); if (!f) { perror(argv[3]); return 1; }
    size_t capd = 1024; g_docs = malloc(capd * sizeof(char *)); g_len = malloc(capd * sizeof(int));
    char *line = NULL; size_t lc = 0; ssize_t n;
    while ((n = getline(&line, &lc, f)) > 0) {
        while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) --n;
        if (n == 0) continue;
        if ((size_t)g_nd == capd) { capd *= 2; g_docs = realloc(g_docs, capd * sizeof(char *)); g_len = realloc(g_len, capd * sizeof(int)); }
        g_docs[g_nd] = malloc((size_t)n + 1); memcpy(g_docs[g_nd], line, (size_t)n); g_docs[g_nd][n] = 0; g_len[g_nd] = (int)n; ++g_nd;
    }
    fclose(f);
    if (!g_nd) { fprintf(stderr, "no documents\n"); return 1; }
    const int Ts[4] = {1, 4, 16, 64};
    for (int i = 0; i < 4; ++i) {
        g_h = load(argv[2]); if (!g_h) { fprintf(stderr, "LoadModel failed\n"); return 1; }
        if (!g_expect) {                                   /* what every call must return: one quiet pass */
            g_expect = malloc((size_t)g_nd * sizeof(uint64_t)); int32_t buf[512];
            for (int d = 0; d < g_nd; ++d) { const int n = g_t2i(g_h, g_docs[d], g_len[d], buf, 512, 100); g_expect[d] = hash_ids(buf, n); }
        }
        const int T = Ts[i]; pthread_t th[64]; struct Arg a[64];
        for (int rep = 0; rep < 2; ++rep) {              /* the first pass warms up */
            g_stop = 0;
            for (int t = 0; t < T; ++t) { a[t].t = t; a[t].T = T; a[t].calls = 0; a[t].ids = 0; a[t].bad = 0; pthread_create(&th[t], NULL, work, &a[t]); }
            const double t0 = now(); struct timespec sl = {(time_t)(rep ? secs : 0.3), (long)(((rep ? secs : 0.3) - (long)(rep ? secs : 0.3)) * 1e9)}; nanosleep(&sl, NULL);
            g_stop = 1; long calls = 0, ids = 0, bad = 0;
            for (int t = 0; t < T; ++t) { pthread_join(th[t], NULL); calls += a[t].calls; ids += a[t].ids; bad += a[t].bad; }
            const double dt = now() - t0;
            if (rep) printf("%3d threads: %10.0f calls/s  (%ld calls, %.1f ids per call, %ld results that differ from the quiet pass)\n", T, calls / dt, calls, calls ? (double)ids / calls : 0.0, bad);
            if (bad) g_fail = 1;
        }
        fflush(stdout);
        fr(g_h);
    }
    return g_fail;
}
