/*
 * cpu_baseline.c -- TEST / BENCH INFRASTRUCTURE: times a CPU TextToIds implementation on host cores.
 *
 * dlopen()s a library that exports the reference C-ABI (LoadModel / TextToIds / FreeModel) -- normally
 * oracle/_ref/libblingfiretokdll_ref.so, the unmodified reference compiled by oracle/Makefile -- and calls
 * TextToIds once per document from T threads sharing one model handle (the usage the reference recommends:
 * README.md:105,215), static interleaved document assignment, per-thread id buffer.
 * Used only by bench.py's cpu_baseline leg and by tests; never by the product.
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef void *(*load_fn)(const char *);
typedef int (*free_fn)(void *);
typedef int (*t2i_fn)(void *, const char *, int, int32_t *, int, int);

#define MAX_THREADS 1024

typedef struct {
    t2i_fn t2i; void *model; const char *text; const int64_t *off; int64_t ndocs; int max_ids, unk;
    int tid, nthreads; int64_t ids; uint64_t checksum; int32_t *out_ids; int64_t *out_counts; int64_t stride;
    uint64_t *out_hash;   /* optional: per-document hash of the ids (bfc_ids_hash) */
} job_t;

/* Order- and length-sensitive 64-bit hash of one document's ids: sum over j of (id_j + C) * (2j + 1), modulo 2^64.
 * bench.py computes the same sum on the device ids (torch int64 arithmetic wraps the same way) for the full-shard check. */
static uint64_t bfc_ids_hash(const int32_t *ids, int n)
{
    uint64_t h = 0;
    for (int j = 0; j < n; ++j) h += ((uint64_t)(int64_t)ids[j] + 0x9E3779B97F4A7C15ull) * (uint64_t)(2 * (int64_t)j + 1);
    return h;
}

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    int32_t *buf = (int32_t *)malloc(sizeof(int32_t) * (size_t)(j->max_ids > 0 ? j->max_ids : 1));
    for (int64_t d = j->tid; d < j->ndocs; d += j->nthreads) {
        int32_t *dst = j->out_ids ? j->out_ids + d * j->stride : buf;
        int n = j->t2i(j->model, j->text + j->off[d], (int)(j->off[d + 1] - j->off[d]), dst, j->max_ids, j->unk);
        j->ids += n;
        for (int k = 0; k < n; ++k) j->checksum = j->checksum * 1099511628211ull + (uint64_t)(uint32_t)dst[k] + (uint64_t)d;
        if (j->out_counts) j->out_counts[d] = n;
        if (j->out_hash) j->out_hash[d] = bfc_ids_hash(dst, n);
    }
    free(buf);
    return NULL;
}

/* Returns seconds of wall time for one pass (negative on error).  If out_ids != NULL it receives the ids of
 * document d at out_ids[d*max_ids ..] and out_counts[d] the count (golden-file generation). */
static double run_passes(const char *lib_path, const char *model_path, const char *text, const int64_t *doc_off,
                         int64_t ndocs, int max_ids, int unk, int nthreads, int passes, int64_t *total_ids,
                         int32_t *out_ids, int64_t *out_counts, uint64_t *out_hash)
{
    void *lib = dlopen(lib_path, RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "cpu_baseline: dlopen(%s): %s\n", lib_path, dlerror()); return -1.0; }
    load_fn load = (load_fn)dlsym(lib, "LoadModel");
    free_fn fre = (free_fn)dlsym(lib, "FreeModel");
    t2i_fn t2i = (t2i_fn)dlsym(lib, "TextToIds");
    if (!load) {   /* the plain-C restatement (oracle/liboracle.so) spells the same three calls bfo_* */
        load = (load_fn)dlsym(lib, "bfo_load_model"); fre = (free_fn)dlsym(lib, "bfo_free_model"); t2i = (t2i_fn)dlsym(lib, "bfo_text_to_ids");
    }
    if (!load || !fre || !t2i) { fprintf(stderr, "cpu_baseline: missing symbols in %s\n", lib_path); return -2.0; }
    void *model = load(model_path);
    if (!model) { fprintf(stderr, "cpu_baseline: LoadModel(%s) failed\n", model_path); return -3.0; }
    if (nthreads < 1) nthreads = 1;
    if (nthreads > MAX_THREADS) nthreads = MAX_THREADS;
    if (passes < 1) passes = 1;
    double best = 1e30; int64_t ids = 0;
    for (int p = 0; p < passes; ++p) {
        static pthread_t th[MAX_THREADS]; static job_t jobs[MAX_THREADS];   /* not re-entrant: one measurement at a time */
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int t = 0; t < nthreads; ++t) {
            memset(&jobs[t], 0, sizeof(job_t));
            jobs[t].t2i = t2i; jobs[t].model = model; jobs[t].text = text; jobs[t].off = doc_off; jobs[t].ndocs = ndocs;
            jobs[t].max_ids = max_ids; jobs[t].unk = unk; jobs[t].tid = t; jobs[t].nthreads = nthreads;
            jobs[t].out_ids = out_ids; jobs[t].out_counts = out_counts; jobs[t].stride = max_ids; jobs[t].out_hash = out_hash;
            pthread_create(&th[t], NULL, worker, &jobs[t]);
        }
        ids = 0;
        for (int t = 0; t < nthreads; ++t) { pthread_join(th[t], NULL); ids += jobs[t].ids; }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        double s = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        if (s < best) best = s;
    }
    if (total_ids) *total_ids = ids;
    fre(model);
    return best;
}

double bfc_time_text_to_ids(const char *lib_path, const char *model_path, const char *text, const int64_t *doc_off,
                            int64_t ndocs, int max_ids, int unk, int nthreads, int passes, int64_t *total_ids,
                            int32_t *out_ids, int64_t *out_counts)
{
    return run_passes(lib_path, model_path, text, doc_off, ndocs, max_ids, unk, nthreads, passes, total_ids, out_ids, out_counts, NULL);
}

/* One pass over the documents that keeps only the id count and bfc_ids_hash of every document (8 + 8 bytes per document
 * instead of max_ids ints): the CPU side of bench.py's full-shard bit-exactness check.  Returns seconds (negative on error). */
double bfc_text_to_ids_hashes(const char *lib_path, const char *model_path, const char *text, const int64_t *doc_off,
                              int64_t ndocs, int max_ids, int unk, int nthreads, int64_t *total_ids,
                              int64_t *out_counts, uint64_t *out_hash)
{
    return run_passes(lib_path, model_path, text, doc_off, ndocs, max_ids, unk, nthreads, 1, total_ids, NULL, out_counts, out_hash);
}

/* ---- exact form of the full-shard check: every id of every document, compact, in document order.
 * Thread t tokenises the contiguous range [t*ndocs/T, (t+1)*ndocs/T) into a buffer of its own; the buffers are then laid end to end.
 * *out_ids is malloc()ed here (release it with bfc_free); out_off[ndocs + 1] receives the id offsets.  Returns seconds (negative on error). */
typedef struct { t2i_fn t2i; void *model; const char *text; const int64_t *off; int64_t d0, d1; int max_ids, unk; int32_t *ids; int64_t n, cap; int64_t *counts; int err; } cjob_t;

static void *cworker(void *arg)
{
    cjob_t *j = (cjob_t *)arg;
    for (int64_t d = j->d0; d < j->d1; ++d) {
        if (j->n + j->max_ids > j->cap) {
            int64_t nc = j->cap * 2 + j->max_ids + 1024;
            int32_t *q = (int32_t *)realloc(j->ids, sizeof(int32_t) * (size_t)nc);
            if (!q) { j->err = 1; return NULL; }
            j->ids = q; j->cap = nc;
        }
        int n = j->t2i(j->model, j->text + j->off[d], (int)(j->off[d + 1] - j->off[d]), j->ids + j->n, j->max_ids, j->unk);
        if (n < 0) n = 0;
        j->counts[d] = n; j->n += n;
    }
    return NULL;
}

double bfc_text_to_ids_compact(const char *lib_path, const char *model_path, const char *text, const int64_t *doc_off, int64_t ndocs,
                               int max_ids, int unk, int nthreads, int32_t **out_ids, int64_t *out_off)
{
    void *lib = dlopen(lib_path, RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "cpu_baseline: dlopen(%s): %s\n", lib_path, dlerror()); return -1.0; }
    load_fn load = (load_fn)dlsym(lib, "LoadModel");
    free_fn fre = (free_fn)dlsym(lib, "FreeModel");
    t2i_fn t2i = (t2i_fn)dlsym(lib, "TextToIds");
    if (!load) { load = (load_fn)dlsym(lib, "bfo_load_model"); fre = (free_fn)dlsym(lib, "bfo_free_model"); t2i = (t2i_fn)dlsym(lib, "bfo_text_to_ids"); }
    if (!load || !fre || !t2i) { fprintf(stderr, "cpu_baseline: missing symbols in %s\n", lib_path); return -2.0; }
    void *model = load(model_path);
    if (!model) { fprintf(stderr, "cpu_baseline: LoadModel(%s) failed\n", model_path); return -3.0; }
    if (nthreads < 1) nthreads = 1;
    if (nthreads > MAX_THREADS) nthreads = MAX_THREADS;
    if (max_ids < 0) max_ids = 0;
    static pthread_t th[MAX_THREADS]; static cjob_t jobs[MAX_THREADS];
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int t = 0; t < nthreads; ++t) {
        memset(&jobs[t], 0, sizeof(cjob_t));
        jobs[t].t2i = t2i; jobs[t].model = model; jobs[t].text = text; jobs[t].off = doc_off; jobs[t].max_ids = max_ids; jobs[t].unk = unk;
        jobs[t].d0 = ndocs * t / nthreads; jobs[t].d1 = ndocs * (t + 1) / nthreads; jobs[t].counts = out_off;     /* counts first, offsets below */
        pthread_create(&th[t], NULL, cworker, &jobs[t]);
    }
    int64_t total = 0; int err = 0;
    for (int t = 0; t < nthreads; ++t) { pthread_join(th[t], NULL); total += jobs[t].n; err |= jobs[t].err; }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    fre(model);
    int32_t *all = err ? NULL : (int32_t *)malloc(sizeof(int32_t) * (size_t)(total > 0 ? total : 1));
    if (!all) err = 1;
    int64_t at = 0;
    for (int t = 0; t < nthreads; ++t) {
        if (!err && jobs[t].n > 0) memcpy(all + at, jobs[t].ids, sizeof(int32_t) * (size_t)jobs[t].n);
        at += jobs[t].n;
        free(jobs[t].ids);
    }
    if (err) { free(all); return -4.0; }
    int64_t acc = 0;
    for (int64_t d = 0; d < ndocs; ++d) { const int64_t c = out_off[d]; out_off[d] = acc; acc += c; }
    out_off[ndocs] = acc;
    *out_ids = all;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
void bfc_free(void *p) { free(p); }

/* ---- TextToWords (BASELINE.json configs[0]): one call per line from T threads, the library's built-in model (tokdll:610-614) ---- */
typedef int (*t2w_fn)(const char *, int, char *, int);
typedef struct { t2w_fn f; const char *text; const int64_t *off; int64_t nlines; int tid, nthreads; int64_t out_bytes; } wjob_t;

static void *wworker(void *arg)
{
    wjob_t *j = (wjob_t *)arg;
    char *buf = (char *)malloc(1 << 16);
    for (int64_t d = j->tid; d < j->nlines; d += j->nthreads) {
        int n = (int)(j->off[d + 1] - j->off[d]);
        int cap = 3 * n + 4; if (cap > (1 << 16)) cap = 1 << 16;
        int r = j->f(j->text + j->off[d], n, buf, cap);
        if (r > 0) j->out_bytes += r;
    }
    free(buf);
    return NULL;
}

/* seconds of the best of `passes` passes over the lines (negative on error); *out_bytes = output bytes of one pass */
double bfc_time_text_to_words(const char *lib_path, const char *text, const int64_t *line_off, int64_t nlines, int nthreads, int passes, int64_t *out_bytes)
{
    void *lib = dlopen(lib_path, RTLD_NOW | RTLD_LOCAL);
    if (!lib) { fprintf(stderr, "cpu_baseline: dlopen(%s): %s\n", lib_path, dlerror()); return -1.0; }
    t2w_fn f = (t2w_fn)dlsym(lib, "TextToWords");
    if (!f) { fprintf(stderr, "cpu_baseline: no TextToWords in %s\n", lib_path); return -2.0; }
    if (nthreads < 1) nthreads = 1;
    if (nthreads > MAX_THREADS) nthreads = MAX_THREADS;
    double best = 1e30; int64_t ob = 0;
    for (int p = 0; p < (passes < 1 ? 1 : passes); ++p) {
        static pthread_t th[MAX_THREADS]; static wjob_t jobs[MAX_THREADS];
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int t = 0; t < nthreads; ++t) {
            memset(&jobs[t], 0, sizeof(wjob_t));
            jobs[t].f = f; jobs[t].text = text; jobs[t].off = line_off; jobs[t].nlines = nlines; jobs[t].tid = t; jobs[t].nthreads = nthreads;
            pthread_create(&th[t], NULL, wworker, &jobs[t]);
        }
        ob = 0;
        for (int t = 0; t < nthreads; ++t) { pthread_join(th[t], NULL); ob += jobs[t].out_bytes; }
        clock_gettime(CLOCK_MONOTONIC, &t1);
        double s = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        if (s < best) best = s;
    }
    if (out_bytes) *out_bytes = ob;
    return best;
}
