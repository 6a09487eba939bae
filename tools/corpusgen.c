/*
 * corpusgen.c -- TEST / BENCH INFRASTRUCTURE: deterministic synthetic corpora.
 *
 * Every document is a pure function of (seed, document index): one splitmix64 stream per
 * document, so any shard of a corpus can be regenerated independently (multi-GPU ranks each
 * build their own contiguous range) and byte-identically on any box.
 *
 * kind "en" (configs 2/3 and the north-star headline): words drawn Zipf(s = 1.07) over a word
 * list, single-space separated; 10 % capitalised, 12 % followed by punctuation, 3 % replaced
 * by a number, 1 % replaced by a random 6-14 letter string (forces multi-piece / UNK words),
 * 0.5 % get a Latin-1 accented vowel (exercises the charmap), flag BFC_MULTIBYTE adds 2 % of
 * documents with CJK / Cyrillic runs.  A document is cut at exactly its target length (the
 * last word may be truncated, which is just another out-of-vocabulary word).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t s; } rng_t;
static uint64_t splitmix64(rng_t *r)
{
    uint64_t z = (r->s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static double rng_unit(rng_t *r) { return (double)(splitmix64(r) >> 11) * (1.0 / 9007199254740992.0); }
static rng_t doc_rng(uint64_t seed, int64_t doc)
{
    rng_t r; r.s = seed * 0xD1342543DE82EF95ull + (uint64_t)doc * 0x2545F4914F6CDD1Dull + 0x1234567ull;
    splitmix64(&r); return r;
}

enum { BFC_LEN_NORMAL = 0, BFC_LEN_LOGUNIFORM = 1 };
enum { BFC_MULTIBYTE = 1 };

typedef struct {
    const uint8_t *words; const int32_t *woff; int nwords; const double *cdf;
    uint64_t seed; int64_t first_doc, ndocs;
    int len_mode; double p0, p1; int minlen, maxlen; int flags;
    uint8_t *out; int64_t *doc_off;
    int tid, nthreads;
} job_t;

static int doc_len(const job_t *j, int64_t doc)
{
    rng_t r = doc_rng(j->seed ^ 0xABCDEF12345ull, doc);
    double v;
    if (j->len_mode == BFC_LEN_LOGUNIFORM) {
        double u = rng_unit(&r);
        v = exp(log((double)j->minlen) + u * (log((double)j->maxlen) - log((double)j->minlen)));
    } else {
        double u1 = rng_unit(&r), u2 = rng_unit(&r);
        if (u1 < 1e-300) u1 = 1e-300;
        v = j->p0 + j->p1 * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    }
    int n = (int)(v + 0.5);
    if (n < j->minlen) n = j->minlen;
    if (n > j->maxlen) n = j->maxlen;
    return n;
}

static int pick_word(const job_t *j, rng_t *r)
{
    double u = rng_unit(r);
    int lo = 0, hi = j->nwords - 1;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (j->cdf[mid] < u) lo = mid + 1; else hi = mid; }
    return lo;
}

static const char PUNCT[] = ".,;:!?'\"()-";
static const char *ACC[5][2] = {{"a", "\xC3\xA1"}, {"e", "\xC3\xA9"}, {"i", "\xC3\xAD"}, {"o", "\xC3\xB6"}, {"u", "\xC3\xBC"}};
static const char *MB_RUNS[] = {"\xE5\xA5\xBD\xE5\xA5\xBD\xE5\xAD\xA6\xE4\xB9\xA0", "\xD0\xBF\xD1\x80\xD0\xB8\xD0\xB2\xD0\xB5\xD1\x82",
                                "\xD0\xBC\xD0\xB8\xD1\x80", "\xC3\xA0 la", "\xE3\x81\x93\xE3\x82\x93\xE3\x81\xAB\xE3\x81\xA1\xE3\x81\xAF",
                                "na\xC3\xAFve", "\xCE\xB1\xCE\xB2\xCE\xB3"};

/* never end on a partial UTF-8 sequence: overwrite a dangling tail with ASCII */
static void fix_tail(uint8_t *dst, int n)
{
    for (int k = n - 1, back = 0; k >= 0 && back < 4; --k, ++back) {
        uint8_t b = dst[k];
        if ((b & 0xC0) == 0x80) continue;
        if (b >= 0xC0) {
            int need = b >= 0xF0 ? 4 : b >= 0xE0 ? 3 : 2;
            if (k + need > n) for (int q = k; q < n; ++q) dst[q] = 'x';
        }
        break;
    }
}

static void fill_doc(const job_t *j, int64_t doc, uint8_t *dst, int n)
{
    rng_t r = doc_rng(j->seed, doc);
    int pos = 0;
    int mb_doc = (j->flags & BFC_MULTIBYTE) && rng_unit(&r) < 0.02;
    uint8_t tmp[64];
    while (pos < n) {
        int len = 0;
        double u = rng_unit(&r);
        if (u < 0.03) {                       /* number */
            int digits = 1 + (int)(splitmix64(&r) % 4);
            for (int k = 0; k < digits; ++k) tmp[len++] = (uint8_t)('0' + splitmix64(&r) % 10);
        } else if (u < 0.04) {                /* random letters */
            int l = 6 + (int)(splitmix64(&r) % 9);
            for (int k = 0; k < l; ++k) tmp[len++] = (uint8_t)('a' + splitmix64(&r) % 26);
        } else if (mb_doc && u < 0.14) {
            const char *s = MB_RUNS[splitmix64(&r) % (sizeof(MB_RUNS) / sizeof(MB_RUNS[0]))];
            len = (int)strlen(s); memcpy(tmp, s, (size_t)len);
        } else {
            int w = pick_word(j, &r);
            int wl = j->woff[w + 1] - j->woff[w];
            if (wl > 40) wl = 40;
            memcpy(tmp, j->words + j->woff[w], (size_t)wl); len = wl;
            if (rng_unit(&r) < 0.10 && tmp[0] >= 'a' && tmp[0] <= 'z') tmp[0] = (uint8_t)(tmp[0] - 32);
            if (rng_unit(&r) < 0.005) {
                for (int k = 0; k < len; ++k) {
                    int done = 0;
                    for (int a = 0; a < 5 && !done; ++a) if (tmp[k] == (uint8_t)ACC[a][0][0]) {
                        memmove(tmp + k + 2, tmp + k + 1, (size_t)(len - k - 1));
                        tmp[k] = (uint8_t)ACC[a][1][0]; tmp[k + 1] = (uint8_t)ACC[a][1][1]; len++; done = 1;
                    }
                    if (done) break;
                }
            }
        }
        if (rng_unit(&r) < 0.12) tmp[len++] = (uint8_t)PUNCT[splitmix64(&r) % (sizeof(PUNCT) - 1)];
        tmp[len++] = ' ';
        if (len > n - pos) len = n - pos;
        memcpy(dst + pos, tmp, (size_t)len);
        pos += len;
    }
    fix_tail(dst, n);
}

static void *worker(void *arg)
{
    const job_t *j = (const job_t *)arg;
    for (int64_t d = j->tid; d < j->ndocs; d += j->nthreads)
        fill_doc(j, j->first_doc + d, j->out + j->doc_off[d], (int)(j->doc_off[d + 1] - j->doc_off[d]));
    return NULL;
}

/* Fills doc_off[ndocs+1] (always) and, if out != NULL, the documents.  Returns the total byte count. */
int64_t bfc_gen(const uint8_t *words, const int32_t *woff, int nwords, const double *cdf, uint64_t seed,
                int64_t first_doc, int64_t ndocs, int len_mode, double p0, double p1, int minlen, int maxlen,
                int flags, uint8_t *out, int64_t *doc_off, int nthreads)
{
    job_t base;
    memset(&base, 0, sizeof(base));
    base.words = words; base.woff = woff; base.nwords = nwords; base.cdf = cdf; base.seed = seed;
    base.first_doc = first_doc; base.ndocs = ndocs; base.len_mode = len_mode; base.p0 = p0; base.p1 = p1;
    base.minlen = minlen; base.maxlen = maxlen; base.flags = flags; base.out = out; base.doc_off = doc_off;
    doc_off[0] = 0;
    for (int64_t d = 0; d < ndocs; ++d) doc_off[d + 1] = doc_off[d] + doc_len(&base, first_doc + d);
    if (out) {
        if (nthreads < 1) nthreads = 1;
        if (nthreads > 64) nthreads = 64;
        pthread_t th[64]; job_t jobs[64];
        for (int t = 0; t < nthreads; ++t) { jobs[t] = base; jobs[t].tid = t; jobs[t].nthreads = nthreads; pthread_create(&th[t], NULL, worker, &jobs[t]); }
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    }
    return doc_off[ndocs];
}

/*
 * kind "multi" (configs 4 / 5): every document picks a script bucket (bucket_cdf has nb + 1 entries,
 * the last one = "mixed": a new bucket for every piece) and is a concatenation of vocabulary pieces of that bucket drawn
 * Zipf(s) over their rank (cdf[] holds one cumulative distribution per bucket, laid end to end like the pieces); a piece
 * that starts with U+2581 starts a new word (the mark becomes a space), any other piece continues the current word; after
 * a piece, with probability 1 % per character of it, one character that is a key of the model's charmap is appended
 * (ncm == 0: none).  Lengths as in bfc_gen (normal, clipped); documents are cut at their target length on a character
 * boundary.  Piece lists: tests/data/pieces_*.tsv.gz (tools/make_piece_lists.py).
 */
typedef struct {
    const uint8_t *blob; const int32_t *poff; const int32_t *boff; int nb; const double *cdf; const double *bucket_cdf;
    const uint8_t *cm_blob; const int32_t *cm_off; int ncm;
    job_t base;
} mjob_t;

static void fill_doc_multi(const mjob_t *m, int64_t doc, uint8_t *dst, int n)
{
    rng_t r = doc_rng(m->base.seed, doc);
    int pos = 0;
    double ub = rng_unit(&r);
    int b = 0;
    while (b < m->nb && m->bucket_cdf[b] < ub) ++b;              /* b == nb: mixed */
    while (pos < n) {
        const int bb = b < m->nb ? b : (int)(splitmix64(&r) % (uint64_t)m->nb);
        const int p0 = m->boff[bb], p1 = m->boff[bb + 1];
        if (p1 <= p0) { dst[pos++] = ' '; continue; }
        const double u = rng_unit(&r);
        int lo = p0, hi = p1 - 1;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (m->cdf[mid] < u) lo = mid + 1; else hi = mid; }
        const uint8_t *s = m->blob + m->poff[lo];
        int len = m->poff[lo + 1] - m->poff[lo], nch = 0;
        if (len >= 3 && s[0] == 0xE2 && s[1] == 0x96 && s[2] == 0x81) { dst[pos++] = ' '; s += 3; len -= 3; nch = 1; }
        for (int k = 0; k < len; ++k) nch += (s[k] & 0xC0) != 0x80;
        if (len > n - pos) len = n - pos;
        memcpy(dst + pos, s, (size_t)len);
        pos += len;
        if (m->ncm > 0 && pos < n && rng_unit(&r) < 0.01 * nch) {
            const int c = (int)(splitmix64(&r) % (uint64_t)m->ncm);
            int cl = m->cm_off[c + 1] - m->cm_off[c];
            if (cl > n - pos) cl = n - pos;
            memcpy(dst + pos, m->cm_blob + m->cm_off[c], (size_t)cl);
            pos += cl;
        }
    }
    fix_tail(dst, n);
}

static void *worker_multi(void *arg)
{
    const mjob_t *m = (const mjob_t *)arg;
    const job_t *j = &m->base;
    for (int64_t d = j->tid; d < j->ndocs; d += j->nthreads)
        fill_doc_multi(m, j->first_doc + d, j->out + j->doc_off[d], (int)(j->doc_off[d + 1] - j->doc_off[d]));
    return NULL;
}

int64_t bfc_gen_multi(const uint8_t *blob, const int32_t *poff, const int32_t *boff, int nb, const double *cdf, const double *bucket_cdf,
                      const uint8_t *cm_blob, const int32_t *cm_off, int ncm, uint64_t seed, int64_t first_doc, int64_t ndocs,
                      double mean, double sd, int minlen, int maxlen, uint8_t *out, int64_t *doc_off, int nthreads)
{
    mjob_t base;
    memset(&base, 0, sizeof(base));
    base.blob = blob; base.poff = poff; base.boff = boff; base.nb = nb; base.cdf = cdf; base.bucket_cdf = bucket_cdf;
    base.cm_blob = cm_blob; base.cm_off = cm_off; base.ncm = ncm;
    base.base.seed = seed; base.base.first_doc = first_doc; base.base.ndocs = ndocs; base.base.len_mode = BFC_LEN_NORMAL;
    base.base.p0 = mean; base.base.p1 = sd; base.base.minlen = minlen; base.base.maxlen = maxlen; base.base.out = out; base.base.doc_off = doc_off;
    doc_off[0] = 0;
    for (int64_t d = 0; d < ndocs; ++d) doc_off[d + 1] = doc_off[d] + doc_len(&base.base, first_doc + d);
    if (out) {
        if (nthreads < 1) nthreads = 1;
        if (nthreads > 64) nthreads = 64;
        pthread_t th[64]; static mjob_t jobs[64];
        for (int t = 0; t < nthreads; ++t) { jobs[t] = base; jobs[t].base.tid = t; jobs[t].base.nthreads = nthreads; pthread_create(&th[t], NULL, worker_multi, &jobs[t]); }
        for (int t = 0; t < nthreads; ++t) pthread_join(th[t], NULL);
    }
    return doc_off[ndocs];
}
