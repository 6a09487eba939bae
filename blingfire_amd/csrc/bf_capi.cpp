// bf_capi.cpp -- extern "C" surface of libblingfiretokdll.so (see include/blingfiretokdll_amd.h)
// and the host runtime around the kernels: device-resident tables, grow-only workspaces,
// stream/event plumbing.  No CPU tokenisation path exists here: every entry point that
// produces ids launches the HIP kernels, and fails loudly if there is no device.
#include "../../include/blingfiretokdll_amd.h"
#include "bf_internal.h"
#include "bf_kernels.h"
#include "bf_model.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

using namespace bfa;

namespace {

thread_local std::string g_last_error;

// a spin-wait hint (x86: pause; elsewhere nothing -- the loops it sits in are bounded and fall back to a futex wait)
static inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#endif
}

bool hip_ok(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    g_last_error = std::string(what) + ": " + hipGetErrorString(e);
    fprintf(stderr, "[blingfire_amd] HIP error in %s: %s\n", what, hipGetErrorString(e));
    return false;
}

struct DevBuf {
    void *p = nullptr; size_t cap = 0;
    // Grow-only.  Growing is a hipMalloc + hipFree (both synchronise the device): callers that must not synchronise size the
    // workspaces once with BfReserve.
    bool reserve(size_t bytes)
    {
        if (bytes <= cap) return true;
        size_t want = bytes + bytes / 8 + 256;
        void *q = nullptr;
        // the contents are never kept across a grow; the old buffer goes first when both do not fit
        if (hipMalloc(&q, want) != hipSuccess) {
            (void)hipGetLastError();
            if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
            if (!hip_ok(hipMalloc(&q, want), "hipMalloc(workspace)")) return false;
        } else if (p) (void)hipFree(p);
        p = q; cap = want; return true;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return (T *)p; }
};

template <class T> bool upload(DevBuf &b, const std::vector<T> &v, size_t pad_elems = 0)
{
    const size_t bytes = (v.size() + pad_elems) * sizeof(T);
    if (!b.reserve(bytes ? bytes : 16)) return false;
    if (pad_elems && !hip_ok(hipMemset(b.p, 0, b.cap), "hipMemset")) return false;
    if (!v.empty() && !hip_ok(hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy(table)")) return false;
    return true;
}

// page-locked host memory (grow-only), the staging side of the chunked host-buffer path
struct PinBuf {
    void *p = nullptr; size_t cap = 0;
    bool reserve(size_t bytes)
    {
        if (bytes <= cap) return true;
        const size_t want = bytes + bytes / 4 + 4096;
        void *q = nullptr;
        if (!hip_ok(hipHostMalloc(&q, want, hipHostMallocDefault), "hipHostMalloc(staging)")) return false;
        if (p) (void)hipHostFree(p);
        p = q; cap = want; return true;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
    template <class T> T *as() const { return (T *)p; }
};

enum { EV_BEGIN = 0, EV_PREP, EV_TOK, EV_SCAN, EV_COMPACT, EV_DOM0, EV_DOM1, EV_COUNT };      // EV_DOM0 / 1: around the dominant kernel of the step (the one a roofline is about)

// TextToIdsBatch on host buffers, large batches: the batch is cut into chunks that flow through NS slots of page-locked staging
// and device buffers -- while chunk k is tokenised, chunk k+1 is copied in (CPU threads -> pinned -> DMA) and the ids of the chunks
// before it are copied out (DMA -> pinned -> CPU threads).  Created on first use.
struct HostPipe {
    static constexpr int NS = 3;        // slots: chunk k is copied in while k-1 waits for / runs its tokenisation and the ids of k-2 leave
    PinBuf pin_text[NS], pin_off[NS], pin_idoff[NS], pin_ids[NS], pin_status;
    DevBuf dev_text[NS], dev_off[NS], dev_ids[NS], dev_idoff[NS];
    hipStream_t s_in = nullptr, s_out = nullptr, s_meta = nullptr;     // copies in, ids out, per-chunk offsets out (never behind a bulk copy)
    hipEvent_t ev_h2d[NS] = {}, ev_cmp[NS] = {}, ev_d2h[NS] = {};
    bool ready = false;
    bool init()
    {
        if (ready) return true;
        if (!hip_ok(hipStreamCreateWithFlags(&s_in, hipStreamNonBlocking), "hipStreamCreate") || !hip_ok(hipStreamCreateWithFlags(&s_out, hipStreamNonBlocking), "hipStreamCreate") ||
            !hip_ok(hipStreamCreateWithFlags(&s_meta, hipStreamNonBlocking), "hipStreamCreate")) return false;
        for (int i = 0; i < NS; ++i)
            if (!hip_ok(hipEventCreateWithFlags(&ev_h2d[i], hipEventDisableTiming), "hipEventCreate") || !hip_ok(hipEventCreateWithFlags(&ev_cmp[i], hipEventDisableTiming), "hipEventCreate") ||
                !hip_ok(hipEventCreateWithFlags(&ev_d2h[i], hipEventDisableTiming), "hipEventCreate")) return false;
        if (!pin_status.reserve(64)) return false;
        ready = true; return true;
    }
    void release()
    {
        for (int i = 0; i < NS; ++i) {
            pin_text[i].release(); pin_off[i].release(); pin_idoff[i].release(); pin_ids[i].release();
            dev_text[i].release(); dev_off[i].release(); dev_ids[i].release(); dev_idoff[i].release();
            if (ev_h2d[i]) (void)hipEventDestroy(ev_h2d[i]);
            if (ev_cmp[i]) (void)hipEventDestroy(ev_cmp[i]);
            if (ev_d2h[i]) (void)hipEventDestroy(ev_d2h[i]);
        }
        pin_status.release();
        if (s_in) (void)hipStreamDestroy(s_in);
        if (s_out) (void)hipStreamDestroy(s_out);
        if (s_meta) (void)hipStreamDestroy(s_meta);
    }
};

// Makes the handle's device current for the duration of a call and puts the caller's device back afterwards (a drop-in
// library must not change the calling thread's current device behind its back).
struct DeviceGuard {
    int prev = -1; bool switched = false, ok = true;
    explicit DeviceGuard(int dev)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) { ok = hip_ok(hipSetDevice(dev), "hipSetDevice"); switched = ok && prev >= 0; }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

struct Handle {
    uint32_t magic = 0xB1F14E01u;
    Model m;
    int device = 0;
    int variant = 3;
    bool last_uni_cut = false;          // the last Unigram batch took the cut form (BfTokeniseKernel / BfStepKernels)
    std::mutex mu;
    // held while the ids of a batch wait in this handle's buffers (w_ids / w_starts / w_ends) for their place in the caller's array: a range of a
    // sharded call from its kernels to its copy out (run_host_sharded), any other host-buffer call for its whole duration -- the second call
    // on a sharded handle (or a call on the handle of range 0 itself) cannot overwrite what the first has not fetched yet
    std::mutex defer_mu;
    // device tables
    DevBuf t_wbd, t_info, t_acts, t_cp_l1, t_cp_pages, t_multi, t_i2w_off, t_i2w_data;
    DevBuf t_kind;                                               // unit-form lexers: what a walk that starts on each class does (bf_wave.h)
    bool lex_stats = false;                                      // BF_LEX_STATS=1 at LoadModel: instrumented kernel instances (experiments)
    DevBuf t_wcp_l1, t_wcp_pages;                                // TextToWords: code point -> class without the charmap
    DevBuf t_bpetab;                                             // _sp BPE: the word table of the wave program
    DevBuf t_segscore, t_segid;                                  // _sp Unigram: the rows' scores alone; their ids alone (the cut form's compaction)
    DevBuf t_dict, t_seginfo;                                    // _sp: Mealy table, I2Info rows (code-point maps reuse t_cp_*/t_multi)
    DevBuf t_dk_l1, t_dk_pages, t_dn_l1, t_dn_pages, t_dn_pool, t_k2i, t_rows;   // key -> info lookup (uploaded on first use)
    bool dict_ready = false;
    DevBuf w_keys, w_keyoff, w_dids, w_dret, w_vals;              // DictGetInfoBatch staging
    DevBuf w_s1, w_s2, w_s3, w_s4, w_perm, w_hist, w_narcs;      // _sp scratch
    DevBuf w_bwflags;                                            // BPE wave program: documents handed back
    DevBuf w_big;                                                // BPE: pool of the documents beyond the per-document arc reserve (k_bpe_seg)
    size_t bpe_pool_bytes = (size_t)64 << 20;                    // its size (BfSetBpePoolBytes; the host-buffer calls grow it when a batch needs more)
    DevBuf t_bpe_prio, t_bpe_place;                              // BPE with merges: the arc order as integers (bf_model.h bpe_prio / bpe_place_id)
    // workspaces
    DevBuf w_cls, w_nchars, w_tmp, w_counts, w_bsums, w_misc, w_flags, w_out, w_outoff;   // w_misc: [0] next_doc (u64), [2] status (int)
    DevBuf w_text, w_docoff, w_ids, w_idoff, w_starts, w_ends;  // host-API staging
    DevBuf w_srcoff, w_span;                                    // offsets API: source-offset stream, staged id spans
    DevBuf w_preplong, w_w2tlong;                               // k_prep_wp_long / k_w2t_copy_long: the documents of more than 2048 bytes / 1024 tokens
    DevBuf w_long;                                              // words modes: the long-document path (bf_kernels.h LexLongParams: list, spec, vis)
    DevBuf w_espan, w_hspan, w_chard;                           // the flat program, offsets API: spans of the entries, of the pieces at the homes; counts of the documents handed back
    DevBuf w_ent, w_home, w_entoff, w_entcnt, w_dstat, w_ranges, w_list, w_wrec, t_flat;   // the flat program (bf_flat.h): entries, homes, per-document records, ranges, the documents handed back; its word table
    bool last_flat = false;                                      // the last batch took the flat program (BfLastKernelMs names the kernels by it)
    // single-document calls that arrive while a batch is in flight are combined into the next launch (text_to_ids_one)
    struct OneReq { const char *s; int n; int32_t *ids; int max_ids, unk; int32_t *starts, *ends; int result; std::atomic<int> state; };
    std::mutex q_mu; std::vector<OneReq *> q; bool q_leader = false; std::atomic<int> q_spinners{0}, q_sleepers{0};
    std::atomic<long long> one_rounds{0}, one_reqs{0}, one_ns{0};          // BF_TRACE_ONE=1: launches for single-document calls, requests served, time inside them
    hipStream_t stream = nullptr;
    hipEvent_t ev[EV_COUNT] = {};
    bool ev_valid = false;
    // BfSetDevices: the devices a host batch is range-sharded over.  shards[g] tokenises range g; entry 0 is this handle itself when the
    // first device is its own, every other entry a handle of its own (tables, workspaces, streams) on its device
    std::vector<Handle *> shards;
    // small host batches of a unit-form WordPiece model (the single-document entry points above all): text, offsets, counts and the id
    // staging live in ONE block of mapped page-locked memory that the wave kernel reads and writes directly -- a call is one launch
    // and one synchronisation, no copies, no scan / compaction kernels (run_host_mapped)
    PinBuf m_small; void *m_small_dev = nullptr;
    int small_status = -1;                                     // status word of the last batch when it took that path (BfLastStatus), else -1
    HostPipe pipe;                                              // chunked host-buffer path (run_host_chunked)
    int64_t host_chunk_bytes = 128ll << 20;                     // its largest chunk (BfSetHostChunkBytes; 0 = never chunk); batches of at least this size take it
    ~Handle()
    {
        for (Handle *c : shards) if (c && c != this) { DeviceGuard dg(c->device); (void)hipDeviceSynchronize(); delete c; }
        shards.clear();
        pipe.release(); m_small.release();
        for (DevBuf *b : {&t_segscore, &t_segid, &t_bpetab, &t_bpe_prio, &t_bpe_place, &t_dk_l1, &t_dk_pages, &t_dn_l1, &t_dn_pages, &t_dn_pool, &t_k2i, &t_rows, &w_keys, &w_keyoff, &w_dids, &w_dret, &w_vals, &t_i2w_off, &t_i2w_data, &t_kind, &t_wbd, &t_info, &t_acts, &t_cp_l1, &t_cp_pages, &t_multi, &t_wcp_l1, &t_wcp_pages, &t_dict, &t_seginfo, &w_s1, &w_s2, &w_s3, &w_s4, &w_big, &w_perm, &w_hist, &w_narcs, &w_bwflags, &w_cls, &w_nchars, &w_tmp, &w_counts, &w_flags, &w_out, &w_outoff,
                          &w_bsums, &w_misc, &w_text, &w_docoff, &w_ids, &w_idoff, &w_starts, &w_ends, &w_srcoff, &w_span, &w_long, &w_preplong, &w_w2tlong, &w_ent, &w_home, &w_entoff, &w_entcnt, &w_dstat, &w_ranges, &w_list, &w_wrec, &t_flat, &w_espan, &w_hspan, &w_chard}) b->release();
        for (auto &e : ev) if (e) (void)hipEventDestroy(e);
        if (stream) (void)hipStreamDestroy(stream);
        magic = 0;
    }
};

Handle *make_handle(const uint8_t *img, size_t size);

// The reference's built-in word-breaking model (tokdll:175-183,426-434: g_DefaultWbd = the bytes of ldbsrc/ldb/wbd.bin compiled
// into the library).  Here the same file (models/wbd.bin, unchanged data) is embedded with .incbin at build time.
#if !defined(__HIP_DEVICE_COMPILE__)
__asm__(".section .rodata\n.balign 16\n.global bf_default_wbd_begin\nbf_default_wbd_begin:\n.incbin \"" BF_DEFAULT_WBD_PATH "\"\n"
        ".global bf_default_wbd_end\nbf_default_wbd_end:\n.byte 0\n.previous\n");
#endif
__asm__(".section .rodata\n.balign 16\n.global bf_default_sbd_begin\nbf_default_sbd_begin:\n.incbin \"" BF_DEFAULT_SBD_PATH "\"\n"
        ".global bf_default_sbd_end\nbf_default_sbd_end:\n.byte 0\n.previous\n");
extern "C" const unsigned char bf_default_wbd_begin[], bf_default_wbd_end[], bf_default_sbd_begin[], bf_default_sbd_end[];

Handle *default_sbd()
{
    // the reference's g_DefaultSbd (tokdll:38,124-136): sbd.bin compiled into the library, set up on first use
    static std::mutex mu; static Handle *h = nullptr; static bool tried = false;
    std::lock_guard<std::mutex> lock(mu);
    if (!tried) { tried = true; h = make_handle(bf_default_sbd_begin, (size_t)(bf_default_sbd_end - bf_default_sbd_begin)); }
    return h;
}

Handle *default_wbd()
{
    static std::mutex mu; static Handle *h = nullptr; static bool tried = false;
    std::lock_guard<std::mutex> lock(mu);
    if (!tried) { tried = true; h = make_handle(bf_default_wbd_begin, (size_t)(bf_default_wbd_end - bf_default_wbd_begin)); }
    return h;
}

// model-free entry points (NormalizeSpaces, TextToHashes) run on a handle that owns only a stream and workspaces
Handle *util_handle()
{
    static std::mutex mu; static Handle *h = nullptr; static bool tried = false;
    std::lock_guard<std::mutex> lock(mu);
    if (tried) return h;
    tried = true;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_last_error = "no HIP device available: this library has no CPU path";
        fprintf(stderr, "[blingfire_amd] %s\n", g_last_error.c_str());
        return nullptr;
    }
    Handle *u = new Handle();
    u->m.kind = KIND_I2W;                                   // no tokenizer behind it
    if (hipGetDevice(&u->device) != hipSuccess) u->device = 0;
    bool ok = hip_ok(hipStreamCreateWithFlags(&u->stream, hipStreamNonBlocking), "hipStreamCreate");
    for (auto &e : u->ev) ok = ok && hip_ok(hipEventCreate(&e), "hipEventCreate");
    ok = ok && u->w_misc.reserve(256) && hip_ok(hipMemset(u->w_misc.p, 0, 256), "hipMemset");
    if (!ok) { fprintf(stderr, "[blingfire_amd] %s\n", g_last_error.c_str()); delete u; return nullptr; }
    h = u;
    return h;
}

Handle *as_handle(void *p)
{
    Handle *h = (Handle *)p;
    return (h && h->magic == 0xB1F14E01u) ? h : nullptr;
}

Handle *make_handle(const uint8_t *img, size_t size)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_last_error = "no HIP device available: this library has no CPU path";
        fprintf(stderr, "[blingfire_amd] %s\n", g_last_error.c_str());
        return nullptr;
    }
    Handle *h = new Handle();
    if (!build_model(h->m, img, size)) {
        g_last_error = "cannot load model: " + h->m.error;
        fprintf(stderr, "[blingfire_amd] %s\n", g_last_error.c_str());
        delete h; return nullptr;
    }
    if (hipGetDevice(&h->device) != hipSuccess) h->device = 0;
    if (const char *v = getenv("BF_LEX_VARIANT")) h->variant = atoi(v);      // experiments only
    h->lex_stats = getenv("BF_LEX_STATS") != nullptr;
    Model &m = h->m;
    bool ok = true;
    if (m.kind == KIND_WP) {
        if (m.acts_pool.size() > 4096) { g_last_error = "lexer action pool exceeds the LDS staging limit (4096 ints)"; fprintf(stderr, "[blingfire_amd] %s\n", g_last_error.c_str()); delete h; return nullptr; }
        if (m.max_depth > LEX_MAX_DEPTH) { g_last_error = "lexer max-depth exceeds the supported 4"; fprintf(stderr, "[blingfire_amd] %s\n", g_last_error.c_str()); delete h; return nullptr; }
        ok = ok && upload(h->t_wbd, m.wbd_t2, 16) && upload(h->t_acts, m.acts_pool, 16) &&
             upload(h->t_cp_l1, m.wbd_cpmap.l1) && upload(h->t_cp_pages, m.wbd_cpmap.pages) && upload(h->t_multi, m.wbd_multi_pool, 16) &&
             upload(h->t_wcp_l1, m.words_cpmap.l1) && upload(h->t_wcp_pages, m.words_cpmap.pages) && upload(h->t_kind, m.wave_kind, 16);
        if (m.flat_ok) ok = ok && upload(h->t_flat, m.flat_tab, 16);
    } else if (m.kind != KIND_I2W) {
        ok = ok && upload(h->t_dict, m.dict.t64, 16) && upload(h->t_seginfo, m.seg_info, 16) &&
             upload(h->t_cp_l1, m.sp_cpmap.l1) && upload(h->t_cp_pages, m.sp_cpmap.pages) && upload(h->t_multi, m.sp_multi_pool, 16);
        if (m.kind == KIND_BPE_MERGES) ok = ok && upload(h->t_bpe_prio, m.bpe_prio, 16) && upload(h->t_bpe_place, m.bpe_place_id, 16);
        if (!m.bpe_tab.empty()) ok = ok && upload(h->t_bpetab, m.bpe_tab, 16);
        if (m.kind == KIND_UNIGRAM) ok = ok && upload(h->t_segscore, m.seg_score, 16) && upload(h->t_segid, m.i2info_id, 16);
    }
    if ((m.kind == KIND_BPE || m.kind == KIND_BPE_OPT || m.kind == KIND_BPE_MERGES) && !m.bpe_seg_ok) {
        // k_bpe_seg (the kernel every BPE document can end up in) packs an arc's id into 20 bits, its length - 1 into 8 and a place into 21
        g_last_error = "BPE model outside the limits of the segmenter (ids in [0, 2^20), entries of at most 256 symbols, ranks that order)";
        fprintf(stderr, "[blingfire_amd] %s\n", g_last_error.c_str()); delete h; return nullptr;
    }
    if (m.has_i2w) ok = ok && upload(h->t_i2w_off, m.i2w_off, 4) && upload(h->t_i2w_data, m.i2w_data, 16);
    ok = ok && hip_ok(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking), "hipStreamCreate");
    for (auto &e : h->ev) ok = ok && hip_ok(hipEventCreate(&e), "hipEventCreate");
    ok = ok && h->w_misc.reserve(256) && hip_ok(hipMemset(h->w_misc.p, 0, 256), "hipMemset");
    if (!ok) { fprintf(stderr, "[blingfire_amd] %s\n", g_last_error.c_str()); delete h; return nullptr; }
    return h;
}

// the Unigram lane program (bf_seg.h UniLane) keeps `depth` window entries in LDS and packs id + 1 into 20 bits
bool uni_lane_ok(const Model &m) { return m.trie_max_depth > 0 && m.trie_max_depth <= 32 && m.seg_info.size() <= (size_t)UNI_MAX_ID && m.max_info_id < 0x7fffffff; }      // the records carry the row's key

// Workspaces of the TextToIds pipeline for a batch of ndocs documents / total_bytes bytes (grow-only; see DevBuf::reserve).

// The WordPiece path of a unit-form lexer (every BERT model) is the wave program of bf_wave.h: ids only.  The offsets API and the
// TextToWords forms, and lexers outside the unit form, take the lane-per-document kernels (bf_lex.h).  Variant 2 (experiments, A/B):
// the lane-per-document kernels for every model.
bool use_wave(const Handle *h, bool want_off, int words)
{
    (void)want_off;                      // the offsets API has an instance of the wave program of its own (bf_wave_body.h OFFS)
    return h->m.kind == KIND_WP && h->m.wave_ok && !words && (h->variant & 0xff) != 2;
}

// Batches of a flat-form model (bf_flat.h; every BERT model) that are large enough to fill the device take the flat program: ids, and ids
// with their byte offsets.  Variant 4 (tests): every batch; variant 5 (A/B): never.
bool use_flat(const Handle *h, bool want_off, int words, int64_t ndocs, int64_t total_bytes)
{
    if (!use_wave(h, want_off, words) || !h->m.flat_ok || ndocs <= 0) return false;
    const int v = h->variant & 0xff;
    if (v == 4) return true;
    return v == 3 && ndocs >= 1024 && total_bytes >= (1 << 20);
}

// the BPE wave program (bf_bpe_wave_body.h) in front of the lane-per-document kernels, for the models its load-time analysis admits
// (Model::bpe_wave_ok); BfSetVariant bit 0x40 switches it off (A/B runs against the lane-per-document kernels alone)
bool use_bpe_wave(const Handle *h, bool want_off) { return h->m.bpe_wave_ok && !want_off && (h->variant & 0x40) == 0; }

// Words modes (TextToWords / TextToSentences): documents of more than `thresh` characters go through the long-document path
// (bf_lex.h lex_one_start), which spends about three times the transitions of the lane kernel on a document of short words but spreads
// them over as many lanes as the document has characters.  It pays where the batch has too few documents to fill the chip or one
// that is much longer than the rest: a lane walks 1.8 us per character when it is the last one running (config 1's 8,396-byte line:
// 4.9 ms), the chip as a whole 75 ps per character (1 M lines: 3.25 ms), so a document of more than total_bytes / 24,000 characters
// would hold the batch up (measured on MI355X, profiles/r06_words_*: 10,000 lines 4.6 -> 0.55 ms at 16, 1 M lines 3.25 ms at 128 and
// 10.2 ms at 16).  BfSetVariant: bit 0x40000000 = off (every document on one lane), bits 12..15 = k > 0: thresh = 8 << k, bit 0x20000000 = a test
// knob: the triple buffer of the words modes holds n / 8 triples instead of n (so that tests reach the position at which it fills), bit 0x10000000 =
// another: the workspace has room for 40 chunks only (so that tests reach the documents that do not fit and stay on lanes).
// The capacities are bounds that hold for any batch of these sizes (a listed document has more than thresh bytes and owns
// (n + 1 + 63) / 64 chunks) unless that is more than LONG_MAX_CHUNKS: then the documents that do not fit stay on lanes.
// The same rule holds for a lexer whose table does not fit LDS (sbd.bin, 1 MB as 100-byte documents: 0.35 ms here, 1.23 ms on lanes; 16 MB as
// 100-byte documents: 3.6 ms here, 2.7 ms on lanes -- the rule sends the first batch here and keeps the second on lanes).  (A first
// measurement said otherwise because of ONE start position: the reference's test file has lines with a run of a hundred spaces, the
// action of sbd.bin's rule at such a line calls a function that starts again at every space -- 6,440 sequential steps, 4 ms on one lane
// whatever runs it; profiles/r06_words_sentences_*.)
constexpr int LONG_THRESH_MIN = 16, LONG_BIG_CELLS = 1 << 19;
constexpr int64_t LONG_BYTES_PER_THRESH = 24000;
constexpr int64_t LONG_MAX_CHUNKS = (int64_t)2 << 20;           // 128 M cells: 5 .. 7 GB of workspace
struct LongCaps { int thresh; int64_t docs, chunks; size_t list_off, spec_off, jump_off, tok2_off, entry_off, jump2_off, entry2_off, bytes; int big_cells; };
LongCaps long_caps(const Handle *h, int64_t ndocs, int64_t total_bytes, int words)
{
    LongCaps c{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (!words || h->m.kind != KIND_WP || h->m.max_depth < 1 || h->m.lexer_void || (h->variant & 0x40000000) || ndocs <= 0) return c;
    const int k = (h->variant >> 12) & 0xf;
    c.thresh = k ? (8 << k) : (int)std::min<int64_t>(std::max<int64_t>(LONG_THRESH_MIN, total_bytes / LONG_BYTES_PER_THRESH), 1 << 30);
    c.docs = std::min<int64_t>(ndocs, total_bytes / ((int64_t)c.thresh + 1)) + 1;
    c.chunks = std::min<int64_t>(total_bytes / 64 + 2 * c.docs + 1, (h->variant & 0x10000000) ? 40 : LONG_MAX_CHUNKS);      // (0x10000000: a test knob -- room for 40 chunks only)
    c.list_off = 0;
    c.spec_off = ((size_t)c.docs * sizeof(LexLongDoc) + 255) & ~(size_t)255;
    c.jump_off = c.spec_off + (size_t)c.chunks * 64 * 16;
    c.tok2_off = c.jump_off + (size_t)c.chunks * 64 * 16;
    c.entry_off = c.tok2_off + (size_t)c.chunks * 64 * 8;
    // documents of more than LONG_BIG_CELLS cells take the chain in two levels (bf_kernels.hip k_lex_long_jump2 / _chain2 / _chain3): 16 more bytes
    // per cell, reserved only for batches that can hold such a document.  BfSetVariant 0x08000000: a test knob -- from 256 cells on
    c.big_cells = (h->variant & 0x08000000) ? 256 : LONG_BIG_CELLS;
    const bool big = total_bytes + 1 > c.big_cells;
    c.jump2_off = c.entry_off + (size_t)c.chunks * 16;
    c.entry2_off = c.jump2_off + (big ? (size_t)c.chunks * 64 * 16 : 0);
    c.bytes = c.entry2_off + (big ? (size_t)c.chunks * 16 : 0);
    return c;
}

bool reserve_ids_workspaces(Handle *h, int64_t ndocs, int64_t total_bytes, bool want_off, int words = 0, bool with_long = true)
{
    if (const LongCaps lc = long_caps(h, ndocs, total_bytes, words); with_long && lc.thresh > 0 && !h->w_long.reserve(lc.bytes)) return false;
    const Model &m = h->m;
    const int nblocks = scan_nblocks(ndocs);
    if (!h->w_nchars.reserve((size_t)(ndocs + 1) * 4) || !h->w_counts.reserve((size_t)(ndocs + 1) * 4) ||
        !h->w_bsums.reserve((size_t)(nblocks + 1) * 8) || !h->w_tmp.reserve((size_t)(total_bytes + 8 * ndocs + 64) * 4)) return false;
    if (m.kind == KIND_WP) {
        if (use_flat(h, want_off, words, ndocs, total_bytes) &&
            (!h->w_ent.reserve((size_t)(total_bytes + 64) * 4) || !h->w_home.reserve((size_t)(total_bytes + 64) * 4) || !h->w_entoff.reserve((size_t)(ndocs + 1) * 8) ||
             !h->w_entcnt.reserve((size_t)(ndocs + 1) * 4) || !h->w_dstat.reserve((size_t)(ndocs + 1) * 4) || !h->w_list.reserve((size_t)(ndocs + 1) * 4) ||
             !h->w_ranges.reserve((size_t)(wp_flat_ranges(ndocs, total_bytes) + 2) * 8) || !h->w_wrec.reserve((size_t)((total_bytes >> WF_REC_SHIFT) + 64) * 16 + (size_t)(wp_flat_ranges(ndocs, total_bytes) + 2) * 8) ||
             (want_off && (!h->w_espan.reserve((size_t)(total_bytes + 64) * 4) || !h->w_hspan.reserve((size_t)(total_bytes + 64) * 8) || !h->w_chard.reserve((size_t)(ndocs + 1) * 4))))) return false;
        if (use_wave(h, want_off, words))                              // no class stream, no dirty flags
            return !want_off || h->w_span.reserve((size_t)(total_bytes + 8 * ndocs + 64) * 8);
        if (!h->w_cls.reserve((size_t)(total_bytes + 64) * 2) || !h->w_flags.reserve((size_t)((total_bytes >> 10) + 2) * 8)) return false;
        if (want_off && (!h->w_srcoff.reserve((size_t)(total_bytes + 64) * 4) || !h->w_span.reserve((size_t)(total_bytes + 8 * ndocs + 64) * 8))) return false;
        return true;
    }
    const size_t cap = (size_t)(m.dict_has_charmap ? 2 : 1) * (size_t)(total_bytes + ndocs) + 64;      // stream elements over all documents
    if (!h->w_cls.reserve(cap * 2 + 512) || !h->w_tmp.reserve(cap * 4)) return false;      // + the sector buffers of the Unigram lane program read up to two sectors past a document
    if (want_off && (!h->w_srcoff.reserve(cap * 4) || !h->w_span.reserve(cap * 8))) return false;
    if (m.kind == KIND_UNIGRAM) {
        // one packed 4-byte record per stream element (bf_seg.h uni_rec; 8 bytes reserved); the sequential / flat variants keep 16-byte records
        const bool lane_form = uni_lane_ok(m);
        if (!h->w_s1.reserve(cap * (lane_form ? 4 : 16) + 256)) return false;
    } else {
        const size_t bm_words = (cap >> 5) + (size_t)ndocs + 4;
        if (use_bpe_wave(h, want_off)) {
            // the wave program + k_bpe_seg: six words per stream cell for a word of more than 64 arcs (unit_huge), the arc pool of k_bpe_seg -- not the
            // 96 + 9 bytes per cell of the lane kernels' arc lists and work arrays (round 6: 111 -> 30 bytes of workspace per stream element)
            if (!h->w_s1.reserve(cap * 24 + 256) || !h->w_big.reserve(h->bpe_pool_bytes)) return false;
        } else if (!h->w_s1.reserve((6 * cap + 32 * (size_t)ndocs + 64) * 16) || !h->w_s2.reserve(std::max(cap * 4, 2 * bm_words * 4) + ((size_t)ndocs + 16) * 4) ||
            !h->w_s3.reserve(cap * 4) || !h->w_s4.reserve(cap) || !h->w_big.reserve(h->bpe_pool_bytes)) return false;
    }
    if (use_bpe_wave(h, want_off) && !h->w_bwflags.reserve((size_t)(ndocs + 1) * 4)) return false;
    return h->w_perm.reserve((size_t)(ndocs + 1) * 4) && h->w_hist.reserve(2048 * 4) && h->w_narcs.reserve((size_t)(ndocs + 1) * 4);
}

// Enqueue the whole pipeline for a batch resident on the device.
int run_device(Handle *h, const char *d_text, const int64_t *d_doc_off, int64_t ndocs, int64_t total_bytes,
               int32_t *d_ids_out, int64_t ids_cap, int64_t *d_id_off, int max_ids, int unk, hipStream_t s,
               int32_t *d_starts = nullptr, int32_t *d_ends = nullptr, int words = 0)
{
    const bool want_off = d_starts && d_ends;                                  // fNeedOffsets (tokdll:1137,1381)
    if (h->m.kind == KIND_I2W) return BF_E_UNSUPPORTED;                        // an [i2w]-only model has no tokenizer
    if (words && (h->m.kind != KIND_WP || !want_off)) return BF_E_ARG;
    if (ndocs < 0 || total_bytes < 0 || !d_doc_off || !d_id_off || (ids_cap > 0 && !d_ids_out) || (total_bytes > 0 && !d_text)) return BF_E_ARG;
    if (max_ids < 0) max_ids = 0;
    Model &m = h->m;
    const int nblocks = scan_nblocks(ndocs);
    if (!reserve_ids_workspaces(h, ndocs, total_bytes, want_off, words)) return BF_E_DEVICE;
    int slot_mul = 0; const int32_t *first = nullptr; bool uni_cut_keys = false, bpe_home = false;
    unsigned long long *next_doc = h->w_misc.as<unsigned long long>();
    int *status = (int *)(h->w_misc.as<char>() + 16);
    Batch b{(const uint8_t *)d_text, d_doc_off, ndocs, total_bytes, status};
    if (!hip_ok(hipMemsetAsync(h->w_misc.p, 0, 64, s), "hipMemsetAsync")) return BF_E_DEVICE;
    h->small_status = -1;
    (void)hipEventRecord(h->ev[EV_BEGIN], s);
    h->last_flat = use_flat(h, want_off, words, ndocs, total_bytes);
    if (h->last_flat) {
        // w_misc: [192] work counter of the ranges, [200] "the batch is not fit for the flat program", [204] documents handed back
        char *misc = h->w_misc.as<char>();
        if (!hip_ok(hipMemsetAsync(misc + 192, 0, 32, s), "hipMemsetAsync") || !hip_ok(hipMemsetAsync(h->w_dstat.p, 0, (size_t)ndocs * 4, s), "hipMemsetAsync") ||
            !hip_ok(hipMemsetAsync(h->w_counts.p, 0, (size_t)ndocs * 4, s), "hipMemsetAsync")) return BF_E_DEVICE;
        int *unsafe = (int *)(misc + 200); unsigned int *list_n = (unsigned int *)(misc + 204);
        const int nranges = wp_flat_ranges(ndocs, total_bytes);
        launch_wp_pre(d_doc_off, ndocs, total_bytes, nranges, h->w_ranges.as<int64_t>(), unsafe, s);
        (void)hipEventRecord(h->ev[EV_PREP], s);
        WpWaveCold cold;
        cold.cpmap = DevCpMap{h->t_cp_l1.as<uint16_t>(), h->t_cp_pages.as<uint32_t>()};
        cold.kind = h->t_kind.as<uint8_t>(); cold.nclasses = m.wbd.nclasses; cold.status = status; cold.no_fast = 0;
        cold.stats = h->lex_stats ? (unsigned long long *)(misc + 64) : nullptr;
        WfParams fp;
        fp.T = h->t_wbd.as<uint64_t>(); fp.W = h->t_flat.as<uint64_t>(); fp.wbits = m.flat_bits; fp.m0 = m.flat_m0; fp.m1 = m.flat_m1; fp.m2 = m.flat_m2;
        fp.ini = m.flat_ini; fp.ini_l = m.flat_ini_l; fp.max_token_length = m.max_token_length; fp.unk = unk;
        fp.text = b.text; fp.doc_off = b.doc_off; fp.ndocs = ndocs; fp.total_bytes = total_bytes;
        fp.range_doc = h->w_ranges.as<int64_t>(); fp.nranges = nranges; fp.next_range = (unsigned long long *)(misc + 192); fp.unsafe = unsafe;
        fp.ent = h->w_ent.as<uint32_t>(); fp.home = h->w_home.as<int32_t>(); fp.ent_off = h->w_entoff.as<int64_t>(); fp.ent_cnt = h->w_entcnt.as<int32_t>();
        fp.dstat = h->w_dstat.as<int32_t>(); fp.cold = cold; fp.espan = want_off ? h->w_espan.as<uint32_t>() : nullptr;
        fp.wrec = h->w_wrec.as<uint32_t>(); fp.wrec_cnt = (int32_t *)(h->w_wrec.as<char>() + (size_t)((total_bytes >> WF_REC_SHIFT) + 64) * 16);
        if (!hip_ok(hipMemsetAsync(fp.wrec_cnt, 0, (size_t)nranges * 8, s), "hipMemsetAsync")) return BF_E_DEVICE;      // (a range without documents writes nothing)
#ifdef BF_EXPERIMENTS
        fp.dbg = (h->variant >> 20) & 0xf;
#endif
        (void)hipEventRecord(h->ev[EV_DOM0], s);
        launch_wp_flat(fp, h->variant, s);
        (void)hipEventRecord(h->ev[EV_DOM1], s);
        // the words the table did not answer: walked by a kernel of their own
        WfUnitParams up;
        up.T = fp.T; up.ini = fp.ini; up.ini_l = fp.ini_l; up.max_token_length = fp.max_token_length; up.text = b.text; up.total_bytes = total_bytes;
        up.wrec = fp.wrec; up.wrec_cnt = fp.wrec_cnt; up.range_doc = fp.range_doc; up.doc_off = b.doc_off; up.nranges = nranges; up.ent = fp.ent; up.home = fp.home; up.extra = h->w_counts.as<int32_t>(); up.espan = fp.espan; up.hspan = want_off ? h->w_hspan.as<uint32_t>() : nullptr; up.cpmap = cold.cpmap; up.kind = cold.kind; up.nclasses = cold.nclasses; up.stats = cold.stats;
        launch_wp_units(up, h->variant, s);
        (void)hipEventRecord(h->ev[EV_TOK], s);
        // the documents it hands back: the wave program, one document at a time
        launch_wp_hardlist(fp.dstat, unsafe, ndocs, h->w_list.as<int32_t>(), list_n, s);
        WpWaveParams wp;
        wp.T = fp.T; wp.acts = h->t_acts.as<int32_t>(); wp.acts_n = (int)m.acts_pool.size();
        wp.initial = m.wbd.initial_base; wp.loop_info = m.loop_info; wp.solo_info = m.wave_solo_info; wp.max_token_length = m.max_token_length;
        wp.text = b.text; wp.doc_off = b.doc_off; wp.ndocs = ndocs; wp.total_bytes = total_bytes;
        wp.ids_tmp = h->w_tmp.as<int32_t>(); wp.counts = h->w_counts.as<int32_t>(); wp.max_ids = max_ids; wp.unk = unk; wp.span_tmp = want_off ? h->w_span.as<int32_t>() : nullptr;
        wp.next_doc = next_doc; wp.doc_list = h->w_list.as<int32_t>(); wp.list_n = list_n;
        wp.cold = cold; wp.cold.stats = nullptr;
        launch_wp_wave(wp, h->variant & ~0x3f000000, s);
        WfMergeParams mp;
        mp.doc_off = b.doc_off; mp.ndocs = ndocs; mp.ent = fp.ent; mp.home = fp.home; mp.ent_off = fp.ent_off; mp.ent_cnt = fp.ent_cnt; mp.dstat = fp.dstat; mp.unsafe = unsafe;
        mp.ids_tmp = wp.ids_tmp; mp.counts = wp.counts; mp.id_off = d_id_off; mp.ids_out = d_ids_out; mp.ids_cap = ids_cap; mp.status = status; mp.max_ids = max_ids; mp.unk = unk;
        mp.espan = fp.espan; mp.hspan = up.hspan; mp.starts_out = d_starts; mp.ends_out = d_ends; mp.counts_hard = want_off ? h->w_chard.as<int32_t>() : nullptr;
        launch_wp_count(mp, s);
        ScanParams sp{h->w_counts.as<int32_t>(), ndocs, d_id_off, h->w_bsums.as<int64_t>(), nblocks};
        launch_scan(sp, s);
        (void)hipEventRecord(h->ev[EV_SCAN], s);
        launch_wp_merge(mp, s);
        if (want_off) {
            // the documents the wave program tokenised: their ids, and the byte offsets from the characters it staged (usually there are none)
            CompactParams cp{b, wp.ids_tmp, mp.counts_hard, d_id_off, d_ids_out, ids_cap, status, 0, nullptr, wp.span_tmp, nullptr, d_starts, d_ends, list_n};
            launch_compact(cp, s);
        }
        (void)hipEventRecord(h->ev[EV_COMPACT], s);
        h->ev_valid = true;
        if (!hip_ok(hipGetLastError(), "kernel launch")) return BF_E_DEVICE;
        return 0;
    }
    if (use_wave(h, want_off, words)) {
        (void)hipEventRecord(h->ev[EV_PREP], s);                       // decoding is part of the wave program
        WpWaveParams wp;
        wp.T = h->t_wbd.as<uint64_t>(); wp.acts = h->t_acts.as<int32_t>(); wp.acts_n = (int)m.acts_pool.size();
        wp.initial = m.wbd.initial_base; wp.loop_info = m.loop_info; wp.solo_info = m.wave_solo_info; wp.max_token_length = m.max_token_length;
        wp.text = b.text; wp.doc_off = b.doc_off; wp.ndocs = b.ndocs; wp.total_bytes = b.total_bytes;
        wp.ids_tmp = h->w_tmp.as<int32_t>(); wp.counts = h->w_counts.as<int32_t>(); wp.max_ids = max_ids; wp.unk = unk;
        wp.span_tmp = want_off ? h->w_span.as<int32_t>() : nullptr;
        wp.next_doc = next_doc;
        wp.cold.cpmap = DevCpMap{h->t_cp_l1.as<uint16_t>(), h->t_cp_pages.as<uint32_t>()};
        wp.cold.kind = h->t_kind.as<uint8_t>(); wp.cold.nclasses = m.wbd.nclasses; wp.cold.status = status; wp.cold.no_fast = 0;
        wp.cold.stats = h->lex_stats ? (unsigned long long *)(h->w_misc.as<char>() + 64) : nullptr;
        (void)hipEventRecord(h->ev[EV_DOM0], s);
        if (ndocs > 0) launch_wp_wave(wp, h->variant, s);
        (void)hipEventRecord(h->ev[EV_DOM1], s);
        (void)hipEventRecord(h->ev[EV_TOK], s);
    } else if (m.kind == KIND_WP) {
        WpPrepParams pp{b, DevCpMap{h->t_cp_l1.as<uint16_t>(), h->t_cp_pages.as<uint32_t>()}, h->t_multi.as<uint16_t>(),
                        m.wbd_charmap_multi ? 1 : 0, h->w_cls.as<uint16_t>(), want_off ? h->w_srcoff.as<int32_t>() : nullptr, h->w_nchars.as<int32_t>()};
        if (words) { pp.cpmap = DevCpMap{h->t_wcp_l1.as<uint16_t>(), h->t_wcp_pages.as<uint32_t>()}; pp.has_multi = 0; }   // no charmap (tokdll:476-499)
        // long documents are decoded by sixteen waves each (k_prep_wp_long); w_misc + 48: their number, zeroed with the status word above
        pp.long_cap = total_bytes / 2048 + 1;
        pp.long_list = h->w_preplong.reserve((size_t)pp.long_cap * 8) ? h->w_preplong.as<int64_t>() : nullptr;
        pp.long_count = (unsigned int *)(h->w_misc.as<char>() + 48);
        if (ndocs > 0) launch_prep_wp(pp, total_bytes, h->w_flags.as<unsigned long long>(), s);
        (void)hipEventRecord(h->ev[EV_PREP], s);
        WpLexParams lp;
        lp.L.T = h->t_wbd.as<uint64_t>(); lp.L.acts = h->t_acts.as<int32_t>();
        lp.L.initial = m.wbd.initial_base; lp.L.initial_l = m.initial_l; lp.L.cls_any = m.cls_any; lp.L.cls_l = m.cls_l; lp.L.cls_r = m.cls_r;
        lp.L.max_depth = m.max_depth; lp.L.max_token_length = m.max_token_length; lp.L.max_frames = m.lex_frames;
        lp.L.loop_state = m.loop_base;
        lp.L.loop_info = m.loop_info; lp.L.loop_final = m.loop_final ? 1 : 0;
        lp.L.fn_no_ra = m.fn_no_ra ? 1 : 0;
        lp.L.two_level = m.two_level ? 1 : 0;
        lp.b = b; lp.cls = h->w_cls.as<uint16_t>(); lp.nchars = h->w_nchars.as<int32_t>();
        lp.ids_tmp = h->w_tmp.as<int32_t>(); lp.counts = h->w_counts.as<int32_t>(); lp.span_tmp = want_off ? h->w_span.as<int32_t>() : nullptr;
        lp.max_ids = max_ids; lp.unk = unk; lp.next_doc = next_doc; lp.status = status; lp.ev_thresh = 0; lp.fetch_thresh = 0; lp.acts_n = (int)m.acts_pool.size(); lp.words = words;
        lp.table_n = (int)(m.wbd_t2.size() > (size_t)LX_T_CLS_MASK + 1 ? m.wbd_t2.size() - ((size_t)LX_T_CLS_MASK + 1) : 0);
        lp.stats = h->lex_stats ? (unsigned long long *)(h->w_misc.as<char>() + 64) : nullptr;
        const LongCaps lc = long_caps(h, ndocs, total_bytes, words);
        lp.lg = LexLongParams{lc.thresh, (words && (h->variant & 0x20000000)) ? 3 : 0, lc.docs, lc.chunks, (unsigned long long *)(h->w_misc.as<char>() + 40) /* zeroed with the status word above */,
                              (LexLongDoc *)(h->w_long.as<char>() + lc.list_off), (int32_t *)(h->w_long.as<char>() + lc.spec_off), (int32_t *)(h->w_long.as<char>() + lc.jump_off),
                              (int32_t *)(h->w_long.as<char>() + lc.tok2_off), lc.big_cells, (int32_t *)(h->w_long.as<char>() + lc.jump2_off),
                              (int32_t *)(h->w_long.as<char>() + lc.entry2_off), (int32_t *)(h->w_long.as<char>() + lc.entry_off)};
        (void)hipEventRecord(h->ev[EV_DOM0], s);
        if (ndocs > 0) {
            launch_lex_long_list(lp, s);
            launch_lex_wp(lp, words ? (h->variant & ~0x7800F000) : h->variant, s);
            launch_lex_long(lp, s);
        }
        (void)hipEventRecord(h->ev[EV_DOM1], s);
        (void)hipEventRecord(h->ev[EV_TOK], s);
    } else {
        const int mul = m.dict_has_charmap ? 2 : 1;
        slot_mul = mul;
        const size_t cap = (size_t)mul * (size_t)(total_bytes + ndocs) + 64;      // elements over all documents
        SpPrepParams pp;
        pp.b = b; pp.cpmap = DevCpMap{h->t_cp_l1.as<uint16_t>(), h->t_cp_pages.as<uint32_t>()}; pp.multi_pool = h->t_multi.as<uint16_t>();
        pp.has_multi = m.sp_has_multi ? 1 : 0; pp.use_bytes = m.use_bytes ? 1 : 0; pp.has_charmap = m.dict_has_charmap ? 1 : 0;
        pp.delim_code = m.sp_delim_code;
        pp.prefix_n = m.no_dummy_prefix ? 0 : (int)m.sp_prefix.size();
        for (int k = 0; k < 10; ++k) pp.prefix[k] = k < pp.prefix_n ? m.sp_prefix[(size_t)k] : 0;
        pp.old_form = (h->variant & 0x80) ? 1 : 0;
        pp.waves = (h->variant >> 24) & 0xf;
        pp.slot_mul = mul; pp.stream = h->w_cls.as<uint16_t>(); pp.lens = h->w_nchars.as<int32_t>(); pp.src_off = want_off ? h->w_srcoff.as<int32_t>() : nullptr;
        if (ndocs > 0) launch_prep_sp(pp, s);
        (void)hipEventRecord(h->ev[EV_PREP], s);
        SpSegParams sg;
        sg.S.T = h->t_dict.as<uint64_t>(); sg.S.info = h->t_seginfo.as<SegInfo>(); sg.S.initial = m.dict.initial_base;
        sg.S.cls_delim = m.sp_delim_code; sg.S.kind = m.kind; sg.S.id_offset = m.id_offset; sg.S.score = m.kind == KIND_UNIGRAM ? h->t_segscore.as<uint32_t>() : nullptr; sg.S.leaf_lo = m.dict.leaf_lo; sg.S.leaf_n = m.dict.leaf_n;
        sg.b = b; sg.stream = h->w_cls.as<uint16_t>(); sg.lens = h->w_nchars.as<int32_t>(); sg.slot_mul = mul;
        sg.ids_tmp = h->w_tmp.as<int32_t>(); sg.counts = h->w_counts.as<int32_t>(); sg.span_tmp = want_off ? h->w_span.as<int32_t>() : nullptr; sg.max_ids = max_ids; sg.unk = unk; sg.status = status;
        sg.best = nullptr; sg.arcs = nullptr; sg.tos = nullptr; sg.idsv = nullptr; sg.inter = nullptr; sg.bm_words = 0; sg.fb_list = nullptr; sg.fb_count = nullptr;
        sg.big_pool = nullptr; sg.big_cap = 0; sg.big_used = (unsigned long long *)(h->w_misc.as<char>() + 32);     // zeroed with the status word above
        sg.big_need = (unsigned long long *)(h->w_misc.as<char>() + 224);      // (not among the words a launch clears: the chunks of a pipelined host call add up in it, run_host clears it)
        sg.bpe_prio = nullptr; sg.bpe_place_id = nullptr; sg.bpe_unk_prio = 0; sg.bpe_prio_bits = m.bpe_prio_bits; sg.seg_stats = nullptr;
        if (m.kind == KIND_UNIGRAM) sg.best = h->w_s1.as<SegBest>();
        else {
            const size_t bm_words = (cap >> 5) + (size_t)ndocs + 4;         // per bitmap: capacity + 1 bits per document (k_bpe_apply_flat)
            sg.bm_words = (int64_t)bm_words;
            sg.arcs = h->w_s1.as<SegArc>(); sg.tos = h->w_s2.as<int32_t>(); sg.idsv = h->w_s3.as<int32_t>(); sg.inter = h->w_s4.as<uint8_t>();
            sg.big_pool = h->w_big.as<uint8_t>(); sg.big_cap = h->w_big.cap;
            if (m.kind == KIND_BPE_MERGES) { sg.bpe_prio = h->t_bpe_prio.as<uint32_t>(); sg.bpe_place_id = h->t_bpe_place.as<int32_t>(); }
            sg.bpe_unk_prio = bpe_unk_prio(m, unk);
        }
        sg.narcs = h->w_narcs.as<int32_t>(); sg.next_doc = next_doc; sg.trie_depth = m.trie_max_depth; sg.variant = h->variant & 0xff; sg.tune = (h->variant >> 8) & 0xff; sg.tune2 = (h->variant >> 16) & 0xff;
        sg.lane_ok = uni_lane_ok(m) ? 1 : 0;
        // ids only: the cut form (bf_seg.h UniCut; BfSetVariant 6: the forward / backward kernels of round 4, which the offsets API still takes)
        sg.uni_cut = (m.kind == KIND_UNIGRAM && sg.lane_ok && !want_off && (h->variant & 0xff) != 6) ? 1 : 0;
        if (m.kind == KIND_UNIGRAM && sg.lane_ok && !sg.uni_cut) first = h->w_narcs.as<int32_t>();
        uni_cut_keys = sg.uni_cut != 0;
        h->last_uni_cut = uni_cut_keys;
        sg.perm = h->w_perm.as<int32_t>(); sg.hist = h->w_hist.as<unsigned int>();
        const bool bwave = use_bpe_wave(h, want_off);
        if (bwave && ndocs > 0) {
            // words that are one vocabulary entry (most are) and short words that are not: the wave program; the documents it hands back
            // (flags: a word of more than 62 elements, a symbol outside the alphabet, ...): one wave per document (bf_bpe_seg_body.h)
            BpeWaveParams bw;
            bw.T = sg.S.T; bw.info = sg.S.info; bw.initial = sg.S.initial; bw.cls_delim = sg.S.cls_delim; bw.id_offset = sg.S.id_offset;
            bw.prio = sg.bpe_prio; bw.place_id = sg.bpe_place_id;
            if (!m.bpe_tab.empty() && (h->variant & 0x100000) == 0) { bw.W = h->t_bpetab.as<uint64_t>(); bw.wbits = m.bpe_tab_bits; bw.m0 = m.bpe_tab_m0; bw.m1 = m.bpe_tab_m1; bw.m2 = m.bpe_tab_m2; }      // (BfSetVariant bit 20: A/B runs without the word table)
            bw.stream = sg.stream; bw.lens = sg.lens; bw.doc_off = b.doc_off; bw.slot_mul = mul; bw.ndocs = ndocs;
            bw.ids_tmp = sg.ids_tmp; bw.counts = sg.counts; bw.flags = h->w_bwflags.as<int32_t>(); bw.max_ids = max_ids; bw.next_doc = next_doc; bw.status = status; bw.scratch = (uint32_t *)sg.arcs; bw.stats = h->lex_stats ? (unsigned long long *)(h->w_misc.as<char>() + 64) : nullptr;
            (void)hipEventRecord(h->ev[EV_DOM0], s);
            launch_bpe_wave(bw, (h->variant >> 8) & 0xf, s);
            (void)hipEventRecord(h->ev[EV_DOM1], s);
            launch_bpe_seg_flags(sg, bw.flags, h->w_perm.as<int32_t>(), h->w_hist.as<unsigned int>(), s);
            bpe_home = bpe_wave_home((h->variant >> 8) & 0xf);
        } else {
            // (the Unigram lane program records the two events around its forward kernel itself: the sort of the documents comes before it)
            sg.ev_dom0 = h->ev[EV_DOM0]; sg.ev_dom1 = h->ev[EV_DOM1];
            if (m.kind != KIND_UNIGRAM || !sg.lane_ok || ndocs <= 0) (void)hipEventRecord(h->ev[EV_DOM0], s);
            if (ndocs > 0) launch_seg_sp(sg, s);
            if (m.kind != KIND_UNIGRAM || !sg.lane_ok || ndocs <= 0) (void)hipEventRecord(h->ev[EV_DOM1], s);
        }
        (void)hipEventRecord(h->ev[EV_TOK], s);
    }
    ScanParams sp{h->w_counts.as<int32_t>(), ndocs, d_id_off, h->w_bsums.as<int64_t>(), nblocks};
    launch_scan(sp, s);
    (void)hipEventRecord(h->ev[EV_SCAN], s);
    CompactParams cp{b, h->w_tmp.as<int32_t>(), h->w_counts.as<int32_t>(), d_id_off, d_ids_out, ids_cap, status, slot_mul, first,
                     want_off ? h->w_span.as<int32_t>() : nullptr, want_off && !use_wave(h, want_off, words) ? h->w_srcoff.as<int32_t>() : nullptr, want_off ? d_starts : nullptr, want_off ? d_ends : nullptr};
    if (uni_cut_keys) {
        // the cut form left tokens, not ids: their ids go straight to the caller's array (k_uni_ids is the compaction of this path)
        UniIdsParams up{b, h->t_dict.as<uint64_t>(), m.dict.initial_base, h->t_segid.as<int32_t>(), h->w_cls.as<uint16_t>(), h->w_nchars.as<int32_t>(), slot_mul,
                        h->w_tmp.as<int32_t>(), h->w_counts.as<int32_t>(), d_id_off, d_ids_out, ids_cap, unk, m.id_offset, status};
        if (ndocs > 0) launch_uni_ids(up, s);
    } else if (bpe_home) {
        BpeHomeParams hp{b, h->w_tmp.as<int32_t>(), h->w_nchars.as<int32_t>(), h->w_bwflags.as<int32_t>(), slot_mul, h->w_counts.as<int32_t>(), d_id_off, d_ids_out, ids_cap, max_ids, status};
        if (ndocs > 0) launch_bpe_home_gather(hp, s);
    } else if (ndocs > 0) launch_compact(cp, s);
    (void)hipEventRecord(h->ev[EV_COMPACT], s);
    h->ev_valid = true;
    if (!hip_ok(hipGetLastError(), "kernel launch")) return BF_E_DEVICE;
    return 0;
}

// memcpy with a few threads: one core moves ~10 GB/s, PCIe wants ~50
void par_memcpy(void *dst, const void *src, size_t n)
{
    if (n == 0) return;
    unsigned hw = std::thread::hardware_concurrency();
    size_t nt = hw >= 16 ? 8 : hw >= 8 ? 4 : hw >= 4 ? 2 : 1;
    if (n < (size_t)(8u << 20)) nt = 1;
    if (nt == 1) { memcpy(dst, src, n); return; }
    const size_t part = ((n + nt - 1) / nt + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    for (size_t t = 1; t < nt; ++t) {
        const size_t a = t * part; if (a >= n) break;
        const size_t len = n - a < part ? n - a : part;
        th.emplace_back([=] { memcpy((char *)dst + a, (const char *)src + a, len); });
    }
    memcpy(dst, src, part < n ? part : n);
    for (auto &x : th) x.join();
}

constexpr int64_t BF_RETRY_POOL = INT64_MIN + 1;            // internal: a BPE document did not fit the arc pool (BF_STATUS_POOL); the need is in w_misc
constexpr int64_t HOST_PIPE_UNAVAILABLE = INT64_MIN;      // run_host_chunked could not set up its staging: the caller takes the one-copy path

// The chunked form of run_host (ids only).  Same results as one TextToIdsBatchDevice call over the whole batch: documents are
// independent, chunk boundaries are document boundaries.  Returns the number of ids or a BF_E_* code; id_off_out (if given) is
// complete even when the ids do not fit ids_cap (BF_E_CAPACITY), like the unchunked path.
int64_t run_host_chunked(Handle *h, const char *text, const int64_t *doc_off, int64_t ndocs, int32_t *ids_out, int64_t ids_cap,
                         int64_t *id_off_out, int max_ids, int unk)
{
    HostPipe &P = h->pipe;
    constexpr int NS = HostPipe::NS;
    if (!P.init()) return HOST_PIPE_UNAVAILABLE;
    hipStream_t s = h->stream;
    // ---- chunks: whole documents, about host_chunk_bytes each
    struct Chunk { int64_t d0, d1; };
    std::vector<Chunk> chunks;
    int64_t max_bytes = 0, max_docs = 0;
    // every chunk costs about a millisecond of stream hand-overs on top of its copies (measured on MI355X: 16 / 32 / 64 / 128 MiB chunks
    // -> 22 / 39 / 60 / 66 M docs/s on the 640 MB sample of the default workload), so chunks are large -- but at least five per batch
    const int64_t total_bytes = doc_off[ndocs] - doc_off[0];
    const int64_t chunk_bytes = std::min(h->host_chunk_bytes, std::max(h->host_chunk_bytes / 2, total_bytes / 5));
    for (int64_t d = 0; d < ndocs;) {
        const int64_t lim = doc_off[d] + chunk_bytes;
        int64_t e = (int64_t)(std::upper_bound(doc_off + d + 1, doc_off + ndocs + 1, lim) - doc_off) - 1;      // last boundary <= lim
        if (e <= d) e = d + 1;                                                                                  // one document larger than a chunk
        if (e - d > (1ll << 30)) e = d + (1ll << 30);
        chunks.push_back({d, e});
        max_bytes = std::max(max_bytes, doc_off[e] - doc_off[d]); max_docs = std::max(max_docs, e - d);
        d = e;
    }
    auto worst_ids = [&](int64_t bytes, int64_t nd) {
        int64_t w = h->m.kind == KIND_WP ? bytes : (int64_t)(h->m.dict_has_charmap ? 2 : 1) * (bytes + nd);
        if (max_ids >= 0 && nd * (int64_t)max_ids < w) w = nd * (int64_t)(max_ids < 0 ? 0 : max_ids);
        return w;
    };
    int64_t max_worst = 0;
    for (const Chunk &c : chunks) max_worst = std::max(max_worst, worst_ids(doc_off[c.d1] - doc_off[c.d0], c.d1 - c.d0));
    for (int i = 0; i < NS; ++i)
        if (!P.pin_text[i].reserve((size_t)max_bytes + 16) || !P.pin_off[i].reserve((size_t)(max_docs + 1) * 8) || !P.pin_idoff[i].reserve((size_t)(max_docs + 1) * 8) ||
            !P.dev_text[i].reserve((size_t)max_bytes + 16) || !P.dev_off[i].reserve((size_t)(max_docs + 1) * 8) || !P.dev_idoff[i].reserve((size_t)(max_docs + 1) * 8) ||
            !P.dev_ids[i].reserve((size_t)(max_worst + 1) * 4)) return HOST_PIPE_UNAVAILABLE;   // (page-locked memory is a limited resource)
    if (!reserve_ids_workspaces(h, max_docs, max_bytes, false)) return BF_E_DEVICE;      // no allocation (= device synchronisation) inside the pipeline
    const int K = (int)chunks.size();
    std::vector<int64_t> nids_of((size_t)K, 0);
    const bool trace = getenv("BF_TRACE_HOST") != nullptr;          // stderr: where the wall time of this call went
    double t_in = 0, t_out = 0, t_wait_cmp = 0, t_meta = 0, t_wait_d2h = 0, t_wait_worker = 0, t_enq = 0;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    // ---- the way out runs on a second host thread, chunk after chunk: wait for the ids of chunk j in pinned memory, copy them and the
    //      rebased offsets into the caller's arrays.  `done` = chunks it has finished (their pinned slots are free again).
    std::mutex wmu; std::condition_variable wcv;
    int issued = 0, done = 0; bool quit = false;                   // guarded by wmu
    int64_t base_ids = 0; bool overflow = false; int64_t worker_err = 0;
    std::thread worker([&] {
        (void)hipSetDevice(h->device);
        for (int j = 0;; ++j) {
            { std::unique_lock<std::mutex> lk(wmu); wcv.wait(lk, [&] { return issued > j || quit; }); if (issued <= j) return; }
            const int sl = j % NS; const Chunk &c = chunks[(size_t)j]; const int64_t nd = c.d1 - c.d0, nids = nids_of[(size_t)j];
            const double t0 = now();
            if (!worker_err && !hip_ok(hipEventSynchronize(P.ev_d2h[sl]), "hipEventSynchronize")) worker_err = BF_E_DEVICE;
            const double t1 = now(); t_wait_d2h += t1 - t0;
            if (!worker_err) {
                if (base_ids + nids > ids_cap) overflow = true;
                if (nids > 0 && !overflow) {
                    if (!ids_out) worker_err = BF_E_ARG;
                    else par_memcpy(ids_out + base_ids, P.pin_ids[sl].p, (size_t)nids * 4);
                }
                if (id_off_out) { const int64_t *o = P.pin_idoff[sl].as<int64_t>(); for (int64_t i = 0; i < nd; ++i) id_off_out[c.d0 + i] = base_ids + o[i]; }
                base_ids += nids;
            }
            t_out += now() - t1;
            { std::lock_guard<std::mutex> lk(wmu); done = j + 1; }
            wcv.notify_all();
        }
    });
    auto stop_worker = [&] { { std::lock_guard<std::mutex> lk(wmu); quit = true; } wcv.notify_all(); if (worker.joinable()) worker.join(); };
    auto fail = [&](int64_t rc) { stop_worker(); (void)hipStreamSynchronize(P.s_in); (void)hipStreamSynchronize(s); (void)hipStreamSynchronize(P.s_meta); (void)hipStreamSynchronize(P.s_out); return rc; };
    int64_t rc_err = 0;
    // ids of chunk j: device -> pinned, started as soon as its tokenisation is done; the worker takes it from there
    auto start_out = [&](int j) -> bool {
        const int sl = j % NS; const Chunk &c = chunks[(size_t)j]; const int64_t nd = c.d1 - c.d0;
        { const double t0 = now(); std::unique_lock<std::mutex> lk(wmu); wcv.wait(lk, [&] { return done >= j - NS + 1; }); t_wait_worker += now() - t0; }   // pin_idoff[sl] / pin_ids[sl]: chunk j-NS is out
        const double t0 = now();
        if (!hip_ok(hipEventSynchronize(P.ev_cmp[sl]), "hipEventSynchronize")) return false;
        const double t0b = now(); t_wait_cmp += t0b - t0;
        if (!hip_ok(hipMemcpyAsync(P.pin_idoff[sl].p, P.dev_idoff[sl].p, (size_t)(nd + 1) * 8, hipMemcpyDeviceToHost, P.s_meta), "D2H offsets") ||
            !hip_ok(hipStreamSynchronize(P.s_meta), "hipStreamSynchronize")) return false;
        t_meta += now() - t0b;
        if (P.pin_status.as<int>()[sl] & (2 | BF_STATUS_INTERNAL)) { rc_err = BF_E_INTERNAL; return false; }
        if (P.pin_status.as<int>()[sl] & BF_STATUS_POOL) { rc_err = BF_RETRY_POOL; return false; }     // run_host grows the pool and runs the batch again
        const int64_t nids = P.pin_idoff[sl].as<int64_t>()[nd];
        nids_of[(size_t)j] = nids;
        if (nids > 0 && ids_out && nids <= ids_cap) {               // (a chunk larger than the whole capacity cannot be delivered anyway)
            if (!P.pin_ids[sl].reserve((size_t)nids * 4)) return false;
            if (!hip_ok(hipMemcpyAsync(P.pin_ids[sl].p, P.dev_ids[sl].p, (size_t)nids * 4, hipMemcpyDeviceToHost, P.s_out), "D2H ids")) return false;
        }
        if (!hip_ok(hipEventRecord(P.ev_d2h[sl], P.s_out), "hipEventRecord")) return false;
        { std::lock_guard<std::mutex> lk(wmu); issued = j + 1; }
        wcv.notify_all();
        return true;
    };
    // the way out of a chunk is STARTED by a third thread (wait for its tokenisation, fetch its id offsets, start the copy of its ids):
    // on the caller's thread those waits were a quarter of the call (0.6 - 1 ms per chunk; measured on the 10 M-document corpus)
    int enq = 0; bool meta_fail = false;                            // guarded by wmu: chunks enqueued; the third thread failed
    std::thread meta([&] {
        (void)hipSetDevice(h->device);
        for (int j = 0; j < K; ++j) {
            { std::unique_lock<std::mutex> lk(wmu); wcv.wait(lk, [&] { return enq > j || quit; }); if (enq <= j) return; }
            if (!start_out(j)) { { std::lock_guard<std::mutex> lk(wmu); meta_fail = true; } wcv.notify_all(); return; }
        }
    });
    auto stop_meta = [&] { { std::lock_guard<std::mutex> lk(wmu); quit = true; } wcv.notify_all(); if (meta.joinable()) meta.join(); };
    auto fail2 = [&](int64_t rc) { stop_meta(); return fail(rc); };
    for (int k = 0; k < K; ++k) {
        {
            const int sl = k % NS; const Chunk &c = chunks[(size_t)k];
            const int64_t nd = c.d1 - c.d0, b0 = doc_off[c.d0], bytes = doc_off[c.d1] - b0;
            if (k >= NS && !hip_ok(hipEventSynchronize(P.ev_h2d[sl]), "hipEventSynchronize")) return fail2(BF_E_DEVICE);             // pin_text[sl] / pin_off[sl]: chunk k-NS is on the device
            const double t0 = now();
            par_memcpy(P.pin_text[sl].p, text + b0, (size_t)bytes);
            { int64_t *o = P.pin_off[sl].as<int64_t>(); for (int64_t i = 0; i <= nd; ++i) o[i] = doc_off[c.d0 + i] - b0; }
            const double t0e = now(); t_in += t0e - t0;
            if (k >= NS && !hip_ok(hipStreamWaitEvent(P.s_in, P.ev_cmp[sl], 0), "hipStreamWaitEvent")) return fail2(BF_E_DEVICE);    // chunk k-NS no longer reads dev_text[sl]
            if ((bytes > 0 && !hip_ok(hipMemcpyAsync(P.dev_text[sl].p, P.pin_text[sl].p, (size_t)bytes, hipMemcpyHostToDevice, P.s_in), "H2D text")) ||
                !hip_ok(hipMemcpyAsync(P.dev_off[sl].p, P.pin_off[sl].p, (size_t)(nd + 1) * 8, hipMemcpyHostToDevice, P.s_in), "H2D offsets") ||
                !hip_ok(hipEventRecord(P.ev_h2d[sl], P.s_in), "hipEventRecord") || !hip_ok(hipStreamWaitEvent(s, P.ev_h2d[sl], 0), "hipStreamWaitEvent")) return fail2(BF_E_DEVICE);
            if (k >= NS) {      // the copy-out of chunk k-NS must have been started (its event recorded) before this chunk may wait for it
                const double tw = now(); std::unique_lock<std::mutex> lk(wmu); wcv.wait(lk, [&] { return issued >= k - NS + 1 || meta_fail; }); t_wait_worker += now() - tw;
                if (meta_fail) { lk.unlock(); return fail2(rc_err ? rc_err : BF_E_DEVICE); }
            }
            if (k >= NS && !hip_ok(hipStreamWaitEvent(s, P.ev_d2h[sl], 0), "hipStreamWaitEvent")) return fail2(BF_E_DEVICE);        // the ids of chunk k-NS have left dev_ids[sl]
            const int rc = run_device(h, P.dev_text[sl].as<char>(), P.dev_off[sl].as<int64_t>(), nd, bytes, P.dev_ids[sl].as<int32_t>(), worst_ids(bytes, nd),
                                      P.dev_idoff[sl].as<int64_t>(), max_ids, unk, s);
            if (rc != 0) return fail2(rc);
            if (!hip_ok(hipMemcpyAsync(P.pin_status.as<int>() + sl, h->w_misc.as<char>() + 16, 4, hipMemcpyDeviceToHost, s), "D2H status") ||
                !hip_ok(hipEventRecord(P.ev_cmp[sl], s), "hipEventRecord")) return fail2(BF_E_DEVICE);
            t_enq += now() - t0e;
            { std::lock_guard<std::mutex> lk(wmu); enq = k + 1; }
            wcv.notify_all();
        }
    }
    { const double t0 = now(); std::unique_lock<std::mutex> lk(wmu); wcv.wait(lk, [&] { return done >= K || meta_fail; }); t_wait_worker += now() - t0; }
    if (meta_fail) return fail2(rc_err ? rc_err : BF_E_DEVICE);
    if (meta.joinable()) meta.join();
    stop_worker();
    if (!hip_ok(hipStreamSynchronize(s), "hipStreamSynchronize")) return BF_E_DEVICE;
    if (worker_err) return worker_err;
    if (id_off_out) id_off_out[ndocs] = base_ids;
    if (trace) fprintf(stderr, "[blingfire_amd] chunked host batch: %d chunks, %.1f ms; caller's thread: copy-in %.1f, enqueueing %.1f, waiting for tokenisation %.1f, offsets D2H %.1f, waiting for the way out %.1f; "
                       "second thread: waiting for D2H %.1f, copy-out %.1f\n", K, now() - t_begin, t_in, t_enq, t_wait_cmp, t_meta, t_wait_worker, t_wait_d2h, t_out);
    return overflow ? BF_E_CAPACITY : base_ids;
}

// ---- small batches through mapped host memory (see Handle::m_small)
constexpr int64_t SMALL_MAX_DOCS = 256, SMALL_MAX_BYTES = 64 * 1024;
struct SmallLayout {
    static constexpr size_t ctrl = 0, off = 64, counts = off + (SMALL_MAX_DOCS + 1) * 8, text = counts + SMALL_MAX_DOCS * 4,
                            ids = (text + SMALL_MAX_BYTES + 63) & ~(size_t)63, total = ids + (SMALL_MAX_BYTES + 8 * SMALL_MAX_DOCS + 64) * 4;
};

bool small_ready(Handle *h)
{
    if (h->m_small.p) return true;
    void *q = nullptr;
    if (!hip_ok(hipHostMalloc(&q, SmallLayout::total, hipHostMallocMapped), "hipHostMalloc(mapped)")) return false;
    void *d = nullptr;
    if (!hip_ok(hipHostGetDevicePointer(&d, q, 0), "hipHostGetDevicePointer")) { (void)hipHostFree(q); return false; }
    h->m_small.p = q; h->m_small.cap = SmallLayout::total; h->m_small_dev = d;
    return true;
}

// caller holds h->mu and has made h->device current; the batch fits SmallLayout.  Returns the id count or BF_E_*.
int64_t run_host_mapped(Handle *h, const char *text, const int64_t *doc_off, int64_t ndocs, int32_t *ids_out, int64_t ids_cap, int64_t *id_off_out, int max_ids, int unk)
{
    const Model &m = h->m;
    char *hp = h->m_small.as<char>(), *dp = (char *)h->m_small_dev;
    const int64_t base = doc_off[0], total = doc_off[ndocs] - base;
    h->last_flat = false;                                          // (the wave program: what BfTokeniseKernel / BfStepKernels report)
    memset(hp + SmallLayout::ctrl, 0, 64);
    int64_t *off = (int64_t *)(hp + SmallLayout::off);
    for (int64_t i = 0; i <= ndocs; ++i) off[i] = doc_off[i] - base;
    if (total > 0) memcpy(hp + SmallLayout::text, text + base, (size_t)total);
    WpWaveParams wp;
    wp.T = h->t_wbd.as<uint64_t>(); wp.acts = h->t_acts.as<int32_t>(); wp.acts_n = (int)m.acts_pool.size();
    wp.initial = m.wbd.initial_base; wp.loop_info = m.loop_info; wp.solo_info = m.wave_solo_info; wp.max_token_length = m.max_token_length;
    wp.text = (const uint8_t *)(dp + SmallLayout::text); wp.doc_off = (const int64_t *)(dp + SmallLayout::off); wp.ndocs = ndocs; wp.total_bytes = total;
    wp.ids_tmp = (int32_t *)(dp + SmallLayout::ids); wp.counts = (int32_t *)(dp + SmallLayout::counts); wp.max_ids = max_ids < 0 ? 0 : max_ids; wp.unk = unk;
    wp.span_tmp = nullptr;
    wp.next_doc = nullptr;              // no work counter: an atomic on host memory costs every wave a trip over the bus
    wp.cold.cpmap = DevCpMap{h->t_cp_l1.as<uint16_t>(), h->t_cp_pages.as<uint32_t>()};
    wp.cold.kind = h->t_kind.as<uint8_t>(); wp.cold.nclasses = m.wbd.nclasses; wp.cold.status = (int *)(dp + SmallLayout::ctrl + 16); wp.cold.no_fast = 0; wp.cold.stats = nullptr;
    launch_wp_wave(wp, h->variant, h->stream);
    h->ev_valid = false;
    if (!hip_ok(hipGetLastError(), "kernel launch") || !hip_ok(hipStreamSynchronize(h->stream), "hipStreamSynchronize")) return BF_E_DEVICE;
    h->small_status = *(const int *)(hp + SmallLayout::ctrl + 16);
    if (h->small_status & (2 | BF_STATUS_INTERNAL)) return BF_E_INTERNAL;
    const int32_t *counts = (const int32_t *)(hp + SmallLayout::counts), *ids = (const int32_t *)(hp + SmallLayout::ids);
    // the offsets are complete whatever ids_cap is (like the other host paths); ids are copied only when all of them fit
    int64_t n = 0;
    for (int64_t d = 0; d < ndocs; ++d) { if (id_off_out) id_off_out[d] = n; n += counts[d] > 0 ? counts[d] : 0; }
    if (id_off_out) id_off_out[ndocs] = n;
    if (n > ids_cap) return BF_E_CAPACITY;
    if (n > 0 && !ids_out) return BF_E_ARG;
    int64_t at = 0;
    for (int64_t d = 0; d < ndocs; ++d) {
        const int c = counts[d];
        if (c > 0) { memcpy(ids_out + at, ids + wv_ids_slot(off[d], d), (size_t)c * 4); at += c; }
    }
    return n;
}

int64_t run_host_locked(Handle *h, const char *text, const int64_t *doc_off, int64_t ndocs, int32_t *ids_out, int64_t ids_cap,
                        int64_t *id_off_out, int max_ids, int unk, int32_t *starts_out, int32_t *ends_out, int words, bool *first_doc_nonempty, bool defer_ids);

// defer_ids (the sharded path): tokenise only -- offsets to the host, ids (and their spans when starts_out / ends_out are non-NULL,
// which are then only flags) stay in the handle's device buffers w_ids / w_starts / w_ends for fetch_deferred_ids()
int64_t run_host(Handle *h, const char *text, const int64_t *doc_off, int64_t ndocs, int32_t *ids_out, int64_t ids_cap,
                 int64_t *id_off_out, int max_ids, int unk, int32_t *starts_out = nullptr, int32_t *ends_out = nullptr, int words = 0,
                 bool *first_doc_nonempty = nullptr /* words modes: the first document decoded to >= 1 character */, bool defer_ids = false)
{
    if (ndocs < 0 || !doc_off || (ndocs > 0 && !text && doc_off[ndocs] > doc_off[0])) return BF_E_ARG;
    if (ndocs > 0 && doc_off[ndocs] - doc_off[0] < 0) return BF_E_ARG;
    std::unique_lock<std::mutex> dl(h->defer_mu, std::defer_lock);
    if (!defer_ids) dl.lock();                                    // (a range of a sharded call holds it already, until its ids are out)
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceGuard dg(h->device); if (!dg.ok) return BF_E_DEVICE;
    // A BPE document whose arcs do not fit the pool costs that document only (count 0, BF_STATUS_POOL) -- a caller of the host-buffer
    // API never sees it: the pool grows by what did not fit and the batch runs again (the reference collects into an unbounded
    // std::vector, ..._bpe_t.h:143-144)
    for (int attempt = 0; attempt < 8; ++attempt) {
        const bool is_bpe = h->m.kind == KIND_BPE || h->m.kind == KIND_BPE_OPT || h->m.kind == KIND_BPE_MERGES;
        if (is_bpe && !hip_ok(hipMemset(h->w_misc.as<char>() + 224, 0, 8), "hipMemset")) return BF_E_DEVICE;      // bytes the pool lacked, over all launches of this attempt
        const int64_t r = run_host_locked(h, text, doc_off, ndocs, ids_out, ids_cap, id_off_out, max_ids, unk, starts_out, ends_out, words, first_doc_nonempty, defer_ids);
        if (r != BF_RETRY_POOL) return r;
        (void)hipDeviceSynchronize();
        unsigned long long need = 0;
        if (!hip_ok(hipMemcpy(&need, h->w_misc.as<char>() + 224, 8, hipMemcpyDeviceToHost), "D2H pool need")) return BF_E_DEVICE;
        const size_t want = std::max(h->w_big.cap * 2, h->w_big.cap + (size_t)need + (size_t)(need >> 2) + ((size_t)1 << 20));
        h->bpe_pool_bytes = want;
        if (!h->w_big.reserve(want)) { g_last_error = "the arc pool of a BPE document does not fit the device memory"; return BF_E_DEVICE; }
    }
    g_last_error = "BPE arc pool: still too small after eight rounds of growing"; return BF_E_INTERNAL;
}

int64_t run_host_locked(Handle *h, const char *text, const int64_t *doc_off, int64_t ndocs, int32_t *ids_out, int64_t ids_cap,
                        int64_t *id_off_out, int max_ids, int unk, int32_t *starts_out, int32_t *ends_out, int words, bool *first_doc_nonempty, bool defer_ids)
{
    const bool want_off = starts_out && ends_out;
    const int64_t base = doc_off[0];
    const int64_t total = ndocs > 0 ? doc_off[ndocs] - base : 0;
    hipStream_t s = h->stream;
    if (!defer_ids && ndocs >= 1 && ndocs <= SMALL_MAX_DOCS && total <= SMALL_MAX_BYTES && !want_off && use_wave(h, want_off, words) && small_ready(h))
        return run_host_mapped(h, text, doc_off, ndocs, ids_out, ids_cap, id_off_out, max_ids, unk);
    if (!defer_ids && !want_off && !words && h->m.kind != KIND_I2W && h->host_chunk_bytes > 0 && total >= h->host_chunk_bytes && ndocs >= 2) {
        const int64_t r = run_host_chunked(h, text, doc_off, ndocs, ids_out, ids_cap, id_off_out, max_ids, unk);
        if (r != HOST_PIPE_UNAVAILABLE) return r;
    }
    // worst-case id count: _wp ids cover >= 1 byte each; _sp tokens cover >= 1 element of <= mul*(n+1) elements
    int64_t worst = h->m.kind == KIND_WP ? total : (int64_t)(h->m.dict_has_charmap ? 2 : 1) * (total + ndocs);
    if (max_ids >= 0 && ndocs * (int64_t)max_ids < worst) worst = ndocs * (int64_t)(max_ids < 0 ? 0 : max_ids);
    if (!h->w_text.reserve((size_t)total + 16) || !h->w_docoff.reserve((size_t)(ndocs + 1) * 8) ||
        !h->w_idoff.reserve((size_t)(ndocs + 1) * 8) || !h->w_ids.reserve((size_t)(worst + 1) * 4)) return BF_E_DEVICE;
    if (want_off && (!h->w_starts.reserve((size_t)(worst + 1) * 4) || !h->w_ends.reserve((size_t)(worst + 1) * 4))) return BF_E_DEVICE;
    std::vector<int64_t> rel;
    const int64_t *src_off = doc_off;
    if (base != 0) { rel.resize((size_t)ndocs + 1); for (int64_t i = 0; i <= ndocs; ++i) rel[(size_t)i] = doc_off[i] - base; src_off = rel.data(); }
    if (total > 0 && !hip_ok(hipMemcpyAsync(h->w_text.p, text + base, (size_t)total, hipMemcpyHostToDevice, s), "H2D text")) return BF_E_DEVICE;
    if (!hip_ok(hipMemcpyAsync(h->w_docoff.p, src_off, (size_t)(ndocs + 1) * 8, hipMemcpyHostToDevice, s), "H2D offsets")) { (void)hipStreamSynchronize(s); return BF_E_DEVICE; }
    int rc = run_device(h, h->w_text.as<char>(), h->w_docoff.as<int64_t>(), ndocs, total, h->w_ids.as<int32_t>(), worst,
                        h->w_idoff.as<int64_t>(), max_ids, unk, s, want_off ? h->w_starts.as<int32_t>() : nullptr,
                        want_off ? h->w_ends.as<int32_t>() : nullptr, words);
    if (rc != 0) { (void)hipStreamSynchronize(s); return rc; }     // `rel` may still be the source of the pending offsets copy
    std::vector<int64_t> tmp_off;
    int64_t *dst_off = id_off_out;
    if (!dst_off) { tmp_off.resize((size_t)ndocs + 1); dst_off = tmp_off.data(); }
    if (!hip_ok(hipMemcpyAsync(dst_off, h->w_idoff.p, (size_t)(ndocs + 1) * 8, hipMemcpyDeviceToHost, s), "D2H offsets")) return BF_E_DEVICE;
    // the id count through a word of its own: in the sharded path dst_off[ndocs] is also the first entry of the next range, which that
    // range's thread writes concurrently
    int64_t nids_word = 0;
    if (!hip_ok(hipMemcpyAsync(&nids_word, h->w_idoff.as<int64_t>() + ndocs, 8, hipMemcpyDeviceToHost, s), "D2H id count")) return BF_E_DEVICE;
    int status = 0;
    if (!hip_ok(hipMemcpyAsync(&status, h->w_misc.as<char>() + 16, 4, hipMemcpyDeviceToHost, s), "D2H status")) return BF_E_DEVICE;
    int32_t nch0 = 0;
    if (words && ndocs == 1 && !hip_ok(hipMemcpyAsync(&nch0, h->w_nchars.p, 4, hipMemcpyDeviceToHost, s), "D2H nchars")) return BF_E_DEVICE;
    if (!hip_ok(hipStreamSynchronize(s), "hipStreamSynchronize")) return BF_E_DEVICE;
    if (first_doc_nonempty) *first_doc_nonempty = nch0 > 0;
    if (status & (2 | BF_STATUS_INTERNAL)) return BF_E_INTERNAL;
    if (status & BF_STATUS_POOL) return BF_RETRY_POOL;
    const int64_t nids = nids_word;
    if (defer_ids) return nids;
    if (nids > ids_cap) return BF_E_CAPACITY;
    if (nids > 0) {
        if (!ids_out) return BF_E_ARG;
        if (!hip_ok(hipMemcpy(ids_out, h->w_ids.p, (size_t)nids * 4, hipMemcpyDeviceToHost), "D2H ids")) return BF_E_DEVICE;
        if (want_off && (!hip_ok(hipMemcpy(starts_out, h->w_starts.p, (size_t)nids * 4, hipMemcpyDeviceToHost), "D2H starts") ||
                         !hip_ok(hipMemcpy(ends_out, h->w_ends.p, (size_t)nids * 4, hipMemcpyDeviceToHost), "D2H ends"))) return BF_E_DEVICE;
    }
    return nids;
}

// TextToWords for a batch resident on the device: lexer in words mode -> spans -> lengths -> scan -> byte gather.
// The word tags / spans stay in the handle's buffers; d_out may be NULL to size only (d_out_off is always filled).
int run_words_device(Handle *h, const char *d_text, const int64_t *d_doc_off, int64_t ndocs, int64_t total_bytes,
                     char *d_out, int64_t out_cap, int64_t *d_out_off, hipStream_t s, bool tokenise, int mode = 1 /* 1 words, 2 sentences */)
{
    if (h->m.kind != KIND_WP) return BF_E_UNSUPPORTED;
    if (h->m.lexer_void) { g_last_error = "TextToWords / TextToSentences: a moore-multi-dfa lexer answers -1 to every input (the single-document calls return that)"; return BF_E_UNSUPPORTED; }
    if (ndocs < 0 || total_bytes < 0 || !d_doc_off || !d_out_off) return BF_E_ARG;
    if (tokenise) {
        if (!h->w_ids.reserve((size_t)(total_bytes + 1) * 4) || !h->w_starts.reserve((size_t)(total_bytes + 1) * 4) ||
            !h->w_ends.reserve((size_t)(total_bytes + 1) * 4) || !h->w_idoff.reserve((size_t)(ndocs + 1) * 8)) return BF_E_DEVICE;
        int rc = run_device(h, d_text, d_doc_off, ndocs, total_bytes, h->w_ids.as<int32_t>(), total_bytes, h->w_idoff.as<int64_t>(), 0x7fffffff, 0, s,
                            h->w_starts.as<int32_t>(), h->w_ends.as<int32_t>(), mode);
        if (rc != 0) return rc;
    }
    const int nblocks = scan_nblocks(ndocs);
    W2tParams p{(const uint8_t *)d_text, d_doc_off, ndocs, h->w_idoff.as<int64_t>(), h->w_starts.as<int32_t>(), h->w_ends.as<int32_t>(),
                h->w_counts.as<int32_t>(), d_out_off, (uint8_t *)d_out, out_cap, mode == 2 ? h->w_nchars.as<int32_t>() : nullptr};
    // documents of many tokens are measured and assembled by sixteen waves each (k_w2t_len_long / k_w2t_copy_long); w_misc + 56 / + 52: their number
    // (+ 56 zeroed with the status word by run_device above, + 52 below)
    p.long_cap = total_bytes / 1024 + 2;
    p.long_list = h->w_w2tlong.reserve((size_t)p.long_cap * 8) ? h->w_w2tlong.as<int64_t>() : nullptr;
    p.long_count = (unsigned int *)(h->w_misc.as<char>() + 56);
    if (tokenise) {
        if (ndocs > 0) { if (mode == 2) launch_s2t_len(p, s); else launch_w2t_len(p, s); }
        ScanParams sp{h->w_counts.as<int32_t>(), ndocs, d_out_off, h->w_bsums.as<int64_t>(), nblocks};
        launch_scan(sp, s);
    }
    if (d_out && ndocs > 0) {
        p.long_count = (unsigned int *)(h->w_misc.as<char>() + 52);
        if (!p.long_list || !hip_ok(hipMemsetAsync(p.long_count, 0, 4, s), "hipMemsetAsync")) p.long_list = nullptr;
        if (mode == 2) launch_s2t_copy(p, s); else launch_w2t_copy(p, s);
    }
    return hip_ok(hipGetLastError(), "TextToWords kernels") ? 0 : BF_E_DEVICE;
}

// IdsToText on device buffers: lengths -> scan -> byte gather (bf_kernels.hip).  d_text may be NULL to size only.
int run_i2t_device(Handle *h, const int32_t *d_ids, const int64_t *d_id_off, int64_t nseq, char *d_text, int64_t text_cap,
                   int64_t *d_text_off, int skip_special, hipStream_t s)
{
    const Model &m = h->m;
    if (!m.has_i2w) return BF_E_UNSUPPORTED;
    if (nseq < 0 || !d_id_off || !d_text_off) return BF_E_ARG;
    const int nblocks = scan_nblocks(nseq);
    if (!h->w_counts.reserve((size_t)(nseq + 1) * 4) || !h->w_bsums.reserve((size_t)(nblocks + 1) * 8)) return BF_E_DEVICE;
    I2tParams p;
    p.tok_off = h->t_i2w_off.as<uint32_t>(); p.tok_data = h->t_i2w_data.as<uint8_t>(); p.ntok = (int)m.i2w_off.size() - 1;
    p.min_id = m.min_token_id; p.max_id = m.max_token_id; p.skip_special = skip_special ? 1 : 0;
    p.ids = d_ids; p.id_off = d_id_off; p.nseq = nseq; p.lens = h->w_counts.as<int32_t>();
    p.text_off = d_text_off; p.text = (uint8_t *)d_text; p.text_cap = text_cap; p.status = (int *)(h->w_misc.as<char>() + 16);
    if (nseq > 0 && !d_text) launch_i2t_len(p, s);
    if (!d_text) {
        ScanParams sp{h->w_counts.as<int32_t>(), nseq, d_text_off, h->w_bsums.as<int64_t>(), nblocks};
        launch_scan(sp, s);
    } else if (nseq > 0) launch_i2t_copy(p, s);
    return hip_ok(hipGetLastError(), "IdsToText kernels") ? 0 : BF_E_DEVICE;
}

// host buffers: text of sequence d = text_out[text_off_out[d] .. text_off_out[d+1]); returns the total byte count
int64_t run_i2t_host(Handle *h, const int32_t *ids, const int64_t *id_off, int64_t nseq, char *text_out, int64_t text_cap,
                     int64_t *text_off_out, int skip_special, bool *unknown_id = nullptr)
{
    if (!h->m.has_i2w) return BF_E_UNSUPPORTED;
    if (nseq < 0 || !id_off || (nseq > 0 && id_off[nseq] > id_off[0] && !ids)) return BF_E_ARG;
    const int64_t base = id_off[0], total_ids = nseq > 0 ? id_off[nseq] - base : 0;
    if (total_ids < 0) return BF_E_ARG;
    std::lock_guard<std::mutex> dlock(h->defer_mu);      // (the id buffers this call uses may hold a sharded range's ids that wait for their copy out)
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceGuard dg(h->device); if (!dg.ok) return BF_E_DEVICE;
    hipStream_t s = h->stream;
    if (!h->w_ids.reserve((size_t)(total_ids + 1) * 4) || !h->w_docoff.reserve((size_t)(nseq + 1) * 8) || !h->w_idoff.reserve((size_t)(nseq + 1) * 8)) return BF_E_DEVICE;
    std::vector<int64_t> rel((size_t)nseq + 1);
    for (int64_t i = 0; i <= nseq; ++i) rel[(size_t)i] = id_off[i] - base;
    if (!hip_ok(hipMemsetAsync(h->w_misc.p, 0, 64, s), "hipMemsetAsync")) return BF_E_DEVICE;
    if (total_ids > 0 && !hip_ok(hipMemcpyAsync(h->w_ids.p, ids + base, (size_t)total_ids * 4, hipMemcpyHostToDevice, s), "H2D ids")) return BF_E_DEVICE;
    if (!hip_ok(hipMemcpyAsync(h->w_docoff.p, rel.data(), (size_t)(nseq + 1) * 8, hipMemcpyHostToDevice, s), "H2D offsets")) return BF_E_DEVICE;
    int rc = run_i2t_device(h, h->w_ids.as<int32_t>(), h->w_docoff.as<int64_t>(), nseq, nullptr, 0, h->w_idoff.as<int64_t>(), skip_special, s);
    if (rc != 0) { (void)hipStreamSynchronize(s); return rc; }
    std::vector<int64_t> tmp_off;
    int64_t *dst_off = text_off_out;
    if (!dst_off) { tmp_off.resize((size_t)nseq + 1); dst_off = tmp_off.data(); }
    int status = 0;
    if (!hip_ok(hipMemcpyAsync(dst_off, h->w_idoff.p, (size_t)(nseq + 1) * 8, hipMemcpyDeviceToHost, s), "D2H offsets") ||
        !hip_ok(hipMemcpyAsync(&status, h->w_misc.as<char>() + 16, 4, hipMemcpyDeviceToHost, s), "D2H status") ||
        !hip_ok(hipStreamSynchronize(s), "hipStreamSynchronize")) return BF_E_DEVICE;
    if (unknown_id) *unknown_id = (status & 4) != 0;
    const int64_t total = dst_off[nseq];
    if (total > text_cap) return BF_E_CAPACITY;
    if (total > 0) {
        if (!text_out) return BF_E_ARG;
        if (!h->w_text.reserve((size_t)total + 16)) return BF_E_DEVICE;
        rc = run_i2t_device(h, h->w_ids.as<int32_t>(), h->w_docoff.as<int64_t>(), nseq, h->w_text.as<char>(), total, h->w_idoff.as<int64_t>(), skip_special, s);
        if (rc != 0) return rc;
        if (!hip_ok(hipMemcpyAsync(text_out, h->w_text.p, (size_t)total, hipMemcpyDeviceToHost, s), "D2H text") ||
            !hip_ok(hipStreamSynchronize(s), "hipStreamSynchronize")) return BF_E_DEVICE;
    }
    return total;
}

// upload a batch of documents (host buffers) into the handle's text / offset buffers; returns the byte total or BF_E_*
int64_t upload_docs(Handle *h, const char *text, const int64_t *doc_off, int64_t ndocs, hipStream_t s)
{
    if (ndocs < 0 || !doc_off || (ndocs > 0 && !text && doc_off[ndocs] > doc_off[0])) return BF_E_ARG;
    const int64_t base = doc_off[0], total = ndocs > 0 ? doc_off[ndocs] - base : 0;
    if (total < 0) return BF_E_ARG;
    if (!h->w_text.reserve((size_t)total + 16) || !h->w_docoff.reserve((size_t)(ndocs + 1) * 8) || !h->w_outoff.reserve((size_t)(ndocs + 1) * 8)) return BF_E_DEVICE;
    std::vector<int64_t> rel((size_t)ndocs + 1);
    for (int64_t i = 0; i <= ndocs; ++i) rel[(size_t)i] = doc_off[i] - base;
    if (total > 0 && !hip_ok(hipMemcpyAsync(h->w_text.p, text + base, (size_t)total, hipMemcpyHostToDevice, s), "H2D text")) return BF_E_DEVICE;
    if (!hip_ok(hipMemcpyAsync(h->w_docoff.p, rel.data(), (size_t)(ndocs + 1) * 8, hipMemcpyHostToDevice, s), "H2D offsets")) return BF_E_DEVICE;
    if (!hip_ok(hipStreamSynchronize(s), "hipStreamSynchronize")) return BF_E_DEVICE;     // `rel` goes out of scope
    return total;
}

static int utf8_encode(int c, unsigned char *o)     // FAIntToUtf8 (cl/src/FAUtf8Utils.cpp:471-527); 0 = cannot be encoded
{
    const unsigned u = (unsigned)c;
    if (u <= 0x7F) { o[0] = (unsigned char)u; return 1; }
    if (u <= 0x7FF) { o[0] = (unsigned char)(0xC0 | (u >> 6)); o[1] = (unsigned char)(0x80 | (u & 0x3F)); return 2; }
    if (u <= 0xFFFF) { if ((u & 0xFFFFF800u) == 0xD800u) return 0; o[0] = (unsigned char)(0xE0 | (u >> 12)); o[1] = (unsigned char)(0x80 | ((u >> 6) & 0x3F)); o[2] = (unsigned char)(0x80 | (u & 0x3F)); return 3; }
    if (u <= 0x10FFFF) { o[0] = (unsigned char)(0xF0 | (u >> 18)); o[1] = (unsigned char)(0x80 | ((u >> 12) & 0x3F)); o[2] = (unsigned char)(0x80 | ((u >> 6) & 0x3F)); o[3] = (unsigned char)(0x80 | (u & 0x3F)); return 4; }
    return 0;
}

// NormalizeSpaces on device buffers: size pass (+ scan) when `size` is set, write pass when d_out is given
int run_normsp_device(Handle *h, const char *d_text, const int64_t *d_doc_off, int64_t ndocs, int u_space, char *d_out, int64_t out_cap,
                      int64_t *d_out_off, hipStream_t s, bool size)
{
    if (ndocs < 0 || !d_doc_off || !d_out_off) return BF_E_ARG;
    const int nblocks = scan_nblocks(ndocs);
    if (!h->w_counts.reserve((size_t)(ndocs + 1) * 4) || !h->w_nchars.reserve((size_t)(ndocs + 1) * 4) || !h->w_bsums.reserve((size_t)(nblocks + 1) * 8)) return BF_E_DEVICE;
    unsigned char ub[4] = {0, 0, 0, 0};
    NormSpParams p;
    p.text = (const uint8_t *)d_text; p.doc_off = d_doc_off; p.ndocs = ndocs; p.u_space = u_space; p.usp_len = utf8_encode(u_space, ub);
    p.usp_bytes = (uint32_t)ub[0] | ((uint32_t)ub[1] << 8) | ((uint32_t)ub[2] << 16) | ((uint32_t)ub[3] << 24);
    p.lens = h->w_counts.as<int32_t>(); p.aux = h->w_nchars.as<int32_t>(); p.out_off = d_out_off; p.out = (uint8_t *)d_out; p.out_cap = out_cap;
    if (size) {
        if (ndocs > 0) launch_normsp(p, false, s);
        ScanParams sp{h->w_counts.as<int32_t>(), ndocs, d_out_off, h->w_bsums.as<int64_t>(), nblocks};
        launch_scan(sp, s);
    }
    if (d_out && ndocs > 0) launch_normsp(p, true, s);
    return hip_ok(hipGetLastError(), "NormalizeSpaces kernels") ? 0 : BF_E_DEVICE;
}

int run_hashes_device(Handle *h, const char *d_text, const int64_t *d_doc_off, int64_t ndocs, int ngrams, int bucket, int32_t *d_out, int64_t out_cap,
                      int64_t *d_out_off, hipStream_t s, bool size)
{
    if (ndocs < 0 || !d_doc_off || !d_out_off || ngrams <= 0 || bucket == 0) return BF_E_ARG;
    const int nblocks = scan_nblocks(ndocs);
    if (!h->w_counts.reserve((size_t)(ndocs + 1) * 4) || !h->w_bsums.reserve((size_t)(nblocks + 1) * 8)) return BF_E_DEVICE;
    HashParams p{(const uint8_t *)d_text, d_doc_off, ndocs, ngrams, bucket, h->w_counts.as<int32_t>(), d_out_off, d_out, out_cap};
    if (size) {
        if (ndocs > 0) launch_hash_count(p, s);
        ScanParams sp{h->w_counts.as<int32_t>(), ndocs, d_out_off, h->w_bsums.as<int64_t>(), nblocks};
        launch_scan(sp, s);
    }
    if (d_out && ndocs > 0) launch_hash_fill(p, s);
    return hip_ok(hipGetLastError(), "TextToHashes kernels") ? 0 : BF_E_DEVICE;
}

// key -> info lookup: tables of the [pos-dict] in their lookup form, uploaded when the first call arrives
bool ensure_dict_tables(Handle *h)
{
    if (h->dict_ready) return true;
    const Model &m = h->m;
    bool ok = upload(h->t_dk_l1, m.dict_clsmap.l1) && upload(h->t_dk_pages, m.dict_clsmap.pages) && upload(h->t_k2i, m.k2i, 4) && upload(h->t_rows, m.info_rows, 4);
    if (m.dict_ignore_case)             // fold + charmap as one map (bf_model.cpp); the pool may be empty (no 1:n entry)
        ok = ok && upload(h->t_dn_l1, m.dict_lookup_map.l1) && upload(h->t_dn_pages, m.dict_lookup_map.pages) && (m.dict_norm_pool.empty() || upload(h->t_dn_pool, m.dict_norm_pool, 4));
    else if (m.dict_direction != 0 && m.dict_has_charmap)
        ok = ok && upload(h->t_dn_l1, m.dict_charmap.l1) && upload(h->t_dn_pages, m.dict_charmap.pages) && upload(h->t_dn_pool, m.dict_norm_pool, 4);
    h->dict_ready = ok;
    return ok;
}

int run_dict_device(Handle *h, const int32_t *d_keys, const int64_t *d_key_off, int64_t nkeys, int32_t *d_ret, int32_t *d_info_ids,
                    int32_t *d_vals, int64_t vals_cap, int64_t *d_val_off, hipStream_t s, bool size)
{
    const Model &m = h->m;
    if (!m.has_seg || m.k2i.empty()) return BF_E_UNSUPPORTED;
    // the reference looks a key up as it is only for (no ignore-case, left-to-right) dictionaries, lower-cases for ignore-case ones and
    // reverses only right-to-left ones (FADictInterpreter_t.h:203-205, FAFsmConst.h DIR_L2R = 0, DIR_R2L = 1): the other combinations are
    // refused here instead of being looked up wrongly
    if (m.dict_direction != 0 && m.dict_direction != 1) { g_last_error = "dictionary lookup: direction other than l2r, r2l is not supported"; return BF_E_UNSUPPORTED; }
    if (nkeys < 0 || !d_key_off || !d_val_off) return BF_E_ARG;
    const int nblocks = scan_nblocks(nkeys);
    if (!ensure_dict_tables(h) || !h->w_counts.reserve((size_t)(nkeys + 1) * 4) || !h->w_bsums.reserve((size_t)(nblocks + 1) * 8) ||
        !h->w_dids.reserve((size_t)(nkeys + 1) * 4) || !h->w_dret.reserve((size_t)(nkeys + 1) * 4)) return BF_E_DEVICE;
    DictParams p;
    p.D.T = h->t_dict.as<uint64_t>(); p.D.initial = m.dict.initial_base; p.D.initial_final = m.dict_raw.is_final[(size_t)m.dict_raw.initial] ? 1 : 0;
    p.D.cls_l1 = h->t_dk_l1.as<uint16_t>(); p.D.cls_pages = h->t_dk_pages.as<uint32_t>();
    // keys are normalised unless (no ignore-case, left-to-right): m_NoNorm, FADictInterpreter_t.h:203-205.  Ignore-case: fold + charmap in one map
    const bool nrm = m.dict_ignore_case || (m.dict_direction != 0 && m.dict_has_charmap);
    p.D.nrm_l1 = nrm ? h->t_dn_l1.as<uint16_t>() : nullptr; p.D.nrm_pages = nrm ? h->t_dn_pages.as<uint32_t>() : nullptr; p.D.nrm_pool = nrm ? h->t_dn_pool.as<int32_t>() : nullptr;
    p.D.k2i = h->t_k2i.as<int32_t>(); p.D.k2i_n = (int)m.k2i.size(); p.D.r2l = m.dict_direction != 0 ? 1 : 0; p.D.ignore_case = m.dict_ignore_case ? 1 : 0;
    p.rows = h->t_rows.as<int32_t>(); p.stride = m.info_stride; p.min_key = m.info_min_key; p.nrows = m.info_stride > 0 ? (int)(m.info_rows.size() / (size_t)m.info_stride) : 0;
    p.keys = d_keys; p.key_off = d_key_off; p.nkeys = nkeys;
    p.info_ids = d_info_ids ? d_info_ids : h->w_dids.as<int32_t>(); p.ret = d_ret ? d_ret : h->w_dret.as<int32_t>(); p.counts = h->w_counts.as<int32_t>();
    p.val_off = d_val_off; p.vals = d_vals; p.vals_cap = vals_cap;
    if (size) {
        if (nkeys > 0) launch_dict_ids(p, s);
        ScanParams sp{h->w_counts.as<int32_t>(), nkeys, d_val_off, h->w_bsums.as<int64_t>(), nblocks};
        launch_scan(sp, s);
    }
    if (d_vals && nkeys > 0) launch_dict_fill(p, s);
    return hip_ok(hipGetLastError(), "dictionary lookup kernels") ? 0 : BF_E_DEVICE;
}

// ids one document of n bytes can produce at most (WordPiece: one per character; SentencePiece-style: the dummy prefix and a
// charmap that expands 1:2 -- the same bound the device workspaces use), capped by the caller's array
static int64_t one_doc_cap(int n, int max_ids) { const int64_t w = 2 * ((int64_t)n + 1); return w < max_ids ? w : (int64_t)max_ids; }

void run_one_group_impl(Handle *h, std::vector<Handle::OneReq *> &g)
{
    const int max_ids = g[0]->max_ids, unk = g[0]->unk; const bool want_off = g[0]->starts && g[0]->ends;
    const int64_t nd = (int64_t)g.size();
    std::vector<int64_t> off((size_t)nd + 1, 0), id_off((size_t)nd + 1, 0);
    int64_t cap = 0;
    for (int64_t i = 0; i < nd; ++i) { off[(size_t)i + 1] = off[(size_t)i] + g[(size_t)i]->n; cap += one_doc_cap(g[(size_t)i]->n, max_ids); }
    std::string packed; const char *text = g[0]->s;
    if (nd > 1) { packed.resize((size_t)off[(size_t)nd]); for (int64_t i = 0; i < nd; ++i) memcpy(&packed[(size_t)off[(size_t)i]], g[(size_t)i]->s, (size_t)g[(size_t)i]->n); text = packed.data(); }
    std::vector<int32_t> ids, st, en; int32_t *pi = g[0]->ids, *ps = g[0]->starts, *pe = g[0]->ends;
    if (nd > 1) { ids.resize((size_t)cap); pi = ids.data(); if (want_off) { st.resize((size_t)cap); en.resize((size_t)cap); ps = st.data(); pe = en.data(); } }
    const int64_t r = run_host(h, text, off.data(), nd, pi, cap, id_off.data(), max_ids, unk, want_off ? ps : nullptr, want_off ? pe : nullptr);
    if (r < 0) fprintf(stderr, "[blingfire_amd] TextToIds failed (%lld): %s\n", (long long)r, g_last_error.c_str());
    for (int64_t i = 0; i < nd; ++i) {
        Handle::OneReq *q = g[(size_t)i];
        if (r < 0) { q->result = 0; continue; }
        const int64_t b = id_off[(size_t)i], c = id_off[(size_t)i + 1] - b;
        if (nd > 1 && c > 0) {
            memcpy(q->ids, pi + b, (size_t)c * 4);
            if (want_off) { memcpy(q->starts, ps + b, (size_t)c * 4); memcpy(q->ends, pe + b, (size_t)c * 4); }
        }
        q->result = (int)c;
    }
}

// One launch for a group of single-document requests that share (max_ids, unk, offsets wanted): the documents are packed,
// run as one batch, and every caller gets exactly what its own TextToIds call would have written.  Nothing may leave through the
// extern "C" entry points: when the temporaries of a combined launch cannot be allocated, the requests are served one by one
// (a batch of one writes straight into the caller's arrays).
void run_one_group(Handle *h, std::vector<Handle::OneReq *> &g)
{
    try { run_one_group_impl(h, g); return; }
    catch (const std::bad_alloc &) {}
    for (Handle::OneReq *q : g) {
        std::vector<Handle::OneReq *> one{q};
        try { run_one_group_impl(h, one); } catch (const std::bad_alloc &) { q->result = 0; }
    }
}

static const int g_max_spinners = []() { const unsigned hc = std::thread::hardware_concurrency(); const int k = (int)(hc / 4); return k < 1 ? 1 : (k > 64 ? 64 : k); }();

// a request's state word: the caller sleeps on it (futex), the leader that served the request -- or hands the lead over -- changes it
enum { ONE_WAITING = 0, ONE_DONE = 1, ONE_LEAD = 2 };
static void one_wait_word(std::atomic<int> *w, int timeout_us)
{
    struct timespec ts = {0, (long)timeout_us * 1000};
    (void)syscall(SYS_futex, (int *)w, FUTEX_WAIT_PRIVATE, ONE_WAITING, &ts, nullptr, 0);
}
static void one_wake_word(std::atomic<int> *w) { (void)syscall(SYS_futex, (int *)w, FUTEX_WAKE_PRIVATE, 1, nullptr, nullptr, 0); }

// TextToIds for one document.  The GPU runs batches: a call that arrives while another one is in flight on the same handle
// waits in a queue, and whoever leads launches everything that queued up as ONE batch (flat combining) -- so that concurrent
// callers of the drop-in entry points (C#, Python threads: SURVEY.md section 8b "Threading") share launches instead of serialising
// one ~30 us launch each.  A lone caller simply runs its own batch of one.  A waiting caller watches its own state word: for about
// the length of a launch without a system call (up to g_max_spinners callers at a time), then asleep on it; whoever served it
// wakes that one word.  (Measured with 64 native threads, profiles/r03_single_calls.txt: one condition variable for all callers
// cost ~340 us per launch in wake-ups and mutex hand-overs, against ~45 us for the launch itself.)
int text_to_ids_one(void *hp, const char *s, int n, int32_t *ids, int max_ids, int unk, int want_kind /* -1 any, 0 wp, 1 sp */,
                    int32_t *starts = nullptr, int32_t *ends = nullptr)
{
    Handle *h = as_handle(hp);
    if (!h) return 0;                                         // tokdll:1629-1631
    if (n <= 0 || n > 1000000000 || !s) return 0;             // tokdll:1121-1123
    if (want_kind == 0 && h->m.kind != KIND_WP) return 0;
    if (want_kind == 1 && h->m.kind == KIND_WP) return 0;
    if (max_ids <= 0 || !ids) return 0;
    Handle::OneReq me{s, n, ids, max_ids, unk, starts, ends, 0, {ONE_WAITING}};
    std::unique_lock<std::mutex> lk(h->q_mu);
    h->q.push_back(&me);
    if (h->q_leader) {
        lk.unlock();
        for (;;) {
            if (h->q_spinners.fetch_add(1) < g_max_spinners) {
                for (int spin = 0; spin < 3000 && me.state.load(std::memory_order_acquire) == ONE_WAITING; ++spin) cpu_relax();
            }
            h->q_spinners.fetch_sub(1);
            if (me.state.load(std::memory_order_acquire) != ONE_WAITING) break;
            h->q_sleepers.fetch_add(1);
            if (me.state.load() == ONE_WAITING) one_wait_word(&me.state, 2000);
            h->q_sleepers.fetch_sub(1);
            if (me.state.load(std::memory_order_acquire) != ONE_WAITING) break;
            // nothing after 2 ms (or a stray wake-up): the lead may be free -- it is handed over, never dropped with requests queued,
            // so this is only a safeguard
            lk.lock();
            if (me.state.load(std::memory_order_acquire) != ONE_WAITING) { lk.unlock(); break; }
            if (!h->q_leader) { h->q_leader = true; me.state.store(ONE_LEAD); lk.unlock(); break; }
            lk.unlock();
        }
        if (me.state.load(std::memory_order_acquire) == ONE_DONE) return me.result;
        lk.lock();                                            // ONE_LEAD: the lead was handed to this caller (q_leader stays set)
    }
    h->q_leader = true;                                       // lead: serve everything queued (at least my own request)
    for (int round = 0; round < 4 && !h->q.empty(); ++round) {
        std::vector<Handle::OneReq *> all; all.swap(h->q);
        lk.unlock();
        while (!all.empty()) {                                // groups of requests with the same call parameters
            std::vector<Handle::OneReq *> g, rest; int64_t bytes = 0, idcap = 0;
            // the grouping allocates: out of memory here must not leave the taken requests waiting for ever -- they are answered 0
            // (the reference's "no ids" result) and the lead goes on
            try { g.reserve(all.size()); rest.reserve(all.size()); }
            catch (const std::bad_alloc &) {
                for (Handle::OneReq *q : all) {
                    q->result = 0;
                    if (q == &me) continue;
                    std::atomic<int> *w = &q->state;
                    w->store(ONE_DONE);
                    if (h->q_sleepers.load() > 0) one_wake_word(w);
                }
                all.clear();
                break;
            }
            for (Handle::OneReq *q : all) {
                const bool same = q->max_ids == all[0]->max_ids && q->unk == all[0]->unk && ((q->starts && q->ends) == (all[0]->starts && all[0]->ends));
                const int64_t c = one_doc_cap(q->n, q->max_ids);
                // the first request of a group is always admitted (whatever its max_ids: INT_MAX means "no limit"); the others while
                // the combined batch stays within the limits of one launch
                if (g.empty() || (same && bytes + q->n <= 1000000000 && idcap + c <= 1000000000)) { g.push_back(q); bytes += q->n; idcap += c; } else rest.push_back(q);
            }
            const auto t0 = std::chrono::steady_clock::now();
            run_one_group(h, g);
            h->one_rounds.fetch_add(1, std::memory_order_relaxed); h->one_reqs.fetch_add((long long)g.size(), std::memory_order_relaxed);
            h->one_ns.fetch_add(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed);
            // a request is gone once its caller has seen ONE_DONE: nothing of *q is read afterwards (waking a word nobody sleeps on
            // is harmless, also when its stack frame is no longer live)
            for (Handle::OneReq *q : g) {
                if (q == &me) continue;
                std::atomic<int> *w = &q->state;
                w->store(ONE_DONE);
                if (h->q_sleepers.load() > 0) one_wake_word(w);
            }
            all.swap(rest);
        }
        lk.lock();
    }
    // requests still queued are led by one of their own callers: the lead is handed to the oldest of them
    if (!h->q.empty()) {
        std::atomic<int> *w = &h->q.front()->state;
        w->store(ONE_LEAD);
        if (h->q_sleepers.load() > 0) one_wake_word(w);
    } else h->q_leader = false;
    return me.result;
}

// Static contiguous range shard of a batch over G devices, byte-balanced: range g = documents
// [bounds[g], bounds[g + 1]), where bounds[g] is the document boundary closest to g/G of the text (the earlier one on a tie).
void shard_bounds(const int64_t *doc_off, int64_t ndocs, int G, int64_t *bounds)
{
    const int64_t base = ndocs > 0 ? doc_off[0] : 0, total = ndocs > 0 ? doc_off[ndocs] - base : 0;
    bounds[0] = 0; bounds[G] = ndocs;
    for (int g = 1; g < G; ++g) {
        const int64_t target = base + (int64_t)((__int128)total * g / G);
        const int64_t *it = std::lower_bound(doc_off, doc_off + ndocs + 1, target);       // first boundary at or behind the target
        int64_t d = it - doc_off;
        if (d > ndocs) d = ndocs;
        if (d > 0 && target - doc_off[d - 1] <= doc_off[d] - target) --d;                 // the boundary before it is at least as close
        if (d < bounds[g - 1]) d = bounds[g - 1];
        bounds[g] = d;
    }
}

// TextToIdsBatch of a handle with several devices: one host thread per range, each through its own handle, in two steps.  Step 1: text
// in, kernels, id offsets out -- the ids stay in the range's device buffer.  As soon as the ranges before it have published their id
// counts a range knows its place in the caller's array, and step 2 copies its ids straight there and rebases its offsets in place (no
// barrier: range 0 is on its way out while later ranges are still on their way in).  Ranges that share a DEVICE take turns for step 1
// (a link moves one direction at full speed: serialising the ways in is what lets the way out of one range overlap the way in of the
// next).  The host holds the caller's input and output and nothing else: the worst-case id array of a range -- 4 bytes per input byte
// for WordPiece -- lives on the device.  The caller sees exactly what one device would have returned.
int64_t run_host_sharded(Handle *h, const std::vector<Handle *> &shards, const char *text, const int64_t *doc_off, int64_t ndocs, int32_t *ids_out, int64_t ids_cap,
                         int64_t *id_off_out, int max_ids, int unk, int32_t *starts_out, int32_t *ends_out)
{
    (void)h;
    const int G = (int)shards.size();
    const bool want_off = starts_out && ends_out;
    if (ndocs < 0 || !doc_off || (ndocs > 0 && !text && doc_off[ndocs] > doc_off[0])) return BF_E_ARG;
    std::vector<int64_t> bounds((size_t)G + 1), nids((size_t)G, 0), base((size_t)G + 1, 0);
    shard_bounds(doc_off, ndocs, G, bounds.data());
    std::vector<int64_t> tmp_off;                                 // only when the caller wants no offsets
    int64_t *offs = id_off_out;
    if (!offs) { try { tmp_off.resize((size_t)ndocs + 1); } catch (const std::bad_alloc &) { g_last_error = "out of host memory"; return BF_E_DEVICE; } offs = tmp_off.data(); }
    std::vector<std::string> errs((size_t)G);
    std::vector<int64_t> rc((size_t)G, 0);
    int32_t flag = 0;                                             // non-NULL marker for "with spans" in step 1
    static std::mutex turn[64];                                   // step 1 of the ranges of one device, one at a time
    std::mutex pm; std::condition_variable pcv; int published = 0; bool failed = false;     // counts of ranges 0 .. published - 1 are known
    {
        std::vector<std::thread> th;
        for (int g = 0; g < G; ++g)
            th.emplace_back([&, g]() {
                Handle *c = shards[(size_t)g];
                const int64_t lo = bounds[(size_t)g], nd = bounds[(size_t)g + 1] - lo;
                int64_t r = 0;
                std::unique_lock<std::mutex> dl(c->defer_mu);           // from the range's kernels to the copy of its ids: nobody else uses this handle's id buffers
                if (nd > 0) {
                    // the range's offsets land in the caller's array at once, relative to the range.  Entry lo + nd is also the first entry of the
                    // next range: the boundary entries are set after the join
                    std::lock_guard<std::mutex> t(turn[c->device & 63]);
                    r = run_host(c, text, doc_off + lo, nd, nullptr, 0, offs + lo, max_ids, unk, want_off ? &flag : nullptr, want_off ? &flag : nullptr, 0, nullptr, true);
                    if (r < 0) errs[(size_t)g] = g_last_error;
                }
                int64_t at = 0;
                {
                    std::unique_lock<std::mutex> lk(pm);
                    pcv.wait(lk, [&] { return published == g || failed; });          // the ranges before this one have published
                    nids[(size_t)g] = r;
                    if (r < 0 || failed) { failed = true; rc[(size_t)g] = r < 0 ? r : 0; published = g + 1; pcv.notify_all(); return; }
                    at = base[(size_t)g]; base[(size_t)g + 1] = at + r; published = g + 1;
                    pcv.notify_all();
                }
                if (r > 0 && at + r <= ids_cap && ids_out) {
                    std::lock_guard<std::mutex> lock(c->mu);
                    DeviceGuard dg(c->device);
                    if (!dg.ok || !hip_ok(hipMemcpy(ids_out + at, c->w_ids.p, (size_t)r * 4, hipMemcpyDeviceToHost), "D2H ids") ||
                        (want_off && (!hip_ok(hipMemcpy(starts_out + at, c->w_starts.p, (size_t)r * 4, hipMemcpyDeviceToHost), "D2H starts") ||
                                      !hip_ok(hipMemcpy(ends_out + at, c->w_ends.p, (size_t)r * 4, hipMemcpyDeviceToHost), "D2H ends")))) { rc[(size_t)g] = BF_E_DEVICE; errs[(size_t)g] = g_last_error; }
                }
                if (id_off_out && at != 0) for (int64_t i = 1; i < nd; ++i) id_off_out[lo + i] += at;      // entry lo itself: after the join
            });
        for (auto &t : th) t.join();
    }
    for (int g = 0; g < G; ++g) if (nids[(size_t)g] < 0 || rc[(size_t)g] != 0) { g_last_error = errs[(size_t)g]; return nids[(size_t)g] < 0 ? nids[(size_t)g] : rc[(size_t)g]; }      // the failing thread's message reaches the caller
    const int64_t total = base[(size_t)G];
    for (int g = 0; g < G; ++g) offs[bounds[(size_t)g]] = base[(size_t)g];
    offs[ndocs] = total;
    if (total > ids_cap) return BF_E_CAPACITY;                    // the offsets are complete even so
    if (total > 0 && !ids_out) return BF_E_ARG;
    return total;
}

int64_t text_batch_host(void *p, const char *text, const int64_t *doc_off, int64_t ndocs, char *text_out, int64_t text_cap, int64_t *text_off_out, int mode)
{
    Handle *h = p ? as_handle(p) : (mode == 2 ? default_sbd() : default_wbd());
    if (!h) return BF_E_ARG;
    if (ndocs < 0 || !doc_off || (ndocs > 0 && !text && doc_off[ndocs] > doc_off[0])) return BF_E_ARG;
    const int64_t base = doc_off[0], total = ndocs > 0 ? doc_off[ndocs] - base : 0;
    if (total < 0) return BF_E_ARG;
    std::lock_guard<std::mutex> dlock(h->defer_mu);      // (the id buffers this call uses may hold a sharded range's ids that wait for their copy out)
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceGuard dg(h->device); if (!dg.ok) return BF_E_DEVICE;
    hipStream_t s = h->stream;
    if (!h->w_text.reserve((size_t)total + 16) || !h->w_docoff.reserve((size_t)(ndocs + 1) * 8) || !h->w_outoff.reserve((size_t)(ndocs + 1) * 8)) return BF_E_DEVICE;
    std::vector<int64_t> rel((size_t)ndocs + 1);
    for (int64_t i = 0; i <= ndocs; ++i) rel[(size_t)i] = doc_off[i] - base;
    if (total > 0 && !hip_ok(hipMemcpyAsync(h->w_text.p, text + base, (size_t)total, hipMemcpyHostToDevice, s), "H2D text")) return BF_E_DEVICE;
    if (!hip_ok(hipMemcpyAsync(h->w_docoff.p, rel.data(), (size_t)(ndocs + 1) * 8, hipMemcpyHostToDevice, s), "H2D offsets")) return BF_E_DEVICE;
    int rc = run_words_device(h, h->w_text.as<char>(), h->w_docoff.as<int64_t>(), ndocs, total, nullptr, 0, h->w_outoff.as<int64_t>(), s, true, mode);
    if (rc != 0) { (void)hipStreamSynchronize(s); return rc; }
    std::vector<int64_t> tmp_off;
    int64_t *dst_off = text_off_out;
    if (!dst_off) { tmp_off.resize((size_t)ndocs + 1); dst_off = tmp_off.data(); }
    if (!hip_ok(hipMemcpyAsync(dst_off, h->w_outoff.p, (size_t)(ndocs + 1) * 8, hipMemcpyDeviceToHost, s), "D2H offsets") ||
        !hip_ok(hipStreamSynchronize(s), "hipStreamSynchronize")) return BF_E_DEVICE;
    const int64_t nout = dst_off[ndocs];
    if (nout > text_cap) return BF_E_CAPACITY;
    if (nout > 0) {
        if (!text_out) return BF_E_ARG;
        if (!h->w_out.reserve((size_t)nout + 16)) return BF_E_DEVICE;
        rc = run_words_device(h, h->w_text.as<char>(), h->w_docoff.as<int64_t>(), ndocs, total, h->w_out.as<char>(), nout, h->w_outoff.as<int64_t>(), s, false, mode);
        if (rc != 0) return rc;
        if (!hip_ok(hipMemcpyAsync(text_out, h->w_out.p, (size_t)nout, hipMemcpyDeviceToHost, s), "D2H text") ||
            !hip_ok(hipStreamSynchronize(s), "hipStreamSynchronize")) return BF_E_DEVICE;
    }
    return nout;
}

} // namespace

extern "C" {

int GetBlingFireTokVersion(void) { return 18000; }

void *LoadModel(const char *path)
{
    g_last_error.clear();
    std::vector<uint8_t> img;
    if (!load_file(path, img)) {
        g_last_error = std::string("cannot read model file: ") + (path ? path : "(null)");
        fprintf(stderr, "[blingfire_amd] %s\n", g_last_error.c_str());
        return nullptr;
    }
    return make_handle(img.data(), img.size());
}

void *SetModel(const unsigned char *img, int size)
{
    g_last_error.clear();
    if (!img || size <= 0) return nullptr;
    return make_handle(img, (size_t)size);
}

int FreeModel(void *p)
{
    Handle *h = as_handle(p);
    if (!h) return 0;
    DeviceGuard dg(h->device);
    (void)hipDeviceSynchronize();
    if (getenv("BF_TRACE_ONE") && h->one_rounds.load() > 0)
        fprintf(stderr, "[blingfire_amd] single-document calls: %lld requests in %lld launches (%.1f per launch), %.1f us per launch\n", h->one_reqs.load(), h->one_rounds.load(),
                (double)h->one_reqs.load() / (double)h->one_rounds.load(), 1e-3 * (double)h->one_ns.load() / (double)h->one_rounds.load());
    delete h;
    return 1;
}

int TextToIds(void *h, const char *s, int n, int32_t *ids, const int max_ids, const int unk) { return text_to_ids_one(h, s, n, ids, max_ids, unk, -1); }
int TextToIds_wp(void *h, const char *s, int n, int32_t *ids, const int max_ids, const int unk) { return text_to_ids_one(h, s, n, ids, max_ids, unk, 0); }
int TextToIds_sp(void *h, const char *s, int n, int32_t *ids, const int max_ids, const int unk) { return text_to_ids_one(h, s, n, ids, max_ids, unk, 1); }

/* reference tokdll:1562-1609, 1108-1118, 1349-1359: ids plus inclusive byte offsets; NULL starts/ends = ids only */
int TextToIdsWithOffsets(void *h, const char *s, int n, int32_t *ids, int *starts, int *ends, const int max_ids, const int unk)
{ return text_to_ids_one(h, s, n, ids, max_ids, unk, -1, starts, ends); }
int TextToIdsWithOffsets_wp(void *h, const char *s, int n, int32_t *ids, int *starts, int *ends, const int max_ids, const int unk)
{ return text_to_ids_one(h, s, n, ids, max_ids, unk, 0, starts, ends); }
int TextToIdsWithOffsets_sp(void *h, const char *s, int n, int32_t *ids, int *starts, int *ends, const int max_ids, const int unk)
{ return text_to_ids_one(h, s, n, ids, max_ids, unk, 1, starts, ends); }

/* ---- TextToWords family (reference tokdll:415-614).  The lexer runs on the GPU in "words" mode (raw tokens, no charmap,
 *      U+0000 fed as U+0020); what remains on the host is the output formatting the reference does with an ostringstream:
 *      copy each word's bytes out of the caller's string (' ' and NUL inside a word -> '_'), join with ' ', terminate with 0. */
int TextToWordsWithOffsetsWithModel(const char *s, int n, char *out, int *starts, int *ends, const int max_out, void *hModel)
{
    Handle *h = hModel ? as_handle(hModel) : default_wbd();
    if (!h || h->m.kind != KIND_WP) return -1;
    if (n == 0) return 0;                                                      // tokdll:447-449
    if (n < 0 || n > 1000000000 || !s) return -1;                              // tokdll:450-455
    if (h->m.lexer_void) return -1;                                            // moore-multi-dfa [wbd]: the reference's lexer answers -1 (tokdll:499-502, bf_model.cpp)
    if (starts && max_out > 0) memset(starts, 0, sizeof(int) * (size_t)max_out);   // tokdll:469-474
    if (ends && max_out > 0) memset(ends, 0, sizeof(int) * (size_t)max_out);
    std::vector<int32_t> tags((size_t)n + 1), ws((size_t)n + 1), we((size_t)n + 1);
    const int64_t off[2] = {0, n};
    int64_t id_off[2] = {0, 0};
    bool nonempty = false;
    const int64_t w = run_host(h, s, off, 1, tags.data(), n + 1, id_off, 0x7fffffff, 0, ws.data(), we.data(), 1, &nonempty);
    if (w < 0) { fprintf(stderr, "[blingfire_amd] TextToWords failed (%lld): %s\n", (long long)w, g_last_error.c_str()); return -1; }
    if (w == 0 && !nonempty) return -1;                                // invalid UTF-8 / nothing decoded (tokdll:477-480)
    std::string os;
    os.reserve((size_t)n + (size_t)w + 1);
    for (int64_t k = 0; k < w; ++k) {
        if (k) os.push_back(' ');
        for (int q = ws[(size_t)k]; q <= we[(size_t)k]; ++q) { const char c = s[q]; os.push_back((c == ' ' || c == 0) ? '_' : c); }
        if (starts && k < max_out) starts[k] = ws[(size_t)k];
        if (ends && k < max_out) ends[k] = we[(size_t)k];
    }
    os.push_back((char)0);                                                     // tokdll:555
    const int len = (int)os.size();
    if (len <= max_out && out) memcpy(out, os.data(), (size_t)len);
    return len;
}
int TextToWordsWithOffsets(const char *s, int n, char *out, int *starts, int *ends, const int max_out)
{ return TextToWordsWithOffsetsWithModel(s, n, out, starts, ends, max_out, nullptr); }
int TextToWordsWithModel(const char *s, int n, char *out, const int max_out, void *hModel)
{ return TextToWordsWithOffsetsWithModel(s, n, out, nullptr, nullptr, max_out, hModel); }
int TextToWords(const char *s, int n, char *out, const int max_out)
{ return TextToWordsWithOffsetsWithModel(s, n, out, nullptr, nullptr, max_out, nullptr); }

/* ---- TextToSentences family (reference tokdll:163-402): the same GPU lexer, every token reported (Tag and From are ignored,
 *      tokdll:262-266): a sentence ends at each token's last character and starts right after the previous one; leading white
 *      space is dropped, '\n' inside a sentence becomes ' ', the rest of the paragraph is the last sentence. */
static bool bf_is_ws(int c)      // blingfiretokdll.h:17-21 __FAIsWhiteSpace__
{
    return c <= 0x20 || c == 0xa0 || (c >= 0x2000 && c <= 0x200f) || c == 0x202f || c == 0x205f || c == 0x2060 || c == 0x2420 || c == 0x2424 ||
           c == 0x3000 || c == 0xfeff;
}
int TextToSentencesWithOffsetsWithModel(const char *s, int n, char *out, int *starts, int *ends, const int max_out, void *hModel)
{
    Handle *h = hModel ? as_handle(hModel) : default_sbd();
    if (!h || h->m.kind != KIND_WP) return -1;
    if (n == 0) return 0;                                                      // tokdll:198-200
    if (n < 0 || n > 1000000000 || !s) return -1;
    if (h->m.lexer_void) return -1;                                            // tokdll:247-250
    if (starts && max_out > 0) memset(starts, 0, sizeof(int) * (size_t)max_out);   // tokdll:220-225
    if (ends && max_out > 0) memset(ends, 0, sizeof(int) * (size_t)max_out);
    std::vector<int32_t> tags((size_t)n + 1), ws((size_t)n + 1), we((size_t)n + 1);
    const int64_t off[2] = {0, n};
    int64_t id_off[2] = {0, 0};
    bool nonempty = false;
    const int64_t w = run_host(h, s, off, 1, tags.data(), n + 1, id_off, 0x7fffffff, 0, ws.data(), we.data(), 2, &nonempty);
    if (w < 0) { fprintf(stderr, "[blingfire_amd] TextToSentences failed (%lld): %s\n", (long long)w, g_last_error.c_str()); return -1; }
    if (w == 0 && !nonempty) return -1;                                // invalid UTF-8 / nothing decoded (tokdll:228-231)
    const unsigned char *u = (const unsigned char *)s;
    std::string os; os.reserve((size_t)n + 1);
    int from = (n >= 3 && u[0] == 0xEF && u[1] == 0xBB && u[2] == 0xBF) ? 3 : 0, sents = 0; bool added = false;
    for (int64_t k = 0; k <= w; ++k) {
        int to;                                                                // last byte of the sentence's last character
        if (k < w) to = we[(size_t)k];
        else { if (!(from < n)) break; to = n - 1; }                           // tokdll:307-311
        int q = from;                                                          // FAGetFirstNonWhiteSpace (tokdll:138-150) on the (valid) UTF-8
        while (q <= to) {
            const unsigned b0 = u[q]; int len = b0 < 0x80 ? 1 : b0 < 0xE0 ? 2 : b0 < 0xF0 ? 3 : 4, cp = b0;
            if (len == 2) cp = ((b0 & 0x1F) << 6) | (u[q + 1] & 0x3F);
            else if (len == 3) cp = ((b0 & 0x0F) << 12) | ((u[q + 1] & 0x3F) << 6) | (u[q + 2] & 0x3F);
            else if (len == 4) cp = ((b0 & 0x07) << 18) | ((u[q + 1] & 0x3F) << 12) | ((u[q + 2] & 0x3F) << 6) | (u[q + 3] & 0x3F);
            if (!bf_is_ws(cp)) break;                                          // (U+0000 counts as U+0020, tokdll:233)
            q += len;
        }
        if (q <= to) {
            if (starts && sents < max_out) starts[sents] = q;
            if (ends && sents < max_out) ends[sents] = to;
            ++sents;
            if (added) os.push_back('\n');
            for (int t = q; t <= to; ++t) { const char c = s[t]; os.push_back(c == '\n' ? ' ' : (c == 0 ? ' ' : c)); }
            if (k < w) added = true;
        }
        from = to + 1;
    }
    os.push_back((char)0);
    const int len = (int)os.size();
    if (len <= max_out && out) memcpy(out, os.data(), (size_t)len);
    return len;
}
int TextToSentencesWithOffsets(const char *s, int n, char *out, int *starts, int *ends, const int max_out)
{ return TextToSentencesWithOffsetsWithModel(s, n, out, starts, ends, max_out, nullptr); }
int TextToSentencesWithModel(const char *s, int n, char *out, const int max_out, void *hModel)
{ return TextToSentencesWithOffsetsWithModel(s, n, out, nullptr, nullptr, max_out, hModel); }
int TextToSentences(const char *s, int n, char *out, const int max_out)
{ return TextToSentencesWithOffsetsWithModel(s, n, out, nullptr, nullptr, max_out, nullptr); }

/* ---- additive: TextToWords for many documents at once; the output string of document d (what TextToWordsWithModel writes,
 *      without the terminating 0) = text_out[text_offsets_out[d] .. text_offsets_out[d+1]) */
int64_t TextToWordsBatch(void *p, const char *text, const int64_t *doc_off, int64_t ndocs, char *text_out, int64_t text_cap, int64_t *text_off_out)
{ return text_batch_host(p, text, doc_off, ndocs, text_out, text_cap, text_off_out, 1); }
int64_t TextToSentencesBatch(void *p, const char *text, const int64_t *doc_off, int64_t ndocs, char *text_out, int64_t text_cap, int64_t *text_off_out)
{ return text_batch_host(p, text, doc_off, ndocs, text_out, text_cap, text_off_out, 2); }

int TextToWordsBatchDevice(void *p, const char *d_text, const int64_t *d_doc_off, int64_t ndocs, int64_t total_bytes, char *d_text_out,
                           int64_t text_cap, int64_t *d_text_off_out, void *stream)
{
    Handle *h = p ? as_handle(p) : default_wbd();
    if (!h) return BF_E_ARG;
    std::lock_guard<std::mutex> dlock(h->defer_mu);      // (the id buffers this call uses may hold a sharded range's ids that wait for their copy out)
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceGuard dg(h->device); if (!dg.ok) return BF_E_DEVICE;
    return run_words_device(h, d_text, d_doc_off, ndocs, total_bytes, d_text_out, text_cap, d_text_off_out, (hipStream_t)stream, true);
}

int TextToSentencesBatchDevice(void *p, const char *d_text, const int64_t *d_doc_off, int64_t ndocs, int64_t total_bytes, char *d_text_out,
                               int64_t text_cap, int64_t *d_text_off_out, void *stream)
{
    Handle *h = p ? as_handle(p) : default_sbd();
    if (!h) return BF_E_ARG;
    std::lock_guard<std::mutex> dlock(h->defer_mu);      // (the id buffers this call uses may hold a sharded range's ids that wait for their copy out)
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceGuard dg(h->device); if (!dg.ok) return BF_E_DEVICE;
    return run_words_device(h, d_text, d_doc_off, ndocs, total_bytes, d_text_out, text_cap, d_text_off_out, (hipStream_t)stream, true, 2);
}

/* ---- NormalizeSpaces (reference tokdll:629-679), model-free; batch of one on the GPU + additive batch forms */
int NormalizeSpaces(const char *s, int n, char *out, const int max_out, const int u_space)
{
    if (n == 0) return -1;                                                     // tokdll:634-636
    if (n < 0 || n > 1000000000 || !s) return -1;
    Handle *h = util_handle();
    if (!h) return -1;
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceGuard dg(h->device); if (!dg.ok) return -1;
    hipStream_t st = h->stream;
    const int64_t off[2] = {0, n};
    if (upload_docs(h, s, off, 1, st) < 0) return -1;
    if (run_normsp_device(h, h->w_text.as<char>(), h->w_docoff.as<int64_t>(), 1, u_space, nullptr, 0, h->w_outoff.as<int64_t>(), st, true) != 0) return -1;
    int32_t len = 0, aux = 0;
    if (!hip_ok(hipMemcpyAsync(&len, h->w_counts.p, 4, hipMemcpyDeviceToHost, st), "D2H") || !hip_ok(hipMemcpyAsync(&aux, h->w_nchars.p, 4, hipMemcpyDeviceToHost, st), "D2H") ||
        !hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize")) return -1;
    if (aux & 1) return -1;                                                    // invalid UTF-8 / nothing decoded (tokdll:646-648)
    unsigned char ub[4];
    if ((aux >> 1) > 0 && utf8_encode(u_space, ub) == 0) return -1;            // a uSpace that cannot be encoded (FAUtf8Utils.cpp:549-552)
    if (len > max_out) return -1;                                              // does not fit: FAArrayToStrUtf8 fails (FAUtf8Utils.cpp:547-552)
    if (len > 0) {
        if (!out || !h->w_out.reserve((size_t)len + 16)) return -1;
        if (run_normsp_device(h, h->w_text.as<char>(), h->w_docoff.as<int64_t>(), 1, u_space, h->w_out.as<char>(), len, h->w_outoff.as<int64_t>(), st, false) != 0) return -1;
        if (!hip_ok(hipMemcpyAsync(out, h->w_out.p, (size_t)len, hipMemcpyDeviceToHost, st), "D2H text") || !hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize")) return -1;
    }
    if (out && len < max_out) out[len] = 0;                                    // tokdll:674-676
    return len;
}

int64_t NormalizeSpacesBatch(const char *text, const int64_t *doc_off, int64_t ndocs, char *text_out, int64_t text_cap, int64_t *text_off_out, int u_space)
{
    Handle *h = util_handle();
    if (!h) return BF_E_DEVICE;
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceGuard dg(h->device); if (!dg.ok) return BF_E_DEVICE;
    hipStream_t s = h->stream;
    const int64_t total = upload_docs(h, text, doc_off, ndocs, s);
    if (total < 0) return total;
    int rc = run_normsp_device(h, h->w_text.as<char>(), h->w_docoff.as<int64_t>(), ndocs, u_space, nullptr, 0, h->w_outoff.as<int64_t>(), s, true);
    if (rc != 0) return rc;
    std::vector<int64_t> tmp_off; int64_t *dst_off = text_off_out;
    if (!dst_off) { tmp_off.resize((size_t)ndocs + 1); dst_off = tmp_off.data(); }
    if (!hip_ok(hipMemcpyAsync(dst_off, h->w_outoff.p, (size_t)(ndocs + 1) * 8, hipMemcpyDeviceToHost, s), "D2H offsets") || !hip_ok(hipStreamSynchronize(s), "hipStreamSynchronize")) return BF_E_DEVICE;
    const int64_t nout = dst_off[ndocs];
    if (nout > text_cap) return BF_E_CAPACITY;
    if (nout > 0) {
        if (!text_out || !h->w_out.reserve((size_t)nout + 16)) return text_out ? BF_E_DEVICE : BF_E_ARG;
        rc = run_normsp_device(h, h->w_text.as<char>(), h->w_docoff.as<int64_t>(), ndocs, u_space, h->w_out.as<char>(), nout, h->w_outoff.as<int64_t>(), s, false);
        if (rc != 0) return rc;
        if (!hip_ok(hipMemcpyAsync(text_out, h->w_out.p, (size_t)nout, hipMemcpyDeviceToHost, s), "D2H text") || !hip_ok(hipStreamSynchronize(s), "hipStreamSynchronize")) return BF_E_DEVICE;
    }
    return nout;
}

/* ---- TextToHashes (reference tokdll:683-815), model-free */
int TextToHashes(const char *s, int n, int32_t *hashes, const int max_hashes, int ngrams, int bucket)
{
    if (ngrams <= 0) return -1;          // the reference only rejects ngrams <= 0 together with a negative length (tokdll:786-789) and would
                                         // otherwise write past the caller's array: refused here
    if (n < 0) return ngrams >= max_hashes ? n * ngrams : 0;                   // one token counted, none hashed (tokdll:718-737,743)
    if (bucket == 0 || !s || n > 1000000000) return -1;
    int tokens = 0;
    Handle *h = util_handle();
    if (!h) return -1;
    std::lock_guard<std::mutex> dlock(h->defer_mu);      // (the id buffers this call uses may hold a sharded range's ids that wait for their copy out)
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceGuard dg(h->device); if (!dg.ok) return -1;
    hipStream_t st = h->stream;
    const int64_t off[2] = {0, n};
    if (upload_docs(h, s, off, 1, st) < 0) return -1;
    if (run_hashes_device(h, h->w_text.as<char>(), h->w_docoff.as<int64_t>(), 1, ngrams, bucket, nullptr, 0, h->w_outoff.as<int64_t>(), st, true) != 0) return -1;
    int32_t cnt = 0;
    if (!hip_ok(hipMemcpyAsync(&cnt, h->w_counts.p, 4, hipMemcpyDeviceToHost, st), "D2H") || !hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize")) return -1;
    tokens = n == 0 ? 0 : cnt / ngrams;                                         // GetTokenCount (tokdll:718-737): 0 for an empty string
    if ((int64_t)tokens * ngrams >= max_hashes) return n * ngrams;             // tokdll:795-798: "requested memory amount"
    if (cnt > max_hashes || !hashes) return -1;                                // (n == 0: one empty token is hashed, tokdll:743-771)
    if (!h->w_ids.reserve((size_t)cnt * 4 + 16)) return -1;
    if (run_hashes_device(h, h->w_text.as<char>(), h->w_docoff.as<int64_t>(), 1, ngrams, bucket, h->w_ids.as<int32_t>(), cnt, h->w_outoff.as<int64_t>(), st, false) != 0) return -1;
    if (!hip_ok(hipMemcpyAsync(hashes, h->w_ids.p, (size_t)cnt * 4, hipMemcpyDeviceToHost, st), "D2H hashes") || !hip_ok(hipStreamSynchronize(st), "hipStreamSynchronize")) return -1;
    return cnt;
}

int64_t TextToHashesBatch(const char *text, const int64_t *doc_off, int64_t ndocs, int32_t *hashes_out, int64_t hashes_cap, int64_t *hash_off_out,
                          int ngrams, int bucket)
{
    if (ngrams <= 0 || bucket == 0) return BF_E_ARG;
    Handle *h = util_handle();
    if (!h) return BF_E_DEVICE;
    std::lock_guard<std::mutex> dlock(h->defer_mu);      // (the id buffers this call uses may hold a sharded range's ids that wait for their copy out)
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceGuard dg(h->device); if (!dg.ok) return BF_E_DEVICE;
    hipStream_t s = h->stream;
    const int64_t total = upload_docs(h, text, doc_off, ndocs, s);
    if (total < 0) return total;
    int rc = run_hashes_device(h, h->w_text.as<char>(), h->w_docoff.as<int64_t>(), ndocs, ngrams, bucket, nullptr, 0, h->w_outoff.as<int64_t>(), s, true);
    if (rc != 0) return rc;
    std::vector<int64_t> tmp_off; int64_t *dst_off = hash_off_out;
    if (!dst_off) { tmp_off.resize((size_t)ndocs + 1); dst_off = tmp_off.data(); }
    if (!hip_ok(hipMemcpyAsync(dst_off, h->w_outoff.p, (size_t)(ndocs + 1) * 8, hipMemcpyDeviceToHost, s), "D2H offsets") || !hip_ok(hipStreamSynchronize(s), "hipStreamSynchronize")) return BF_E_DEVICE;
    const int64_t nout = dst_off[ndocs];
    if (nout > hashes_cap) return BF_E_CAPACITY;
    if (nout > 0) {
        if (!hashes_out || !h->w_ids.reserve((size_t)nout * 4 + 16)) return hashes_out ? BF_E_DEVICE : BF_E_ARG;
        rc = run_hashes_device(h, h->w_text.as<char>(), h->w_docoff.as<int64_t>(), ndocs, ngrams, bucket, h->w_ids.as<int32_t>(), nout, h->w_outoff.as<int64_t>(), s, false);
        if (rc != 0) return rc;
        if (!hip_ok(hipMemcpyAsync(hashes_out, h->w_ids.p, (size_t)nout * 4, hipMemcpyDeviceToHost, s), "D2H hashes") || !hip_ok(hipStreamSynchronize(s), "hipStreamSynchronize")) return BF_E_DEVICE;
    }
    return nout;
}

/* ---- IdsToText (reference tokdll:1689-1745) and its batch forms: a variable-length byte gather on the GPU */
int IdsToText(void *p, const int32_t *ids, const int n, char *out, const int max_out, bool skip_special)
{
    Handle *h = as_handle(p);
    if (!h) return 0;
    if (n == 0 || !ids) return 0;
    if (!h->m.has_i2w || n < 0) return 0;
    const int64_t off[2] = {0, n};
    int64_t toff[2] = {0, 0};
    std::vector<char> tmp;
    bool unknown = false;
    // the text is assembled on the device at full length; what fits is handed to the caller
    int64_t r = run_i2t_host(h, ids, off, 1, nullptr, 0, toff, skip_special ? 1 : 0, &unknown);
    if (r == BF_E_CAPACITY) { tmp.resize((size_t)toff[1]); r = run_i2t_host(h, ids, off, 1, tmp.data(), toff[1], toff, skip_special ? 1 : 0, &unknown); }
    if (r < 0) { fprintf(stderr, "[blingfire_amd] IdsToText failed (%lld): %s\n", (long long)r, g_last_error.c_str()); return 0; }
    if (unknown) return 0;                                                     // unknown id (tokdll:1719-1721)
    const int64_t len = toff[1];
    if (out && max_out > 0 && len > 0) memcpy(out, tmp.data(), (size_t)std::min<int64_t>(len, max_out));
    if (out && max_out > len) out[len] = 0;                                    // tokdll:1737-1739
    return (int)(len + 1);
}

int64_t IdsToTextBatch(void *p, const int32_t *ids, const int64_t *id_offsets, int64_t nseq, char *text_out, int64_t text_cap,
                       int64_t *text_offsets_out, int skip_special)
{
    Handle *h = as_handle(p);
    if (!h) return BF_E_ARG;
    return run_i2t_host(h, ids, id_offsets, nseq, text_out, text_cap, text_offsets_out, skip_special);
}

int IdsToTextBatchDevice(void *p, const int32_t *d_ids, const int64_t *d_id_offsets, int64_t nseq, char *d_text_out, int64_t text_cap,
                         int64_t *d_text_offsets_out, int skip_special, void *stream)
{
    Handle *h = as_handle(p);
    if (!h) return BF_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceGuard dg(h->device); if (!dg.ok) return BF_E_DEVICE;
    hipStream_t s = (hipStream_t)stream;
    if (!hip_ok(hipMemsetAsync(h->w_misc.p, 0, 64, s), "hipMemsetAsync")) return BF_E_DEVICE;
    int rc = run_i2t_device(h, d_ids, d_id_offsets, nseq, nullptr, 0, d_text_offsets_out, skip_special, s);
    if (rc != 0 || !d_text_out) return rc;
    return run_i2t_device(h, d_ids, d_id_offsets, nseq, d_text_out, text_cap, d_text_offsets_out, skip_special, s);
}

/* ---- additive: FADictInterpreter_t<int>::GetInfo for many keys at once over the model's [pos-dict] */
int DictGetInfoBatchDevice(void *p, const int32_t *d_keys, const int64_t *d_key_offsets, int64_t nkeys, int32_t *d_ret_out, int32_t *d_info_ids_out,
                           int32_t *d_values_out, int64_t values_cap, int64_t *d_value_offsets_out, void *stream)
{
    Handle *h = as_handle(p);
    if (!h) return BF_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceGuard dg(h->device); if (!dg.ok) return BF_E_DEVICE;
    return run_dict_device(h, d_keys, d_key_offsets, nkeys, d_ret_out, d_info_ids_out, d_values_out, values_cap, d_value_offsets_out, (hipStream_t)stream, true);
}

int64_t DictGetInfoBatch(void *p, const int32_t *keys, const int64_t *key_offsets, int64_t nkeys, int32_t *ret_out, int32_t *info_ids_out,
                         int32_t *values_out, int64_t values_cap, int64_t *value_offsets_out)
{
    Handle *h = as_handle(p);
    if (!h) return BF_E_ARG;
    if (nkeys < 0 || !key_offsets || (nkeys > 0 && key_offsets[nkeys] > key_offsets[0] && !keys)) return BF_E_ARG;
    const int64_t base = key_offsets[0], total = nkeys > 0 ? key_offsets[nkeys] - base : 0;
    if (total < 0) return BF_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceGuard dg(h->device); if (!dg.ok) return BF_E_DEVICE;
    hipStream_t s = h->stream;
    if (!h->w_keys.reserve((size_t)(total + 1) * 4) || !h->w_keyoff.reserve((size_t)(nkeys + 1) * 8) || !h->w_outoff.reserve((size_t)(nkeys + 1) * 8)) return BF_E_DEVICE;
    std::vector<int64_t> rel((size_t)nkeys + 1);
    for (int64_t i = 0; i <= nkeys; ++i) rel[(size_t)i] = key_offsets[i] - base;
    if (total > 0 && !hip_ok(hipMemcpyAsync(h->w_keys.p, keys + base, (size_t)total * 4, hipMemcpyHostToDevice, s), "H2D keys")) { (void)hipStreamSynchronize(s); return BF_E_DEVICE; }
    if (!hip_ok(hipMemcpyAsync(h->w_keyoff.p, rel.data(), (size_t)(nkeys + 1) * 8, hipMemcpyHostToDevice, s), "H2D offsets")) { (void)hipStreamSynchronize(s); return BF_E_DEVICE; }
    int rc = run_dict_device(h, h->w_keys.as<int32_t>(), h->w_keyoff.as<int64_t>(), nkeys, nullptr, nullptr, nullptr, 0, h->w_outoff.as<int64_t>(), s, true);
    if (rc != 0) { (void)hipStreamSynchronize(s); return rc; }
    std::vector<int64_t> tmp_off; int64_t *dst_off = value_offsets_out;
    if (!dst_off) { tmp_off.resize((size_t)nkeys + 1); dst_off = tmp_off.data(); }
    if (!hip_ok(hipMemcpyAsync(dst_off, h->w_outoff.p, (size_t)(nkeys + 1) * 8, hipMemcpyDeviceToHost, s), "D2H offsets") ||
        (ret_out && nkeys > 0 && !hip_ok(hipMemcpyAsync(ret_out, h->w_dret.p, (size_t)nkeys * 4, hipMemcpyDeviceToHost, s), "D2H ret")) ||
        (info_ids_out && nkeys > 0 && !hip_ok(hipMemcpyAsync(info_ids_out, h->w_dids.p, (size_t)nkeys * 4, hipMemcpyDeviceToHost, s), "D2H ids")) ||
        !hip_ok(hipStreamSynchronize(s), "hipStreamSynchronize")) return BF_E_DEVICE;
    const int64_t nvals = dst_off[nkeys];
    if (nvals > values_cap) return BF_E_CAPACITY;
    if (nvals > 0) {
        if (!values_out || !h->w_vals.reserve((size_t)nvals * 4 + 16)) return values_out ? BF_E_DEVICE : BF_E_ARG;
        rc = run_dict_device(h, h->w_keys.as<int32_t>(), h->w_keyoff.as<int64_t>(), nkeys, nullptr, nullptr, h->w_vals.as<int32_t>(), nvals, h->w_outoff.as<int64_t>(), s, false);
        if (rc != 0) return rc;
        if (!hip_ok(hipMemcpyAsync(values_out, h->w_vals.p, (size_t)nvals * 4, hipMemcpyDeviceToHost, s), "D2H values") || !hip_ok(hipStreamSynchronize(s), "hipStreamSynchronize")) return BF_E_DEVICE;
    }
    return nvals;
}

/* reference tokdll:818-915.  Not on the TextToIds path, no hyphenation engine here: resolves, fails loudly. */
int WordHyphenationWithModel(const char *, int n, char *, const int, void *, const int)
{
    if (n == 0) return 0;                                                      // tokdll:832-834
    static bool warned = false;
    if (!warned) { warned = true; fprintf(stderr, "[blingfire_amd] WordHyphenationWithModel: the hyphenation engine is not part of this library (TextToIds path only); returning -1\n"); }
    g_last_error = "WordHyphenationWithModel is not implemented by this library";
    return -1;
}

int BfReserve(void *p, int64_t max_docs, int64_t max_bytes, int want_offsets)
{
    Handle *h = as_handle(p);
    if (!h || max_docs < 0 || max_bytes < 0) return BF_E_ARG;
    if (h->m.kind == KIND_I2W) return BF_E_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceGuard dg(h->device); if (!dg.ok) return BF_E_DEVICE;
    // both forms of the WordPiece path: the wave program's workspaces and (words = 1 skips use_wave()'s early return) the class stream and flags of
    // the lane-per-document kernels, which TextToWords / TextToSentences, lexers outside the unit form and BfSetVariant(2) run -- no hipMalloc
    // (= device synchronisation) inside a later call of either kind
    if (!reserve_ids_workspaces(h, max_docs, max_bytes, want_offsets != 0)) return BF_E_DEVICE;
    // (not the long-document workspace of the words modes, w_long: 40 .. 56 bytes per cell, allocated by the first words call that is that large)
    if (h->m.kind == KIND_WP && !reserve_ids_workspaces(h, max_docs, max_bytes, true, 1, false)) return BF_E_DEVICE;
    return 0;
}

int SetNoDummyPrefix(void *p, bool flag)          /* reference signature: blingfiretokdll.h:103 */
{
    Handle *h = as_handle(p);
    if (!h) return 0;
    std::lock_guard<std::mutex> lock(h->mu);
    h->m.no_dummy_prefix = flag;
    for (Handle *c : h->shards) if (c && c != h) { std::lock_guard<std::mutex> lc(c->mu); c->m.no_dummy_prefix = flag; }      // every range of a sharded batch sees the same setting
    return 1;
}

int64_t TextToIdsBatch(void *p, const char *text, const int64_t *doc_offsets, int64_t ndocs, int32_t *ids_out, int64_t ids_cap,
                       int64_t *id_offsets_out, int max_ids_per_doc, int unk)
{
    Handle *h = as_handle(p);
    if (!h) return BF_E_ARG;
    std::vector<Handle *> shards;
    { std::lock_guard<std::mutex> lock(h->mu); shards = h->shards; }      // a snapshot: BfSetDevices may replace the list (not while a batch call runs: documented)
    if (shards.size() > 1 || (shards.size() == 1 && shards[0] != h)) return run_host_sharded(h, shards, text, doc_offsets, ndocs, ids_out, ids_cap, id_offsets_out, max_ids_per_doc, unk, nullptr, nullptr);
    return run_host(h, text, doc_offsets, ndocs, ids_out, ids_cap, id_offsets_out, max_ids_per_doc, unk);
}

/* Range-shards the host-buffer batch calls of this handle over n devices: the model's tables are replicated
 * on every listed device, TextToIdsBatch / TextToIdsWithOffsetsBatch split a batch into n contiguous byte-balanced document ranges,
 * one host thread, stream set and workspace per device, and return exactly what one device would have returned.  A device may be
 * listed more than once (logical shards).  n == 1 with the handle's own device ends sharding.  The ...BatchDevice calls and the
 * single-document calls stay on the handle's own device; BfShardHandle(h, g) is the handle of range g for callers that keep their
 * shards resident on the devices themselves.  Returns 0 or BF_E_*. */
int BfSetDevices(void *p, const int *device_ids, int n)
{
    Handle *h = as_handle(p);
    if (!h || !device_ids || n < 1 || n > 64) return BF_E_ARG;
    if (h->m.kind == KIND_I2W) return BF_E_UNSUPPORTED;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) return BF_E_DEVICE;
    for (int i = 0; i < n; ++i) if (device_ids[i] < 0 || device_ids[i] >= ndev) return BF_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    for (Handle *c : h->shards) if (c && c != h) { DeviceGuard dg(c->device); (void)hipDeviceSynchronize(); delete c; }
    h->shards.clear();
    if (n == 1 && device_ids[0] == h->device) return 0;
    std::vector<Handle *> sh;
    for (int i = 0; i < n; ++i) {
        if (i == 0 && device_ids[0] == h->device) { sh.push_back(h); continue; }
        DeviceGuard dg(device_ids[i]);
        Handle *c = dg.ok ? make_handle(h->m.image.data(), h->m.image.size()) : nullptr;
        if (!c) { for (Handle *q : sh) if (q != h) delete q; return BF_E_DEVICE; }
        c->m.no_dummy_prefix = h->m.no_dummy_prefix; c->variant = h->variant; c->host_chunk_bytes = h->host_chunk_bytes;
        sh.push_back(c);
    }
    h->shards.swap(sh);
    return 0;
}

void *BfShardHandle(void *p, int g)
{
    Handle *h = as_handle(p);
    if (!h) return nullptr;
    if (h->shards.empty()) return g == 0 ? (void *)h : nullptr;
    return (g >= 0 && (size_t)g < h->shards.size()) ? (void *)h->shards[(size_t)g] : nullptr;
}

/* the document ranges a batch would be split into over G devices: bounds[0 .. G], range g = [bounds[g], bounds[g + 1]) (pure host arithmetic) */
int BfShardRanges(const int64_t *doc_offsets, int64_t ndocs, int G, int64_t *bounds)
{
    if (!doc_offsets || !bounds || ndocs < 0 || G < 1) return BF_E_ARG;
    shard_bounds(doc_offsets, ndocs, G, bounds);
    return 0;
}

int64_t TextToIdsWithOffsetsBatch(void *p, const char *text, const int64_t *doc_offsets, int64_t ndocs, int32_t *ids_out, int32_t *starts_out,
                                  int32_t *ends_out, int64_t cap, int64_t *id_offsets_out, int max_ids_per_doc, int unk)
{
    Handle *h = as_handle(p);
    if (!h) return BF_E_ARG;
    std::vector<Handle *> shards;
    { std::lock_guard<std::mutex> lock(h->mu); shards = h->shards; }
    if ((shards.size() > 1 || (shards.size() == 1 && shards[0] != h)) && starts_out && ends_out) return run_host_sharded(h, shards, text, doc_offsets, ndocs, ids_out, cap, id_offsets_out, max_ids_per_doc, unk, starts_out, ends_out);
    return run_host(h, text, doc_offsets, ndocs, ids_out, cap, id_offsets_out, max_ids_per_doc, unk, starts_out, ends_out);
}

int TextToIdsWithOffsetsBatchDevice(void *p, const char *d_text, const int64_t *d_doc_offsets, int64_t ndocs, int64_t total_bytes,
                                    int32_t *d_ids_out, int32_t *d_starts_out, int32_t *d_ends_out, int64_t cap, int64_t *d_id_offsets_out,
                                    int max_ids_per_doc, int unk, void *stream)
{
    Handle *h = as_handle(p);
    if (!h) return BF_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceGuard dg(h->device); if (!dg.ok) return BF_E_DEVICE;
    return run_device(h, d_text, d_doc_offsets, ndocs, total_bytes, d_ids_out, cap, d_id_offsets_out, max_ids_per_doc, unk, (hipStream_t)stream,
                      d_starts_out, d_ends_out);
}

int TextToIdsBatchDevice(void *p, const char *d_text, const int64_t *d_doc_offsets, int64_t ndocs, int64_t total_bytes,
                         int32_t *d_ids_out, int64_t ids_cap, int64_t *d_id_offsets_out, int max_ids_per_doc, int unk, void *stream)
{
    Handle *h = as_handle(p);
    if (!h) return BF_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    DeviceGuard dg(h->device); if (!dg.ok) return BF_E_DEVICE;
    return run_device(h, d_text, d_doc_offsets, ndocs, total_bytes, d_ids_out, ids_cap, d_id_offsets_out, max_ids_per_doc, unk, (hipStream_t)stream);
}

int BfLastKernelMs(void *p, float *ms, int n)
{
    Handle *h = as_handle(p);
    if (!h || !ms || n <= 0) return BF_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (!h->ev_valid) return 0;
    if (!hip_ok(hipEventSynchronize(h->ev[EV_COMPACT]), "hipEventSynchronize")) return BF_E_DEVICE;
    float v[6] = {0, 0, 0, 0, 0, 0};
    (void)hipEventElapsedTime(&v[0], h->ev[EV_BEGIN], h->ev[EV_PREP]);
    (void)hipEventElapsedTime(&v[1], h->ev[EV_PREP], h->ev[EV_TOK]);
    (void)hipEventElapsedTime(&v[2], h->ev[EV_TOK], h->ev[EV_SCAN]);
    (void)hipEventElapsedTime(&v[3], h->ev[EV_SCAN], h->ev[EV_COMPACT]);
    (void)hipEventElapsedTime(&v[4], h->ev[EV_BEGIN], h->ev[EV_COMPACT]);
    (void)hipEventElapsedTime(&v[5], h->ev[EV_DOM0], h->ev[EV_DOM1]);
    int k = n < 6 ? n : 6;
    for (int i = 0; i < k; ++i) ms[i] = v[i];
    return k;
}

int64_t BfSetBpePoolBytes(void *p, int64_t bytes)
{
    Handle *h = as_handle(p);
    if (!h || bytes < 0) return BF_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    const int64_t old = (int64_t)h->bpe_pool_bytes;
    h->bpe_pool_bytes = (size_t)bytes;
    for (Handle *c : h->shards) if (c && c != h) { std::lock_guard<std::mutex> lc(c->mu); c->bpe_pool_bytes = (size_t)bytes; }
    return old;
}

int64_t BfSetHostChunkBytes(void *p, int64_t bytes)
{
    Handle *h = as_handle(p);
    if (!h || bytes < 0) return BF_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    const int64_t old = h->host_chunk_bytes;
    h->host_chunk_bytes = bytes;
    for (Handle *c : h->shards) if (c && c != h) { std::lock_guard<std::mutex> lc(c->mu); c->host_chunk_bytes = bytes; }
    return old;
}

int BfLastStatus(void *p)
{
    Handle *h = as_handle(p);
    if (!h) return BF_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    if (h->small_status >= 0) return h->small_status;
    if (h->ev_valid) (void)hipEventSynchronize(h->ev[EV_COMPACT]);
    int status = 0;
    if (!hip_ok(hipMemcpy(&status, h->w_misc.as<char>() + 16, 4, hipMemcpyDeviceToHost), "D2H status")) return BF_E_DEVICE;
    return status;
}

const char *BfLastError(void) { return g_last_error.c_str(); }

/* experiments: instrumentation counters of the lexer kernel (BF_LEX_STATS=1), accumulated since LoadModel */
int BfLexStats(void *p, unsigned long long *out, int n)
{
    Handle *h = as_handle(p);
    if (!h || !out || n <= 0) return BF_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    (void)hipDeviceSynchronize();
    if (n > 16) n = 16;
    if (!hip_ok(hipMemcpy(out, h->w_misc.as<char>() + 64, (size_t)n * 8, hipMemcpyDeviceToHost), "D2H stats")) return BF_E_DEVICE;
    return n;
}

/* diagnostics: documents of the last BPE batch that k_bpe_fused handed to the full path */
long long BfBpeFallbackDocs(void *p)
{
    Handle *h = as_handle(p);
    if (!h || !h->w_hist.p) return BF_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    (void)hipDeviceSynchronize();
    unsigned int v = 0;
    if (!hip_ok(hipMemcpy(&v, h->w_hist.p, 4, hipMemcpyDeviceToHost), "D2H")) return BF_E_DEVICE;
    return (long long)v;
}

int BfModelKind(void *p) { Handle *h = as_handle(p); return h ? h->m.kind : BF_E_ARG; }

int BfSetVariant(void *p, int variant)
{
    Handle *h = as_handle(p);
    if (!h) return BF_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
#ifndef BF_EXPERIMENTS
    // the bits that select a measurement instance (compiled with BF_EXPERIMENTS only): WordPiece -- units configuration (8..11), waves per SIMD of
    // the flat program (16..19), transitions per vote of the lane kernel (20..23); _sp -- waves per SIMD of the prologue (24..27)
    if (variant & (h->m.kind == KIND_WP ? 0x00FF0F00 : 0x0F000000)) return BF_E_UNSUPPORTED;
#endif
    int old = h->variant; h->variant = variant;
    for (Handle *c : h->shards) if (c && c != h) { std::lock_guard<std::mutex> lc(c->mu); c->variant = variant; }
    return old;
}

/* experiments: switches the instrumented kernel instances on / off (what BF_LEX_STATS=1 at LoadModel does) and clears the counters */
int BfSetLexStats(void *p, int on)
{
    Handle *h = as_handle(p);
    if (!h) return BF_E_ARG;
    std::lock_guard<std::mutex> lock(h->mu);
    (void)hipDeviceSynchronize();
    const int old = h->lex_stats ? 1 : 0;
    h->lex_stats = on != 0;
    if (!hip_ok(hipMemset(h->w_misc.as<char>() + 64, 0, 128), "hipMemset(stats)")) return BF_E_DEVICE;
    return old;
}

/* the dominant kernel of a plain TextToIds batch of this model, as the last batch ran it (the one a bench line's roofline is about) */
const char *BfTokeniseKernel(void *p)
{
    Handle *h = as_handle(p);
    if (!h) return "";
    std::lock_guard<std::mutex> lock(h->mu);          // (last_flat / last_uni_cut / variant are written under it by the batch calls)
    switch (h->m.kind) {
    case KIND_WP: return h->last_flat ? "k_wp_flat" : use_wave(h, false, 0) ? "k_wp_wave" : (h->m.two_level ? "k_lex_wp_plain" : "k_lex_wp_flat");
    case KIND_UNIGRAM: return h->last_uni_cut ? "k_uni_cut" : "k_seg_unigram_lane";
    case KIND_I2W: return "";
    default: return use_bpe_wave(h, false) ? "k_bpe_wave" : "k_bpe_fused";
    }
}

/* every kernel of a plain TextToIds step of this model, by the segment of BfLastKernelMs it is timed in */
const char *BfStepKernels(void *p)
{
    Handle *h = as_handle(p);
    if (!h) return "";
    std::lock_guard<std::mutex> lock(h->mu);
    switch (h->m.kind) {
    case KIND_WP:
        if (h->last_flat) return "prep: k_wp_pre | tokenise: k_wp_flat, k_wp_units | scan: k_wp_hardlist, k_wp_wave (the documents handed back), k_wp_count, k_scan_block_sums, k_scan_top, k_scan_apply | compact: k_wp_merge";
        if (use_wave(h, false, 0)) return "prep: - | tokenise: k_wp_wave | scan: k_scan_block_sums, k_scan_top, k_scan_apply | compact: k_compact_ids (offsets: k_compact_text)";
        return "prep: k_prep_wp_flat, k_prep_wp_docs | tokenise: k_lex_wp_plain / k_lex_wp_flat | scan: k_scan_block_sums, k_scan_top, k_scan_apply | compact: k_compact_ids (offsets: k_compact)";
    case KIND_UNIGRAM:
        if (h->last_uni_cut) return "prep: k_prep_sp8 | tokenise: k_sp_hist, k_sp_hist_scan, k_sp_scatter, k_uni_cut | scan: k_scan_block_sums, k_scan_top, k_scan_apply | compact: k_uni_ids";
        return "prep: k_prep_sp8 | tokenise: k_sp_hist, k_sp_hist_scan, k_sp_scatter, k_seg_unigram_lane, k_uni_back | scan: k_scan_block_sums, k_scan_top, k_scan_apply | compact: k_compact_ids";
    case KIND_I2W: return "";
    default:
        if (use_bpe_wave(h, false)) return bpe_wave_home((h->variant >> 8) & 0xf) ? "prep: k_prep_sp8 | tokenise: k_bpe_wave, k_bpe_flag_list, k_bpe_seg | scan: k_scan_block_sums, k_scan_top, k_scan_apply | compact: k_bpe_home_gather"
                                                                              : "prep: k_prep_sp8 | tokenise: k_bpe_wave, k_bpe_flag_list, k_bpe_seg | scan: k_scan_block_sums, k_scan_top, k_scan_apply | compact: k_compact_ids";
        return "prep: k_prep_sp8 | tokenise: k_sp_hist, k_sp_hist_scan, k_sp_scatter, k_bpe_fused, k_bpe_collect_list, k_bpe_sort, k_bpe_apply_flat, k_bpe_seg | scan: k_scan_block_sums, k_scan_top, k_scan_apply | compact: k_compact_ids";
    }
}

} // extern "C"
