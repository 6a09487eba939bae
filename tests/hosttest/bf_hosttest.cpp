// bf_hosttest.cpp -- TEST-ONLY library (never linked into the product).
//
//  * bft_verify_tables: exhaustive (state x symbol) equivalence of the GPU table layout built
//    by bf_model.cpp against the oracle's packed-image readers -- the idea of the reference's
//    `fa_fsm2fsm_pack --auto-test` (blingfirecompile.library/inc/FATestCmpDfa.h:29-54).
//  * bft_emu_*: runs the per-lane device programs of bf_lex.h / bf_seg.h on the host, fed by a
//    scalar restatement of the prep kernels, so that the lane logic can be fuzzed against the
//    oracle on millions of documents without a GPU.
#include <stdint.h>
// lookup-index histogram of the lexer table (design evidence for the LDS-resident prefix, tools/lookup_profile.py)
static unsigned long long g_lookup_hist[4096];
#define BF_LEX_PROFILE_HOOK(idx) (g_lookup_hist[((idx) >> 8) & 4095]++)
static unsigned long long g_uc_stat[8];       // tokens of the Unigram cut form: [0] inline with their key, [1] inline to be walked, [2] through the emission phase
#define BF_UC_STAT(k) (g_uc_stat[(k)]++)
#include "../../blingfire_amd/csrc/bf_model.h"
#include "../../blingfire_amd/csrc/bf_lex.h"
#include "../../blingfire_amd/csrc/bf_seg.h"
#include "../../oracle/bf_oracle.h"

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace bfa;

#include "hosttest.h"

static long g_big_pool = 64l << 20;   // bft_set_big_pool: bytes of the pool behind seg_bpe_doc_big (the device's is 64 MiB per batch)
static int g_uni_cut_w = 0;   // bft_set_uni_cut(W, period): Unigram through the cut form (bf_seg.h UniCut) with a record ring of W positions; emission every `period` steps
static int g_uni_cut_period = 1;
static int g_uni_cut_k = 3;         // bft_set_uni_cut_k: transitions per trip of the driver (the device: 3)
static int g_uni_cut_quick = 1;     // bft_set_uni_cut_quick(0): every chunk through the emission phase
static unsigned long long g_uni_cut_stats[8 + 260];   // [0] documents, [1] given up (hard), [2] chunks, [3] tokens, [4] restarts; [8 + k] documents whose widest pending span (start + depth - cut0, emitted as soon as possible) was k
static int g_uni_seq = 0;     // bft_set_uni_seq(1): Unigram through the plain sequential restatement (seg_unigram_doc) instead of UniLane
static int g_general = 0;     // bft_set_general(1): the general lexer machine even for two-level models (A/B in tests)
static int g_no_ff = 0;       // bft_set_no_ff(1): run the lexer emulation without the loop-state fast-forward (A/B in tests)

// class stream of the host emulation, seen through the same 8-element aligned windows as the device's ClsWin: run() never
// looks past the window that holds position i (nor past the stream, which ends with one unflagged element of padding)
static unsigned long long g_cls_reads = 0;      // class-stream accesses of the lexer emulation (tools: the work of one start position)
struct HostCls {
    const uint16_t *cp; int n;
    uint32_t operator()(int i) const { ++g_cls_reads; return cp[i]; }
    void prefetch(int) const {}
    bool has(int) const { return true; }
    int run(int i) const { ++g_cls_reads; int k = 0; while (i + k < n && ((i + k) >> 3) == (i >> 3) && (cp[i + k] & LX_C_LOOP)) ++k; return k; }
};

#include "../../blingfire_amd/csrc/bf_tolower.h"
extern "C" {

void *bft_load(const char *path)
{
    Handle *h = new Handle();
    h->path = path ? path : "";
    std::vector<uint8_t> img;
    if (!load_file(path, img)) { h->m.error = "cannot read file"; return h; }
    build_model(h->m, img.data(), img.size());
    return h;
}
const char *bft_error(void *hv) { return ((Handle *)hv)->m.error.c_str(); }
// the fold of ignore-case models as the product has it (bf_tolower.h)
int bft_tolower(int cp) { return bfa::bf_tolower(cp); }
int bft_lexer_void(void *hv) { return ((Handle *)hv)->m.lexer_void ? 1 : 0; }
void bft_free(void *hv) { delete (Handle *)hv; }
void bft_set_no_ff(int v) { g_no_ff = v; }
void bft_set_uni_cut_quick(int v) { g_uni_cut_quick = v; }
void bft_set_uni_cut_k(int k) { g_uni_cut_k = k < 1 ? 1 : (k > 4 ? 4 : k); }
void bft_set_uni_cut(int w, int period) { g_uni_cut_w = w; g_uni_cut_period = period > 0 ? period : 1; }
void bft_uni_cut_stats(unsigned long long *out, int reset)
{
    g_uni_cut_stats[6] = g_uc_stat[0]; g_uni_cut_stats[7] = g_uc_stat[1]; g_uni_cut_stats[3] = g_uc_stat[2]; g_uni_cut_stats[0 + 268 - 2] = g_uc_stat[3]; g_uni_cut_stats[268 - 1] = g_uc_stat[4];      // [6] inline with key, [7] inline walked, [3] through the emission phase
    memcpy(out, g_uni_cut_stats, sizeof(g_uni_cut_stats));
    if (reset) { memset(g_uni_cut_stats, 0, sizeof(g_uni_cut_stats)); memset(g_uc_stat, 0, sizeof(g_uc_stat)); }
}
void bft_set_general(int v) { g_general = v; }
int bft_two_level(void *hv) { return ((Handle *)hv)->m.two_level ? 1 : 0; }
int bft_fn_no_ra(void *hv) { return ((Handle *)hv)->m.fn_no_ra ? 1 : 0; }
void bft_set_uni_seq(int v) { g_uni_seq = v; }
void bft_set_big_pool(long v) { g_big_pool = v; }
// loop state facts: out[0] = base (-1: none), [1] = number of flagged classes, [2] = final, [3] = info
void bft_loop_state(void *hv, long *out) { Model &m = ((Handle *)hv)->m; out[0] = m.loop_base == 0xFFFFFFFFu ? -1 : (long)m.loop_base; long n = 0; for (uint8_t b : m.loop_cls) n += b; out[1] = n; out[2] = m.loop_final; out[3] = m.loop_info; }
void bft_lookup_hist(unsigned long long *out, int n, int reset) { for (int i = 0; i < n && i < 4096; ++i) out[i] = g_lookup_hist[i]; if (reset) memset(g_lookup_hist, 0, sizeof(g_lookup_hist)); }
long bft_table_len(void *hv) { return (long)((Handle *)hv)->m.wbd_t2.size(); }
int bft_trie_depth(void *hv) { return ((Handle *)hv)->m.trie_max_depth; }
int bft_kind(void *hv) { return ((Handle *)hv)->m.kind; }

// sizes for reports: out[0]=wbd states, [1]=wbd transitions, [2]=wbd table entries, [3]=wbd classes,
// [4..7] same for dict, [8]=trie depth
void bft_stats(void *hv, long *out)
{
    Model &m = ((Handle *)hv)->m;
    out[0] = (long)m.wbd_raw.state_off.size(); out[1] = (long)m.wbd_raw.tr_sym.size(); out[2] = m.wbd.table_len(); out[3] = m.wbd.nclasses;
    out[4] = (long)m.dict_raw.state_off.size(); out[5] = (long)m.dict_raw.tr_sym.size(); out[6] = m.dict.table_len(); out[7] = m.dict.nclasses;
    out[8] = m.trie_max_depth; out[9] = m.lex_frames;
}

static long verify_dfa(const Model &m, const bfo_model *o, int which, int verbose)
{
    const RawDfa &raw = which ? m.dict_raw : m.wbd_raw;
    const PackedDfa &pk = which ? m.dict : m.wbd;
    long bad = 0;
    auto report = [&](const char *what, int s, int sym, long a, long b) {
        if (++bad <= 10 && verbose) fprintf(stderr, "  [%s] state@%d sym %d: oracle %ld, table %ld\n", what, s, sym, a, b);
    };
    if (raw.state_off[(size_t)raw.initial] != bfo_dfa_initial(o, which)) report("initial", 0, 0, bfo_dfa_initial(o, which), raw.state_off[(size_t)raw.initial]);
    // representative raw symbol per class
    std::vector<int> rep((size_t)pk.nclasses, -1);
    if (raw.remap) {
        for (size_t i = 0; i < raw.iw_from.size(); ++i)
            for (int k = 0; k <= raw.iw_to[i] - raw.iw_from[i]; ++k) {
                int c = raw.iw_cls[i][(size_t)k];
                if (c >= 0 && c < pk.nclasses && rep[(size_t)c] < 0) rep[(size_t)c] = raw.iw_from[i] + k;
            }
    } else {
        for (int c = 0; c < pk.nclasses; ++c) rep[(size_t)c] = pk.sym_of_class[(size_t)c];
    }
    // base -> expected oracle state offset
    std::vector<int> off_of_base(pk.table_len(), -9);
    for (size_t s = 0; s < raw.state_off.size(); ++s) off_of_base[pk.state_base[s]] = raw.state_off[s];
    off_of_base[pk.dead_base] = DFA_DEAD_STATE;
    for (size_t s = 0; s < raw.state_off.size(); ++s) {
        const int soff = raw.state_off[s];
        for (int c = 0; c < pk.nclasses; ++c) {
            if (rep[(size_t)c] < 0) continue;
            int fin = 0, ow = 0;
            long nb = pk.step(pk.state_base[s], (uint32_t)c, &fin, &ow);
            long got = nb < 0 ? -1 : off_of_base[(size_t)nb];
            long want; int want_ow = 0;
            if (which) want = bfo_mealy_dest_ow(o, soff, rep[(size_t)c], &want_ow);
            else want = bfo_dfa_dest(o, which, soff, rep[(size_t)c]);
            if (got != want) { report("dest", soff, rep[(size_t)c], want, got); continue; }
            if (want >= 0) {
                if ((int)bfo_dfa_is_final(o, which, (int)want) != fin) report("final", soff, rep[(size_t)c], bfo_dfa_is_final(o, which, (int)want), fin);
                if (which && want_ow != ow) report("ow", soff, rep[(size_t)c], want_ow, ow);
            }
        }
    }
    return bad;
}

// returns the number of mismatches (0 = equivalent)
long bft_verify_tables(void *hv, int verbose)
{
    Handle *h = (Handle *)hv;
    Model &m = h->m;
    bfo_model *o = bfo_load_model(h->path.c_str());
    if (!o) return -1;
    long bad = 0;
    if (m.has_wbd && bfo_has_dfa(o, 0)) {
        bad += verify_dfa(m, o, 0, verbose);
        // fused code point map == charmap o class map, for every code point
        for (int cp = 0; cp <= 0x10FFFF; ++cp) {
            int norm[10]; int c = bfo_charmap_get(o, 0, cp, norm, 10);
            std::vector<uint32_t> want;
            auto cls = [&](int x) { int k = bfo_wbd_iw_class(o, x < 3 ? 3 : x); if (k >= 0 && !m.wbd_raw.remap) { auto &S = m.wbd.sym_of_class; auto it = std::lower_bound(S.begin(), S.end(), k); k = (it != S.end() && *it == k) ? (int)(it - S.begin()) : -1; } return k < 0 ? CLS_NONE : (uint32_t)k; };
            if (c == -1) want.push_back(cls(cp));
            else if (c >= 1 && c <= 10) for (int k = 0; k < c; ++k) want.push_back(cls(norm[k]));
            uint32_t v = m.wbd_cpmap.get(cp);
            std::vector<uint32_t> got;
            if (v & FUSED_MULTI) { size_t off = v & 0x7fffffffu; int n = m.wbd_multi_pool[off]; for (int k = 0; k < n; ++k) got.push_back(m.wbd_multi_pool[off + 1 + (size_t)k] & ~LX_C_LOOP); }
            else got.push_back(v & ~LX_C_LOOP);
            // the loop flag marks exactly the self-loop classes of the loop state
            if (!(v & FUSED_MULTI)) {
                const uint32_t k = v & T32_CLS_MASK;
                const bool want_flag = k < m.loop_cls.size() && m.loop_cls[k];
                if (want_flag != ((v & LX_C_LOOP) != 0) && ++bad <= 10 && verbose) fprintf(stderr, "  [cpmap] U+%04X loop flag differs\n", cp);
            }
            if (got != want) { if (++bad <= 10 && verbose) fprintf(stderr, "  [cpmap] U+%04X differs\n", cp); }
        }
        // actions of every final state
        for (size_t s = 0; s < m.wbd_raw.state_off.size(); ++s) {
            if (!m.wbd_raw.is_final[s]) continue;
            int act[64]; int n = bfo_wbd_action(o, bfo_state2ow(o, m.wbd_raw.state_off[s]), act, 64);
            uint32_t inf = m.wbd_info[m.wbd.state_base[s]];
            bool ok = n >= 3;
            if (ok) {
                if (inf & INFO_SIMPLE_BIT) ok = n == 3 && act[0] == 0 && act[1] == 0 && act[2] == (int)(inf & 0x7fffffffu);
                else {
                    const int32_t *a = m.acts_pool.data() + inf;
                    int fi = act[2] != 0 ? (n > 3 ? 4 : 3) : 3;
                    ok = a[0] == act[0] && a[1] == act[1] && a[2] == act[2] && a[3] == n - fi;
                    for (int k = 0; ok && k < a[3]; ++k) ok = a[4 + LX_ACT_FN_STRIDE * k] == act[fi + k];
                }
            }
            if (!ok && ++bad <= 10 && verbose) fprintf(stderr, "  [action] state@%d differs\n", m.wbd_raw.state_off[s]);
        }
    }
    if (m.has_seg && bfo_has_dfa(o, 1)) {
        bad += verify_dfa(m, o, 1, verbose);
        for (size_t k = 0; k < m.i2info_id.size(); ++k) {
            int id = 0; uint32_t bits = 0; int c = bfo_i2info_get(o, m.i2info_min_key + (int)k, &id, &bits);
            if (c < 1) { if (m.i2info_valid[k] && ++bad <= 10 && verbose) fprintf(stderr, "  [i2info] key %zu validity\n", k); continue; }
            if ((c > 2 ? 0 : c) != m.i2info_valid[k] || (m.i2info_valid[k] && (id != m.i2info_id[k] || (c >= 2 && bits != m.i2info_score[k]))))
                if (++bad <= 10 && verbose) fprintf(stderr, "  [i2info] key %zu differs\n", k);
        }
        if (m.dict_has_charmap) {
            for (int cp = 0; cp <= 0x10FFFF; ++cp) {
                int norm[10]; int c = bfo_charmap_get(o, 1, cp, norm, 10);
                uint32_t v = m.dict_charmap.get(cp);
                bool ok;
                if (c == -1) ok = v == NORM_NONE;
                else {
                    if (c < 0 || c > 10) c = 0;
                    uint32_t cnt = v >> 24, pay = v & 0xffffffu;
                    if (cnt == 11) { ok = c == 1 && m.dict_norm_pool[pay] == norm[0]; }
                    else if (cnt == 1) ok = c == 1 && (int)pay == norm[0];
                    else { ok = (int)cnt == c && v != NORM_NONE; for (int k = 0; ok && k < c; ++k) ok = m.dict_norm_pool[pay + (size_t)k] == norm[k]; }
                }
                if (!ok && ++bad <= 10 && verbose) fprintf(stderr, "  [charmap] U+%04X differs\n", cp);
            }
        }
    }
    bfo_free_model(o);
    return bad;
}

// ---- host emulation of the GPU pipeline (scalar prep + the device lane programs)
static int emu_wp(const Model &m, const char *s, int n, int32_t *ids, int32_t *spans, std::vector<int> *src_off, int max_ids, int unk)
{
    if (n <= 0 || !s) return 0;
    std::vector<int> cps((size_t)n);
    int len = bfo_utf8_to_utf32(s, n, cps.data(), n);   // the prep KERNEL has its own parallel decoder; GPU tests cover it
    if (len <= 0) return 0;
    std::vector<uint16_t> cls;
    // byte offset of every decoded character (BOM included, like FAStrUtf8ToArray's offsets form)
    std::vector<int> boff; { int p = (n >= 3 && (unsigned char)s[0] == 0xEF && (unsigned char)s[1] == 0xBB && (unsigned char)s[2] == 0xBF) ? 3 : 0;
        for (int i = 0; i < len; ++i) { boff.push_back(p); int c = cps[(size_t)i]; p += c < 0x80 ? 1 : c < 0x800 ? 2 : c < 0x10000 ? 3 : 4; } }
    for (int i = 0; i < len; ++i) {
        uint32_t v = m.wbd_cpmap.get(cps[(size_t)i]);
        if (v & FUSED_MULTI) { size_t off = v & 0x7fffffffu; int c = m.wbd_multi_pool[off]; for (int k = 0; k < c; ++k) { cls.push_back(m.wbd_multi_pool[off + 1 + (size_t)k]); if (src_off) src_off->push_back(boff[(size_t)i]); } }
        else { cls.push_back((uint16_t)v); if (src_off) src_off->push_back(boff[(size_t)i]); }
    }
    if (cls.empty() || (int)cls.size() > n) return 0;
    LexTables L;
    L.T = m.wbd_t2.data(); L.acts = m.acts_pool.data();
    L.initial = m.wbd.initial_base; L.initial_l = m.initial_l; L.cls_any = m.cls_any; L.cls_l = m.cls_l; L.cls_r = m.cls_r;
    L.max_depth = m.max_depth; L.max_token_length = m.max_token_length; L.max_frames = m.lex_frames;
    L.loop_state = g_no_ff ? LX_NO_STATE : m.loop_base; L.loop_info = m.loop_info; L.loop_final = m.loop_final ? 1 : 0; L.two_level = (m.two_level && !g_general) ? 1 : 0; L.fn_no_ra = (m.fn_no_ra && !g_general) ? 1 : 0;
    const int nch = (int)cls.size();
    cls.push_back((uint16_t)CLS_NONE);        // one element of padding: step() reads (and ignores) position InSize under the right anchor
    HostCls cls_at{cls.data(), nch + 1};
    IdOutDirect out{ids, spans};
    FramesArray frames;
    return lex_doc(L, cls_at, nch, out, max_ids, unk, frames);
}

static std::vector<int32_t> *g_arc_dump = nullptr;

// scalar restatement of the _sp prologue on the fused element-code map (the prep KERNEL is wave-parallel; GPU tests cover it):
// the class stream of one document.  false: TextToIds returns 0 for it (tokdll:1409-1411,1440-1444)
bool bft_sp_stream(const Model &m, const char *s, int n, std::vector<uint16_t> &st, std::vector<int> *src_off)
{
    st.clear();
    if (n <= 0 || !s) return false;
    std::vector<int> cps((size_t)n);
    int len;
    if (m.use_bytes) {
        const unsigned char *p = (const unsigned char *)s; int k = 0;
        if (n >= 3 && p[0] == 0xEF && p[1] == 0xBB && p[2] == 0xBF) k = 3;
        len = 0; for (; k < n; ++k) cps[(size_t)len++] = p[k];
    } else len = bfo_utf8_to_utf32(s, n, cps.data(), n);
    if (len <= 0) return false;
    std::vector<uint16_t> el; std::vector<int> eoff;
    if (!m.no_dummy_prefix) { el = m.sp_prefix; eoff.assign(el.size(), -1); }
    {
        int p = (n >= 3 && (unsigned char)s[0] == 0xEF && (unsigned char)s[1] == 0xBB && (unsigned char)s[2] == 0xBF) ? 3 : 0;
        for (int i = 0; i < len; ++i) {
            const int c0 = cps[(size_t)i];
            uint32_t v = m.sp_cpmap.get(c0);
            if (v & FUSED_MULTI) { size_t off = v & 0x7fffffffu; int c = m.sp_multi_pool[off]; for (int k = 0; k < c; ++k) { el.push_back(m.sp_multi_pool[off + 1 + (size_t)k]); eoff.push_back(p); } }
            else { el.push_back((uint16_t)v); eoff.push_back(p); }
            p += m.use_bytes ? 1 : (c0 < 0x80 ? 1 : c0 < 0x800 ? 2 : c0 < 0x10000 ? 3 : 4);
        }
    }
    if (m.dict_has_charmap && (el.empty() || (long)el.size() > 2L * (n + 1))) return false;
    const uint16_t D = m.sp_delim_code;
    for (size_t i = 0; i < el.size(); ++i) {
        const uint16_t e = el[i];
        if (e != SP_WS) { st.push_back(e); if (src_off) src_off->push_back(eoff[i]); }
        else if (i == 0 || !(el[i - 1] == SP_WS || el[i - 1] == D)) { st.push_back(D); if (src_off) src_off->push_back(eoff[i]); }
    }
    if (st.size() > 1 && st.back() == D) st.pop_back();
    return true;
}

static int emu_sp(const Model &m, const char *s, int n, int32_t *ids, int32_t *spans, std::vector<int> *src_off, int max_ids, int unk)
{
    std::vector<uint16_t> st;
    if (!bft_sp_stream(m, s, n, st, src_off)) return 0;
    const uint16_t D = m.sp_delim_code;
    const int L = (int)st.size();
    SegTables S;
    S.T = m.dict.t64.data(); S.info = (const SegInfo *)m.seg_info.data(); S.initial = m.dict.initial_base; S.cls_delim = D;
    S.kind = m.kind; S.id_offset = m.id_offset; S.score = m.seg_score.data(); S.leaf_lo = m.dict.leaf_lo; S.leaf_n = m.dict.leaf_n;
    const uint16_t *cp = st.data();
    auto cls_at = [cp](int i) -> uint32_t { return cp[i]; };
    IdOutDirect out{ids, spans};
    if (m.kind == KIND_UNIGRAM && g_uni_seq) {
        std::vector<SegBest> best((size_t)L + 1);
        return seg_unigram_doc(S, cls_at, L, best.data(), out, max_ids, unk);
    }
    if (m.kind == KIND_UNIGRAM && g_uni_cut_w > 0 && !spans && L > 0) {
        // the cut form (bf_seg.h UniCut): records in a ring of W positions, tokens leave at the cuts; a document that outgrows the ring falls
        // through to the lane program below, like on the device
        struct CutRing {
            std::vector<double> v; std::vector<uint8_t> r; int smask, rmask;
            double score(int pos) const { return v[(size_t)(pos & smask)]; }
            uint32_t rec(int pos) const { return r[(size_t)(pos & rmask)]; }
            void set(int pos, double x, uint32_t rr) { v[(size_t)(pos & smask)] = x; r[(size_t)(pos & rmask)] = (uint8_t)rr; }
            void setrec(int pos, uint32_t rr) { r[(size_t)(pos & rmask)] = (uint8_t)rr; }
            void setscore(int pos, double x) { v[(size_t)(pos & smask)] = x; }
            void fill(double x) { for (auto &e : v) e = x; for (auto &e : r) e = 0x5A; }      // (record slots are not initialised on the device either)
        };
        struct HostSeek { const uint16_t *cp; uint32_t operator()(int i) const { return cp[i]; } void seek(int) const {} void advance(int) const {} };
        int ring_n = 1; while (ring_n < m.trie_max_depth) ring_n <<= 1;
        CutRing ring{std::vector<double>((size_t)ring_n), std::vector<uint8_t>((size_t)g_uni_cut_w), ring_n - 1, g_uni_cut_w - 1};
        HostSeek hs{cp};
        UniCut<HostSeek, CutRing> uc(S, hs, ring);
        std::vector<uint8_t> spill_all((size_t)L + 64, (uint8_t)0xEE);
        uc.init(L, m.trie_max_depth, g_uni_cut_w, spill_all.data() + 32);
        std::vector<uint32_t> toks((size_t)L + 1, 0xDEADBEEFu);
        struct HostPut {
            std::vector<uint32_t> &toks; int align;
            void operator()(int k, uint32_t v) const { toks[(size_t)k] = v; }
            void quad(int k, uint32_t a, uint32_t b, uint32_t c, uint32_t d) const { if (((k + align) & 3) != 0) abort(); toks[(size_t)k] = a; toks[(size_t)k + 1] = b; toks[(size_t)k + 2] = c; toks[(size_t)k + 3] = d; }
        } put{toks, n % 4};
        uc.tok_align = n % 4;                                           // (the document's token array starts at any alignment, like a slot on the device)
        int widest = 0; unsigned long long chunks = 0, restarts = 0, spills = 0;
        // the trip of k_uni_cut: K transitions of the walk (a walk that is over waits), the end of the start position once, the short way out; the
        // emission phase when the document is done, the ring has no room, or every `period` trips
        const int K = g_uni_cut_k;
        for (unsigned long long trip = 1;; ++trip) {
            const int span = uc.i - uc.cut0 + 1;
            if (g_uni_cut_period == 1 && span > widest) widest = span;
            const bool can = uc.room(K);
            bool done = false;
            if (can) {
                for (int u = 0; u < K; ++u) if (uc.walking) uc.step();
                if (!uc.walking) done = uc.finish_start() == UC_DONE;
                if (g_uni_cut_quick) uc.quick(put);
            }
            if (done || !can || trip % (unsigned long long)g_uni_cut_period == 0) {
                if (uc.pending()) { const int before = uc.nout; uc.emit(put); ++chunks; if (uc.nout <= before) ++restarts; }
                else if (!can) { uc.spill(K); ++spills; }
            }
            if (done) break;
            if (trip > 64ull * (unsigned long long)L + 1024ull) return -5;      // the driver does not make progress
        }
        if (spill_all[31] != 0xEE || spill_all[(size_t)L + 32] != 0xEE) return -3;     // a spill left the document's own range
        ++g_uni_cut_stats[0];
        if (spills) ++g_uni_cut_stats[1];
        g_uni_cut_stats[5] += spills;
        {
            g_uni_cut_stats[2] += chunks; g_uni_cut_stats[4] += restarts;
            if (g_uni_cut_period == 1) ++g_uni_cut_stats[8 + (widest < 259 ? widest : 259)];
            const int nout = uc.nout < max_ids ? uc.nout : max_ids;
            for (int k = 0; k < nout; ++k) {                                // what k_uni_ids does
                const uint32_t v = toks[(size_t)k];
                if (v == 0xDEADBEEFu) return -3;                            // a place no chunk wrote
                out.put(k, uni_token_id(S, m.i2info_id.data(), hs, v, unk));
            }
            out.finish(nout);
            return nout;
        }
    }
    if (m.kind == KIND_UNIGRAM) {
        // the default GPU form (bf_seg.h UniLane: score ring + deferred relaxation), driven sequentially
        if (L <= 0) return 0;
        struct HostRing {
            std::vector<double> v; std::vector<uint32_t> r; int mask;
            double score(int pos) const { return v[(size_t)(pos & mask)]; }
            uint32_t rec(int pos) const { return r[(size_t)(pos & mask)]; }
            void set(int pos, double x, uint32_t rr) { v[(size_t)(pos & mask)] = x; r[(size_t)(pos & mask)] = rr; }
            void fill(double x) { for (auto &e : v) e = x; for (auto &e : r) e = UNI_REC_NONE; }
        };
        struct HostSeek { const uint16_t *cp; uint32_t operator()(int i) const { return cp[i]; } void seek(int) const {} void advance(int) const {} };
        int ring_n = 1; while (ring_n < m.trie_max_depth) ring_n <<= 1;
        HostRing ring{std::vector<double>((size_t)ring_n), std::vector<uint32_t>((size_t)ring_n), ring_n - 1};
        HostSeek hs{cp};
        // the record array starts at an arbitrary element of a 64-byte aligned array, like a document slot on the device; the forward pass
        // runs in both forms of the record queue (groups of 4: one kernel; groups of 16: the split form) and must leave the same records
        const int64_t abs0 = (int64_t)(n % 23);
        std::vector<uint32_t> recs_all((size_t)L + 64, 0xDEADBEEFu), recs_all16((size_t)L + 64, 0xDEADBEEFu);
        uint32_t *recs = recs_all.data() + 32;
        UniLane<HostSeek, HostRing> ul(S, hs, ring);
        ul.init(L, m.trie_max_depth, recs, abs0);
        while (ul.wstep()) {}
        {
            HostRing ring16{std::vector<double>((size_t)ring_n), std::vector<uint32_t>((size_t)ring_n), ring_n - 1};
            UniLane<HostSeek, HostRing, 16> ul16(S, hs, ring16);
            ul16.init(L, m.trie_max_depth, recs_all16.data() + 32, abs0);
            while (ul16.wstep()) {}
            if (recs_all16 != recs_all) return -3;
        }
        if (recs_all[31] != 0xDEADBEEFu || recs_all[(size_t)L + 32] != 0xDEADBEEFu) return -3;     // a group store left the document's own range
        ul.begin_back();
        std::vector<int32_t> rid, rfrom, rto;              // ids in backward order, like the device's right-aligned slot
        auto put = [&](int, int id, int from, int to) { rid.push_back(id); rfrom.push_back(from); rto.push_back(to); };
        for (;;) { const uint32_t br = recs[(size_t)ul.end]; if (!ul.bstep(br, put, unk)) break; }
        const int cnt = (int)rid.size(), nout = cnt < max_ids ? cnt : max_ids;
        for (int k = 0; k < nout; ++k) { out.put(k, rid[(size_t)(cnt - 1 - k)]); out.span(k, rfrom[(size_t)(cnt - 1 - k)], rto[(size_t)(cnt - 1 - k)]); }
        out.finish(nout);
        return nout;
    }
    const int cap = 6 * L + 32;
    std::vector<SegArc> arcs((size_t)cap); std::vector<int32_t> tos((size_t)L + 1), idsv((size_t)L + 1); std::vector<uint8_t> inter((size_t)L + 1);
    if (g_arc_dump) {       // tests/test_parallel_formulations.py: the arc list in collection order, before the sort
        const int na = L > 0 ? seg_bpe_collect(S, cls_at, L, arcs.data(), cap, unk) : 0;
        g_arc_dump->assign({L, na, m.kind});
        for (int k = 0; k < na; ++k) { g_arc_dump->push_back(arcs[(size_t)k].start); g_arc_dump->push_back(arcs[(size_t)k].end); g_arc_dump->push_back(arcs[(size_t)k].id); g_arc_dump->push_back((int32_t)arcs[(size_t)k].rank_bits); }
    }
    int r = seg_bpe_doc(S, cls_at, L, arcs.data(), cap, tos.data(), idsv.data(), inter.data(), out, max_ids, unk);
    if (r == -1) {
        // more arcs than the per-document reserve: the pool path of the device (k_bpe_big), with a host pool and canaries around every claim
        struct HostClaim {
            std::vector<std::vector<uint8_t>> blocks; size_t budget;
            uint8_t *operator()(size_t bytes) {
                if (bytes > budget) return nullptr;
                budget -= bytes;
                blocks.emplace_back(bytes + 32, (uint8_t)0xCD);
                return blocks.back().data() + 16;
            }
            bool intact() const { for (auto &b : blocks) for (int k = 0; k < 16; ++k) if (b[(size_t)k] != 0xCD || b[b.size() - 1 - (size_t)k] != 0xCD) return false; return true; }
        } claim{{}, (size_t)g_big_pool};
        r = seg_bpe_doc_big(S, cls_at, L, claim, out, max_ids, unk);
        if (!claim.intact()) return -3;
    }
    return r < 0 ? -2 : r;
}

int bft_emu_sp_doc(const Model &m, const char *s, int n, int32_t *ids, int max_ids, int unk) { return emu_sp(m, s, n, ids, nullptr, nullptr, max_ids, unk); }

// BPE models: ids as bft_emu_text_to_ids + the collected arc list: out = [L, narcs, kind, (start, end, id, rank bits) * narcs]; returns
// the id count (-2: the reference would not terminate), *out_ints = ints written (0 if the buffer is too small)
int bft_emu_bpe_arcs(void *hv, const char *s, int n, int32_t *ids, int max_ids, int unk, int32_t *out, int out_cap, int *out_ints)
{
    Model &m = ((Handle *)hv)->m;
    *out_ints = 0;
    if (!m.error.empty() || m.kind == KIND_WP || m.kind == KIND_UNIGRAM) return -1;
    std::vector<int32_t> dump;
    g_arc_dump = &dump;
    const int r = emu_sp(m, s, n, ids, nullptr, nullptr, max_ids, unk);
    g_arc_dump = nullptr;
    if ((int)dump.size() <= out_cap) { memcpy(out, dump.data(), dump.size() * sizeof(int32_t)); *out_ints = (int)dump.size(); }
    return r;
}

// Structural fuzz of the cut form (bf_seg.h UniCut) against the sequential restatement (seg_unigram_doc, itself pinned to the oracle on the real
// models): small random dictionaries over 2..4 symbols with entries of 1..7 symbols, scores with ties and -- rarely -- a huge POSITIVE value, so
// that the paths no shipped model reaches are walked too: positions without incoming arc that the backward pass LANDS on (the reference emits
// <UnkId, -1, end> and stops: the output restarts), dictionaries without single-symbol entries, spills out of rings as small as depth + 8,
// emission at random times.  Returns the number of (dictionary, text) pairs whose ids differ; *restarts / *spills: how often those paths ran.
int bft_uni_cut_fuzz(unsigned seed, int ndicts, int ntexts, unsigned long long *restarts, unsigned long long *spills)
{
    auto rnd = [&seed]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    int bad = 0; unsigned long long nrestart = 0, nspill = 0;
    for (int dct = 0; dct < ndicts; ++dct) {
        const int ncls = 2 + (int)(rnd() % 3u), npieces = 1 + (int)(rnd() % 12u), maxlen = 1 + (int)(rnd() % 7u);
        // trie: node 0 = root; child[node * ncls + c]
        std::vector<int> child((size_t)ncls, -1); std::vector<char> fin(1, 0);
        for (int k = 0; k < npieces; ++k) {
            int node = 0; const int len = 1 + (int)(rnd() % (unsigned)maxlen);
            for (int j = 0; j < len; ++j) {
                const int c = (int)(rnd() % (unsigned)ncls);
                if (child[(size_t)node * ncls + c] < 0) { child[(size_t)node * ncls + c] = (int)fin.size(); fin.push_back(0); child.resize(fin.size() * (size_t)ncls, -1); }
                node = child[(size_t)node * ncls + c];
            }
            fin[(size_t)node] = 1;
        }
        const int nn = (int)fin.size();
        // keys in preorder (growing along every path), the output weight of a transition = key difference to the nearest final ancestor
        std::vector<int> key((size_t)nn, -1), base((size_t)nn, 0); int nkeys = 0, depth = 0;
        std::vector<uint64_t> T((size_t)nn * ncls + 8, (uint64_t)0xFFFFFull);
        std::vector<int> order; std::vector<int> dep((size_t)nn, 0); { std::vector<int> s2(1, 0); while (!s2.empty()) { const int n = s2.back(); s2.pop_back(); order.push_back(n); for (int c = ncls - 1; c >= 0; --c) { const int ch = child[(size_t)n * ncls + c]; if (ch >= 0) { dep[(size_t)ch] = dep[(size_t)n] + 1; if (dep[(size_t)ch] > depth) depth = dep[(size_t)ch]; s2.push_back(ch); } } } }
        for (int n : order) if (fin[(size_t)n]) key[(size_t)n] = nkeys++;
        std::vector<int> accv((size_t)nn, 0);
        for (int n : order) {
            for (int c = 0; c < ncls; ++c) {
                const int ch = child[(size_t)n * ncls + c];
                if (ch < 0) continue;
                const int ow = fin[(size_t)ch] ? key[(size_t)ch] - accv[(size_t)n] : 0;
                accv[(size_t)ch] = accv[(size_t)n] + ow;
                T[(size_t)n * ncls + c] = (uint64_t)c | (fin[(size_t)ch] ? SG_FINAL : 0ull) | ((uint64_t)((size_t)ch * ncls) << SG_NEXT_SHIFT) | ((uint64_t)ow << SG_OW_SHIFT);
            }
        }
        if (depth < 1) depth = 1;
        std::vector<SegInfo> info((size_t)nkeys + 1); std::vector<uint32_t> score((size_t)nkeys + 1);
        for (int k = 0; k < nkeys; ++k) {
            float sc = -(float)(1 + rnd() % 6u) * 0.5f;                   // few distinct values: ties
            const unsigned r = rnd() % 40u;
            if (r == 0) sc = 3.0e38f; else if (r == 1) sc = -3.0e38f; else if (r == 2) sc = 0.0f;
            union { float f; uint32_t u; } x; x.f = sc;
            info[(size_t)k].id = (int32_t)(rnd() % 1000u); info[(size_t)k].score_bits = x.u; score[(size_t)k] = x.u;
        }
        SegTables S; S.T = T.data(); S.info = info.data(); S.initial = 0; S.cls_delim = 0xFFFDu; S.kind = SG_KIND_UNIGRAM; S.id_offset = (int)(rnd() % 3u) - 1; S.score = score.data();
        for (int t = 0; t < ntexts; ++t) {
            const int L = 1 + (int)(rnd() % 90u);
            std::vector<uint16_t> cls((size_t)L + 1);
            for (int j = 0; j < L; ++j) { const unsigned r = rnd() % 16u; cls[(size_t)j] = r == 0 ? (uint16_t)SG_CLS_NONE : (uint16_t)(rnd() % (unsigned)ncls); }
            const uint16_t *cp = cls.data();
            auto cls_at = [cp](int i) -> uint32_t { return cp[i]; };
            const int max_ids = 1 + (int)(rnd() % 100u), unk = (int)(rnd() % 1000u);
            std::vector<int32_t> want((size_t)max_ids + 1, -7), got((size_t)max_ids + 1, -7);
            std::vector<SegBest> best((size_t)L + 1);
            IdOutDirect ow{want.data(), nullptr};
            const int nw = seg_unigram_doc(S, cls_at, L, best.data(), ow, max_ids, unk);
            struct CutRing {
                std::vector<double> v; std::vector<uint8_t> r; int smask, rmask;
                double score(int pos) const { return v[(size_t)(pos & smask)]; }
                uint32_t rec(int pos) const { return r[(size_t)(pos & rmask)]; }
                void set(int pos, double x, uint32_t rr) { v[(size_t)(pos & smask)] = x; r[(size_t)(pos & rmask)] = (uint8_t)rr; }
                void setrec(int pos, uint32_t rr) { r[(size_t)(pos & rmask)] = (uint8_t)rr; }
                void setscore(int pos, double x) { v[(size_t)(pos & smask)] = x; }
                void fill(double x) { for (auto &e : v) e = x; for (auto &e : r) e = 0x5A; }
            };
            struct HostSeek { const uint16_t *cp; uint32_t operator()(int i) const { return cp[i]; } void seek(int) const {} void advance(int) const {} };
            int ring_n = 1; while (ring_n < depth) ring_n <<= 1;
            int W = 16; while (W < depth + UC_SPILL + 4) W <<= 1;
            if (rnd() % 4u == 0) W <<= 1;
            CutRing ring{std::vector<double>((size_t)ring_n), std::vector<uint8_t>((size_t)W), ring_n - 1, W - 1};
            HostSeek hs{cp};
            UniCut<HostSeek, CutRing> uc(S, hs, ring);
            std::vector<uint8_t> sp((size_t)L + 2, (uint8_t)0xEE);
            uc.init(L, depth, W, sp.data());
            std::vector<uint32_t> toks((size_t)L + 1, 0xDEADBEEFu);
            struct HostPut {
                std::vector<uint32_t> &toks; int align;
                void operator()(int k, uint32_t v) const { toks[(size_t)k] = v; }
                void quad(int k, uint32_t a, uint32_t b, uint32_t c, uint32_t d) const { if (((k + align) & 3) != 0) abort(); toks[(size_t)k] = a; toks[(size_t)k + 1] = b; toks[(size_t)k + 2] = c; toks[(size_t)k + 3] = d; }
            } put{toks, (int)(rnd() % 4u)};
            uc.tok_align = put.align;
            const unsigned period = 1 + rnd() % 40u; const int K = 1 + (int)(rnd() % 4u); const bool use_quick = rnd() % 4u != 0;
            bool hang = false;
            for (unsigned trip = 1;; ++trip) {                            // the trip of k_uni_cut (emu_sp above)
                const bool can = uc.room(K);
                bool done = false;
                if (can) {
                    for (int u = 0; u < K; ++u) if (uc.walking) uc.step();
                    if (!uc.walking) done = uc.finish_start() == UC_DONE;
                    if (use_quick) uc.quick(put);
                }
                if (done || !can || trip % period == 0) {
                    if (uc.pending()) { const int before = uc.nout; uc.emit(put); if (uc.nout <= before) ++nrestart; }
                    else if (!can) { uc.spill(K); ++nspill; }
                }
                if (done) break;
                if (trip > 64u * (unsigned)L + 1024u) { hang = true; break; }
            }
            if (hang) { ++bad; continue; }
            const int ng = uc.nout < max_ids ? uc.nout : max_ids;
            bool ok = ng == nw;
            std::vector<int32_t> idcol((size_t)nkeys + 1);
            for (int k = 0; k < nkeys; ++k) idcol[(size_t)k] = info[(size_t)k].id;
            for (int k = 0; ok && k < ng; ++k) {
                const uint32_t v = toks[(size_t)k];
                if (v == 0xDEADBEEFu) { ok = false; break; }
                got[(size_t)k] = uni_token_id(S, idcol.data(), hs, v, unk);
                ok = got[(size_t)k] == want[(size_t)k];
            }
            if (!ok) ++bad;
        }
    }
    if (restarts) *restarts = nrestart;
    if (spills) *spills = nspill;
    return bad;
}

int bft_emu_text_to_ids(void *hv, const char *s, int n, int32_t *ids, int max_ids, int unk)
{
    Model &m = ((Handle *)hv)->m;
    if (!m.error.empty()) return -1;
    if (m.kind == KIND_WP) return emu_wp(m, s, n, ids, nullptr, nullptr, max_ids, unk);
    return emu_sp(m, s, n, ids, nullptr, nullptr, max_ids, unk);
}

// key -> info id through the device program of bf_seg.h (dict_info_id) on the host, and the row count GetInfo would return
int bft_emu_dict_get_info(void *hv, const int32_t *key, int n, int32_t *info_id, int32_t *vals, int max_vals)
{
    Model &m = ((Handle *)hv)->m;
    if (!m.error.empty() || !m.has_seg || m.k2i.empty()) return -2;
    DictTables D;
    D.T = m.dict.t64.data(); D.initial = m.dict.initial_base; D.initial_final = m.dict_raw.is_final[(size_t)m.dict_raw.initial] ? 1 : 0;
    D.cls_l1 = m.dict_clsmap.l1.data(); D.cls_pages = m.dict_clsmap.pages.data();
    // as bf_capi.cpp run_dict_device / ensure_dict_tables: an ignore-case dictionary normalises through fold + charmap in one map
    const bool nrm = m.dict_ignore_case || (m.dict_direction != 0 && m.dict_has_charmap);
    const TwoLevelMap &nm = m.dict_ignore_case ? m.dict_lookup_map : m.dict_charmap;
    D.nrm_l1 = nrm ? nm.l1.data() : nullptr; D.nrm_pages = nrm ? nm.pages.data() : nullptr; D.nrm_pool = nrm ? m.dict_norm_pool.data() : nullptr;
    D.k2i = m.k2i.data(); D.k2i_n = (int)m.k2i.size(); D.r2l = m.dict_direction != 0; D.ignore_case = m.dict_ignore_case ? 1 : 0;
    const int id = dict_info_id(D, key, n);
    *info_id = id;
    const int nrows = m.info_stride > 0 ? (int)(m.info_rows.size() / (size_t)m.info_stride) : 0;
    if (id == -1 || id < m.info_min_key || id - m.info_min_key >= nrows) return -1;
    const int32_t *row = m.info_rows.data() + (size_t)(id - m.info_min_key) * (size_t)m.info_stride;
    for (int q = 0; q < row[0] && q < max_vals; ++q) vals[q] = row[1 + q];
    return row[0];
}

// TextToWords on the host: words-mode lane program on the unfused class map + the output formatting of the product
int bft_emu_text_to_words(void *hv, const char *s, int n, char *out, int32_t *starts, int32_t *ends, int max_out)
{
    Model &m = ((Handle *)hv)->m;
    if (!m.error.empty() || m.kind != KIND_WP) return -1;
    if (n == 0) return 0;
    if (n < 0 || !s) return -1;
    if (starts && max_out > 0) memset(starts, 0, sizeof(int32_t) * (size_t)max_out);
    if (ends && max_out > 0) memset(ends, 0, sizeof(int32_t) * (size_t)max_out);
    std::vector<int> cps((size_t)n);
    const int len = bfo_utf8_to_utf32(s, n, cps.data(), n);
    if (len <= 0) return -1;
    std::vector<uint16_t> cls; std::vector<int> boff;
    { int p = (n >= 3 && (unsigned char)s[0] == 0xEF && (unsigned char)s[1] == 0xBB && (unsigned char)s[2] == 0xBF) ? 3 : 0;
      for (int i = 0; i < len; ++i) { boff.push_back(p); int c = cps[(size_t)i]; p += c < 0x80 ? 1 : c < 0x800 ? 2 : c < 0x10000 ? 3 : 4; cls.push_back((uint16_t)m.words_cpmap.get(c)); } }
    LexTables L;
    L.T = m.wbd_t2.data(); L.acts = m.acts_pool.data(); L.initial = m.wbd.initial_base; L.initial_l = m.initial_l; L.cls_any = m.cls_any; L.cls_l = m.cls_l; L.cls_r = m.cls_r;
    L.max_depth = m.max_depth; L.max_token_length = m.max_token_length; L.max_frames = m.lex_frames;
    L.loop_state = g_no_ff ? LX_NO_STATE : m.loop_base; L.loop_info = m.loop_info; L.loop_final = m.loop_final ? 1 : 0; L.two_level = (m.two_level && !g_general) ? 1 : 0; L.fn_no_ra = (m.fn_no_ra && !g_general) ? 1 : 0;
    cls.push_back((uint16_t)CLS_NONE);        // padding (see emu_wp)
    HostCls cls_at{cls.data(), len + 1};
    std::vector<int32_t> tags((size_t)len + 1), spans(2 * (size_t)len + 2);
    IdOutDirect o{tags.data(), spans.data()};
    FramesArray frames;
    const int w = lex_doc(L, cls_at, len, o, len, 0, frames, true);
    std::string os;
    for (int k = 0; k < w; ++k) {
        const int f = spans[2 * (size_t)k], t = spans[2 * (size_t)k + 1];
        const int so = boff[(size_t)f], eo0 = boff[(size_t)t];
        const unsigned char b = (unsigned char)s[eo0];
        const int sz = (b & 0x80) == 0 ? 1 : (b & 0xE0) == 0xC0 ? 2 : (b & 0xF0) == 0xE0 ? 3 : (b & 0xF8) == 0xF0 ? 4 : 0;
        const int eo = eo0 + (sz > 0 ? sz - 1 : 0);
        if (k) os.push_back(' ');
        for (int q = so; q <= eo; ++q) os.push_back((s[q] == ' ' || s[q] == 0) ? '_' : s[q]);
        if (starts && k < max_out) starts[k] = so;
        if (ends && k < max_out) ends[k] = eo;
    }
    os.push_back((char)0);
    if ((int)os.size() <= max_out && out) memcpy(out, os.data(), os.size());
    return (int)os.size();
}

// The raw tokens of the words modes (mode 1: TextToWords, 2: TextToSentences) as <tag, first, last> over characters, two ways:
// long_form 0 = the sequential lane program (lex_doc), 1 = the long-document form (bf_lex.h lex_one_start / lex_chain_visit: every
// start position on its own, the chain, the visited positions again).  cap > 0 pretends the reference's triple buffer holds `cap`
// triples instead of n (both forms), to exercise the position at which it fills.  visited (optional) = positions the chain visits.
int bft_emu_lex_tokens(void *hv, const char *s, int n, int mode, int long_form, int cap, int32_t *tags, int32_t *spans, int max_out, int *visited)
{
    Model &m = ((Handle *)hv)->m;
    if (!m.error.empty() || m.kind != KIND_WP || n <= 0 || !s) return -1;
    std::vector<int> cps((size_t)n);
    const int len = bfo_utf8_to_utf32(s, n, cps.data(), n);
    if (len <= 0) return -1;
    std::vector<uint16_t> cls;
    for (int i = 0; i < len; ++i) cls.push_back((uint16_t)m.words_cpmap.get(cps[(size_t)i]));
    LexTables L;
    L.T = m.wbd_t2.data(); L.acts = m.acts_pool.data(); L.initial = m.wbd.initial_base; L.initial_l = m.initial_l; L.cls_any = m.cls_any; L.cls_l = m.cls_l; L.cls_r = m.cls_r;
    L.max_depth = m.max_depth; L.max_token_length = m.max_token_length; L.max_frames = m.lex_frames;
    L.loop_state = g_no_ff ? LX_NO_STATE : m.loop_base; L.loop_info = m.loop_info; L.loop_final = m.loop_final ? 1 : 0; L.two_level = 0; L.fn_no_ra = (m.fn_no_ra && !g_general) ? 1 : 0;
    cls.push_back((uint16_t)CLS_NONE);
    HostCls cls_at{cls.data(), len + 1};
    const int room_all = cap > 0 ? cap : len;
    std::vector<int32_t> tg((size_t)len + 1), sp(2 * (size_t)len + 2);
    FramesArray frames;
    const bool any = L.cls_any != LX_CLS_NONE;
    TabDirect tab{L.T};
    int w = 0;
    if (!long_form) {
        IdOutDirect o{tg.data(), sp.data()};
        auto run = [&](auto has_any) {
            LexLane<HostCls, IdOutDirect, FramesArray, decltype(has_any)::value> lane(L, cls_at, o, frames);
            lane.init(len, 0x7fffffff, 0, mode); lane.max_triples = room_all;
            while (lane.prepare()) { while (lane.step_r()) {} lane.after_walk(); }
            return lane.finish();
        };
        w = any ? run(std::true_type{}) : run(std::false_type{});
    } else {
        auto one = [&](int p0, auto &out, int room) {
            return any ? lex_one_start<true>(L, cls_at, len, p0, out, frames, tab, mode, room) : lex_one_start<false>(L, cls_at, len, p0, out, frames, tab, mode, room);
        };
        std::vector<LexStart> st((size_t)len + 1);
        IdOutNull none;
        for (int p = -1; p < len; ++p) st[(size_t)p + 1] = one(p, none, room_all);
        int pos = -1, ob = 0, eb = 0, nv = 0;
        for (;;) {
            int room; const int here = pos, base = ob;
            const bool more = lex_chain_visit(len, room_all, st[(size_t)pos + 1], pos, ob, eb, room);
            if (visited) visited[nv] = here;
            ++nv;
            IdOutDirect o{tg.data() + base, sp.data() + 2 * (size_t)base};
            const LexStart r = one(here, o, room >= 0 ? room : room_all);
            if (room >= 0) { ob = base + r.n_out; break; }
            if (r.next != st[(size_t)here + 1].next || r.n_out != st[(size_t)here + 1].n_out) return -3;      // the two runs of a position differ
            if (!more) break;
        }
        w = ob;
        if (visited) visited[nv] = -2;
    }
    for (int k = 0; k < w && k < max_out; ++k) { tags[k] = tg[(size_t)k]; spans[2 * k] = sp[2 * (size_t)k]; spans[2 * k + 1] = sp[2 * (size_t)k + 1]; }
    return w;
}

// what every start position of a document does on its own (bf_lex.h lex_one_start): out[4 * (p + 1) ..] = next position, tokens output,
// triples produced, first - p << 16 | last - p of the first token (tools: the shape of a lexer's chain); returns the number of characters
int bft_emu_lex_starts(void *hv, const char *s, int n, int mode, int32_t *out, int max_pos)
{
    Model &m = ((Handle *)hv)->m;
    if (!m.error.empty() || m.kind != KIND_WP || n <= 0 || !s) return -1;
    std::vector<int> cps((size_t)n);
    const int len = bfo_utf8_to_utf32(s, n, cps.data(), n);
    if (len <= 0) return -1;
    std::vector<uint16_t> cls;
    for (int i = 0; i < len; ++i) cls.push_back((uint16_t)m.words_cpmap.get(cps[(size_t)i]));
    LexTables L;
    L.T = m.wbd_t2.data(); L.acts = m.acts_pool.data(); L.initial = m.wbd.initial_base; L.initial_l = m.initial_l; L.cls_any = m.cls_any; L.cls_l = m.cls_l; L.cls_r = m.cls_r;
    L.max_depth = m.max_depth; L.max_token_length = m.max_token_length; L.max_frames = m.lex_frames;
    L.loop_state = m.loop_base; L.loop_info = m.loop_info; L.loop_final = m.loop_final ? 1 : 0; L.two_level = 0; L.fn_no_ra = m.fn_no_ra ? 1 : 0;
    cls.push_back((uint16_t)CLS_NONE);
    HostCls cls_at{cls.data(), len + 1};
    FramesArray frames;
    TabDirect tab{L.T};
    const bool any = L.cls_any != LX_CLS_NONE;
    for (int p = -1; p < len && p + 1 < max_pos; ++p) {
        IdOutFirst o;
        const unsigned long long reads0 = g_cls_reads;
        const int md = mode & 0xff;
        const LexStart r = any ? lex_one_start<true>(L, cls_at, len, p, o, frames, tab, md, len) : lex_one_start<false>(L, cls_at, len, p, o, frames, tab, md, len);
        int32_t *q = out + 4 * (size_t)(p + 1);
        if (mode & 0x100) { q[0] = r.next; q[1] = r.n_out; q[2] = r.n_emit; q[3] = (int32_t)(g_cls_reads - reads0); continue; }      // (the position's class-stream accesses instead of the span)
        q[0] = r.next; q[1] = r.n_out; q[2] = r.n_emit; q[3] = r.n_out > 0 ? (int32_t)(((unsigned)(o.from0 - p) << 16) | (unsigned)((o.to0 - p) & 0xffff)) : 0;
    }
    return len;
}

// offsets form: the lane programs report stream positions, the source-offset stream maps them to bytes and the end
// offset adds the UTF-8 size of the last character (tokdll:1263-1273,1519-1529) -- what k_compact does on the GPU
int bft_emu_text_to_ids_with_offsets(void *hv, const char *s, int n, int32_t *ids, int32_t *starts, int32_t *ends, int max_ids, int unk)
{
    Model &m = ((Handle *)hv)->m;
    if (!m.error.empty()) return -1;
    std::vector<int32_t> spans(2 * (size_t)(max_ids > 0 ? max_ids : 1) + 2, 0);
    std::vector<int> src;
    int c = m.kind == KIND_WP ? emu_wp(m, s, n, ids, spans.data(), &src, max_ids, unk) : emu_sp(m, s, n, ids, spans.data(), &src, max_ids, unk);
    for (int k = 0; k < c; ++k) {
        const int f = spans[2 * (size_t)k], t = spans[2 * (size_t)k + 1];
        const int so = src[(size_t)f], eo = src[(size_t)t];
        int sz = 0;
        if (eo >= 0) { const unsigned char b = (unsigned char)s[eo]; sz = (b & 0x80) == 0 ? 1 : (b & 0xE0) == 0xC0 ? 2 : (b & 0xF0) == 0xE0 ? 3 : (b & 0xF8) == 0xF0 ? 4 : 0; }
        starts[k] = so; ends[k] = eo + (sz > 0 ? sz - 1 : 0);
    }
    return c;
}

} // extern "C"
