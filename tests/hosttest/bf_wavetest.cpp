// bf_wavetest.cpp -- TEST-ONLY (part of libbf_hosttest.so, never linked into the product).
//
// Runs the wave kernel of bf_wave.h -- the very source the GPU executes -- inside the 64-fibre wave simulator of wave_emu.h,
// followed by a scalar restatement of the scan + compaction kernels, so that the whole WordPiece batch path of a unit-form model
// can be fuzzed against the oracle without a GPU.
#include "wave_emu.h"
#include "../../blingfire_amd/csrc/bf_wave_body.h"
#include "../../blingfire_amd/csrc/bf_bpe_wave_body.h"
#include "../../blingfire_amd/csrc/bf_bpe_seg_body.h"
#include "../../blingfire_amd/csrc/bf_uni_walk_body.h"
#include "hosttest.h"

#include <vector>
#include <functional>
#include <algorithm>

using namespace bfa;

template <class LDS, int NU>
static void run_cfg(const WpWaveParams &p, int nwaves, int grab)
{
    std::vector<LDS *> lds;
    std::vector<uint16_t> ascii(128);
    for (int i = 0; i < 128; ++i) ascii[(size_t)i] = (uint16_t)wv_element(p.cold, i);
    int next_wave = 0;
    std::vector<LDS *> of_wave((size_t)nwaves);
    for (int i = 0; i < nwaves; ++i) { of_wave[(size_t)i] = new LDS(); memset((void *)of_wave[(size_t)i], 0xA5, sizeof(LDS)); }
    // every fibre of a wave must see the same LDS block: the wave object's address identifies the wave
    std::vector<const void *> wave_ids;
    auto body = [&]() {
        const void *wid = (const void *)wvemu::g_cur->wave;
        size_t k = 0;
        for (; k < wave_ids.size(); ++k) if (wave_ids[k] == wid) break;
        if (k == wave_ids.size()) { wave_ids.push_back(wid); (void)next_wave; }
        WpWave<LDS, NU, true> w(p, p.cold, *of_wave[k], ascii.data(), p.acts);
        w.run(grab, (int)k, nwaves);
    };
    wvemu::run_waves(nwaves, body);
    for (auto *q : of_wave) delete q;
}

extern "C" {

int bft_wave_ok(void *hv) { return ((Handle *)hv)->m.wave_ok ? 1 : 0; }
const char *bft_wave_why(void *hv) { return ((Handle *)hv)->m.wave_why.c_str(); }
int bft_bpe_wave_ok(void *hv) { return ((Handle *)hv)->m.bpe_wave_ok ? 1 : 0; }

// TextToIdsBatch through the wave kernel on the host.  cfg: 0 = the shipped configuration, 1 = two units per lane, the smallest ring and queue, a two-entry
// document table, every token with an explicit action, 2 = three units per lane, a large ring and queue.  Returns the total id count, or < 0 (-1: model not in unit form, -5: the kernel raised a status bit).
// stats (optional, 16 counters): see bf_wave.h WpWaveParams::stats.
long bft_emu_wave_batch(void *hv, const uint8_t *text, long text_bytes, const int64_t *doc_off, long ndocs, int max_ids, int unk, int nwaves, int grab, int cfg,
                        int32_t *ids_out, long ids_cap, int64_t *id_off, unsigned long long *stats)
{
    Model &m = ((Handle *)hv)->m;
    if (!m.error.empty() || m.kind != KIND_WP || !m.wave_ok) return -1;
    if (max_ids < 0) max_ids = 0;
    const int64_t total = text_bytes;                              // what the caller's buffer really holds
    std::vector<int32_t> tmp((size_t)(total + 8 * ndocs + 64 + 8), -77), counts((size_t)ndocs + 1, -55);
    unsigned long long next_doc = 0; int status = 0;
    WpWaveParams p;
    p.T = m.wbd_t2.data(); p.acts = m.acts_pool.data(); p.acts_n = (int)m.acts_pool.size();
    p.initial = m.wbd.initial_base; p.loop_info = m.loop_info; p.solo_info = m.wave_solo_info; p.max_token_length = m.max_token_length;
    p.text = text; p.doc_off = doc_off; p.ndocs = ndocs; p.total_bytes = total;
    p.ids_tmp = tmp.data(); p.counts = counts.data(); p.max_ids = max_ids; p.unk = unk; p.next_doc = &next_doc;
    if (cfg >= 16) { p.next_doc = nullptr; cfg -= 16; }            // cfg + 16: no work counter, the waves take their ranges round-robin
    p.cold.cpmap = DevCpMap{m.wbd_cpmap.l1.data(), m.wbd_cpmap.pages.data()};
    p.cold.kind = m.wave_kind.data(); p.cold.nclasses = m.wbd.nclasses; p.cold.status = &status; p.cold.stats = stats; p.cold.no_fast = cfg == 1 ? 1 : 0;
    if (ndocs > 0) {
        if (cfg == 1) run_cfg<WvLds<1024, 128, 2>, 2>(p, nwaves, grab);
        else if (cfg == 2) run_cfg<WvLds<4096, 512, 64>, 3>(p, nwaves, grab);
        else run_cfg<WvLds<1024, 256, 8>, 1>(p, nwaves, grab);
    }
    if (status) return -5;
    // k_scan + k_compact, restated
    long o = 0;
    for (long d = 0; d < ndocs; ++d) {
        id_off[d] = o;
        const int c = counts[(size_t)d];
        if (c < 0) return -6;                                // a document nobody wrote a count for
        if (c == 0) continue;
        const int64_t slot = wv_ids_slot(doc_off[d], d);
        for (int i = 0; i < c; ++i) { if (o + i < ids_cap) ids_out[o + i] = tmp[(size_t)(slot + i)]; }
        o += c;
    }
    id_off[ndocs] = o;
    return o;
}

// The same for a bpe-opt model (bf_bpe_wave_body.h): the _sp prologue restated per document (bft_sp_stream) lays the class streams out as
// the prologue kernel does (slot of document d = mul * (doc_off[d] + d)), the BPE wave program runs in the simulator, the documents it
// hands back (flags) are redone by the sequential restatement of the lane-per-document path, then scan + compaction.  cfg: 0 = ring 1,024 /
// queue 256 / 8 open documents, 1 = queue 128 / 2 open documents; + 16 = no work counter.  flags_out[d] = 1: document d was handed back.
// Returns the total id count, -1: the model is not eligible, -5: the kernel raised a status bit.
long bft_emu_bpe_wave_batch(void *hv, const uint8_t *text, long text_bytes, const int64_t *doc_off, long ndocs, int max_ids, int unk, int nwaves, int grab, int cfg,
                            int32_t *ids_out, long ids_cap, int64_t *id_off, int32_t *flags_out, unsigned long long *stats)
{
    Model &m = ((Handle *)hv)->m;
    if (!m.error.empty() || !m.bpe_wave_ok) return -1;
    if (max_ids < 0) max_ids = 0;
    (void)text_bytes;
    const int mul = m.dict_has_charmap ? 2 : 1;
    const int64_t total = ndocs > 0 ? doc_off[ndocs] : 0;
    const size_t cells = (size_t)(mul * (total + ndocs + 1) + 64);
    std::vector<uint16_t> stream(cells, (uint16_t)0xEEEE); std::vector<int32_t> tmp(cells, -77), lens((size_t)ndocs + 1, 0), counts((size_t)ndocs + 1, -55), flags((size_t)ndocs + 1, -55);
    std::vector<uint16_t> st;
    for (long d = 0; d < ndocs; ++d) {
        const int n = (int)(doc_off[d + 1] - doc_off[d]);
        if (!bft_sp_stream(m, (const char *)text + doc_off[d], n, st, nullptr)) { lens[(size_t)d] = 0; continue; }
        lens[(size_t)d] = (int32_t)st.size();
        if ((int64_t)st.size() > (int64_t)mul * (n + 1)) return -6;
        memcpy(stream.data() + (size_t)mul * (size_t)(doc_off[d] + d), st.data(), st.size() * 2);
    }
    unsigned long long next_doc = 0; int status = 0;
    std::vector<uint32_t> scratch(6 * cells, 0xABABABABu);
    BpeWaveParams p;
    p.T = m.dict.t64.data(); p.info = (const SegInfo *)m.seg_info.data(); p.initial = m.dict.initial_base; p.cls_delim = m.sp_delim_code; p.id_offset = m.id_offset;
    p.prio = m.kind == KIND_BPE_MERGES ? m.bpe_prio.data() : nullptr; p.place_id = m.kind == KIND_BPE_MERGES ? m.bpe_place_id.data() : nullptr;
    p.stream = stream.data(); p.lens = lens.data(); p.doc_off = doc_off; p.slot_mul = mul; p.ndocs = ndocs;
    p.ids_tmp = tmp.data(); p.counts = counts.data(); p.flags = flags.data(); p.max_ids = max_ids; p.next_doc = &next_doc; p.status = &status; p.stats = stats; p.scratch = scratch.data();
    if (cfg >= 16) { p.next_doc = nullptr; cfg -= 16; }
    if (ndocs > 0) {
        auto run = [&](auto *lds_tag) {
            typedef typename std::remove_pointer<decltype(lds_tag)>::type LDS;
            std::vector<LDS *> of_wave((size_t)nwaves);
            for (int i = 0; i < nwaves; ++i) { of_wave[(size_t)i] = new LDS(); memset((void *)of_wave[(size_t)i], 0xA5, sizeof(LDS)); }
            std::vector<const void *> wave_ids;
            auto body = [&]() {
                const void *wid = (const void *)wvemu::g_cur->wave;
                size_t k = 0;
                for (; k < wave_ids.size(); ++k) if (wave_ids[k] == wid) break;
                if (k == wave_ids.size()) wave_ids.push_back(wid);
                BpeWave<LDS> w(p, *of_wave[k]);
                w.run(grab, (int)k, nwaves);
            };
            wvemu::run_waves(nwaves, body);
            for (auto *q : of_wave) delete q;
        };
        if (cfg == 1) run((BwLds<1024, 128, 2> *)nullptr); else run((BwLds<1024, 256, 8> *)nullptr);
    }
    if (status) return -5;
    long o = 0;
    std::vector<int32_t> one((size_t)(max_ids > 0 ? max_ids : 1));
    for (long d = 0; d < ndocs; ++d) {
        id_off[d] = o;
        if (counts[(size_t)d] < 0 || flags[(size_t)d] < 0) return -7;                 // a document the kernel never settled
        if (flags_out) flags_out[d] = flags[(size_t)d];
        int c = counts[(size_t)d];
        const int32_t *src = tmp.data() + (size_t)mul * (size_t)(doc_off[d] + d);
        if (flags[(size_t)d]) {                                                        // handed back: the lane-per-document path
            c = bft_emu_sp_doc(m, (const char *)text + doc_off[d], (int)(doc_off[d + 1] - doc_off[d]), one.data(), max_ids, unk);
            if (c < 0) return -8;
            src = one.data();
        }
        if (o + c > ids_cap) return -9;
        for (int k = 0; k < c; ++k) ids_out[o + k] = src[k];
        o += c;
    }
    id_off[ndocs] = o;
    return o;
}


int bft_bpe_seg_ok(void *hv) { return ((Handle *)hv)->m.bpe_seg_ok ? 1 : 0; }

// Every document of the batch through the one-wave-per-document BPE program (bf_bpe_seg_body.h) in the simulator: prologue restated per
// document, the program, scan + compaction restated.  pool_bytes: the pool the documents claim their blocks from.  spans_out (optional):
// [first, last] stream position of every id.  Returns the total id count; -1: model not eligible; -5: a status bit other than
// BF_STATUS_DOC_FAILED / BF_STATUS_POOL was raised; *status_out = the status word; *pool_used_out = bytes claimed + bytes of the claims that did not fit (a pool of that size holds the batch).
long bft_emu_bpe_seg_batch(void *hv, const uint8_t *text, const int64_t *doc_off, long ndocs, int max_ids, int unk, int nwaves, long pool_bytes,
                           int32_t *ids_out, long ids_cap, int64_t *id_off, int32_t *spans_out, int *status_out, unsigned long long *pool_used_out, unsigned long long *stats)
{
    Model &m = ((Handle *)hv)->m;
    if (!m.error.empty() || !m.bpe_seg_ok) return -1;
    if (max_ids < 0) max_ids = 0;
    const int mul = m.dict_has_charmap ? 2 : 1;
    const int64_t total = ndocs > 0 ? doc_off[ndocs] : 0;
    const size_t cells = (size_t)(mul * (total + ndocs + 1) + 64);
    std::vector<uint16_t> stream(cells, (uint16_t)0xEEEE); std::vector<int32_t> tmp(cells, -77), spans(2 * cells, -77), lens((size_t)ndocs + 1, 0), counts((size_t)ndocs + 1, -55);
    std::vector<uint16_t> st;
    for (long d = 0; d < ndocs; ++d) {
        const int n = (int)(doc_off[d + 1] - doc_off[d]);
        if (!bft_sp_stream(m, (const char *)text + doc_off[d], n, st, nullptr)) { lens[(size_t)d] = 0; continue; }
        lens[(size_t)d] = (int32_t)st.size();
        if ((int64_t)st.size() > (int64_t)mul * (n + 1)) return -6;
        memcpy(stream.data() + (size_t)mul * (size_t)(doc_off[d] + d), st.data(), st.size() * 2);
    }
    unsigned long long next_doc = 0, pool_used = 0, pool_need = 0; int status = 0;
    std::vector<uint8_t> pool((size_t)pool_bytes + 64, (uint8_t)0xCD);
    BpeSegParams p;
    p.T = m.dict.t64.data(); p.info = (const SegInfo *)m.seg_info.data(); p.initial = m.dict.initial_base; p.cls_delim = m.sp_delim_code; p.id_offset = m.id_offset; p.kind = m.kind;
    p.prio = m.kind == KIND_BPE_MERGES ? m.bpe_prio.data() : nullptr; p.place_id = m.kind == KIND_BPE_MERGES ? m.bpe_place_id.data() : nullptr;
    p.unk_prio = bpe_unk_prio(m, unk); p.prio_bits = m.bpe_prio_bits;
    p.stream = stream.data(); p.lens = lens.data(); p.doc_off = doc_off; p.slot_mul = mul;
    p.list = nullptr; p.list_n = nullptr; p.narcs = nullptr; p.narcs_want = 0; p.ndocs = ndocs;
    p.ids_tmp = tmp.data(); p.span_tmp = spans_out ? spans.data() : nullptr; p.counts = counts.data(); p.max_ids = max_ids; p.unk = unk;
    p.next_doc = &next_doc; p.status = &status;
    p.pool = pool.data(); p.pool_bytes = (unsigned long long)pool_bytes; p.pool_used = &pool_used; p.pool_need = &pool_need; p.stats = stats;
    if (ndocs > 0) {
        std::vector<BsLds *> of_wave((size_t)nwaves);
        for (int i = 0; i < nwaves; ++i) { of_wave[(size_t)i] = new BsLds(); memset((void *)of_wave[(size_t)i], 0xA5, sizeof(BsLds)); }
        std::vector<const void *> wave_ids;
        auto body = [&]() {
            const void *wid = (const void *)wvemu::g_cur->wave;
            size_t k = 0;
            for (; k < wave_ids.size(); ++k) if (wave_ids[k] == wid) break;
            if (k == wave_ids.size()) wave_ids.push_back(wid);
            BpeSeg<BsLds> w(p, *of_wave[k]);
            w.run();
        };
        wvemu::run_waves(nwaves, body);
        for (auto *q : of_wave) delete q;
    }
    for (size_t k = 0; k < 64; ++k) if (pool[(size_t)pool_bytes + k] != 0xCD) return -10;       // a block that left the pool
    if (status_out) *status_out = status;
    if (pool_used_out) *pool_used_out = pool_used + pool_need;
    if (status & ~(BF_STATUS_DOC_FAILED | BF_STATUS_POOL)) return -5;
    long o = 0;
    for (long d = 0; d < ndocs; ++d) {
        id_off[d] = o;
        const int c = counts[(size_t)d];
        if (c < 0) return -7;
        const size_t slot = (size_t)mul * (size_t)(doc_off[d] + d);
        if (o + c > ids_cap) return -9;
        for (int k = 0; k < c; ++k) { ids_out[o + k] = tmp[slot + (size_t)k]; if (spans_out) { spans_out[2 * (o + k)] = spans[2 * (slot + (size_t)k)]; spans_out[2 * (o + k) + 1] = spans[2 * (slot + (size_t)k) + 1]; } }
        o += c;
    }
    id_off[ndocs] = o;
    return o;
}


// Unigram-LM through the two-stage device path on the host: the walks of bf_uni_walk_body.h in the wave simulator (arc records + round
// table), then bf_seg.h UniArcLane per document, driven sequentially (what k_uni_dp runs per lane), backward pass, scan + compaction
// restated.  pool_recs: records the pool holds (documents that do not fit are flagged and redone by the sequential restatement of the
// lane-per-document path, as the device does).  rows: 16 or 32 (entries per start the stage holds).  flags_out[d] = 1: redone.
// Returns the total id count; -1: the model is not eligible (not Unigram, entries longer than `rows`, ids >= 2^20 - 2).
long bft_emu_uni_walk_batch(void *hv, const uint8_t *text, const int64_t *doc_off, long ndocs, int max_ids, int unk, int nwaves, long pool_recs, int rows,
                            int32_t *ids_out, long ids_cap, int64_t *id_off, int32_t *flags_out, unsigned long long *stats)
{
    Model &m = ((Handle *)hv)->m;
    if (!m.error.empty() || m.kind != KIND_UNIGRAM || m.trie_max_depth <= 0 || m.trie_max_depth > UA_MAX_DEPTH || (rows < 100 && m.trie_max_depth > rows) || m.max_info_id > UNI_MAX_ID) return -1;
    if (max_ids < 0) max_ids = 0;
    const int mul = m.dict_has_charmap ? 2 : 1;
    const int64_t total = ndocs > 0 ? doc_off[ndocs] : 0;
    const size_t cells = (size_t)(mul * (total + ndocs + 1) + 64);
    std::vector<uint16_t> stream(cells, (uint16_t)0xEEEE); std::vector<int32_t> lens((size_t)ndocs + 1, 0), flags((size_t)ndocs + 1, -55);
    std::vector<uint16_t> st;
    for (long d = 0; d < ndocs; ++d) {
        const int n = (int)(doc_off[d + 1] - doc_off[d]);
        if (!bft_sp_stream(m, (const char *)text + doc_off[d], n, st, nullptr)) { lens[(size_t)d] = 0; continue; }
        lens[(size_t)d] = (int32_t)st.size();
        memcpy(stream.data() + (size_t)mul * (size_t)(doc_off[d] + d), st.data(), st.size() * 2);
    }
    unsigned long long next_doc = 0, cursor = 0;
    std::vector<uint64_t> pool((size_t)pool_recs + 8, 0xCDCDCDCDCDCDCDCDull), rounds((cells >> 6) + (size_t)ndocs + 8, ~0ull);
    UniWalkParams p;
    p.T = m.dict.t64.data(); p.info = (const SegInfo *)m.seg_info.data(); p.initial = m.dict.initial_base;
    p.stream = stream.data(); p.lens = lens.data(); p.doc_off = doc_off; p.slot_mul = mul; p.ndocs = ndocs; p.perm = nullptr;
    p.pool = pool.data(); p.pool_recs = (unsigned long long)pool_recs; p.pool_cursor = &cursor; p.rounds = rounds.data(); p.flags = flags.data();
    p.next_doc = &next_doc; p.stats = stats;
    if (ndocs > 0) {
        auto run = [&](auto *tag, auto rows_c, auto ns_c) {
            typedef typename std::remove_pointer<decltype(tag)>::type LDS;
            std::vector<LDS *> of_wave((size_t)nwaves);
            for (int i = 0; i < nwaves; ++i) { of_wave[(size_t)i] = new LDS(); memset((void *)of_wave[(size_t)i], 0xA5, sizeof(LDS)); }
            std::vector<const void *> wave_ids;
            auto body = [&]() {
                const void *wid = (const void *)wvemu::g_cur->wave;
                size_t k = 0;
                for (; k < wave_ids.size(); ++k) if (wave_ids[k] == wid) break;
                if (k == wave_ids.size()) wave_ids.push_back(wid);
                UniWalk<LDS, decltype(rows_c)::value, decltype(ns_c)::value> w(p, *of_wave[k]);
                w.run();
            };
            wvemu::run_waves(nwaves, body);
            for (auto *q : of_wave) delete q;
        };
        // rows + 100 * starts per lane: the device instances (16 x 2, 8 x 4, 32 x 1) and one that overflows its stage often (2 x 3)
        if (rows == 216) run((UwLds<16, 2> *)nullptr, std::integral_constant<int, 16>(), std::integral_constant<int, 2>());
        else if (rows == 408) run((UwLds<8, 4> *)nullptr, std::integral_constant<int, 8>(), std::integral_constant<int, 4>());
        else if (rows == 302) run((UwLds<2, 3> *)nullptr, std::integral_constant<int, 2>(), std::integral_constant<int, 3>());
        else if (rows <= 16) run((UwLds<16, 1> *)nullptr, std::integral_constant<int, 16>(), std::integral_constant<int, 1>());
        else run((UwLds<32, 1> *)nullptr, std::integral_constant<int, 32>(), std::integral_constant<int, 1>());
    }
    for (size_t k = 0; k < 8; ++k) if (pool[(size_t)pool_recs + k] != 0xCDCDCDCDCDCDCDCDull) return -10;
    // ---- the relaxations per document (k_uni_dp, one lane), the backward pass, compaction
    struct HostRing {
        std::vector<double> v; std::vector<uint32_t> r; int mask;
        double score(int pos) const { return v[(size_t)(pos & mask)]; }
        uint32_t rec(int pos) const { return r[(size_t)(pos & mask)]; }
        void set(int pos, double x, uint32_t rr) { v[(size_t)(pos & mask)] = x; r[(size_t)(pos & mask)] = rr; }
        void fill(double x) { for (auto &e : v) e = x; for (auto &e : r) e = UNI_REC_NONE; }
    };
    int ring_n = 1; while (ring_n < m.trie_max_depth) ring_n <<= 1;
    long o = 0;
    std::vector<int32_t> one((size_t)(max_ids > 0 ? max_ids : 1));
    for (long d = 0; d < ndocs; ++d) {
        id_off[d] = o;
        const int L = lens[(size_t)d];
        if (flags_out) flags_out[d] = L > 0 ? flags[(size_t)d] : 0;
        if (L <= 0) continue;
        if (flags[(size_t)d] < 0) return -7;
        if (flags[(size_t)d]) {
            const int c = bft_emu_sp_doc(m, (const char *)text + doc_off[d], (int)(doc_off[d + 1] - doc_off[d]), one.data(), max_ids, unk);
            if (c < 0) return -8;
            if (o + c > ids_cap) return -9;
            for (int k = 0; k < c; ++k) ids_out[o + k] = one[(size_t)k];
            o += c; continue;
        }
        const int64_t slot = (int64_t)mul * (doc_off[d] + d);
        HostRing ring{std::vector<double>((size_t)ring_n), std::vector<uint32_t>((size_t)ring_n), ring_n - 1};
        std::vector<uint32_t> recs_all((size_t)L + 16, 0xDEADBEEFu);
        uint32_t *recs = recs_all.data() + 8;
        UniArcLane<HostRing> ul(ring, m.id_offset);
        ul.init(L, m.trie_max_depth, recs, (int64_t)(d % 5));
        const uint64_t *rt = rounds.data() + uw_round_base(slot, d);
        bool more = true;
        for (int r = 0; more; ++r) {
            if (rt[r] == ~0ull || rt[r] >= (uint64_t)pool_recs) return -11;
            const uint64_t *q = pool.data() + rt[r];
            const int starts = L - r * 64 < 64 ? L - r * 64 : 64;
            for (int done = 0; done < starts && more;) {
                const uint64_t rec = *q++;
                if ((uint32_t)rec & (UA_LAST | UA_UNK)) ++done;
                more = ul.astep((uint32_t)rec, (uint32_t)(rec >> 32));
            }
        }
        if (recs_all[7] != 0xDEADBEEFu || recs_all[(size_t)L + 8] != 0xDEADBEEFu) return -3;
        ul.begin_back();
        std::vector<int32_t> rid;
        auto put = [&](int, int id, int, int) { rid.push_back(id); };
        for (;;) { const uint32_t br = recs[(size_t)ul.end]; if (!ul.bstep(br, put, unk)) break; }
        const int cnt = (int)rid.size(), nout = cnt < max_ids ? cnt : max_ids;
        if (o + nout > ids_cap) return -9;
        for (int k = 0; k < nout; ++k) ids_out[o + k] = rid[(size_t)(cnt - 1 - k)];
        o += nout;
    }
    id_off[ndocs] = o;
    return o;
}


// EXPERIMENT (design aid, not a test of the product): what a direct-mapped cache of 2^log_slots table entries (index low bits = slot) would
// catch of the Unigram walks' table gathers and I2Info gathers on a batch.  out[0..5] = T gathers, T cache hits, gathers at depth 0,
// info gathers, info hits, starts; out[8 + j] = gathers at depth j (j < 8), out[16 + j] = hits (transitions made) at depth j
void bft_uni_cache_sim(void *hv, const uint8_t *text, const int64_t *doc_off, long ndocs, int log_slots, unsigned long long *out)
{
    Model &m = ((Handle *)hv)->m;
    std::vector<uint32_t> tagT((size_t)1 << log_slots, 0xFFFFFFFFu), tagI((size_t)1 << log_slots, 0xFFFFFFFFu);
    const uint32_t mask = (1u << log_slots) - 1u;
    std::vector<uint16_t> st;
    const uint64_t *T = m.dict.t64.data();
    for (long d = 0; d < ndocs; ++d) {
        if (!bft_sp_stream(m, (const char *)text + doc_off[d], (int)(doc_off[d + 1] - doc_off[d]), st, nullptr)) continue;
        const int L = (int)st.size();
        for (int s = 0; s < L; ++s) {
            ++out[5];
            uint32_t state = m.dict.initial_base; int sum = 0;
            for (int i = s, j = 0; i < L; ++i, ++j) {
                const uint32_t c = st[(size_t)i];
                if (c >= SG_CLS_DELIM_ABSENT) break;
                const uint32_t idx = state + c;
                ++out[0]; if (j == 0) ++out[2]; if (j < 8) ++out[8 + j];
                if (tagT[idx & mask] == idx) ++out[1]; else tagT[idx & mask] = idx;
                const uint64_t e = T[idx];
                if ((e & SG_CLS_MASK) != c) break;
                if (j < 8) ++out[16 + j];
                state = (uint32_t)((e >> SG_NEXT_SHIFT) & SG_NEXT_MASK); sum += (int)(e >> SG_OW_SHIFT);
                if (e & SG_FINAL) { ++out[3]; const uint32_t k = (uint32_t)sum; if (tagI[k & mask] == k) ++out[4]; else tagI[k & mask] = k; }
            }
        }
    }
}


// EXPERIMENT (design aid): a STATIC direct-mapped table of the 2^log_slots hottest transitions / I2Info rows, chosen at load from the
// model alone (mass of an edge = sum of exp(score) over the entries below it; a slot keeps the heaviest edge that maps to it), against
// the walks of a batch.  out[0..5] = T gathers, T hits, -, info gathers, info hits, starts
#include <cmath>
void bft_uni_static_cache_sim(void *hv, const uint8_t *text, const int64_t *doc_off, long ndocs, int log_slots, unsigned long long *out)
{
    Model &m = ((Handle *)hv)->m;
    const RawDfa &rw = m.dict_raw;
    const size_t ns = rw.state_off.size();
    const uint32_t mask = (1u << log_slots) - 1u;
    std::vector<uint32_t> tagT((size_t)1 << log_slots, 0xFFFFFFFFu), tagI((size_t)1 << log_slots, 0xFFFFFFFFu);
    std::vector<double> massT((size_t)1 << log_slots, -1.0), massI((size_t)1 << log_slots, -1.0);
    // class of a raw symbol
    std::vector<int> cls_of_sym;
    { int mx = 0; for (int sy : m.dict.sym_of_class) mx = std::max(mx, sy); cls_of_sym.assign((size_t)mx + 1, -1); for (size_t c = 0; c < m.dict.sym_of_class.size(); ++c) cls_of_sym[(size_t)m.dict.sym_of_class[c]] = (int)c; }
    // subtree mass by DFS with the accumulated MPH index
    std::function<double(int, int)> dfs = [&](int st, int sum) -> double {
        double mass = 0;
        for (uint32_t t = rw.tr_begin[(size_t)st]; t < rw.tr_begin[(size_t)st + 1]; ++t) {
            const int dst = rw.tr_dst[t]; if (dst < 0) continue;
            const int sum2 = sum + rw.tr_ow[t];
            double sub = dfs(dst, sum2);
            if (rw.is_final[(size_t)dst]) {
                float f; const uint32_t b = m.i2info_score[(size_t)sum2]; memcpy(&f, &b, 4);
                const double pm = std::exp((double)f);
                sub += pm;
                const uint32_t k = (uint32_t)sum2;
                if (pm > massI[k & mask]) { massI[k & mask] = pm; tagI[k & mask] = k; }
            }
            const int sy = rw.tr_sym[t];
            const int c = sy >= 0 && (size_t)sy < cls_of_sym.size() ? cls_of_sym[(size_t)sy] : -1;
            if (c >= 0) { const uint32_t idx = m.dict.state_base[(size_t)st] + (uint32_t)c; if (sub > massT[idx & mask]) { massT[idx & mask] = sub; tagT[idx & mask] = idx; } }
            mass += sub;
        }
        return mass;
    };
    (void)ns;
    dfs(rw.initial, 0);
    std::vector<uint16_t> st;
    const uint64_t *T = m.dict.t64.data();
    for (long d = 0; d < ndocs; ++d) {
        if (!bft_sp_stream(m, (const char *)text + doc_off[d], (int)(doc_off[d + 1] - doc_off[d]), st, nullptr)) continue;
        const int L = (int)st.size();
        for (int s = 0; s < L; ++s) {
            ++out[5];
            uint32_t state = m.dict.initial_base; int sum = 0;
            for (int i = s; i < L; ++i) {
                const uint32_t c = st[(size_t)i];
                if (c >= SG_CLS_DELIM_ABSENT) break;
                const uint32_t idx = state + c;
                ++out[0];
                if (tagT[idx & mask] == idx) ++out[1];
                const uint64_t e = T[idx];
                if ((e & SG_CLS_MASK) != c) break;
                state = (uint32_t)((e >> SG_NEXT_SHIFT) & SG_NEXT_MASK); sum += (int)(e >> SG_OW_SHIFT);
                if (e & SG_FINAL) { ++out[3]; const uint32_t k = (uint32_t)sum; if (tagI[k & mask] == k) ++out[4]; }
            }
        }
    }
}

} // extern "C"
