// bf_wavetest.cpp -- TEST-ONLY (part of libbf_hosttest.so, never linked into the product).
//
// Runs the wave kernel of bf_wave.h -- the very source the GPU executes -- inside the 64-fibre wave simulator of wave_emu.h,
// followed by a scalar restatement of the scan + compaction kernels, so that the whole WordPiece batch path of a unit-form model
// can be fuzzed against the oracle without a GPU.
#include "wave_emu.h"
#include "../../blingfire_amd/csrc/bf_wave_body.h"
#include "../../blingfire_amd/csrc/bf_bpe_wave_body.h"
#include "../../blingfire_amd/csrc/bf_flat_body.h"
#include "../../blingfire_amd/csrc/bf_bpe_seg_body.h"
#include "hosttest.h"

#include <vector>

using namespace bfa;

template <class LDS, int NU, bool OFFS = false, int TRIM = 0>
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
        WpWave<LDS, NU, true, 0, 3, 4, 0, OFFS, TRIM> w(p, p.cold, *of_wave[k], ascii.data(), p.acts);
        w.run(grab, (int)k, nwaves);
    };
    wvemu::run_waves(nwaves, body);
    for (auto *q : of_wave) delete q;
}

extern "C" {

int bft_wave_ok(void *hv) { return ((Handle *)hv)->m.wave_ok ? 1 : 0; }
const char *bft_wave_why(void *hv) { return ((Handle *)hv)->m.wave_why.c_str(); }
int bft_bpe_wave_ok(void *hv) { return ((Handle *)hv)->m.bpe_wave_ok ? 1 : 0; }

// TextToIdsBatch through the wave kernel on the host.  cfg: 3 / 4 = configurations 0 / 1 with the TRIM bits; 0 = the shipped configuration, 1 = two units per lane, the smallest ring and queue, a two-entry
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
    p.ids_tmp = tmp.data(); p.counts = counts.data(); p.max_ids = max_ids; p.unk = unk; p.next_doc = &next_doc; p.span_tmp = nullptr;
    if (cfg >= 16) { p.next_doc = nullptr; cfg -= 16; }            // cfg + 16: no work counter, the waves take their ranges round-robin
    p.cold.cpmap = DevCpMap{m.wbd_cpmap.l1.data(), m.wbd_cpmap.pages.data()};
    p.cold.kind = m.wave_kind.data(); p.cold.nclasses = m.wbd.nclasses; p.cold.status = &status; p.cold.stats = stats; p.cold.no_fast = (cfg == 1 || cfg == 4) ? 1 : 0;
    if (ndocs > 0) {
        if (cfg == 1) run_cfg<WvLds<1024, 128, 2>, 2>(p, nwaves, grab);
        else if (cfg == 2) run_cfg<WvLds<4096, 512, 64>, 3>(p, nwaves, grab);
        else if (cfg == 3) run_cfg<WvLds<1024, 256, 8>, 1, false, 15>(p, nwaves, grab);          // the TRIM bits (bf_wave_body.h) on the shipped configuration ...
        else if (cfg == 4) run_cfg<WvLds<1024, 128, 2>, 2, false, 15>(p, nwaves, grab);          // ... and on the smallest one, every token with an explicit action
        else if (cfg == 5) run_cfg<WvLds<1024, 256, 8>, 1, false, 3>(p, nwaves, grab);          // ... bits 1 + 2 alone
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

// TextToIdsWithOffsetsBatch through the OFFS instance of the wave program: ids + byte offsets of every id (first byte, last byte), the
// span -> byte-offset step of k_compact restated (tokdll:1263-1273).  cfg as above (0, 1, 2).
long bft_emu_wave_batch_offsets(void *hv, const uint8_t *text, long text_bytes, const int64_t *doc_off, long ndocs, int max_ids, int unk, int nwaves, int grab, int cfg,
                                int32_t *ids_out, int32_t *starts_out, int32_t *ends_out, long ids_cap, int64_t *id_off)
{
    Model &m = ((Handle *)hv)->m;
    if (!m.error.empty() || m.kind != KIND_WP || !m.wave_ok) return -1;
    if (max_ids < 0) max_ids = 0;
    const int64_t total = text_bytes;
    std::vector<int32_t> tmp((size_t)(total + 8 * ndocs + 64 + 8), -77), counts((size_t)ndocs + 1, -55), span(2 * (size_t)(total + 8 * ndocs + 64 + 8), -77);
    unsigned long long next_doc = 0, stats[16] = {0}; int status = 0;
    WpWaveParams p;
    p.T = m.wbd_t2.data(); p.acts = m.acts_pool.data(); p.acts_n = (int)m.acts_pool.size();
    p.initial = m.wbd.initial_base; p.loop_info = m.loop_info; p.solo_info = m.wave_solo_info; p.max_token_length = m.max_token_length;
    p.text = text; p.doc_off = doc_off; p.ndocs = ndocs; p.total_bytes = total;
    p.ids_tmp = tmp.data(); p.counts = counts.data(); p.max_ids = max_ids; p.unk = unk; p.next_doc = &next_doc;
    p.span_tmp = span.data();
    p.cold.cpmap = DevCpMap{m.wbd_cpmap.l1.data(), m.wbd_cpmap.pages.data()};
    p.cold.kind = m.wave_kind.data(); p.cold.nclasses = m.wbd.nclasses; p.cold.status = &status; p.cold.stats = stats; p.cold.no_fast = cfg == 1 ? 1 : 0;
    if (ndocs > 0) {
        if (cfg == 1) run_cfg<WvLds<1024, 128, 2, true>, 2, true>(p, nwaves, grab);
        else if (cfg == 2) run_cfg<WvLds<4096, 512, 64, true>, 3, true, 15>(p, nwaves, grab);
        else if (cfg == 3) run_cfg<WvLds<1024, 256, 8, true>, 1, true>(p, nwaves, grab);           // the instance without the TRIM bits
        else run_cfg<WvLds<1024, 256, 8, true>, 1, true, 15>(p, nwaves, grab);                     // shipped
    }
    if (status) return -5;
    long o = 0;
    for (long d = 0; d < ndocs; ++d) {
        id_off[d] = o;
        const int c = counts[(size_t)d];
        if (c < 0) return -6;
        const int64_t b = doc_off[d], slot = wv_ids_slot(b, d);
        // characters -> bytes from the text itself (what k_compact_text does by ballots): a character starts at every byte that is not a
        // continuation byte, a leading BOM is none
        const int n = (int)(doc_off[d + 1] - b);
        std::vector<int> byte_of;
        const int bom = (n >= 3 && text[b] == 0xEF && text[b + 1] == 0xBB && text[b + 2] == 0xBF) ? 3 : 0;
        for (int q = bom; q < n; ++q) if ((text[b + q] & 0xC0) != 0x80) byte_of.push_back(q);
        for (int i = 0; i < c; ++i) {
            if (o + i >= ids_cap) return -9;
            ids_out[o + i] = tmp[(size_t)(slot + i)];
            const int from = span[2 * (size_t)(slot + i)], to = span[2 * (size_t)(slot + i) + 1];
            if (from < 0 || to < from || (size_t)to >= byte_of.size()) return -12;
            const int so = byte_of[(size_t)from], eo = byte_of[(size_t)to];
            const uint32_t ch = text[b + eo];
            const int sz = (ch & 0x80) == 0 ? 1 : (ch & 0xE0) == 0xC0 ? 2 : (ch & 0xF0) == 0xE0 ? 3 : (ch & 0xF8) == 0xF0 ? 4 : 0;
            starts_out[o + i] = so; ends_out[o + i] = eo + (sz > 0 ? sz - 1 : 0);
        }
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
    // the word table (cfg bit 3: without it -- the program must give the same ids either way)
    if (!(cfg & 8) && !m.bpe_tab.empty()) { p.W = m.bpe_tab.data(); p.wbits = m.bpe_tab_bits; p.m0 = m.bpe_tab_m0; p.m1 = m.bpe_tab_m1; p.m2 = m.bpe_tab_m2; }
    cfg &= ~8;
    const bool home = (cfg & 32) != 0;                                   // the HOME form (ids at their words' homes; count + gather restated below)
    cfg &= ~32;
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
                if (home) { BpeWave<LDS, 3, 4, true> w(p, *of_wave[k]); w.run(grab, (int)k, nwaves); }
                else { BpeWave<LDS> w(p, *of_wave[k]); w.run(grab, (int)k, nwaves); }
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
        std::vector<int32_t> squeezed;
        if (home && !flags[(size_t)d]) {                                               // k_bpe_home_gather: the cells that hold an id, front to back
            for (int i = 0; i < lens[(size_t)d] && (int)squeezed.size() < max_ids; ++i) if (src[i] != BW_HOME_NONE) squeezed.push_back(src[i]);
            if ((int)squeezed.size() != c) return -10;                                // the count the program added up is not the number of cells that hold an id
            src = squeezed.data();
        }
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

// TextToIdsBatch through the FLAT program (bf_flat.h) on the host: k_wp_pre restated (ranges, fitness of the batch), the flat wave program,
// the list of documents it hands back, the wave program's LIST instance on those, k_wp_count / k_wp_merge (the device sources, in the
// simulator) and the scan restated.  nranges <= 0: as many ranges as documents allow up to 4 per wave.  Returns the total id count or < 0;
// stats (optional, 16 counters): [0] chunks [1] plain-ASCII chunks [2] tokens [3] table hits [4] words for the list [7] documents handed back;
// [8] = documents on the list of the wave program, [9] = 1 when the batch was not fit, [10] unit rounds, [11] words on the list.
// wrec_cap_in > 0: capacity of the word list (tests: a list that is too small).
// starts_out / ends_out (optional): the offsets API -- the byte offsets of every id (the listed documents': the OFFS + LIST instance of the wave
// program and the characters -> bytes step of k_compact_text restated).
static long emu_flat_batch(void *hv, const uint8_t *text, long text_bytes, const int64_t *doc_off, long ndocs, int max_ids, int unk, int nwaves, int nranges,
                           int32_t *ids_out, long ids_cap, int64_t *id_off, unsigned long long *stats, long wrec_cap_in, int32_t *starts_out, int32_t *ends_out)
{
    const bool offs = starts_out && ends_out;
    Model &m = ((Handle *)hv)->m;
    if (!m.error.empty() || m.kind != KIND_WP || !m.wave_ok || !m.flat_ok) return -1;
    if (max_ids < 0) max_ids = 0;
    const int64_t total = text_bytes;
    // ---- k_wp_pre
    int unsafe = 0;
    for (long d = 0; d < ndocs; ++d) { const int64_t a = doc_off[d], b = doc_off[d + 1]; if (a < 0 || b < a || b > total || b - a > WF_DOC_MAX) unsafe = 1; }
    if (nranges <= 0) nranges = nwaves * 4;
    if (nranges > ndocs) nranges = (int)ndocs;
    if (nranges < 1) nranges = 1;
    std::vector<int64_t> range_doc((size_t)nranges + 1, 0);
    if (!unsafe) {
        const int64_t first = doc_off[0], span = doc_off[ndocs] - first;
        for (int r = 0; r <= nranges; ++r) {
            int64_t lo = 0, hi = ndocs;
            if (r == 0) hi = 0; else if (r == nranges) lo = ndocs;
            else { const int64_t target = first + (int64_t)((__int128)span * r / nranges); while (lo < hi) { const int64_t mid = lo + (hi - lo) / 2; if (doc_off[mid] >= target) hi = mid; else lo = mid + 1; } }
            range_doc[(size_t)r] = lo;
        }
    }
    std::vector<uint32_t> ent((size_t)total + 64, 0xDEADBEEFu);
    std::vector<int32_t> home((size_t)total + 64, -77), entcnt((size_t)ndocs + 1, -55), dstat((size_t)ndocs + 1, 0), list((size_t)ndocs + 1, -1);
    std::vector<int64_t> entoff((size_t)ndocs + 1, -1);
    std::vector<int32_t> tmp((size_t)(total + 8 * ndocs + 64 + 8), -77), counts((size_t)ndocs + 1, 0), counts_hard((size_t)ndocs + 1, -55);
    std::vector<int32_t> span(offs ? 2 * (size_t)(total + 8 * ndocs + 64 + 8) : 2, -77);
    std::vector<uint32_t> espan(offs ? (size_t)total + 64 : 1, 0xDEADBEEFu), hspan(offs ? 2 * ((size_t)total + 64) : 1, 0xDEADBEEFu);
    unsigned long long next_range = 0, next_doc = 0; int status = 0;
    WpWaveCold cold;
    cold.cpmap = DevCpMap{m.wbd_cpmap.l1.data(), m.wbd_cpmap.pages.data()};
    cold.kind = m.wave_kind.data(); cold.nclasses = m.wbd.nclasses; cold.status = &status; cold.stats = stats; cold.no_fast = 0;
    WfParams fp;
    fp.T = m.wbd_t2.data(); fp.W = m.flat_tab.data(); fp.wbits = m.flat_bits; fp.m0 = m.flat_m0; fp.m1 = m.flat_m1; fp.m2 = m.flat_m2;
    fp.ini = m.flat_ini; fp.ini_l = m.flat_ini_l; fp.max_token_length = m.max_token_length; fp.unk = unk;
    fp.text = text; fp.doc_off = doc_off; fp.ndocs = ndocs; fp.total_bytes = total;
    fp.range_doc = range_doc.data(); fp.nranges = nranges; fp.next_range = &next_range; fp.unsafe = &unsafe;
    fp.ent = ent.data(); fp.home = home.data(); fp.ent_off = entoff.data(); fp.ent_cnt = entcnt.data(); fp.dstat = dstat.data(); fp.cold = cold;
    fp.espan = offs ? espan.data() : nullptr;
    (void)wrec_cap_in;
    std::vector<uint32_t> wrec((size_t)((total >> WF_REC_SHIFT) + 64) * 4, 0xDEADBEEFu); std::vector<int32_t> wrec_cnt((size_t)nranges * 2 + 2, 0);
    fp.wrec = wrec.data(); fp.wrec_cnt = wrec_cnt.data();
    if (ndocs > 0) {
        std::vector<uint32_t> lut(WF_LUT);
        for (int i = 0; i < 128; ++i) lut[(size_t)i] = wf_lut_value(cold, i);
        for (int i = 128; i < WF_LUT; ++i) lut[(size_t)i] = wf_kmask_value(i - 128);
        std::vector<WfLds *> of_wave((size_t)nwaves);
        for (int i = 0; i < nwaves; ++i) { of_wave[(size_t)i] = new WfLds(); memset((void *)of_wave[(size_t)i], 0xA5, sizeof(WfLds)); }
        std::vector<const void *> wave_ids;
        auto body = [&]() {
            const void *wid = (const void *)wvemu::g_cur->wave;
            size_t k = 0;
            for (; k < wave_ids.size(); ++k) if (wave_ids[k] == wid) break;
            if (k == wave_ids.size()) wave_ids.push_back(wid);
            WfWave<true> w(fp, *of_wave[k], lut.data(), fp.cold);
            w.run((int)k, nwaves);
        };
        wvemu::run_waves(nwaves, body);
        for (auto *q : of_wave) delete q;
    }
    if (status) return -5;
    // ---- k_wp_units
    if (!unsafe && ndocs > 0) {
        WfUnitParams up;
        up.T = fp.T; up.ini = fp.ini; up.ini_l = fp.ini_l; up.max_token_length = fp.max_token_length; up.text = text; up.total_bytes = total;
        up.wrec = wrec.data(); up.wrec_cnt = wrec_cnt.data(); up.range_doc = range_doc.data(); up.doc_off = doc_off; up.nranges = nranges;
        up.ent = ent.data(); up.home = home.data(); up.extra = counts.data(); up.espan = fp.espan; up.hspan = offs ? hspan.data() : nullptr; up.cpmap = cold.cpmap; up.kind = cold.kind; up.nclasses = cold.nclasses; up.stats = stats;
        std::vector<uint32_t> lut(128);
        for (int i = 0; i < 128; ++i) lut[(size_t)i] = wf_lut_value(cold, i);
        unsigned long long rounds = 0, nf_all = 0, ns_all = 0;
        std::vector<uint16_t> cbuf(64 * 16 + 8);
        wvemu::run_waves(1, [&]() {
            for (int r = 0; r < nranges; ++r) {
                const int64_t dlo = range_doc[(size_t)r], dhi = range_doc[(size_t)r + 1];
                if (dlo >= dhi) continue;
                const int64_t b0 = doc_off[dlo], b1 = doc_off[dhi];
                const unsigned long long nfast = (unsigned long long)wrec_cnt[2 * (size_t)r], nslow = (unsigned long long)wrec_cnt[2 * (size_t)r + 1];
                const uint32_t *fl = up.wrec + 4 * ((b0 + (1 << WF_REC_SHIFT) - 1) >> WF_REC_SHIFT), *sl = up.wrec + 4 * ((b1 >> WF_REC_SHIFT) - (int64_t)nslow);
                if (offs) {
                    for (unsigned long long first = 0; first < nfast; first += 128) wf_units<2, true, 0, true>(up, lut.data(), cbuf.data(), fl, b0, dlo, first, nfast, &rounds);
                    for (unsigned long long first = 0; first < nslow; first += 64) { wf_units<1, true, 1, true>(up, lut.data(), cbuf.data(), sl, b0, dlo, first, nslow, &rounds); wf_units<1, true, 2, true>(up, lut.data(), cbuf.data(), sl, b0, dlo, first, nslow, &rounds); }
                } else {
                    for (unsigned long long first = 0; first < nfast; first += 128) wf_units<2, true, 0, false>(up, lut.data(), cbuf.data(), fl, b0, dlo, first, nfast, &rounds);
                    for (unsigned long long first = 0; first < nslow; first += 64) { wf_units<1, true, 1, false>(up, lut.data(), cbuf.data(), sl, b0, dlo, first, nslow, &rounds); wf_units<1, true, 2, false>(up, lut.data(), cbuf.data(), sl, b0, dlo, first, nslow, &rounds); }
                }
                if (wvemu::g_cur->lane == 0) { nf_all += nfast; ns_all += nslow; }
            }
        });
        if (stats) { stats[10] = rounds; stats[11] = nf_all; stats[12] = ns_all; }
    }
    // ---- k_wp_hardlist
    unsigned int list_n = 0;
    for (long d = 0; d < ndocs; ++d) if (unsafe || (dstat[(size_t)d] & WF_D_HARD)) list[list_n++] = (int32_t)d;
    if (stats) { stats[8] = list_n; stats[9] = (unsigned long long)unsafe; }
    // ---- the wave program on the listed documents
    if (list_n > 0) {
        WpWaveParams p;
        p.T = m.wbd_t2.data(); p.acts = m.acts_pool.data(); p.acts_n = (int)m.acts_pool.size();
        p.initial = m.wbd.initial_base; p.loop_info = m.loop_info; p.solo_info = m.wave_solo_info; p.max_token_length = m.max_token_length;
        p.text = text; p.doc_off = doc_off; p.ndocs = ndocs; p.total_bytes = total;
        p.ids_tmp = tmp.data(); p.counts = counts.data(); p.max_ids = max_ids; p.unk = unk; p.next_doc = &next_doc; p.span_tmp = offs ? span.data() : nullptr;
        p.doc_list = list.data(); p.list_n = &list_n;
        p.cold = cold; p.cold.stats = nullptr;
        typedef WvLds<1024, 256, 8> L;
        typedef WvLds<1024, 256, 8, true> LO;
        std::vector<uint16_t> ascii(128);
        for (int i = 0; i < 128; ++i) ascii[(size_t)i] = (uint16_t)wv_element(p.cold, i);
        std::vector<L *> of_wave((size_t)nwaves); std::vector<LO *> of_wave_o((size_t)nwaves);
        for (int i = 0; i < nwaves; ++i) { of_wave[(size_t)i] = new L(); memset((void *)of_wave[(size_t)i], 0xA5, sizeof(L)); of_wave_o[(size_t)i] = new LO(); memset((void *)of_wave_o[(size_t)i], 0xA5, sizeof(LO)); }
        std::vector<const void *> wave_ids;
        auto body = [&]() {
            const void *wid = (const void *)wvemu::g_cur->wave;
            size_t k = 0;
            for (; k < wave_ids.size(); ++k) if (wave_ids[k] == wid) break;
            if (k == wave_ids.size()) wave_ids.push_back(wid);
            if (offs) { WpWave<LO, 1, false, 0, 3, 4, 0, true, 15, true> w(p, p.cold, *of_wave_o[k], ascii.data(), p.acts); w.run(1, (int)k, nwaves); }
            else { WpWave<L, 1, false, 0, 3, 4, 0, false, 15, true> w(p, p.cold, *of_wave[k], ascii.data(), p.acts); w.run(1, (int)k, nwaves); }
        };
        wvemu::run_waves(nwaves, body);
        for (auto *q : of_wave) delete q;
        for (auto *q : of_wave_o) delete q;
        if (status) return -5;
    }
    // ---- k_wp_count, scan, k_wp_merge
    WfMergeParams mp;
    mp.doc_off = doc_off; mp.ndocs = ndocs; mp.ent = ent.data(); mp.home = home.data(); mp.ent_off = entoff.data(); mp.ent_cnt = entcnt.data(); mp.dstat = dstat.data(); mp.unsafe = &unsafe;
    mp.ids_tmp = tmp.data(); mp.counts = counts.data(); mp.id_off = id_off; mp.ids_out = ids_out; mp.ids_cap = ids_cap; mp.status = &status; mp.max_ids = max_ids; mp.unk = unk;
    mp.espan = offs ? espan.data() : nullptr; mp.hspan = offs ? hspan.data() : nullptr; mp.starts_out = starts_out; mp.ends_out = ends_out; mp.counts_hard = offs ? counts_hard.data() : nullptr;
    wvemu::run_waves(1, [&]() { for (int64_t base = 0; base < ndocs; base += 64) wf_count_docs(mp, base); });
    long o = 0;
    for (long d = 0; d < ndocs; ++d) { id_off[d] = o; if (counts[(size_t)d] < 0) return -6; o += counts[(size_t)d]; }
    id_off[ndocs] = o;
    bool over_any = false;
    if (offs) {
        WfMergeLds<true> *mlds = new WfMergeLds<true>();
        wvemu::run_waves(1, [&]() { bool over = false; for (int64_t base = 0; base < ndocs; base += 64) wf_merge_docs<true>(mp, base, over, *mlds); if (over) over_any = true; });
        delete mlds;
        // k_compact_text restated for the documents the wave program tokenised (counts_hard): characters -> bytes from the text itself
        for (long d = 0; d < ndocs && list_n > 0; ++d) {
            const int c = counts_hard[(size_t)d];
            if (c < 0) return -6;
            if (c == 0) continue;
            const int64_t b = doc_off[d], slot = wv_ids_slot(b, d), od = id_off[d];
            const int n = (int)(doc_off[d + 1] - b);
            std::vector<int> byte_of;
            const int bom = (n >= 3 && text[b] == 0xEF && text[b + 1] == 0xBB && text[b + 2] == 0xBF) ? 3 : 0;
            for (int q = bom; q < n; ++q) if ((text[b + q] & 0xC0) != 0x80) byte_of.push_back(q);
            for (int i = 0; i < c; ++i) {
                if (od + i >= ids_cap) return -9;
                ids_out[od + i] = tmp[(size_t)(slot + i)];
                const int from = span[2 * (size_t)(slot + i)], to = span[2 * (size_t)(slot + i) + 1];
                if (from < 0 || to < from || (size_t)to >= byte_of.size()) return -12;
                const int so = byte_of[(size_t)from], eo = byte_of[(size_t)to];
                const uint32_t ch = text[b + eo];
                const int sz = (ch & 0x80) == 0 ? 1 : (ch & 0xE0) == 0xC0 ? 2 : (ch & 0xF0) == 0xE0 ? 3 : (ch & 0xF8) == 0xF0 ? 4 : 0;
                starts_out[od + i] = so; ends_out[od + i] = eo + (sz > 0 ? sz - 1 : 0);
            }
        }
    } else {
        WfMergeLds<false> *mlds = new WfMergeLds<false>();
        wvemu::run_waves(1, [&]() { bool over = false; for (int64_t base = 0; base < ndocs; base += 64) wf_merge_docs<false>(mp, base, over, *mlds); if (over) over_any = true; });
        delete mlds;
    }
    if (over_any) return -9;
    return o;
}

long bft_emu_flat_batch(void *hv, const uint8_t *text, long text_bytes, const int64_t *doc_off, long ndocs, int max_ids, int unk, int nwaves, int nranges,
                        int32_t *ids_out, long ids_cap, int64_t *id_off, unsigned long long *stats, long wrec_cap_in)
{
    return emu_flat_batch(hv, text, text_bytes, doc_off, ndocs, max_ids, unk, nwaves, nranges, ids_out, ids_cap, id_off, stats, wrec_cap_in, nullptr, nullptr);
}

long bft_emu_flat_batch_offsets(void *hv, const uint8_t *text, long text_bytes, const int64_t *doc_off, long ndocs, int max_ids, int unk, int nwaves, int nranges,
                                int32_t *ids_out, int32_t *starts_out, int32_t *ends_out, long ids_cap, int64_t *id_off, unsigned long long *stats)
{
    return emu_flat_batch(hv, text, text_bytes, doc_off, ndocs, max_ids, unk, nwaves, nranges, ids_out, ids_cap, id_off, stats, 0, starts_out, ends_out);
}

int bft_flat_ok(void *hv) { return ((Handle *)hv)->m.flat_ok ? 1 : 0; }
int bft_flat_words(void *hv) { return ((Handle *)hv)->m.flat_words; }

} // extern "C"
