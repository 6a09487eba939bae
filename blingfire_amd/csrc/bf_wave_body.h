// bf_wave_body.h -- the wave program of bf_wave.h (see there).  Include AFTER a definition of namespace wv:
//   bf_kernels.hip   wave intrinsics of gfx950
//   tests/hosttest   wave_emu.h, the 64-fibre simulator (test only)
//
// One wave works in batches, each phase a tight loop of its own so that every instruction is issued for (nearly) 64 busy lanes:
//   fill     documents are opened, decoded into the LDS ring and resolved into tokens (phase A) until the token queue is full;
//   units    every lane owns NU word units; a unit walks the vocabulary automaton over one token, one table gather per trip for
//            each unit (NU independent gathers in flight per lane), and takes the next queued token when its word is finished.
//            The loop ends when the queue is handed out and only a few long words are still walking: those keep their units
//            and go on together with the next batch;
//   retire   the finished tokens at the head of the queue, 64 at a time and in order: ids to their documents' staging slots.
// Measured on MI355X (profiles/r03_*): a first form that drained every batch ran the units at 17 % lane occupancy; a form with
// one producing action, one transition and one retire check per trip issued 4,000 VALU + 4,000 SALU instructions per 512-byte
// document (21 trips).  Instruction count per document is what this structure is about.
#pragma once
#include "bf_wave.h"

namespace bfa {

constexpr uint32_t WV_TK_INFO = 1u << 30, WV_TK_LEN_MASK = 0x1FFu;
// the action info of a top-level token in 16 bits: a SIMPLE action has a tag in 1..4, any other is an index into <= WV_ACTS_MAX ints (bf_model.cpp "unit form")
BF_WVD uint16_t wv_pack_info(uint32_t info) { return (uint16_t)((info & LX_INFO_SIMPLE) ? (0x8000u | (info & 0x7FFFu)) : (info & 0x7FFFu)); }
BF_WVD uint32_t wv_unpack_info(uint32_t v) { return (v & 0x8000u) ? (LX_INFO_SIMPLE | (v & 0x7FFFu)) : v; }    // token flags (WV_TK_INFO: the action is in qi[]; else the common word kind)

// LDS of one wave.  QCAP / DTN / RING are powers of two.  A token: q[].pos = absolute ring position of its first character (later,
// for a word of one piece: its id, unit_event); q[].w = length | document table entry (low 8 bits of the absolute entry number) << 16 |
// WV_TK_* flags; qc = set by the unit: 0 while it walks, then 1 + the number of ids -- and, until the unit starts, the action of a
// WV_TK_INFO token (wv_pack_info: tags and action indices are small).
struct WvTok { uint32_t pos, w; };
constexpr int WV_GRAB_MAX = 8;         // documents a wave takes from the work counter at once, at most
template <int RING_, int QCAP_, int DTN_, bool OFFS_ = false>
struct WvLds {
    static constexpr int RING = RING_, QCAP = QCAP_, DTN = DTN_;
    uint32_t qid[OFFS_ ? QCAP_ : 1];  // offsets API: the id of a word of one piece (its queue entry keeps the word's position: the span is derived from it at retire)
    alignas(16) uint16_t ring[RING];
    int64_t dt_slot[DTN], dt_doc[DTN];
    alignas(8) WvTok q[QCAP];
    int32_t dt_cap[DTN], dt_cnt[DTN]; uint32_t dt_flags[DTN], dt_rbase[DTN];      // dt_rbase: ring position of the document's first character
    uint16_t qc[QCAP];
    int64_t doff[WV_GRAB_MAX + 1];   // text offsets of the documents taken from the work counter
    alignas(8) WvTok spare_tok;
    uint32_t spare32;
    uint16_t spare;                  // spare, spare32, spare_tok: where a lane writes when it has nothing to write
};

// DBG (experiments, wrong results by design): 1 = units finish at once without walking, 2 = also nothing is moved at retire:
// instruction counts of the phases by difference (profiles/r03_phase_costs.txt)
// OFFS: the offsets API -- every id carries the first / last character of its sub-token (of its word, for UnkId: tokdll:1263-1297) through the
// provisional homes to its place, and the decoder records the byte every character starts at
// TRIM (bits; each measured on its own before it became part of the shipped instance): 1 = no settle at the top of a trip (the one at the bottom of
// the trip before has just run), 2 = the chunk-wide pass writes its tokens with selects (a lane without a token left writes to a spare entry)
// instead of an execution-mask branch per token, 4 = fill as nested loops (fill_nested) instead of the re-entered fill_step,
// 8 = the retire pass waits once for the ids it loaded (wv::arrived) so that its stores leave back to back
// LIST: the documents are those of p.doc_list[0 .. *p.list_n) (the ones the flat program, bf_flat.h, hands back), taken one at a time
template <class LDS, int NU = 2, bool STATS = false, int DBG = 0, int STEPS = 3, int UMIN = 4, int CROOM = 0, bool OFFS = false, int TRIM = 0, bool LIST = false>
struct WpWave {
    static constexpr int RING = LDS::RING, QCAP = LDS::QCAP, DTN = LDS::DTN;
    static constexpr uint32_t RMASK = RING - 1, QMASK = QCAP - 1, DMASK = DTN - 1;
    static_assert((RING & (RING - 1)) == 0 && RING >= 1024 && RING <= 32768, "ring size");
    static_assert((QCAP & (QCAP - 1)) == 0 && QCAP >= 128 && (DTN & (DTN - 1)) == 0 && DTN <= 64, "queue / document table size");

    const WpWaveParams &p; const WpWaveCold &cold; LDS &S; const uint16_t *ascii; const int32_t *acts;      // cold, ascii, acts: per-workgroup LDS copies / tables
    // the launch constants the loops use, each as a value of its own: the kernel arguments arrive as 8- and 16-dword tuples, and a tuple
    // that does not stay in scalar registers is reloaded whole for one member (wv::own() hides where the value came from)
    const uint64_t *T; int32_t *ids_tmp; int unk, maxtok;
    int lane;
    // ---- wave-uniform state
    uint32_t u_need;                 // distance from rlo to the lowest ring position a busy unit reads (0xFFFFFFFF: none is busy)
    uint32_t rhi, rlo;               // absolute ring positions: next element to write / oldest element still needed
    uint32_t q_tail, q_issue, q_retire;   // tokens: queued / handed to a unit / retired (absolute counters; slot = counter & QMASK)
    uint32_t dt_head, dt_tail;       // document table entries in use (absolute counters)
    int64_t dbase; int di, dn;       // the range of documents this wave took from the work counter: [dbase, dbase + dn), di of them opened (offsets: S.doff[])
    int64_t nd_all, list_doc;        // documents of the launch (LIST: *p.list_n); LIST: the document of list entry dbase
    int st_round, st_wave, st_waves; // without a work counter: range number st_wave + st_round * st_waves is this wave's next one
    bool have_doc, exiting;
    // current document
    const uint8_t *s; int n; uint32_t rbase; int dec_bytes, dec, done, open_start, bom; uint32_t curk;
    uint32_t fn_ini, fn_ini_l;       // the vocabulary function of the common word kinds (run and solo tokens), when fast_ok
    bool fast_ok;                    // run and solo tokens are both "WORD, call the same function": their units need no action lookup
    bool err;                        // per lane: this lane saw invalid UTF-8 in the current document
    unsigned long long st_trips, st_win, st_slow, st_tok, st_steps, st_ret, st_rewalk, st_idle, st_dec, st_gath, st_trans;

    BF_WVD WpWave(const WpWaveParams &p_, const WpWaveCold &cold_, LDS &S_, const uint16_t *ascii_, const int32_t *acts_) : p(p_), cold(cold_), S(S_), ascii(ascii_), acts(acts_)
    {
        T = p.T; ids_tmp = p.ids_tmp;       // pointers stay what they are: behind wv::own() the compiler would no longer know they point to global memory
        unk = wv::own(p.unk); maxtok = wv::own(p.max_token_length);
        lane = wv::lane(); rhi = rlo = 0; u_need = 0xFFFFFFFFu; q_tail = q_issue = q_retire = 0; dt_head = dt_tail = 0;
        dbase = 0; di = dn = 0; st_round = st_wave = 0; st_waves = 1; have_doc = exiting = false;
        nd_all = LIST ? (int64_t)*p.list_n : p.ndocs; list_doc = 0;
        s = nullptr; n = 0; rbase = 0; dec_bytes = dec = done = bom = 0; open_start = -1; curk = 0; err = false;
        st_trips = st_win = st_slow = st_tok = st_steps = st_ret = st_rewalk = st_idle = st_dec = st_gath = st_trans = 0;
        // the action of a run token and of a solo token (bf_model.cpp): the usual case is one calling WORD action for both
        fast_ok = false; fn_ini = 0; fn_ini_l = LX_NO_STATE;
        if (!cold.no_fast && !(p.loop_info & LX_INFO_SIMPLE) && !(p.solo_info & LX_INFO_SIMPLE)) {
            const int32_t *a = acts + p.loop_info, *c = acts + p.solo_info;
            if (a[2] == WBD_WORD_TAG && c[2] == WBD_WORD_TAG && a[5] == c[5] && a[6] == c[6]) { fast_ok = true; fn_ini = (uint32_t)a[5]; fn_ini_l = (uint32_t)a[6]; }
        }
    }

    BF_WVD int ring_free() const { return RING - (int)(rhi - rlo); }
    BF_WVD uint32_t ring_at(uint32_t abs_pos) const { return S.ring[abs_pos & RMASK]; }
    BF_WVD void put_token(uint32_t t, int pos, int len, uint32_t flags)
    {
        const uint32_t sl = t & QMASK;
        WvTok e; e.pos = rbase + (uint32_t)pos; e.w = (uint32_t)len | ((curk & 0xFFu) << 16) | flags;
        S.q[sl] = e;
    }
    BF_WVD void put_word(uint32_t t, int pos, int len, bool solo)                 // a run / solo token of the mask form
    {
        if (fast_ok) put_token(t, pos, len, 0u); else put_token_info(t, pos, len, solo ? p.solo_info : p.loop_info);
    }
    BF_WVD void put_token_info(uint32_t t, int pos, int len, uint32_t info)      // general form: any action
    {
        put_token(t, pos, len, WV_TK_INFO);
        S.qc[t & QMASK] = wv_pack_info(info);
    }

    // ------------------------------------------------------------------------------------------------------------------
    // decode: the next WV_CHUNK bytes of the current document -> ring elements.  Strict UTF-8, the rules of the sequential
    // decoder restated per byte position (FAUtf8Utils.cpp:121-196,233-270; same scheme as k_prep_wp): a continuation byte is
    // no character and must be covered by a lead 1..3 bytes before it; a lead byte gives length, checks its continuation
    // bytes, truncation (:167-171), overlong / > U+10FFFF (:185-188), surrogates (:190-193).  One leading BOM is skipped.
    // ------------------------------------------------------------------------------------------------------------------
    // ------------------------------------------------------------------------------------------------------------------
    // phase A, wide form: the next (up to) 512 elements at once, eight per lane, none of them WK_GENERAL .  Kinds as 2-bit fields of a 16-bit word per lane; a run of WK_LOOP elements
    // that crosses lanes is closed with one ballot (which lanes hold a run start) and one shuffle (the start position of the
    // nearest one); token numbers from a prefix sum of the per-lane counts; every lane then writes its own tokens.  Returns false
    // and leaves everything untouched (the window forms take over) when the elements hold a WK_GENERAL one, a run that
    // max-length cuts, or more tokens than the queue has room for.
    // ------------------------------------------------------------------------------------------------------------------
    BF_WVD bool phase_a_wide(bool fully, bool have_kk = false, uint32_t kk_in = 0)
    {
        const int cb = done;
        const int total = dec - cb < WV_CHUNK ? dec - cb : WV_CHUNK;
        int nb = total - lane * 8; nb = nb < 0 ? 0 : (nb > 8 ? 8 : nb);
        const uint32_t r = rbase + (uint32_t)cb + (uint32_t)(lane * 8);
        uint32_t kk = 0;
        if (have_kk) kk = kk_in;                                       // straight from the decoder: these are the elements it just wrote
        else if (((rbase + (uint32_t)cb) & 7u) == 0) {
            const uint32_t *src = (const uint32_t *)(S.ring + (r & RMASK));          // 8 elements = one 16-byte row, never wraps
            const uint32_t d0 = src[0], d1 = src[1], d2 = src[2], d3 = src[3];
            // kinds: bits 15:14 and 31:30 of every dword -> 2-bit fields 2k, 2k+1
            kk = ((d0 >> 14) & 3u) | ((d0 >> 28) & 0xCu) | (((d1 >> 14) & 3u) << 4) | (((d1 >> 28) & 0xCu) << 4) |
                 (((d2 >> 14) & 3u) << 8) | (((d2 >> 28) & 0xCu) << 8) | (((d3 >> 14) & 3u) << 12) | (((d3 >> 28) & 0xCu) << 12);
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) kk |= ((uint32_t)S.ring[(r + (uint32_t)k) & RMASK] >> WK_SHIFT) << (2 * k);
        }
        const uint32_t vm = nb >= 8 ? 0x5555u : (((1u << (2 * nb)) - 1u) & 0x5555u);
        const uint32_t lo = kk & 0x5555u, hi = (kk >> 1) & 0x5555u;
        const uint32_t loopm = lo & ~hi & vm, solom = lo & hi & vm, genm = ~lo & ~hi & vm;
        if (wv::any(genm != 0)) return false;
        const bool at_end = fully && cb + total == dec;                // the document ends with these elements
        const bool cont = open_start >= 0;
        uint32_t prev_last = wv::shfl_up((loopm >> 14) & 1u, 1), next_first = wv::shfl_down(loopm & 1u, 1);
        if (lane == 0) prev_last = cont ? 1u : 0u;
        if (lane == 63) next_first = 0u;
        const uint32_t h = loopm & ~((loopm << 2) | prev_last);        // run starts
        uint32_t en = loopm & ~((loopm >> 2) | (next_first << 14));    // run ends
        const int last_lane = (total - 1) >> 3; const uint32_t last_bit = 1u << (2 * ((total - 1) & 7));
        const bool stays_open = !at_end && wv::any(lane == last_lane && (en & last_bit) != 0);
        if (stays_open && lane == last_lane) en &= ~last_bit;
        // the start of the run a lane's first end belongs to, when it lies in an earlier lane (or before these elements)
        const int hl = h ? cb + lane * 8 + ((31 - __builtin_clz(h)) >> 1) : -1;           // this lane's last run start
        const unsigned long long HB = wv::ballot(h != 0);
        const unsigned long long hb_lt = HB & ((1ull << lane) - 1ull);
        const int hsrc = hb_lt ? 63 - __builtin_clzll(hb_lt) : 0;
        int hprev = wv::shfl(hl, hsrc);
        if (!hb_lt) hprev = open_start;
        const int new_open = stays_open ? (HB ? wv::bcast(hl, 63 - __builtin_clzll(HB)) : open_start) : -1;
        const bool carry_end = cont && !wv::any(lane == 0 && (loopm & 1u) != 0);           // the open run ended just before these elements
        uint32_t tk = en | solom;
        const int c = __builtin_popcount(tk);
        const int inc = wv::incl_scan(c);
        const uint32_t ntok = (uint32_t)wv::bcast(inc, 63) + (carry_end ? 1u : 0u);
        if ((q_tail - q_retire) + ntok > (uint32_t)QCAP) return false;
        if ((carry_end && cb - open_start > maxtok) || (stays_open && cb + total - new_open >= maxtok)) return false;
        // a finished run longer than max-length: found while writing, nothing is committed then
        uint32_t t = q_tail + (carry_end ? 1u : 0u) + (uint32_t)(inc - c);
        bool toolong = false;
        if (carry_end && lane == 0) put_word(q_tail, open_start, cb - open_start, false);
        // every token of the lane, lowest first; a run token (its end is here) starts at the nearest run start before the end, a solo
        // token is its own element.  Straight-line code: the only divergence is between lanes that have a token left and those that do not
        const uint32_t w_hi = ((curk & 0xFFu) << 16) | (fast_ok ? 0u : WV_TK_INFO);
        const uint16_t li = wv_pack_info(p.loop_info), si = wv_pack_info(p.solo_info);
        const int lane0 = cb + lane * 8;
        if (TRIM & 2) {
            int mlen = 0;                                                  // the longest token written (an integer: a per-lane bool carried through the loop is a lane mask, its updates scalar instructions)
            while (wv::any(tk != 0)) {
                const bool on = tk != 0;
                const int bit = __builtin_ctz(tk | 0x10000u); tk &= tk - 1u;           // (tk == 0: bit 16, nothing of it is kept)
                const int bpos = lane0 + (bit >> 1);
                const bool is_end = (en >> bit) & 1u;
                const uint32_t hm = h & ((2u << bit) - 1u);
                const int hs = hm ? lane0 + ((31 - __builtin_clz(hm | 1u)) >> 1) : hprev;
                const int start = is_end ? hs : bpos;
                const int len = bpos - start + 1;
                const int lenon = on ? len : 0;
                mlen = lenon > mlen ? lenon : mlen;
                const uint32_t sl = t & QMASK;
                WvTok e; e.pos = rbase + (uint32_t)start; e.w = (uint32_t)len | w_hi;
                WvTok *dst = on ? &S.q[sl] : &S.spare_tok;
                *dst = e;
                if (!fast_ok) { uint16_t *dq = on ? &S.qc[sl] : &S.spare; *dq = is_end ? li : si; }
                t += on ? 1u : 0u;
            }
            toolong = mlen > maxtok;
        } else
        while (wv::any(tk != 0)) {
            if (tk) {
                const int bit = __builtin_ctz(tk); tk &= tk - 1u;
                const int bpos = lane0 + (bit >> 1);
                const bool is_end = (en >> bit) & 1u;
                const uint32_t hm = h & ((2u << bit) - 1u);
                const int hs = hm ? lane0 + ((31 - __builtin_clz(hm | 1u)) >> 1) : hprev;
                const int start = is_end ? hs : bpos;
                const int len = bpos - start + 1;
                toolong |= len > maxtok;
                const uint32_t sl = t & QMASK;
                WvTok e; e.pos = rbase + (uint32_t)start; e.w = (uint32_t)len | w_hi;
                S.q[sl] = e;
                if (!fast_ok) S.qc[sl] = is_end ? li : si;
                ++t;
            }
        }
        wv::sync();
        if (wv::any(toolong)) return false;
        if (STATS) ++st_win;
        q_tail += ntok; done = cb + total; open_start = new_open;
        return true;
    }

    // this lane's 8 bytes of the chunk at `pos` of the text [t, t + tn)
    BF_WVD uint64_t load_chunk(const uint8_t *t, int tn, int pos) const
    {
        const int q0 = pos + lane * 8;
        uint64_t own = 0;
        int nb = tn - q0; nb = nb < 0 ? 0 : (nb > 8 ? 8 : nb);
        if (nb == 8) __builtin_memcpy(&own, t + q0, 8);
        else for (int k = 0; k < nb; ++k) own |= (uint64_t)t[q0 + k] << (8 * k);
        return own;
    }
    // Returns true when the chunk was plain ASCII; kk_out then holds the kinds of this lane's eight elements as 2-bit fields (what
    // phase_a_wide() would read back from the ring).
    BF_WVD bool decode_chunk(uint32_t &kk_out)
    {
        kk_out = 0;
        if (STATS) ++st_dec;
        const int pos = dec_bytes;
        const int q0 = pos + lane * 8;
        // (Issuing this load one chunk -- or one document -- ahead was built and measured: the two registers that then live through
        // every phase cost 7 % more vector instructions in spill code at 64 VGPRs, 6.86 against 6.66 ms per 2.5 M documents.)
        const uint64_t own = load_chunk(s, n, pos);
        int nb = n - q0; nb = nb < 0 ? 0 : (nb > 8 ? 8 : nb);
        const uint32_t w0 = rbase + (uint32_t)dec;                  // absolute ring position of this chunk's first element
        if (!wv::any((own & 0x8080808080808080ull) != 0)) {
            // plain ASCII (a BOM is not): stream position == byte position, every byte through the 128-entry table
            uint32_t e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = ascii[(uint32_t)(own >> (8 * k)) & 0x7f];
            const uint32_t r = w0 + (uint32_t)(lane * 8);
            if ((w0 & 7) == 0) {                                     // (wave-uniform) rows of 8 elements: what lies behind the text is never read
                uint32_t *dst = (uint32_t *)(S.ring + (r & RMASK));  // one 16-byte row, never wraps (RING % 8 == 0)
                dst[0] = e[0] | (e[1] << 16); dst[1] = e[2] | (e[3] << 16); dst[2] = e[4] | (e[5] << 16); dst[3] = e[6] | (e[7] << 16);
            } else {
#pragma unroll
                for (int k = 0; k < 8; ++k) if (k < nb) S.ring[(r + (uint32_t)k) & RMASK] = (uint16_t)e[k];
            }
            const int total = n - pos < WV_CHUNK ? n - pos : WV_CHUNK;
            dec += total; dec_bytes = pos + WV_CHUNK; rhi = rbase + (uint32_t)dec;
#pragma unroll
            for (int k = 0; k < 8; ++k) kk_out |= (e[k] >> WK_SHIFT) << (2 * k);
            wv::sync();
            return true;
        }
        // ---- a chunk with bytes >= 0x80.  Every byte that is not a continuation byte is an element; the ASCII ones come from the table
        //      as above, the lead bytes are decoded one per lane and trip (a lane of Latin text holds one or two, a lane of CJK three);
        //      a continuation byte is legal exactly when it is one of the (length - 1) bytes behind a lead byte
        uint32_t nxt = wv::shfl_down((uint32_t)own, 1);
        if (lane == 63) { nxt = 0; for (int k = 0; k < 3; ++k) if (q0 + 8 + k < n) nxt |= (uint32_t)s[q0 + 8 + k] << (8 * k); }
        if (pos == 0) {                                                   // FAUtf8Utils.cpp:247-252
            const int has_bom = (n >= 3 && ((uint32_t)own & 0xFFFFFFu) == 0xBFBBEFu) ? 3 : 0;
            bom = wv::bcast(has_bom, 0);
        }
        const uint64_t h80 = own & 0x8080808080808080ull, h40 = (own << 1) & 0x8080808080808080ull;
        // one bit per byte: >= 0x80, continuation (10xxxxxx), lead (11xxxxxx); bytes of this lane that belong to the text (a BOM does not)
        uint32_t m80 = 0, m40 = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { m80 |= (uint32_t)((h80 >> (8 * k + 7)) & 1ull) << k; m40 |= (uint32_t)((h40 >> (8 * k + 7)) & 1ull) << k; }
        uint32_t vmask = nb >= 8 ? 0xFFu : ((1u << nb) - 1u);
        if (pos == 0 && lane == 0 && bom) vmask &= ~7u;
        const uint32_t contm = m80 & ~m40 & vmask, leadm = m80 & m40 & vmask, em = vmask & ~contm;
        const int cnt = __builtin_popcount(em);
        const int inc = wv::incl_scan(cnt);
        const uint32_t base = w0 + (uint32_t)(inc - cnt);
        uint32_t e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = ascii[(uint32_t)(own >> (8 * k)) & 0x7f];
#pragma unroll
        for (int k = 0; k < 8; ++k) {                                     // the ASCII elements (a lane without one writes into the spare slot)
            const bool on = ((em & ~leadm) >> k) & 1u;
            uint16_t *dst = on ? &S.ring[(base + (uint32_t)__builtin_popcount(em & ((1u << k) - 1u))) & RMASK] : &S.spare;
            *dst = (uint16_t)e[k];
        }
        uint32_t cov = 0;                                                 // bytes behind a lead that belong to its character (bits 8..10: in the next lane)
        bool e_any = false;
        for (uint32_t lm = leadm; wv::any(lm != 0);) {
            if (lm) {
                const int k = __builtin_ctz(lm); lm &= lm - 1u;
                const int q = q0 + k;
                uint64_t w = own >> (8 * k);
                if (k) w |= (uint64_t)nxt << (64 - 8 * k);
                const uint32_t b0 = (uint32_t)w & 0xFF, b1 = (uint32_t)(w >> 8) & 0xFF, b2 = (uint32_t)(w >> 16) & 0xFF, b3 = (uint32_t)(w >> 24) & 0xFF;
                int len, cp; bool er = false;
                if ((b0 & 0xE0) == 0xC0) { len = 2; cp = (int)(b0 & 0x1F); }
                else if ((b0 & 0xF0) == 0xE0) { len = 3; cp = (int)(b0 & 0x0F); }
                else if ((b0 & 0xF8) == 0xF0) { len = 4; cp = (int)(b0 & 0x07); }
                else { len = 1; cp = 0; er = true; }                                            // F8 .. FF
                if (q + len > n) er = true;                                                    // truncated tail (:167-171)
                if (len >= 2) { if ((b1 & 0xC0) != 0x80) er = true; cp = (cp << 6) | (int)(b1 & 0x3F); }
                if (len >= 3) { if ((b2 & 0xC0) != 0x80) er = true; cp = (cp << 6) | (int)(b2 & 0x3F); }
                if (len >= 4) { if ((b3 & 0xC0) != 0x80) er = true; cp = (cp << 6) | (int)(b3 & 0x3F); }
                const int need = cp <= 0x7F ? 1 : cp <= 0x7FF ? 2 : cp <= 0xFFFF ? 3 : cp <= 0x10FFFF ? 4 : 0;
                if (need != len) er = true;                                                    // overlong / > U+10FFFF (:185-188)
                if ((cp & 0xFFFFF800) == 0xD800) er = true;                                    // surrogate (:190-193)
                e_any |= er;
                cov |= (((1u << len) - 1u) & ~1u) << k;
                S.ring[(base + (uint32_t)__builtin_popcount(em & ((1u << k) - 1u))) & RMASK] = (uint16_t)(er ? (LX_CLS_NONE | (WK_NOMATCH << WK_SHIFT)) : wv_element(cold, cp));
            }
        }
        // continuation bytes at the start of the lane may belong to a lead in the lane before (lane 0: in the bytes before the chunk)
        uint32_t spill = wv::shfl_up(cov >> 8, 1);
        if (lane == 0) {
            spill = 0;
            for (int back = 1; back <= 3 && pos - back >= bom; ++back) {
                const uint32_t c0 = s[pos - back];
                if ((c0 & 0xC0) == 0x80) continue;                                             // a continuation byte: look further back
                const int len = (c0 & 0xE0) == 0xC0 ? 2 : (c0 & 0xF0) == 0xE0 ? 3 : (c0 & 0xF8) == 0xF0 ? 4 : 1;
                if (c0 >= 0xC0 && len > back) spill = (1u << (len - back)) - 1u;
                break;
            }
        }
        if (contm & ~(cov | spill)) e_any = true;                          // a continuation byte no lead accounts for (:152-165)
        err |= e_any;
        const int total = wv::bcast(inc, 63);
        dec += total; dec_bytes = pos + WV_CHUNK; rhi = rbase + (uint32_t)dec;
        wv::sync();
        return false;
    }

    // ------------------------------------------------------------------------------------------------------------------
    // phase A, general form: one window of up to 64 start positions, the first of which (`done`) is a start position of the
    // reference's loop (FALexTools_t.h:229); every lane finds the token a walk from its position gives (the automaton itself for
    // WK_GENERAL elements), the chain of start positions is followed through the window.  Queues the tokens whose extent is known
    // and moves `done` behind them.  Returns false when the token at `done` itself needs elements that are not decoded yet.
    // `fully`: the whole document is decoded.  Used for what phase_a_wide() declines.
    // ------------------------------------------------------------------------------------------------------------------
    BF_WVD bool phase_a_general(bool fully)
    {
        const int w0 = done;
        const int nv = dec - w0 < 64 ? dec - w0 : 64;
        const bool valid = lane < nv;
        const int pos = w0 + lane;
        const uint32_t el = valid ? ring_at(rbase + (uint32_t)pos) : (WK_NOMATCH << WK_SHIFT);
        const uint32_t kind = el >> WK_SHIFT;
        const unsigned long long M = wv::ballot(valid && kind == WK_LOOP);
        const unsigned long long G = wv::ballot(valid && kind == WK_GENERAL);
        const bool at_end = fully && w0 + nv == dec;                    // the window reaches the end of the document
        if (STATS) ++st_slow;
        // ---- per lane: the token a walk from this position finds (has / len / info), whether its extent is certain, next start
        bool has = false, complete = true; int len = 1; uint32_t info = 0;
        if (kind == WK_LOOP) {
            // the run of WK_LOOP elements from here (closed, final loop state: FALexTools_t.h:255-277 stays in it, fp follows)
            const unsigned long long x = ~(M >> lane);                  // bits above 63 - lane of M >> lane are 0: r <= 64 - lane
            const int r = x ? __builtin_ctzll(x) : 64;
            has = true; info = p.loop_info;
            len = r < maxtok ? r : maxtok;
            complete = r >= maxtok || lane + r < nv || at_end;          // a run that touches the end of the window may go on
        } else if (kind == WK_SOLO) {
            has = true; info = p.solo_info;
        }
        // a run that fills the whole window: look further (words of 64 characters and more)
        if (M == ~0ull && nv < maxtok) {
            int ext = nv; bool term = false;
            while (ext < maxtok && w0 + ext < dec) {
                const int nv2 = dec - (w0 + ext) < 64 ? dec - (w0 + ext) : 64;
                const uint32_t e2 = lane < nv2 ? ring_at(rbase + (uint32_t)(w0 + ext + lane)) : 0u;
                const unsigned long long m2 = wv::ballot(lane < nv2 && (e2 >> WK_SHIFT) == WK_LOOP);
                const int rr = m2 == ~0ull ? 64 : __builtin_ctzll(~m2);      // <= nv2: the lanes behind nv2 are clear
                ext += rr;
                if (rr < nv2) { term = true; break; }
            }
            if (lane == 0) {
                len = ext < maxtok ? ext : maxtok;
                complete = term || ext >= maxtok || (fully && w0 + ext >= dec);
            }
        }
        if (G) {
            if (kind == WK_GENERAL && valid) {
                // the automaton itself (FALexTools_t.h:255-277 without anchors: none apply, bf_model.cpp "unit form")
                uint32_t state = p.initial; int j = pos, fp = -1; uint32_t finfo = 0;
                const int b = pos + maxtok;
                for (;;) {
                    if (j >= b) break;
                    if (j >= dec) { complete = fully; break; }
                    const uint32_t c = ring_at(rbase + (uint32_t)j) & LX_T_CLS_MASK;
                    const uint64_t e64 = T[state + c];
                    const uint32_t e = (uint32_t)e64;
                    if ((e & LX_T_CLS_MASK) != c) break;
                    if ((int32_t)e < 0) { fp = j; finfo = (uint32_t)(e64 >> 32); }
                    state = (e >> LX_T_NEXT_SHIFT) & LX_T_NEXT_MASK;
                    ++j;
                }
                has = fp >= 0; len = has ? fp - pos + 1 : 1; info = finfo;
            }
        }
        const int nxt = pos + len;
        // ---- the chain of start positions inside the window
        const unsigned long long V = nv == 64 ? ~0ull : ((1ull << nv) - 1ull);
        unsigned long long St;
        const bool odd = (kind == WK_GENERAL && has && len > 1) || (kind == WK_LOOP && len == maxtok && lane + len < nv && ((M >> (lane + len)) & 1ull));
        if (wv::ballot(valid && odd) == 0) {
            St = (V & ~M) | (M & ~(M << 1));            // every non-run element, and the first element of every run
        } else {
            // a multi-element general token or a run cut by max-length: follow the chain (FALexTools_t.h:229, 390-393)
            St = 0; int cur = w0;
            while (cur < w0 + nv) { St |= 1ull << (cur - w0); cur = wv::bcast(nxt, cur - w0); }
        }
        // ---- cut at the first start whose token is not certain yet
        const unsigned long long inc = wv::ballot(valid && !complete) & St;
        int new_done;
        if (inc) {
            const int fi = __builtin_ctzll(inc);
            if (fi == 0) return false;
            St &= (1ull << fi) - 1ull;
            new_done = w0 + fi;
        } else new_done = wv::bcast(nxt, 63 - __builtin_clzll(St));
        // ---- queue the tokens
        const unsigned long long TK = St & wv::ballot(has);
        if ((TK >> lane) & 1ull) put_token_info(q_tail + (uint32_t)__builtin_popcountll(TK & ((1ull << lane) - 1ull)), pos, len, info);
        q_tail += (uint32_t)__builtin_popcountll(TK);
        done = new_done;
        wv::sync();
        return true;
    }

    // ------------------------------------------------------------------------------------------------------------------
    // units: the function frame of Process_int (FALexTools_t.h:229-393 at depth 1) on one word, and the post-pass on its
    // sub-tokens (tokdll:1239-1301): from = -1 takes the left anchor (resolved at load), a walk that matches emits a piece
    // and continues behind it, a walk that fails at from >= 0 leaves a gap, so the pieces cannot tile the word: UNK.
    // A unit is resumable: issue() starts the gather of its next transition, complete() consumes it.
    // ------------------------------------------------------------------------------------------------------------------
    // Ids have a provisional home in global memory: piece k of the word whose first character is at position f of document d goes
    // to ids_tmp[slot(d) + f + k] (a piece is at least one character, so homes never collide and never leave the slot); retire moves
    // them down to their place in the document.  The id of a word that is a single piece (most are) never leaves LDS before that: it takes
    // the place of the word's ring position in its queue entry (unit_event).
    //
    // A unit runs the frame of ONE function call: walks start at `j0` (first the anchored walk from ini_l at character 0, if the
    // function has a left-anchor transition; then the plain walks from ini), read letters while j < lim = min(start + max-length, L).
    // Written with selects: the walk loop is what every lane executes on every trip.
    struct Unit {
        int tok;                         // token (absolute queue counter, as int), -1: idle
        uint32_t rs; int Lk; uint32_t ini;       // Lk: length of the word | its document's table entry << 16
        int j, lim; uint32_t state; int fp; uint32_t ftag;
        int ca;                          // pieces so far | 1 << 16 while the anchored walk runs
        int from;                        // OFFS: where the walk under way started (a piece is [from, fp])
        // a walk is under way while j < lim; j >= lim with tok >= 0: its end waits for unit_event (a miss sets j = lim)
    };

    // What follows runs for all lanes of the wave at once and is written with selects: a lane that has nothing to do (pred false)
    // executes the same instructions, keeps what it has and writes to the spare words.  (Measured, profiles/r03_*: as nested
    // conditionals the code of a round cost 127 scalar instructions, most of them execution-mask bookkeeping.)
    // OFFS: id k of the unit's word and the characters [from, to] of the word it covers, to the provisional home
    BF_WVD void home_put(const Unit &u, int k, int32_t id, int from, int to)
    {
        const uint32_t ke = (uint32_t)u.Lk >> 16;
        const int f = (int)(u.rs - S.dt_rbase[ke]);
        const int64_t at = S.dt_slot[ke] + (int64_t)f + k;
        ids_tmp[at] = id; p.span_tmp[2 * at] = f + from; p.span_tmp[2 * at + 1] = f + to;
    }
    BF_WVD void unit_finish(Unit &u, int cnt)
    {
        S.qc[(uint32_t)u.tok & QMASK] = (uint16_t)(cnt + 1);
        u.tok = -1; u.j = u.lim;
    }
    // the frame of a call to the function (ini, ini_l) on the unit's word
    BF_WVD void unit_call(Unit &u, uint32_t ini, uint32_t ini_l, bool pred)
    {
        const bool anchored = ini_l != LX_NO_STATE && maxtok > 1;     // else "from = -1" goes straight on to from = 0 (FALexTools_t.h:244-252)
        const int cap = anchored ? maxtok - 1 : maxtok;
        const int L = u.Lk & 0xFFFF;
        u.ini = pred ? ini : u.ini; u.state = pred ? (anchored ? ini_l : ini) : u.state; u.ca = pred ? (anchored ? 0x10000 : 0) : u.ca;
        u.j = pred ? 0 : u.j; u.lim = pred ? (cap < L ? cap : L) : u.lim; u.fp = pred ? -1 : u.fp;
        if (OFFS) u.from = pred ? 0 : u.from;
    }
    // starts the unit of token t (take: this lane takes one)
    BF_WVD void unit_begin(Unit &u, uint32_t t, bool take)
    {
        const uint32_t sl = t & QMASK;
        const WvTok e = S.q[sl];
        const uint32_t w1 = e.w;
        const uint32_t info16 = wv_unpack_info(S.qc[sl]);
        uint16_t *zq = take ? &S.qc[sl] : &S.spare;
        *zq = 0;
        u.tok = take ? (int)t : u.tok;
        u.rs = take ? e.pos : u.rs; u.Lk = take ? (int)((w1 & WV_TK_LEN_MASK) | (((w1 >> 16) & DMASK) << 16)) : u.Lk;
        if (DBG >= 1) { if (take) unit_finish(u, 1); return; }
        const bool other = take && (w1 & WV_TK_INFO) != 0;
        unit_call(u, fn_ini, fn_ini_l, take && !other);                            // a word of the common kind: the vocabulary function
        if (wv::any(other)) {
            // any other action (general form of phase A; lexers whose run / solo actions differ)
            if (other) {
                const uint32_t info = info16;
                int tag; bool call = false; uint32_t ini = 0, ini_l = LX_NO_STATE;
                if (info & LX_INFO_SIMPLE) tag = (int)(info & 0x7FFFFFFFu);
                else { const int32_t *a = acts + info; tag = a[2]; ini = (uint32_t)a[5]; ini_l = (uint32_t)a[6]; call = true; }
                if (tag != WBD_WORD_TAG) unit_finish(u, 0);                            // tags 2..4: neither a word nor a sub-token
                else if (!call) {                                                      // a word without sub-tokens (tokdll:1282-1301)
                    if (OFFS) S.qid[sl] = (uint32_t)unk; else S.q[sl].pos = (uint32_t)unk;
                    unit_finish(u, 1);
                }
                else unit_call(u, ini, ini_l, true);
            }
        }
    }
    // One transition (FALexTools_t.h:255-277) for every lane at once, written without a branch: a lane whose unit is not walking
    // feeds its old state to the table as well and keeps everything it has.  j >= lim afterwards: the walk is over (a miss, or the
    // next position is not < lim) and waits for unit_event().
    BF_WVD void unit_step(Unit &u) const
    {
        const uint32_t c = (uint32_t)S.ring[(u.rs + (uint32_t)u.j) & RMASK] & LX_T_CLS_MASK;
        // a lane that is not walking reads entry 0 like every other such lane (one cache line for all of them: a divergent gather costs
        // the memory pipeline about a cycle per distinct lane address, MI355X tools/microbench/gather.hip)
        const bool act = u.j < u.lim;
        const uint64_t e64 = T[act ? u.state + c : 0u];
        const uint32_t e = (uint32_t)e64;
        const bool hit = act && (e & LX_T_CLS_MASK) == c;
        const bool fin = hit && (int32_t)e < 0;
        u.fp = fin ? u.j : u.fp; u.ftag = fin ? (uint32_t)(e64 >> 32) : u.ftag;
        u.state = hit ? ((e >> LX_T_NEXT_SHIFT) & LX_T_NEXT_MASK) : u.state;
        u.j = hit ? u.j + 1 : u.lim;                                      // a miss ends the walk
    }
    // The end of a walk (ev: this lane's unit has one): a match is a piece and the next walk starts behind it (FALexTools_t.h:390-393);
    // the anchored walk without a match is followed by the plain walk at 0 (:293); any other walk without a match leaves a gap, the
    // pieces cannot tile the word: UNK (tokdll:1252-1301)
    BF_WVD void unit_event(Unit &u, bool ev)
    {
        const int cnt0 = u.ca & 0xFFFF, L = u.Lk & 0xFFFF;
        const bool matched = ev && u.fp >= 0, gap = ev && u.fp < 0 && !(u.ca >> 16);
        const uint32_t sl = (uint32_t)u.tok & QMASK;
        const int32_t id = (int32_t)(u.ftag & 0x7FFFFFFFu);
        // The id of a word's only piece stays in LDS (in the queue entry's pos: the unit holds the word's ring position itself); with
        // the second piece the first one moves to its provisional home and the entry gets the position back (retire finds the home by it)
        if (OFFS) {
            // the entry keeps the word's position.  A word's only id stays in LDS (qid) and takes the word's own span at retire (also UnkId:
            // tokdll:1282-1297); with the second piece the first one moves to its provisional home, its span is [0, the second one's begin - 1]
            // (the pieces tile the word), and every further piece goes there with its span
            const bool more = matched && cnt0 >= 1;
            if (wv::any(more)) { if (more) { if (cnt0 == 1) home_put(u, 0, (int32_t)S.qid[sl], 0, u.from - 1); home_put(u, cnt0, id, u.from, u.fp); } }
            uint32_t *wq = (gap || (matched && cnt0 == 0)) ? &S.qid[OFFS ? sl : 0] : &S.spare32;
            *wq = gap ? (uint32_t)unk : (uint32_t)id;
        } else {
        const bool more = matched && cnt0 >= 1;
        if (wv::any(more)) {
            if (more) {
                const uint32_t ke = (uint32_t)u.Lk >> 16;
                int32_t *home = ids_tmp + S.dt_slot[ke] + (int64_t)(u.rs - S.dt_rbase[ke]);
                if (cnt0 == 1) { if (!(TRIM & 16)) home[0] = (int32_t)S.q[sl].pos; S.q[sl].pos = u.rs; }
                if (!(TRIM & 16)) home[cnt0] = id;                    // (TRIM 16: an experiment, wrong results by design: what the stores to the provisional homes cost)
            }
        }
        uint32_t *wp = (gap || (matched && cnt0 == 0)) ? &S.q[sl].pos : &S.spare32;
        *wp = gap ? (uint32_t)unk : (uint32_t)id;
        }
        const int cnt = cnt0 + (matched ? 1 : 0);
        const int nf = matched ? u.fp + 1 : 0;
        const bool fin = gap || (ev && nf >= L), go = ev && !fin;
        uint16_t *cq = fin ? &S.qc[sl] : &S.spare;
        *cq = (uint16_t)((gap ? 1 : cnt) + 1);
        const int b = nf + maxtok;
        u.state = go ? u.ini : u.state; u.j = go ? nf : u.j; u.lim = go ? (b < L ? b : L) : u.lim; u.fp = go ? -1 : u.fp;
        if (OFFS) u.from = go ? nf : u.from;
        u.ca = go ? cnt : u.ca; u.tok = fin ? -1 : u.tok;
    }
    // Runs the units until the queue is handed out and fewer than UNIT_MIN of them are still busy (`drain`: until all are done).
    // A round: the units whose walk is over take its result (piece / next walk / word finished), idle units take the next queued
    // tokens (rank among the idle lanes = order in the queue), then every walking unit makes STEPS transitions.  The event code
    // is branchy and runs once per round for all the lanes that need it; the transitions are straight-line code.
    // (Measured on MI355X, profiles/r03_phase_costs.txt: with the event code inside every transition the loop cost 1,340 issued
    // instructions per 512-byte document, a third of them for the transitions themselves.)  Returns whether a transition was made.
    BF_WVD bool units_phase(Unit (&u)[NU], bool drain)
    {
        bool ran = false;
        const uint32_t tail = wv::uni(q_tail);
        uint32_t issue = wv::uni(q_issue);
        for (;;) {
            int nb = 0;
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                const bool ev = u[i].tok >= 0 && u[i].j >= u[i].lim;
                if (wv::any(ev)) unit_event(u[i], ev);
                const uint32_t avail = tail - issue;
                unsigned long long idle = wv::ballot(u[i].tok < 0);
                if (avail != 0 && idle != 0) {
                    const uint32_t r = wv::mbcnt(idle);
                    const bool take = u[i].tok < 0 && r < avail;
                    unit_begin(u[i], issue + r, take);
                    const uint32_t k = (uint32_t)__builtin_popcountll(idle);
                    issue += k < avail ? k : avail;
                    idle = wv::ballot(u[i].tok < 0);
                }
                nb += 64 - __builtin_popcountll(idle);
            }
            if (nb == 0) break;
            if (!drain && issue == tail && nb < UNIT_MIN) break;
            if (STATS) { ++st_trips; st_steps += (unsigned long long)nb; }
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
#pragma unroll
                for (int i = 0; i < NU; ++i) {
                    if (STATS) { st_gath += 64; st_trans += (unsigned long long)__builtin_popcountll(wv::ballot(u[i].j < u[i].lim)); }
                    unit_step(u[i]);
                }
            }
            ran = true;
        }
        q_issue = issue;
        // what the units that stay busy still read of the ring (the queue entry of a token a unit has taken no longer holds its position)
        u_need = 0xFFFFFFFFu;
        {
            uint32_t need = 0xFFFFFFFFu;
#pragma unroll
            for (int i = 0; i < NU; ++i) if (u[i].tok >= 0) { const uint32_t d = u[i].rs - rlo; need = d < need ? d : need; }
            if (wv::any(need != 0xFFFFFFFFu)) u_need = wv::min_all(need);
        }
        wv::sync();
        return ran;
    }
    static constexpr int UNIT_MIN = UMIN * NU;
    static constexpr int CHUNK_ROOM = CROOM > 0 ? CROOM : (QCAP >= 512 ? 256 : (QCAP * 15) / 32);

    // ------------------------------------------------------------------------------------------------------------------
    // retire (phase C): the finished tokens at the head of the queue, in order.  Position of a unit's ids = ids its document has
    // so far + ids of the units before it in the document (segmented prefix sum); the ids move from their provisional homes down
    // to that position (never up: a document has at most as many ids as characters before any point), all lanes reading before
    // any lane writes; nothing is written at or behind the document's cap (tokdll:1308-1310 is then the prefix rule: what a full
    // output array cuts off does not change what came before).  `all`: any finished prefix; else only a full group of 64.
    // Returns the number of tokens retired.
    // ------------------------------------------------------------------------------------------------------------------
    BF_WVD int retire(bool all)
    {
        const uint32_t navail = q_issue - q_retire < 64u ? q_issue - q_retire : 64u;       // tokens a unit has taken
        if (navail == 0 || (!all && navail < 64u)) return 0;
        const uint32_t t = q_retire + (uint32_t)lane, sl = t & QMASK;
        const int cnt0 = (uint32_t)lane < navail ? (int)S.qc[sl] - 1 : -1;
        const unsigned long long fin = wv::ballot(cnt0 >= 0);
        const int nret = fin == ~0ull ? 64 : __builtin_ctzll(~fin);                  // the finished prefix
        if (nret == 0 || (!all && nret < 64)) return 0;
        if (DBG >= 2) { q_retire += (uint32_t)nret; return nret; }
        if (STATS) ++st_ret;
        const bool act = lane < nret;
        const int cnt = act ? cnt0 : 0;
        const WvTok te = S.q[sl];
        const int k = act ? (int)((te.w >> 16) & 0xFFu) : -1;
        const uint32_t ke = (uint32_t)k & DMASK;
        int64_t slot = 0; int cap = 0, dcnt = 0; uint32_t f = 0;
        uint32_t w0 = 0;
        if (act) { slot = S.dt_slot[ke]; cap = S.dt_cap[ke]; dcnt = S.dt_cnt[ke]; w0 = te.pos; f = w0 - S.dt_rbase[ke]; }
        const int32_t *src = ids_tmp + slot + (int64_t)f;          // used by words of two and more pieces only (unit_event); OFFS: by every word
        int32_t v0 = (int32_t)w0, v1 = 0, v2 = 0, v3 = 0;            // a single piece is in q0 itself
        if (cnt > 1) { v0 = src[0]; v1 = src[1]; }
        if (OFFS && cnt == 1) v0 = (int32_t)S.qid[OFFS ? sl : 0];
        if (cnt > 2) v2 = src[2];
        if (cnt > 3) v3 = src[3];
        int sa0 = 0, sb0 = 0, sa1 = 0, sb1 = 0, sa2 = 0, sb2 = 0, sa3 = 0, sb3 = 0;
        if (OFFS) {
            const int32_t *sp = p.span_tmp + 2 * (slot + (int64_t)f);
            if (cnt == 1) { sa0 = (int)f; sb0 = (int)f + (int)(te.w & WV_TK_LEN_MASK) - 1; }      // the word's own span
            if (cnt > 1) { sa0 = sp[0]; sb0 = sp[1]; sa1 = sp[2]; sb1 = sp[3]; }
            if (cnt > 2) { sa2 = sp[4]; sb2 = sp[5]; }
            if (cnt > 3) { sa3 = sp[6]; sb3 = sp[7]; }
        }
        const int inc = wv::incl_scan(cnt), exc = inc - cnt;
        const int kp = wv::shfl_up(k, 1), kn = wv::shfl_down(k, 1);
        const unsigned long long hm = wv::ballot(lane == 0 || k != kp);
        const int head = 63 - __builtin_clzll(hm & ((2ull << lane) - 1ull));
        const int segbase = wv::shfl(exc, head);
        const int pos = dcnt + (exc - segbase);
        if (TRIM & 8) {                                               // one wait for the loads above; the stores below then leave back to back
            wv::arrived(v0, v1, v2, v3);
            if (OFFS) { wv::arrived(sa0, sb0, sa1, sb1); wv::arrived(sa2, sb2, sa3, sb3); }
        }
        wv::sync();                                                   // every lane has read its document's count and its first four ids
        if (act && (lane == 63 || k != kn)) S.dt_cnt[ke] = pos + cnt;
        const int room = cap - pos;
        // words of more than four pieces first, one after the other and in order, the whole wave moving 64 ids at a time (reads
        // before writes, lowest first): the place of a LATER token may reach into the provisional home of such a word's last ids
        unsigned long long big = wv::ballot(cnt > 4);
        while (big) {
            const int l = __builtin_ctzll(big); big &= big - 1ull;
            const int bc = wv::bcast(cnt, l), br = wv::bcast(room, l);
            const int64_t bs = wv::bcast(slot + (int64_t)f, l), bd = wv::bcast(slot + (int64_t)pos, l);
            if (STATS && lane == 0) ++st_rewalk;
            for (int o = 4; o < bc; o += 64) {
                const int i = o + lane;
                int32_t v = 0, va = 0, vb = 0;
                if (i < bc) v = ids_tmp[bs + i];
                if (OFFS && i < bc) { va = p.span_tmp[2 * (bs + i)]; vb = p.span_tmp[2 * (bs + i) + 1]; }
                wv::sync();
                if (i < bc && i < br) ids_tmp[bd + i] = v;
                if (OFFS && i < bc && i < br) { p.span_tmp[2 * (bd + i)] = va; p.span_tmp[2 * (bd + i) + 1] = vb; }
                wv::sync();
            }
        }
        int32_t *dst = ids_tmp + slot + pos;
        if (cnt > 0 && room > 0) dst[0] = v0;
        if (cnt > 1 && room > 1) dst[1] = v1;
        if (cnt > 2 && room > 2) dst[2] = v2;
        if (cnt > 3 && room > 3) dst[3] = v3;
        if (OFFS) {
            int32_t *sd = p.span_tmp + 2 * (slot + pos);
            if (cnt > 0 && room > 0) { sd[0] = sa0; sd[1] = sb0; }
            if (cnt > 1 && room > 1) { sd[2] = sa1; sd[3] = sb1; }
            if (cnt > 2 && room > 2) { sd[4] = sa2; sd[5] = sb2; }
            if (cnt > 3 && room > 3) { sd[6] = sa3; sd[7] = sb3; }
        }
        q_retire += (uint32_t)nret;
        wv::sync();
        return nret;
    }
    // documents that are complete (closed, and in front of the document of the oldest token still queued) get their count; the
    // ring is needed from the oldest queued token on (or from the first unresolved position of the open document)
    BF_WVD bool settle()
    {
        bool moved = false;
        uint32_t limit;
        if (q_retire != q_tail) limit = dt_head + ((((S.q[q_retire & QMASK].w >> 16) & 0xFFu) - dt_head) & 0xFFu);
        else limit = have_doc ? curk : dt_tail;
        if (limit != dt_head) {
            const uint32_t kk = dt_head + (uint32_t)lane;
            if (kk - dt_head < limit - dt_head) {
                const uint32_t e = kk & DMASK, f = S.dt_flags[e];
                const int c = S.dt_cnt[e], cap = S.dt_cap[e];
                p.counts[S.dt_doc[e]] = (f & WV_DT_BAD) ? 0 : (c < cap ? c : cap);
            }
            dt_head = limit; moved = true;
        }
        // the ring is needed from: the units still busy, the oldest token no unit has taken, the open document's unresolved part
        const uint32_t old_lo = rlo;
        uint32_t keep = q_issue != q_tail ? S.q[q_issue & QMASK].pos - old_lo : (have_doc ? rbase + (uint32_t)(open_start >= 0 ? open_start : done) : rhi) - old_lo;
        if (u_need < keep) keep = u_need;
        rlo = old_lo + keep; u_need -= u_need == 0xFFFFFFFFu ? 0u : keep;
        return moved || keep != 0;
    }

    // starts the document [b, e) of the text; false: nothing to tokenise (its count is written here)
    BF_WVD bool open_document(int64_t d, int64_t b, int64_t e)
    {
        const int64_t n64 = e - b;
        if (n64 <= 0 || n64 > 1000000000) { if (lane == 0) p.counts[d] = 0; return false; }                                // tokdll:1121
        if (b < 0 || b + n64 > p.total_bytes) { if (lane == 0) { p.counts[d] = 0; wv::atomic_or(cold.status, BF_STATUS_BAD_OFFSETS); } return false; }
        n = (int)n64; s = p.text + b;
        int cap = p.max_ids; if ((int64_t)cap > n64) cap = n; if (cap < 0) cap = 0;
        curk = dt_tail++;
        const uint32_t ke = curk & DMASK;
        rhi = (rhi + 7u) & ~7u; rbase = rhi;
        if (lane == 0) { S.dt_slot[ke] = wv_ids_slot(b, d); S.dt_doc[ke] = d; S.dt_cap[ke] = cap; S.dt_cnt[ke] = 0; S.dt_flags[ke] = 0; S.dt_rbase[ke] = rbase; }
        dec_bytes = dec = done = bom = 0; open_start = -1; err = false;
        wv::sync();
        return true;
    }
    BF_WVD void close_document()
    {
        // a run that was still open when the decoded text ended without another element (possible only behind invalid UTF-8 or a
        // character that straddles the last chunk boundary)
        if (open_start >= 0) { if (lane == 0) put_word(q_tail, open_start, dec - open_start, false); ++q_tail; open_start = -1; }
        const bool bad = wv::any(err);
        if (lane == 0) S.dt_flags[curk & DMASK] = WV_DT_CLOSED | (bad ? WV_DT_BAD : 0u);
        wv::sync();
    }

    // one producing action: take a range of documents, open / decode / resolve a window / close.  false: nothing can be produced now
    // (the queue, the document table or the ring is full, or the input is exhausted)
    BF_WVD bool fill_step(int grab)
    {
        if (!have_doc) {
            if (exiting || dt_tail - dt_head >= (uint32_t)DTN) return false;
            if (di >= dn) {
                wv::sync();                                             // every lane has read the offsets of the range before (an empty last document returns without a hand-off)
                unsigned long long base = 0;
                if (p.next_doc) {
                    if (lane == 0) base = wv::atomic_add(p.next_doc, (unsigned long long)grab);
                    base = wv::bcast(base, 0);
                } else { base = ((unsigned long long)st_wave + (unsigned long long)st_round * (unsigned long long)st_waves) * (unsigned long long)grab; ++st_round; }   // no work counter: ranges dealt out round-robin (small batches)
                if ((int64_t)base >= nd_all) { exiting = true; return false; }
                dbase = (int64_t)base; di = 0; dn = dbase + grab < nd_all ? grab : (int)(nd_all - dbase);
                if (LIST) { list_doc = (int64_t)p.doc_list[dbase]; if (lane <= 1) S.doff[lane] = p.doc_off[list_doc + lane]; }
                else if (lane <= dn) S.doff[lane] = p.doc_off[dbase + lane];
                wv::sync();
            }
            const int64_t b = S.doff[di], e = S.doff[di + 1];
            have_doc = open_document(LIST ? list_doc : dbase + di, b, e);
            ++di;
            return true;
        }
        const bool fully = dec_bytes >= n;
        const bool room_q = (q_tail - q_retire) + 66u <= (uint32_t)QCAP;
        if (done < dec) {
            if (phase_a_wide(fully)) return true;
            if (!room_q) return false;
            if (open_start >= 0) { done = open_start; open_start = -1; }     // the general form starts at a certain start position
            if (phase_a_general(fully)) return true;
        }
        if (dec_bytes < n) {
            // a chunk is taken when the ring has room for it and the queue for the tokens it usually holds (the chunk-wide pass writes them at once)
            if (ring_free() < WV_CHUNK || (q_tail - q_retire) + (uint32_t)CHUNK_ROOM > (uint32_t)QCAP) return false;
            const bool fresh = done == dec;                            // nothing older is unresolved: the wide pass covers exactly the new chunk
            uint32_t kk = 0;
            const bool ascii_chunk = decode_chunk(kk);
            if (fresh && ascii_chunk) (void)phase_a_wide(dec_bytes >= n, true, kk);    // declined (a long run, a full queue): the next step sorts it out
            return true;
        }
        if (done >= dec) { if (open_start >= 0 && !room_q) return false; close_document(); have_doc = false; return true; }
        return false;
    }

    // The same producing actions in the same order as `while (fill_step(grab))`, written as nested loops (TRIM bit 4): documents outside, the
    // chunks of a document inside, the usual chunk -- decoded, resolved by the chunk-wide pass, the document closed behind it -- in one
    // run of straight-line code.  (fill_step() is re-entered through one dispatching head per action, and every way into that head
    // carries the wave-uniform state in another set of scalar registers: ~28 scalar copies per action, five actions per document.)
    // Returns whether anything was produced.
    BF_WVD bool fill_nested(int grab)
    {
        bool filled = false;
        for (;;) {
            if (!have_doc) {
                if (exiting || dt_tail - dt_head >= (uint32_t)DTN) return filled;
                if (di >= dn) {
                    wv::sync();                                             // every lane has read the offsets of the range before
                    unsigned long long base = 0;
                    if (p.next_doc) {
                        if (lane == 0) base = wv::atomic_add(p.next_doc, (unsigned long long)grab);
                        base = wv::bcast(base, 0);
                    } else { base = ((unsigned long long)st_wave + (unsigned long long)st_round * (unsigned long long)st_waves) * (unsigned long long)grab; ++st_round; }
                    if ((int64_t)base >= nd_all) { exiting = true; return filled; }
                    dbase = (int64_t)base; di = 0; dn = dbase + grab < nd_all ? grab : (int)(nd_all - dbase);
                    if (LIST) { list_doc = (int64_t)p.doc_list[dbase]; if (lane <= 1) S.doff[lane] = p.doc_off[list_doc + lane]; }
                    else if (lane <= dn) S.doff[lane] = p.doc_off[dbase + lane];
                    wv::sync();
                    filled = true;
                }
                const int64_t b = S.doff[di], e = S.doff[di + 1];
                have_doc = open_document(LIST ? list_doc : dbase + di, b, e);
                ++di;
                filled = true;
                if (!have_doc) continue;
            }
            // ---- the open document: resolve what is decoded, decode the next chunk, close
            for (;;) {
                const bool room_q = (q_tail - q_retire) + 66u <= (uint32_t)QCAP;
                if (done < dec) {
                    const bool fully = dec_bytes >= n;
                    if (phase_a_wide(fully)) { filled = true; continue; }
                    if (!room_q) return filled;
                    if (open_start >= 0) { done = open_start; open_start = -1; }     // the general form starts at a certain start position
                    if (phase_a_general(fully)) { filled = true; continue; }
                }
                if (dec_bytes < n) {
                    if (ring_free() < WV_CHUNK || (q_tail - q_retire) + (uint32_t)CHUNK_ROOM > (uint32_t)QCAP) return filled;
                    const bool fresh = done == dec;
                    uint32_t kk = 0;
                    const bool ascii_chunk = decode_chunk(kk);
                    if (fresh && ascii_chunk) (void)phase_a_wide(dec_bytes >= n, true, kk);
                    filled = true;
                    continue;
                }
                if (done >= dec) {
                    if (open_start >= 0 && !room_q) return filled;
                    close_document(); have_doc = false; filled = true;
                    break;
                }
                return filled;                                             // the token at `done` waits for room (cannot happen once the document is fully decoded: see run())
            }
        }
    }

    // wave_id / n_waves: this wave's number and the number of waves of the launch (used when the batch has no work counter)
    BF_WVD void run(int grab, int wave_id, int n_waves)
    {
        grab = LIST ? 1 : grab < 1 ? 1 : (grab > WV_GRAB_MAX ? WV_GRAB_MAX : grab);
        st_wave = wave_id; st_waves = n_waves; st_round = 0;
        Unit u[NU];
#pragma unroll
        for (int i = 0; i < NU; ++i) { u[i].tok = -1; u[i].rs = 0; u[i].j = 0; u[i].state = 0; u[i].lim = 0; u[i].fp = -1; u[i].ftag = 0; u[i].Lk = 0; u[i].ca = 0; u[i].ini = 0; u[i].from = 0; }
        for (;;) {
            bool moved = (TRIM & 1) ? false : settle();       // (the settle at the bottom of the trip before has just run; the first trip has nothing to settle)
            bool filled = false;
            if (TRIM & 4) filled = fill_nested(grab); else
            while (fill_step(grab)) filled = true;
            const bool drain = !filled;          // nothing could be produced: what blocks is freed only by finishing and retiring tokens
            if (units_phase(u, drain)) moved = true;
            for (;;) { const int r = retire(drain); if (r == 0) break; moved = true; if (r < 64) break; }
            if (settle()) moved = true;
            if (exiting && !have_doc && q_tail == q_retire && dt_head == dt_tail) break;
            if (!filled && !moved) {
                // nothing can move: cannot happen (the ring holds a whole token and a chunk, the queue a whole window; checked at load)
                if (STATS) ++st_idle;
                if (lane == 0) wv::atomic_or(cold.status, BF_STATUS_INTERNAL);
                break;
            }
        }
        if (STATS && lane == 0) {
            wv::atomic_add(&cold.stats[0], st_trips); wv::atomic_add(&cold.stats[1], st_win); wv::atomic_add(&cold.stats[2], st_slow); wv::atomic_add(&cold.stats[3], (unsigned long long)q_tail);
            wv::atomic_add(&cold.stats[4], st_steps); wv::atomic_add(&cold.stats[5], st_ret); wv::atomic_add(&cold.stats[7], st_idle); wv::atomic_add(&cold.stats[8], st_dec);
            wv::atomic_add(&cold.stats[9], st_gath); wv::atomic_add(&cold.stats[10], st_trans);
        }
        if (STATS) { const unsigned long long r = st_rewalk; if (r) wv::atomic_add(&cold.stats[6], r); }
    }
};

} // namespace bfa
