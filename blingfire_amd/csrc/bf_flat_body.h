// bf_flat_body.h -- the flat wave program of bf_flat.h (see there).  Include AFTER a definition of namespace wv (bf_kernels.hip: the wave
// intrinsics of gfx950; tests/hosttest/wave_emu.h: the 64-fibre simulator).
//
// One wave takes ranges of documents from a work counter.  A range is a stream of 512-byte chunks, eight bytes per lane, positions counted
// in BYTES from the range's first byte:
//   decode    plain ASCII (one vote): eight look-ups per lane in a 128-entry table give class, kind and key code of every byte; else every lead
//             byte is decoded by its lane (strict UTF-8, the rules of FAUtf8Utils.cpp:121-196 per byte position), its continuation bytes
//             belong to its character.  The class of every byte goes to a ring in LDS (what a unit reads, should the word need one);
//   tokens    kinds as one bit per byte in 4-bit fields: run starts / ends from shifts, a document boundary is one more bit that cuts
//             runs (FALexTools_t.h:229-393 on a unit-form lexer is a function of the kinds alone, bf_wave.h); every lane owns the tokens
//             that END in its eight bytes; a prefix sum gives each its entry;
//   keys      every lane cuts the keys of its tokens out of the packed codes of its bytes and the eight before them (a run of <= 9 plain
//             characters: the key IS the word) and writes them, in token order, to a list in LDS;
//   look-up   the list two tokens per lane: four 12-byte gathers in flight per lane, ids to the entries as whole rows;
//   units     the tokens the table did not answer wait as records (a plain-ASCII word: where its bytes are in the text; any other: its
//             characters copied to an arena); 64 of them are walked at once (wf_drain: the frame of one call of the vocabulary function,
//             bf_wave_body.h Unit restated without a queue), pieces to the word's home, the entry says how many.
// What couples documents is left to the kernels behind (k_wp_count, k_wp_merge): a document's entries are dense, its ids are not yet.
#pragma once
#include "bf_flat.h"

namespace bfa {

constexpr int WF_TQ = 192;                // tokens of a chunk the list holds (three per lane); a chunk with more hands its documents on
constexpr uint64_t WF_KEY_NONE = 1ull << 62;     // "no key": matches no entry of the table (an entry's lowest field is never 0)
constexpr uint32_t WF_REC_TEXT = 0x80000000u;    // record word 2: the word's characters are its bytes in the text (plain ASCII, <= 16 bytes); else arena base | characters << 16

struct WfLds {
    alignas(16) uint16_t ring[WF_RING];  // class of every byte position of this chunk and the one before (WF_CONT: no character starts there)
    uint32_t tq_lo[WF_TQ], tq_hi[WF_TQ]; uint16_t tq_pos[WF_TQ];      // the chunk's tokens in order: key, (first byte - (chunk - 64)) | bytes << 10 (0 bytes: a run of more than WF_RUN_MAX; then its last byte)
    uint32_t rec[WF_REC * 3];            // a word that waits for a unit: entry (range-relative), first byte (range-relative), where its characters are
    uint16_t arena[WF_ARENA];            // characters of the waiting words that are not plain text
    int arena_n;                         // characters in the arena
    uint32_t spare32; uint16_t spare;
};

// the 128-entry table of the ASCII bytes: [12:0] class, [18:16] kind bits (loop, solo, general), [26:20] key code
BF_WV uint32_t wf_lut_value(const WpWaveCold &p, int b)
{
    const uint32_t el = wv_element(p, b), c = el & LX_T_CLS_MASK, k = el >> WK_SHIFT;
    const uint32_t nib = k == WK_LOOP ? 1u : k == WK_SOLO ? 2u : k == WK_GENERAL ? 4u : 0u;
    return c | (nib << 16) | (wf_code(c, k) << 20);
}

#if defined(__HIPCC__)
#define BF_WF_NOINLINE __device__ __noinline__
#else
#define BF_WF_NOINLINE static __attribute__((noinline))
#endif

// ------------------------------------------------------------------------------------------------------------------
// units: the words that wait (records 0 .. n), all at once.  The frame of ONE call of the vocabulary function on a word of L characters
// (FALexTools_t.h:229-393 at depth 1; L < max-token-length, so every walk's limit is the word's end): first the walk from the state behind
// the left anchor at character 0, if the function has one; a walk that ends with a match is a piece and the next walk starts behind it
// (:390-393); the anchored walk without a match is followed by the plain walk at 0 (:293); any other walk without a match leaves a gap:
// the pieces cannot tile the word, its id is UnkId (tokdll:1252-1301).  Restated from bf_wave_body.h Unit / unit_step / unit_event.
// A function of its own (not inlined): it runs once per ~10 chunks, and what it keeps in registers must not press on the chunk code.
// ------------------------------------------------------------------------------------------------------------------
template <bool STATS>
BF_WF_NOINLINE void wf_drain(const uint64_t *T, uint32_t ini, uint32_t ini_l, int max_token_length, WfLds &S, const uint32_t *lut, const uint8_t *txt, uint32_t *ent, int32_t *home,
                             int n, unsigned long long *rounds)
{
    const int lane = wv::lane();
    const bool have = lane < n;
    const uint32_t e_rel = have ? S.rec[3 * lane] : 0u, p_rel = have ? S.rec[3 * lane + 1] : 0u, w2 = have ? S.rec[3 * lane + 2] : 0u;
    const bool is_text = (w2 & WF_REC_TEXT) != 0u;
    const uint32_t abase = w2 & 0xFFFFu; const int L = is_text ? (int)(w2 & 0xFFu) : (int)(w2 >> 16);
    // a plain-text word: its (at most 16) bytes in two registers, every byte's class through the table of the ASCII bytes
    uint64_t t_lo = 0, t_hi = 0;
    if (is_text) { __builtin_memcpy(&t_lo, txt + p_rel, 8); __builtin_memcpy(&t_hi, txt + p_rel + 8, 8); }
    const bool anchored0 = ini_l != LX_NO_STATE && max_token_length > 1;
    uint32_t state = anchored0 ? ini_l : ini; int j = 0, fp = -1, cnt = 0; uint32_t ftag = 0; int32_t id0 = 0;
    bool anch = anchored0, act = have, missed = false;
    int32_t *hm = home + p_rel;
    while (wv::any(act)) {
        if (STATS) ++*rounds;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool walking = act && !missed && j < L;
            const uint32_t tb = (uint32_t)((j < 8 ? t_lo : t_hi) >> (8 * (j & 7))) & 0x7Fu;
            const uint16_t *src = is_text ? (const uint16_t *)(lut + tb) : &S.arena[walking ? abase + (uint32_t)j : 0u];
            const uint32_t c = (uint32_t)*src & LX_T_CLS_MASK;
            const uint64_t e64 = T[walking ? state + c : 0u];
            const uint32_t e = (uint32_t)e64;
            const bool hit = walking && (e & LX_T_CLS_MASK) == c;
            const bool fin = hit && (int32_t)e < 0;
            fp = fin ? j : fp; ftag = fin ? (uint32_t)(e64 >> 32) : ftag;
            state = hit ? ((e >> LX_T_NEXT_SHIFT) & LX_T_NEXT_MASK) : state;
            j = hit ? j + 1 : j;
            missed = missed || (walking && !hit);
        }
        const bool ev = act && (missed || j >= L);
        if (wv::any(ev)) {
            if (ev) {
                const bool matched = fp >= 0;
                if (matched) {
                    const int32_t id = (int32_t)(ftag & 0x7FFFFFFFu);
                    if (cnt == 0) id0 = id;
                    else { if (cnt == 1) hm[0] = id0; hm[cnt] = id; }
                    ++cnt;
                    const int nf = fp + 1;
                    if (nf >= L) {
                        ent[e_rel] = cnt == 1 ? (uint32_t)id0 : (WF_ENT_FLAG | ((uint32_t)cnt << WF_ENT_CNT_SHIFT) | (p_rel - e_rel));
                        act = false;
                    } else { state = ini; j = nf; fp = -1; anch = false; missed = false; }
                } else if (anch) { state = ini; j = 0; fp = -1; anch = false; missed = false; }
                else { ent[e_rel] = WF_ENT_FLAG; act = false; }            // a gap: UnkId
            }
        }
    }
    wv::sync();
}

// The tokens the table did not answer (`miss`: this lane holds one; MB = the lanes that do) become waiting words.  A word of plain
// ASCII that lies inside this chunk is walked from the text itself (`text_ok`); of any other the characters (not the continuation
// bytes) are copied from the ring to the arena.  A full record table or arena is drained first; `flush`: and at the end.  nrec_in: words that
// wait already; returns how many do now | drains << 8.
template <bool STATS>
BF_WF_NOINLINE int wf_words(unsigned long long MB, bool miss, uint32_t start, int blen, uint32_t rank, bool text_ok, bool flush, int nrec_in,
                            const uint64_t *T, uint32_t ini, uint32_t ini_l, int max_token_length, WfLds &S, const uint32_t *lut, const uint8_t *txt, uint32_t *ent, int32_t *home, unsigned long long *rounds)
{
    constexpr uint32_t RMASK = WF_RING - 1;
    const int lane = wv::lane();
    wv::sync();                                                  // (records may have been added in line)
    int nrec = wv::uni(nrec_in), arena_n = wv::uni(S.arena_n), drains = 0;
    unsigned long long todo = MB;
    bool full = false;
    while (todo || flush) {
        if (full || !todo) {                                     // no room for the next word -- or, at the end of a range, nothing left to add: the units run
            if (nrec) { wf_drain<STATS>(T, ini, ini_l, max_token_length, S, lut, txt, ent, home, nrec, rounds); ++drains; }
            nrec = 0; arena_n = 0; full = false;
            if (!todo) break;
            continue;
        }
        const bool mine_todo = miss && ((todo >> lane) & 1ull) != 0ull;
        const int need = (mine_todo && !text_ok) ? blen : 0;
        const int inc = wv::incl_scan(need);
        const int ridx = (int)wv::mbcnt(todo);
        const unsigned long long fit = wv::ballot(mine_todo && inc <= WF_ARENA - arena_n && ridx < WF_REC - nrec);     // a prefix of `todo`: both grow with the lane
        if (!fit) { full = true; continue; }
        const bool mine = mine_todo && ((fit >> lane) & 1ull) != 0ull;
        const uint32_t ab = (uint32_t)(arena_n + inc - need);
        int w = 0;
        if (wv::any(mine && !text_ok)) {
            for (int t = 0; wv::any(mine && !text_ok && t < blen); ++t) {
                const bool on = mine && !text_ok && t < blen;
                const uint32_t el = S.ring[(start + (uint32_t)t) & RMASK];
                const bool ch = on && el != WF_CONT;
                uint16_t *dst = ch ? &S.arena[ab + (uint32_t)w] : &S.spare;
                *dst = (uint16_t)el;
                w += ch ? 1 : 0;
            }
        }
        if (mine) { uint32_t *r = S.rec + 3 * (nrec + ridx); r[0] = rank; r[1] = start; r[2] = text_ok ? (WF_REC_TEXT | (uint32_t)blen) : (ab | ((uint32_t)w << 16)); }
        arena_n += wv::bcast(inc, 63 - __builtin_clzll(fit)); nrec += __builtin_popcountll(fit); todo &= ~fit;
        wv::sync();
    }
    if (lane == 0) S.arena_n = arena_n;
    wv::sync();
    return nrec | (drains << 8);
}

// A chunk with bytes >= 0x80 (the caller has put the ASCII bytes' classes and WF_CONT for all others into the ring).  Every lead byte is decoded
// by its lane, one per trip; a continuation byte is legal exactly when it is one of the (length - 1) bytes behind a lead byte OF ITS DOCUMENT (a
// document boundary inside a character truncates it, FAUtf8Utils.cpp:167-171).  S8: the documents that begin in this lane's bytes; peek: in the
// three bytes behind the chunk.  Returns what the characters add to the kind bits, the bytes that are invalid, and what the last character of
// the chunk covers of the next one.
struct WfMb { uint32_t acc, errm, cov_carry, loop_carry; unsigned long long na; };      // na: the lanes that hold a byte >= 0x80
BF_WVD WfMb wf_decode_multibyte(uint64_t own, uint32_t S8, uint32_t peek, int c, int len, const uint8_t *txt, uint16_t *ring, const uint16_t *cp_l1, const uint32_t *cp_pages,
                                        const uint8_t *kind, int nclasses, uint32_t cov_carry, uint32_t loop_carry)
{
    const int lane = wv::lane(), lane0 = c + lane * 8;
    constexpr uint32_t RMASK = WF_RING - 1;
    int nb = len - lane0; nb = nb < 0 ? 0 : (nb > 8 ? 8 : nb);
    uint32_t acc = 0, errm = 0;
    WpWaveCold cold; cold.cpmap.l1 = cp_l1; cold.cpmap.pages = cp_pages; cold.kind = kind; cold.nclasses = nclasses; cold.status = nullptr; cold.stats = nullptr; cold.no_fast = 0;
    const uint32_t vm8 = nb >= 8 ? 0xFFu : ((1u << nb) - 1u);
    const uint64_t h80 = own & 0x8080808080808080ull, h40 = (own << 1) & 0x8080808080808080ull;
    uint32_t m80 = 0, m40 = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) { m80 |= (uint32_t)((h80 >> (8 * i + 7)) & 1ull) << i; m40 |= (uint32_t)((h40 >> (8 * i + 7)) & 1ull) << i; }
    const uint32_t contm = m80 & ~m40 & vm8, leadm = m80 & m40 & vm8;
    uint32_t nxt = wv::shfl_down((uint32_t)own, 1);
    uint32_t s_nx = wv::shfl_down(S8, 1) & 7u;                         // boundaries at the three bytes behind this lane's
    if (lane == 63) {
        nxt = 0; for (int i = 0; i < 3; ++i) if (lane0 + 8 + i < len) nxt |= (uint32_t)txt[lane0 + 8 + i] << (8 * i);
        s_nx = 0;
    }
    if (lane == 63) s_nx = peek;
    const uint32_t S11 = S8 | (s_nx << 8);
    uint32_t cov = 0, lsp = 0;                                // bytes behind a lead that belong to its character; the same for run membership
    for (uint32_t lm = leadm; wv::any(lm != 0);) {
        if (lm) {
            const int i = __builtin_ctz(lm); lm &= lm - 1u;
            const int q = lane0 + i;
            uint64_t w = own >> (8 * i);
            if (i) w |= (uint64_t)nxt << (64 - 8 * i);
            const uint32_t c0 = (uint32_t)w & 0xFF, c1 = (uint32_t)(w >> 8) & 0xFF, c2 = (uint32_t)(w >> 16) & 0xFF, c3 = (uint32_t)(w >> 24) & 0xFF;
            int cl, cp; bool er = false;
            if ((c0 & 0xE0) == 0xC0) { cl = 2; cp = (int)(c0 & 0x1F); }
            else if ((c0 & 0xF0) == 0xE0) { cl = 3; cp = (int)(c0 & 0x0F); }
            else if ((c0 & 0xF8) == 0xF0) { cl = 4; cp = (int)(c0 & 0x07); }
            else { cl = 1; cp = 0; er = true; }                                            // F8 .. FF
            if (q + cl > len) er = true;                                                   // the range ends inside the character: so does its document
            if (cl >= 2) { if ((c1 & 0xC0) != 0x80) er = true; cp = (cp << 6) | (int)(c1 & 0x3F); }
            if (cl >= 3) { if ((c2 & 0xC0) != 0x80) er = true; cp = (cp << 6) | (int)(c2 & 0x3F); }
            if (cl >= 4) { if ((c3 & 0xC0) != 0x80) er = true; cp = (cp << 6) | (int)(c3 & 0x3F); }
            const int need = cp <= 0x7F ? 1 : cp <= 0x7FF ? 2 : cp <= 0xFFFF ? 3 : cp <= 0x10FFFF ? 4 : 0;
            if (need != cl) er = true;                                                     // overlong / > U+10FFFF (:185-188)
            if ((cp & 0xFFFFF800) == 0xD800) er = true;                                    // surrogate (:190-193)
            uint32_t cb = (((1u << cl) - 1u) & ~1u) << i;                                  // its continuation bytes
            const uint32_t bb = S11 & cb;
            if (bb) { er = true; cb &= (bb & (0u - bb)) - 1u; }                            // a document begins inside the character
            cov |= cb;
            const bool is_bom = !er && cp == 0xFEFF && ((S8 >> i) & 1u);                    // one leading BOM is skipped (:247-252)
            uint32_t el = WF_CONT, nib = 0;
            if (!er && !is_bom) {
                const uint32_t e2 = wv_element(cold, cp), kd = e2 >> WK_SHIFT;
                el = e2 & LX_T_CLS_MASK; nib = kd == WK_LOOP ? 1u : kd == WK_SOLO ? 2u : kd == WK_GENERAL ? 4u : 0u;
            }
            ring[((uint32_t)lane0 + (uint32_t)i) & RMASK] = (uint16_t)el;
            acc |= nib << (4 * i);
            if (nib & 1u) lsp |= cb;                                                       // the continuation bytes of a run member are run members
            if (er) errm |= 1u << i;
        }
    }
    uint32_t sp_cov = wv::shfl_up(cov >> 8, 1), sp_loop = wv::shfl_up(lsp >> 8, 1);
    if (lane == 0) { sp_cov = cov_carry; sp_loop = loop_carry; }
    const uint32_t cov_out = wv::bcast(cov >> 8, 63), loop_out = wv::bcast(lsp >> 8, 63);
    errm |= contm & ~(cov | sp_cov);                                    // a continuation byte no lead accounts for (:152-165)
    const uint32_t lb = (lsp | sp_loop) & 0xFFu & contm;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc |= ((lb >> i) & 1u) << (4 * i);
    WfMb r; r.acc = acc; r.errm = errm; r.cov_carry = cov_out; r.loop_carry = loop_out; r.na = wv::ballot(m80 != 0);
    return r;
}

// the document of the range [dlo, dlo + dn) that owns byte `pos` (range-relative, wave-uniform) gets `flag`; returns where that document ends
BF_WF_NOINLINE int wf_mark_doc(const int64_t *doc_off, int32_t *dstat, int64_t dlo, int dn, int64_t b0, int pos, int flag)
{
    int lo = 0, hi = dn - 1;                                   // the last document whose first byte is <= pos
    while (lo < hi) {
        const int mid = lo + (hi - lo + 1) / 2;
        if (doc_off[dlo + mid] - b0 <= (int64_t)pos) lo = mid; else hi = mid - 1;
    }
    if (wv::lane() == 0) wv::atomic_or(&dstat[dlo + lo], flag);
    return (int)(doc_off[dlo + lo + 1] - b0);
}

template <bool STATS = false>
struct WfWave {
    static constexpr uint32_t RMASK = WF_RING - 1;
    const WfParams &p; WfLds &S; const uint32_t *lut; const WpWaveCold &cold;
    int lane;
    // ---- the range (wave-uniform)
    int64_t dlo, b0;                     // its first document; its first byte in the text
    int dn, dnext, wlo;                  // its documents; the next one whose first byte has not been met; the first one of the window (all relative to dlo)
    int len;                             // its bytes
    int32_t win;                         // per lane: first byte (range-relative) of document dlo + wlo + lane
    const uint8_t *txt; uint32_t *ent; int32_t *home;
    int k, kdoc;                         // tokens so far; tokens before the open document
    int open_start;                      // first byte of the run that reaches the end of the chunk before (-1: none)
    unsigned long long na_prev;          // the lanes of the chunk before that hold a byte >= 0x80
    uint64_t pcc63; uint32_t cov_carry, loop_carry;      // of lane 63 of the chunk before: packed codes; bytes of the next chunk that belong to its last character
    int nrec;                            // words that wait for a unit
    int bad_lo, bad_hi, hard_lo, hard_hi; // [lo, hi): bytes of the document that got the flag last (one look-up per document, mostly)
    unsigned long long st_chunks, st_ascii, st_tok, st_hit, st_notes, st_drains, st_rounds, st_hard;

    BF_WVD WfWave(const WfParams &p_, WfLds &S_, const uint32_t *lut_, const WpWaveCold &cold_) : p(p_), S(S_), lut(lut_), cold(cold_)
    {
        lane = wv::lane(); nrec = 0; if (lane == 0) S.arena_n = 0; wv::sync(); bad_lo = bad_hi = hard_lo = hard_hi = 0;
        st_chunks = st_ascii = st_tok = st_hit = st_notes = st_drains = st_rounds = st_hard = 0;
        dlo = 0; dn = dnext = wlo = 0; b0 = 0; len = 0; win = 0; txt = nullptr; ent = nullptr; home = nullptr; k = kdoc = 0; open_start = -1; pcc63 = 0; cov_carry = loop_carry = 0; na_prev = 0;
    }

    // first byte of document dlo + d (0 <= d <= dn), range-relative; d is wave-uniform
    BF_WVD int off_rel(int d)
    {
        const uint32_t i = (uint32_t)(d - wlo);
        if (i < 64u) return wv::bcast(win, (int)i);
        return (int)(p.doc_off[dlo + d] - b0);
    }
    BF_WVD void load_window(int from)
    {
        wlo = from;
        const int d = from + lane;
        win = (int)(p.doc_off[dlo + (d <= dn ? d : dn)] - b0);
    }
    // this lane's 8 bytes of the chunk at c
    BF_WVD uint64_t load_chunk(int c) const
    {
        const int q0 = c + lane * 8;
        uint64_t own = 0;
        int nb = len - q0; nb = nb < 0 ? 0 : (nb > 8 ? 8 : nb);
        if (nb == 8) __builtin_memcpy(&own, txt + q0, 8);
        else for (int i = 0; i < nb; ++i) own |= (uint64_t)txt[q0 + i] << (8 * i);
        return own;
    }
    BF_WVD void mark(int pos, int flag)
    {
        if (flag == WF_D_BAD ? (pos >= bad_lo && pos < bad_hi) : (pos >= hard_lo && pos < hard_hi)) return;      // between a byte of a marked document and its end
        const int e = wf_mark_doc(p.doc_off, p.dstat, dlo, dn, b0, pos, flag);
        if (flag == WF_D_BAD) { bad_lo = pos; bad_hi = e; } else { hard_lo = pos; hard_hi = e; }
        if (STATS && flag == WF_D_HARD) ++st_hard;
    }
    // every byte of the chunk at c whose bit is set in a lane's mask (bit 8 of lane 0: the byte before the chunk)
    BF_WVD void mark_bytes(int c, uint32_t mm, int flag)
    {
        unsigned long long lb = wv::ballot(mm != 0);
        while (lb) {
            const int l = __builtin_ctzll(lb); lb &= lb - 1ull;
            uint32_t m = wv::bcast(mm, l);
            while (m) { const int bit = __builtin_ctz(m); m &= m - 1u; mark(c + l * 8 + (bit == 8 ? -1 : bit), flag); }
        }
    }
    // (see wf_words)
    BF_WVD void add_words(unsigned long long MB, bool miss, uint32_t start, int blen, uint32_t rank, bool text_ok, bool flush)
    {
        const int r = wf_words<STATS>(MB, miss, start, blen, rank, text_ok, flush, nrec, p.T, p.ini, p.ini_l, p.max_token_length, S, lut, txt, ent, home, STATS ? &st_rounds : nullptr);
        nrec = r & 0xFF;
        if (STATS) st_drains += (unsigned long long)(r >> 8);
    }

    BF_WVD void emit_boundary(int d, int kd)
    {
        if (lane == 0) { p.ent_off[dlo + d] = b0 + (int64_t)kd; if (d > 0) p.ent_cnt[dlo + d - 1] = kd - kdoc; }
        kdoc = kd;
    }

    // ------------------------------------------------------------------------------------------------------------------
    // one chunk: `own` = this lane's eight bytes at c + 8 * lane
    // ------------------------------------------------------------------------------------------------------------------
    BF_WVD void chunk(const int c, const uint64_t own)
    {
        if (STATS) ++st_chunks;
        const int lane0 = c + lane * 8;
        int nb = len - lane0; nb = nb < 0 ? 0 : (nb > 8 ? 8 : nb);
        const uint32_t vm4 = nb >= 8 ? 0x11111111u : (((1u << (4 * nb)) - 1u) & 0x11111111u);
        const bool at_end = c + WF_CHUNK >= len;
        // ---- the documents that begin in this chunk: one bit per byte, as 4-bit fields (S4) and packed (S8)
        uint32_t S4 = 0, S8 = 0;
        const int dfirst = dnext;
        while (dnext < dn) {
            if (dnext - wlo >= 64) load_window(dnext);
            const int o = off_rel(dnext);
            if (o >= c + WF_CHUNK) break;
            const int r = o - c;
            if (lane == (r >> 3)) { S4 |= 1u << (4 * (r & 7)); S8 |= 1u << (r & 7); }
            ++dnext;
        }
        // ---- decode
        uint32_t acc = 0, cclo = 0, cchi = 0;
        unsigned long long na = 0;
        const bool ascii_chunk = !wv::any((own & 0x8080808080808080ull) != 0);
        {
            uint32_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint32_t b = (uint32_t)(own >> (8 * i)) & 0xFFu;
                v[i] = lut[b & 0x7Fu];
                if (!ascii_chunk) v[i] = b < 0x80u ? v[i] : WF_CONT;
            }
            uint32_t *row = (uint32_t *)(S.ring + ((uint32_t)lane0 & RMASK));        // 8 positions = one 16-byte row, never wraps
            row[0] = (v[0] & 0xFFFFu) | (v[1] << 16); row[1] = (v[2] & 0xFFFFu) | (v[3] << 16); row[2] = (v[4] & 0xFFFFu) | (v[5] << 16); row[3] = (v[6] & 0xFFFFu) | (v[7] << 16);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc |= ((v[i] >> 16) & 7u) << (4 * i);
#pragma unroll
            for (int i = 0; i < 4; ++i) { cclo |= ((v[i] >> 20) & 0x7Fu) << (7 * i); cchi |= ((v[i + 4] >> 20) & 0x7Fu) << (7 * i); }
        }
        if (ascii_chunk) { if (STATS) ++st_ascii; cov_carry = 0; loop_carry = 0; }
        else {
            // lane 63: the documents that begin in the first three bytes of the next chunk (not consumed here)
            uint32_t peek = 0;
            for (int d = dnext; d < dn; ++d) { const int o = off_rel(d); if (o >= c + WF_CHUNK + 3) break; peek |= 1u << (o - (c + WF_CHUNK)); }
            const WfMb r = wf_decode_multibyte(own, S8, peek, c, len, txt, S.ring, cold.cpmap.l1, cold.cpmap.pages, cold.kind, cold.nclasses, cov_carry, loop_carry);
            acc |= r.acc; cov_carry = r.cov_carry; loop_carry = r.loop_carry; na = r.na;
            if (wv::any(r.errm != 0)) mark_bytes(c, r.errm, WF_D_BAD);          // invalid UTF-8: the document has no ids (tokdll:1151-1153)
        }
        wv::sync();                                                             // the ring is written
        const uint64_t cc = (uint64_t)cclo | ((uint64_t)cchi << 28);
        uint64_t pcc = (uint64_t)wv::shfl_up((unsigned long long)cc, 1);
        if (lane == 0) pcc = pcc63;
        pcc63 = (uint64_t)wv::bcast((unsigned long long)cc, 63);

        // ---- tokens: every lane owns the runs that END in its bytes and its one-element tokens
        const uint32_t Lm = acc & 0x11111111u & vm4, SO = (acc >> 1) & 0x11111111u & vm4;
        {
            const uint32_t G = (acc >> 2) & 0x11111111u & vm4;                  // an element the automaton itself must decide: the document goes to the wave program
            if (wv::any(G != 0)) { uint32_t g8 = 0; for (int i = 0; i < 8; ++i) g8 |= ((G >> (4 * i)) & 1u) << i; mark_bytes(c, g8, WF_D_HARD); }
        }
        uint32_t pl = wv::shfl_up(Lm >> 28, 1), nf = wv::shfl_down(Lm & ~S4 & 1u, 1);
        if (lane == 0) pl = open_start >= 0 ? 1u : 0u;
        if (lane == 63) nf = 0u;
        const uint32_t h = Lm & (~((Lm << 4) | pl) | S4);                       // run starts
        uint32_t en = Lm & ~(((Lm & ~S4) >> 4) | (nf << 28));                   // run ends
        const bool stays_open = !at_end && wv::any(lane == 63 && (en >> 28) != 0u);
        if (stays_open && lane == 63) en &= 0x0FFFFFFFu;
        const bool carry_end = open_start >= 0 && !wv::any(lane == 0 && (Lm & ~S4 & 1u) != 0u);    // the open run ended with the chunk before
        const int hl = h ? lane0 + ((31 - __builtin_clz(h)) >> 2) : -1;          // this lane's last run start
        const unsigned long long HB = wv::ballot(h != 0);
        const unsigned long long hb_lt = HB & ((1ull << lane) - 1ull);
        int hprev = wv::shfl(hl, hb_lt ? 63 - __builtin_clzll(hb_lt) : 0);
        if (!hb_lt) hprev = open_start;
        const int new_open = stays_open ? (HB ? wv::bcast(hl, 63 - __builtin_clzll(HB)) : open_start) : -1;
        const uint32_t tk0 = en | SO;
        const int cnt = __builtin_popcount(tk0);
        const int inc = wv::incl_scan(cnt);
        const int coff = carry_end ? 1 : 0;
        const int ntok = wv::bcast(inc, 63) + coff;
        const int excl = inc - cnt;
        if (STATS) st_tok += (unsigned long long)ntok;
        const bool over = ntok > WF_TQ;                                          // more tokens than the list holds: every document of the chunk is handed on
        // (1) every lane writes the keys of ITS tokens to the list, in order (no memory but LDS)
        {
            // this lane's tokens as bits 4 i + 1 (the token that ends at byte i); bit 0 of lane 0: the run that ended with the chunk before
            uint32_t tk = (tk0 << 1) | ((carry_end && lane == 0) ? 1u : 0u);
            int slot = (lane == 0 ? 0 : coff) + excl;
            while (wv::any(tk != 0)) {
                const bool on = tk != 0;
                const int bit = __builtin_ctz(tk | 0x80000000u);
                tk &= tk - 1u;
                const bool cr = (bit & 3) == 0;
                const int idx = bit >> 2;
                const int bpos = cr ? c - 1 : lane0 + idx;
                const bool is_end = cr || ((en >> (4 * idx)) & 1u) != 0u;
                const uint32_t hm4 = h & ((2u << (4 * idx)) - 1u);
                const int hs = hm4 ? lane0 + ((31 - __builtin_clz(hm4 | 1u)) >> 2) : hprev;
                const int start = cr ? open_start : (is_end ? hs : bpos);
                const int blen = bpos - start + 1;
                // the key: a run of <= 9 bytes that starts in this lane's bytes or the eight before them (a byte that is not an ASCII run
                // member has code 0: a key with a zero field matches no entry, only a zero at the TOP would look like a shorter word)
                const int rel = start - (lane0 - 8);
                const bool inown = rel >= 8;
                const uint64_t src = inown ? cc : pcc;
                const int sha = 7 * (inown ? rel - 8 : (rel < 0 ? 0 : rel));
                uint64_t key = src >> sha;
                if (!inown) key |= cc << (56 - sha);
                const int kl = blen > WF_KEY_CHARS ? WF_KEY_CHARS : (blen < 1 ? 1 : blen);
                key &= (1ull << (7 * kl)) - 1ull;
                // (a field of the key is 0 where the byte is no plain run member: the classic zero-field test, 7-bit fields; it may also flag the
                // field above a zero one -- no harm: such a run takes the way of the runs without a key)
                const uint64_t km = (1ull << (7 * kl)) - 1ull;
                const bool haszero = (((key | ~km) - 0x0102040810204081ull) & ~key & 0x4081020408102040ull & km) != 0ull;
                const bool fast = blen <= WF_KEY_CHARS && rel >= 0 && !haszero;
                if (!fast) {
                    // no key.  Is the run plain ASCII all the same (then a unit can read it from the text)?  No lane it touches holds a byte >= 0x80
                    bool plain = true;
                    if (na | na_prev) {
                        const int ls = (start - (c - WF_CHUNK)) >> 3, le = (bpos - (c - WF_CHUNK)) >> 3;       // lanes counted from the chunk before (a run is <= 48 bytes)
                        const unsigned long long wlo_ = ls < 64 ? ((na_prev >> (ls & 63)) | ((ls & 63) ? na << (64 - (ls & 63)) : 0ull)) : (na >> ((ls - 64) & 63));
                        plain = ls >= 0 && (wlo_ & ((2ull << ((le - ls) & 63)) - 1ull)) == 0ull;
                    }
                    key = WF_KEY_NONE | (plain ? 1ull : 0ull);
                }
                if (!is_end) key = WF_KEY_SOLO | WF_KEY_SOLO_CLS | (uint64_t)(S.ring[(uint32_t)bpos & RMASK] & LX_T_CLS_MASK);     // a one-element token: by class
                const bool put = on && (uint32_t)slot < (uint32_t)WF_TQ;
                uint32_t *d0 = put ? &S.tq_lo[slot] : &S.spare32;
                *d0 = (uint32_t)key;
                uint32_t *d1 = put ? &S.tq_hi[slot] : &S.spare32;
                *d1 = (uint32_t)(key >> 32);
                uint16_t *d2 = put ? &S.tq_pos[slot] : &S.spare;
                *d2 = (uint16_t)(blen > WF_RUN_MAX ? (uint32_t)(bpos - (c - 64)) : ((uint32_t)(start - (c - 64)) | ((uint32_t)blen << 10)));      // (a run that long may begin anywhere: its LAST byte is kept)
                slot += on ? (cr ? coff : 1) : 0;
            }
        }
        // ---- the documents that begin here: their first entry; the one before each is complete
        for (int d = dfirst; d < dnext; ++d) {
            const int r = off_rel(d) - c, sl = r >> 3;
            const uint32_t t_sl = wv::bcast(tk0, sl);
            const int kd = k + coff + wv::bcast(excl, sl) + __builtin_popcount(t_sl & ((1u << (4 * (r & 7))) - 1u));
            emit_boundary(d, kd);
        }
        if (over) { mark(c > 0 ? c - 1 : 0, WF_D_HARD); for (int d = dfirst; d < dnext; ++d) { const int o = off_rel(d); if (o < len) mark(o, WF_D_HARD); } }
        wv::sync();
        // (2) the list is looked up, one token per lane and trip: two 12-byte gathers per lane in flight, the ids go to their entries as whole rows
        const int nlist = over ? 0 : ntok;
        uint32_t *eout = ent + k;
        const bool last = at_end;
        for (int t0 = 0; t0 < nlist || (t0 == 0 && last); t0 += 64) {
            const bool have = t0 + lane < nlist;
            const uint64_t key = have ? ((uint64_t)S.tq_lo[t0 + lane] | ((uint64_t)S.tq_hi[t0 + lane] << 32)) : WF_KEY_NONE;
            const uint32_t ps = have ? (uint32_t)S.tq_pos[t0 + lane] : 0u;
            const uint32_t x = wf_mix(key, p.m0);
            const uint32_t *ea = (const uint32_t *)p.W + 4u * wf_h(x, p.m1, p.wbits), *eb = (const uint32_t *)p.W + 4u * wf_h(x, p.m2, p.wbits);
            const uint32_t al = ea[0], ah = ea[1], ai = ea[2], bl_ = eb[0], bh = eb[1], bi = eb[2];
            const bool hita = al == (uint32_t)key && ah == (uint32_t)(key >> 32), hitb = bl_ == (uint32_t)key && bh == (uint32_t)(key >> 32);
            const bool hit = have && (hita || hitb);
            if (hit) eout[t0 + lane] = hita ? ai : bi;
            if (STATS) st_hit += (unsigned long long)__builtin_popcountll(wv::ballot(hit));
            const bool rest = have && !hit;
            const int blen = (int)(ps >> 10);
            const unsigned long long RB = wv::ballot(rest), TL = wv::ballot(rest && blen == 0);
            const bool fin = last && t0 + 64 >= nlist;
            if (RB) {
                // a run of more than WF_RUN_MAX bytes: its document is handed on
                for (unsigned long long tl = TL; tl;) { const int l = __builtin_ctzll(tl); tl &= tl - 1ull; mark(c - 64 + (int)(wv::bcast(ps, l) & 0x3FFu), WF_D_HARD); }
                const int s0 = c - 64 + (int)(ps & 0x3FFu);
                // a run with a key is plain ASCII; one without says so (bit 0); a one-element token the table does not hold takes the general way
                const bool text_ok = !(key >> 63) && key != WF_KEY_NONE && blen <= 16 && s0 + 16 <= len;
                const bool word = rest && blen != 0;
                const unsigned long long WB = wv::ballot(word), XB = wv::ballot(word && !text_ok);
                const int nw = __builtin_popcountll(WB);
                if (STATS) st_notes += (unsigned long long)nw;
                if (!XB && nrec + nw <= WF_REC) {
                    // the usual case, in line: plain words that fit the record table
                    if (word) { uint32_t *r = S.rec + 3 * (nrec + (int)wv::mbcnt(WB)); r[0] = (uint32_t)(k + t0 + lane); r[1] = (uint32_t)s0; r[2] = WF_REC_TEXT | (uint32_t)blen; }
                    nrec += nw;
                } else if (WB) add_words(WB, word, (uint32_t)s0, blen, (uint32_t)(k + t0 + lane), text_ok, false);
            }
            if (fin) { wv::sync(); add_words(0ull, false, 0u, 0, 0u, false, true); }
        }
        k += ntok; open_start = new_open; na_prev = na;
    }

    BF_WVD void range(int64_t r)
    {
        dlo = p.range_doc[r];
        const int64_t dhi = p.range_doc[r + 1];
        if (dlo >= dhi) return;
        dn = (int)(dhi - dlo);
        b0 = p.doc_off[dlo]; len = (int)(p.doc_off[dhi] - b0);
        txt = p.text + b0; ent = p.ent + b0; home = p.home + b0;
        k = kdoc = 0; dnext = 0; open_start = -1; pcc63 = 0; cov_carry = loop_carry = 0; na_prev = 0; bad_lo = bad_hi = hard_lo = hard_hi = 0;
        load_window(0);
        uint64_t own = load_chunk(0);
        for (int c = 0; c < len; c += WF_CHUNK) {
            const uint64_t nxt = c + WF_CHUNK < len ? load_chunk(c + WF_CHUNK) : 0ull;       // the next chunk is on its way while this one is worked on
            chunk(c, own);
            own = nxt;
        }
        // documents that begin where the range ends (empty ones), then the last document's count
        for (; dnext < dn; ++dnext) emit_boundary(dnext, k);
        if (lane == 0) p.ent_cnt[dhi - 1] = k - kdoc;
    }

    BF_WVD void run(int wave_id, int n_waves)
    {
        if (*p.unsafe) return;
        for (int round = 0;; ++round) {
            unsigned long long r = 0;
            if (p.next_range) { if (lane == 0) r = wv::atomic_add(p.next_range, 1ull); r = wv::bcast(r, 0); }
            else r = (unsigned long long)wave_id + (unsigned long long)round * (unsigned long long)n_waves;
            if (r >= (unsigned long long)p.nranges) break;
            range((int64_t)r);
        }
        if (STATS && cold.stats && lane == 0) {
            wv::atomic_add(&cold.stats[0], st_chunks); wv::atomic_add(&cold.stats[1], st_ascii); wv::atomic_add(&cold.stats[2], st_tok); wv::atomic_add(&cold.stats[3], st_hit);
            wv::atomic_add(&cold.stats[4], st_notes); wv::atomic_add(&cold.stats[5], st_drains); wv::atomic_add(&cold.stats[6], st_rounds); wv::atomic_add(&cold.stats[7], st_hard);
        }
    }
};

// ----------------------------------------------------------------------------------------------------------------------
// k_wp_count / k_wp_merge: a wave takes 64 consecutive documents, one per lane for what is read once (first entry, number of entries, flags,
// place in the output), then works through them together.  An entry is an id, or -- bit 31 -- a word of n ids at its home (n = 0: UnkId).
// ----------------------------------------------------------------------------------------------------------------------
BF_WVD int wf_entry_ids(uint32_t e) { const int n = (int)((e & ~WF_ENT_FLAG) >> WF_ENT_CNT_SHIFT); return (e & WF_ENT_FLAG) ? (n ? n : 1) : 1; }

// counts[d] = ids of document d (tokdll:1308-1310: at most max_ids; 0 for invalid UTF-8: :1151-1153)
BF_WVD void wf_count_docs(const WfMergeParams &p, int64_t base)
{
    const int lane = wv::lane();
    const int64_t d = base + lane;
    const bool unsafe = *p.unsafe != 0;
    int ec = 0, st = 0, old = 0; int64_t eo = 0;
    if (d < p.ndocs) { st = unsafe ? WF_D_HARD : p.dstat[d]; old = p.counts[d]; if (!(st & WF_D_HARD)) { eo = p.ent_off[d]; ec = p.ent_cnt[d]; } }
    if (st & (WF_D_BAD | WF_D_HARD)) ec = 0;
    int extra = 0;
    const int nd = p.ndocs - base < 64 ? (int)(p.ndocs - base) : 64;
    for (int i = 0; i < nd; ++i) {
        const int n = wv::bcast(ec, i);
        if (n == 0) continue;
        const int64_t o = wv::bcast(eo, i);
        int sum = 0;
        for (int t = 0; t < n; t += 64) {
            const uint32_t e = t + lane < n ? p.ent[o + t + lane] : 0u;
            const int x = wf_entry_ids(e) - 1;
            if (wv::any(x != 0)) sum += wv::bcast(wv::incl_scan(x), 63);
        }
        if (lane == i) extra = sum;
    }
    if (d < p.ndocs) {
        int cnt = (st & WF_D_HARD) ? old : (st & WF_D_BAD) ? 0 : ec + extra;
        if (!(st & WF_D_HARD) && cnt > p.max_ids) cnt = p.max_ids;
        p.counts[d] = cnt;
    }
}

BF_WVD void wf_merge_docs(const WfMergeParams &p, int64_t base, bool &over)
{
    const int lane = wv::lane();
    const int64_t d = base + lane;
    const bool unsafe = *p.unsafe != 0;
    int ec = 0, st = 0, cnt = 0; int64_t eo = 0, o = 0;
    if (d < p.ndocs) {
        st = unsafe ? WF_D_HARD : p.dstat[d]; cnt = p.counts[d]; o = p.id_off[d];
        if (st & WF_D_HARD) { eo = wv_ids_slot(p.doc_off[d], d); ec = cnt; } else { eo = p.ent_off[d]; ec = p.ent_cnt[d]; }
        if (o + cnt > p.ids_cap) { over = true; cnt = o < p.ids_cap ? (int)(p.ids_cap - o) : 0; }
        if (cnt == 0) ec = 0;
    }
    const int nd = p.ndocs - base < 64 ? (int)(p.ndocs - base) : 64;
    for (int i = 0; i < nd; ++i) {
        const int n = wv::bcast(ec, i);
        if (n == 0) continue;
        const int64_t src = wv::bcast(eo, i), dst = wv::bcast(o, i);
        const int cap = wv::bcast(cnt, i);
        if (wv::bcast(st, i) & WF_D_HARD) {                                   // tokenised by the wave program: its ids are in place
            for (int t = lane; t < cap; t += 64) p.ids_out[dst + t] = p.ids_tmp[src + t];
            continue;
        }
        int run = 0;
        for (int t = 0; t < n && run < cap; t += 64) {
            const bool have = t + lane < n;
            const uint32_t e = have ? p.ent[src + t + lane] : 0u;
            const int x = have ? wf_entry_ids(e) : 0;
            if (!wv::any(have && (e & WF_ENT_FLAG) != 0u)) {
                const int pos = run + lane;
                if (have && pos < cap) p.ids_out[dst + pos] = (int32_t)e;
                run += n - t < 64 ? n - t : 64;
                continue;
            }
            const int inc = wv::incl_scan(x), pos = run + inc - x;
            if (have && !(e & WF_ENT_FLAG)) { if (pos < cap) p.ids_out[dst + pos] = (int32_t)e; }
            else if (have) {
                const int nn = (int)((e & ~WF_ENT_FLAG) >> WF_ENT_CNT_SHIFT);
                if (nn == 0) { if (pos < cap) p.ids_out[dst + pos] = p.unk; }
                else {
                    const int32_t *hm = p.home + (src + t + lane) + (int64_t)(e & WF_ENT_DELTA_MASK);
                    for (int j = 0; j < nn && pos + j < cap; ++j) p.ids_out[dst + pos + j] = hm[j];
                }
            }
            run += wv::bcast(inc, 63);
        }
    }
}

} // namespace bfa
