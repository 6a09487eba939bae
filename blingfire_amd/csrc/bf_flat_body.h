// bf_flat_body.h -- the flat wave program of bf_flat.h (see there).  Include AFTER a definition of namespace wv (bf_kernels.hip: the wave
// intrinsics of gfx950; tests/hosttest/wave_emu.h: the 64-fibre simulator).
//
// One wave takes ranges of documents from a work counter.  A range is a stream of 512-byte chunks, eight bytes per lane, positions counted
// in BYTES from the range's first byte:
//   classes   eight look-ups per lane in a 128-entry table give class, kind and key code of every byte; a chunk with a byte >= 0x80 (one vote)
//             then has every lead byte decoded by its lane (strict UTF-8, the rules of FAUtf8Utils.cpp:121-196 per byte position), its
//             continuation bytes belong to its character.  Class and key code of every byte position go to two rings in LDS;
//   tokens    kinds as one bit per byte in 4-bit fields: run starts / ends from shifts, a document boundary is one more bit that cuts
//             runs (FALexTools_t.h:229-393 on a unit-form lexer is a function of the kinds alone, bf_wave.h); every lane owns the tokens
//             that END in its eight bytes; a prefix sum gives each its entry;
//   list      every lane writes where its tokens are, in token order, to a list in LDS;
//   look-up   the list one token per lane and trip: the key of a run of <= 12 plain characters is its bytes of the code ring (the key IS
//             the word), two 16-byte gathers per lane, ids to the entries as whole rows;
//   records   the tokens the table did not answer become records (entry, first byte, bytes, document) in the range's own list in global
//             memory, 64 at a time: the words are walked by a kernel of their own (k_wp_units, wf_units below), where nothing waits for them.
// What couples documents is left to the kernels behind (k_wp_count, k_wp_merge): a document's entries are dense, its ids are not yet.
#pragma once
#include "bf_flat.h"

namespace bfa {

constexpr int WF_TQ = 512 + 8;            // tokens of a chunk, at most: every byte one, and the run that ended with the chunk before (513; the rest keeps the alignment)
constexpr int WF_RING_DUP = 16;           // the first positions of the code ring once more behind its end: the 12 bytes of a key are read without a wrap
constexpr uint32_t WF_TQ_SOLO = 63;       // list entry: bytes == 63: a one-element token
// a record of the word list (16 bytes): [0] the word's entry, [1] its first byte, [3] its document -- counted from the first entry / byte / document of the range -- [2] bytes | WF_REC_PLAIN
constexpr uint32_t WF_REC_PLAIN = 1u << 8;       // the word's bytes are plain ASCII (its characters are its bytes)

struct alignas(16) WfRow { uint32_t k0lo, k0hi, k1, id; };      // an entry of the word table (bf_flat_key.h)
struct WfLds {
    alignas(16) uint16_t ring[WF_RING];  // class of every byte position of this chunk and the one before (WF_CONT: no character starts there)
    alignas(16) uint8_t cring[WF_RING + WF_RING_DUP];   // the same positions: the class's code inside a key (bf_flat_key.h; 0: it has none, or no character starts there)
    uint16_t tq_pos[WF_TQ];              // the chunk's tokens in order: (first byte - (chunk - 64)) | bytes << 10; bytes == 0: a run of more than WF_RUN_MAX bytes, its LAST byte
    alignas(16) uint32_t rec[WF_REC * 4]; // words on their way to the list
    uint32_t spare32; uint16_t spare;
};

// the 128-entry table of the ASCII bytes: [2:0] kind bits (loop, solo, general), [14:8] the class's code inside a key, [28:16] class -- where one
// v_perm_b32 / v_alignbit_b32 picks them up for the rows of the two rings and the kind mask of a lane's eight bytes
BF_WV uint32_t wf_lut_value(const WpWaveCold &p, int b)
{
    const uint32_t el = wv_element(p, b), c = el & LX_T_CLS_MASK, k = el >> WK_SHIFT;
    const uint32_t nib = k == WK_LOOP ? 1u : k == WK_SOLO ? 2u : k == WK_GENERAL ? 4u : 0u;
    return nib | ((c < 127u ? c + 1u : 0u) << 8) | (c << 16);
}
BF_WV uint32_t wf_lut_class(uint32_t v) { return v >> 16; }
// behind it (entries 128 + 4 n .. + 2): the bytes of a key a word of n characters has (bf_flat_key.h), as three masks
constexpr int WF_LUT = 128 + 4 * 16;
BF_WV uint32_t wf_kmask_value(int i)
{
    const int n = (i >> 2) - (i & 3) * 4;                 // bytes of dword i & 3 that belong to a word of i >> 2 characters
    return (i & 3) == 3 || n <= 0 ? 0u : n >= 4 ? 0xFFFFFFFFu : (1u << (8 * n)) - 1u;
}

#if defined(__HIPCC__)
#define BF_WF_NOINLINE __device__ __forceinline__
// (left to itself the compiler fetches the second row's key only after the first row has come back and did not match, or waits for the first row
// before it sends for the second: two trips to the table for one)
typedef uint32_t wf_u32x4 __attribute__((ext_vector_type(4)));
#define BF_WF_LOAD_ROWS(A, B, PA, PB) { wf_u32x4 ra_, rb_; const WfRow *pa_ = (PA), *pb_ = (PB); \
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off\n\ts_waitcnt vmcnt(0)" : "=&v"(ra_), "=&v"(rb_) : "v"(pa_), "v"(pb_) : "memory"); \
    A.k0lo = ra_.x; A.k0hi = ra_.y; A.k1 = ra_.z; A.id = ra_.w; B.k0lo = rb_.x; B.k0hi = rb_.y; B.k1 = rb_.z; B.id = rb_.w; }
#else
#define BF_WF_NOINLINE static __attribute__((noinline))
#define BF_WF_LOAD_ROWS(A, B, PA, PB) { A = *(PA); B = *(PB); }
#endif

// A chunk with bytes >= 0x80 (the caller has put every byte through the table of the ASCII bytes).  Every lead byte is decoded
// by its lane, one per trip; a continuation byte is legal exactly when it is one of the (length - 1) bytes behind a lead byte OF ITS DOCUMENT (a
// document boundary inside a character truncates it, FAUtf8Utils.cpp:167-171).  S8: the documents that begin in this lane's bytes; peek: in the
// three bytes behind the chunk.  Returns what the characters add to the kind bits, the bytes that are invalid, and what the last character of
// the chunk covers of the next one.
struct WfMb { uint32_t acc, clr, errm, cov_carry, loop_carry; unsigned long long na; };      // na: the lanes that hold a byte >= 0x80; clr: the kind bits of those bytes
BF_WVD WfMb wf_decode_multibyte(uint64_t own, uint32_t S8, uint32_t peek, int c, int len, const uint8_t *txt, uint16_t *ring, uint8_t *cring, const uint16_t *cp_l1, const uint32_t *cp_pages,
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
    // what the caller made of the bytes >= 0x80 (their low seven bits through the table of the ASCII bytes) is taken back: no character starts
    // there until a lead byte below says so
    uint32_t clr = 0;
    for (uint32_t m = m80; m;) {
        const int i = __builtin_ctz(m); m &= m - 1u;
        const uint32_t rp = ((uint32_t)lane0 + (uint32_t)i) & RMASK;
        ring[rp] = (uint16_t)WF_CONT; cring[rp] = 0; if (rp < (uint32_t)WF_RING_DUP) cring[rp + WF_RING] = 0;
        clr |= 0xFu << (4 * i);
    }
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
            {
                const uint32_t rp = ((uint32_t)lane0 + (uint32_t)i) & RMASK; const uint8_t code = (uint8_t)(el < 127u ? el + 1u : 0u);
                ring[rp] = (uint16_t)el; cring[rp] = code; if (rp < (uint32_t)WF_RING_DUP) cring[rp + WF_RING] = code;
            }
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
    WfMb r; r.acc = acc; r.clr = clr; r.errm = errm; r.cov_carry = cov_out; r.loop_carry = loop_out; r.na = wv::ballot(m80 != 0);
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
    const uint8_t *txt; uint32_t *ent, *esp; int32_t *home;
    int k, kdoc;                         // tokens so far; tokens before the open document
    int open_start;                      // first byte of the run that reaches the end of the chunk before (-1: none)
    int doc0;                            // first byte of the document that is open where the chunk begins (offsets API)
    unsigned long long na_prev;          // the lanes of the chunk before that hold a byte >= 0x80
    uint32_t cov_carry, loop_carry;      // of lane 63 of the chunk before: bytes of the next chunk that belong to its last character
    int nrec, wf_n, ws_n;                // words in S.rec; words of this range on its two lists
    int bad_lo, bad_hi, hard_lo, hard_hi; // [lo, hi): bytes of the document that got the flag last (one look-up per document, mostly)
    unsigned long long st_chunks, st_ascii, st_tok, st_hit, st_notes, st_drains, st_rounds, st_hard;

    BF_WVD WfWave(const WfParams &p_, WfLds &S_, const uint32_t *lut_, const WpWaveCold &cold_) : p(p_), S(S_), lut(lut_), cold(cold_)
    {
        lane = wv::lane(); nrec = 0; wf_n = ws_n = 0; bad_lo = bad_hi = hard_lo = hard_hi = 0;
        st_chunks = st_ascii = st_tok = st_hit = st_notes = st_drains = st_rounds = st_hard = 0;
        dlo = 0; dn = dnext = wlo = 0; b0 = 0; len = 0; win = 0; txt = nullptr; ent = esp = nullptr; home = nullptr; k = kdoc = 0; open_start = -1; doc0 = 0; cov_carry = loop_carry = 0; na_prev = 0;
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
    // the words in S.rec go to the list (one atomic for all of them; a list that is full hands their documents on: cannot be but for text
    // that is all words the table does not hold)
    // The words in S.rec go to the lists of the range.  A range owns the records [b0 / 16, (b0 + len) / 16) (WF_REC_SHIFT) (no counter is shared between waves: a
    // counter that 8,000 waves add to costs more than the walks): the words a unit holds in registers (plain ASCII, <= 16 bytes, 16 readable bytes
    // behind their first) from the front, the others from the back.  A range whose words do not fit hands their documents on (cannot be but
    // for text that is all words the table does not hold).
    BF_WVD void flush_records()
    {
        wv::sync();
        if (nrec > 0) {
            uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;
            if (lane < nrec) { const uint32_t *r = S.rec + 4 * lane; r0 = r[0]; r1 = r[1]; r2 = r[2]; r3 = r[3]; }
            const bool fastw = lane < nrec && (r2 & WF_REC_PLAIN) != 0u && (r2 & 0xFFu) <= 16u && b0 + (int64_t)r1 + 16 <= p.total_bytes;
            const unsigned long long FB = wv::ballot(fastw), SB = wv::ballot(lane < nrec && !fastw);
            const int nf = __builtin_popcountll(FB), ns = __builtin_popcountll(SB);
            const int64_t lo = (b0 + (1 << WF_REC_SHIFT) - 1) >> WF_REC_SHIFT, hi = (b0 + (int64_t)len) >> WF_REC_SHIFT;           // the range's records
            if ((int64_t)(wf_n + nf + ws_n + ns) <= hi - lo) {
                if (lane < nrec) {
                    uint32_t *d = p.wrec + 4 * (fastw ? lo + wf_n + (int64_t)wv::mbcnt(FB) : hi - 1 - ws_n - (int64_t)wv::mbcnt(SB));
                    d[0] = r0; d[1] = r1; d[2] = r2; d[3] = r3;
                }
                wf_n += nf; ws_n += ns;
            } else {
                for (unsigned long long lb = FB | SB; lb;) {
                    const int l = __builtin_ctzll(lb); lb &= lb - 1ull;
                    hard_lo = hard_hi = 0; mark((int)wv::bcast(r1, l), WF_D_HARD);
                }
            }
        }
        nrec = 0;
        wv::sync();
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
        uint32_t acc = 0;
        unsigned long long na = 0;
        const bool ascii_chunk = !wv::any((own & 0x8080808080808080ull) != 0);
        {
            // every byte through the table of the ASCII bytes (a byte >= 0x80 gets what its low seven bits say: a chunk that holds one puts that
            // right below, wf_decode_multibyte): class rows, code rows, kind bits -- one instruction per pair, per pair and per byte
            uint32_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = lut[(uint32_t)(own >> (8 * i)) & 0x7Fu];
            const uint32_t rp = (uint32_t)lane0 & RMASK;
            uint32_t *row = (uint32_t *)(S.ring + rp);                          // 8 positions = one 16-byte row, never wraps
            row[0] = wv::perm(v[1], v[0], 0x07060302u); row[1] = wv::perm(v[3], v[2], 0x07060302u);
            row[2] = wv::perm(v[5], v[4], 0x07060302u); row[3] = wv::perm(v[7], v[6], 0x07060302u);
            const uint32_t c0 = wv::perm(wv::perm(v[3], v[2], 0x05010c0cu), wv::perm(v[1], v[0], 0x0c0c0501u), 0x07060100u);
            const uint32_t c1 = wv::perm(wv::perm(v[7], v[6], 0x05010c0cu), wv::perm(v[5], v[4], 0x0c0c0501u), 0x07060100u);
            uint32_t *crow = (uint32_t *)(S.cring + rp);
            crow[0] = c0; crow[1] = c1;
            if (rp < (uint32_t)WF_RING_DUP) { uint32_t *dup = (uint32_t *)(S.cring + rp + WF_RING); dup[0] = c0; dup[1] = c1; }
#pragma unroll
            for (int i = 0; i < 8; ++i) acc = wv::alignbit(v[i], acc, 4);         // (the kind bits of byte i end up in bits 4 i ..)
        }
        if (ascii_chunk) { if (STATS) ++st_ascii; cov_carry = 0; loop_carry = 0; }
        else {
            // lane 63: the documents that begin in the first three bytes of the next chunk (not consumed here)
            uint32_t peek = 0;
            for (int d = dnext; d < dn; ++d) { const int o = off_rel(d); if (o >= c + WF_CHUNK + 3) break; peek |= 1u << (o - (c + WF_CHUNK)); }
            const WfMb r = wf_decode_multibyte(own, S8, peek, c, len, txt, S.ring, S.cring, cold.cpmap.l1, cold.cpmap.pages, cold.kind, cold.nclasses, cov_carry, loop_carry);
            acc = (acc & ~r.clr) | r.acc; cov_carry = r.cov_carry; loop_carry = r.loop_carry; na = r.na;
            if (wv::any(r.errm != 0)) mark_bytes(c, r.errm, WF_D_BAD);          // invalid UTF-8: the document has no ids (tokdll:1151-1153)
        }
        wv::sync();                                                             // the ring is written
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
        // (1) every lane writes where ITS tokens are to the list, in order
#ifdef BF_EXPERIMENTS
        if (!(p.dbg & 2))
#endif
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
                const uint32_t pw = !is_end ? ((uint32_t)(bpos - (c - 64)) | (WF_TQ_SOLO << 10))
                                  : blen > WF_RUN_MAX ? (uint32_t)(bpos - (c - 64)) : ((uint32_t)(start - (c - 64)) | ((uint32_t)blen << 10));
                uint16_t *d2 = on ? &S.tq_pos[slot] : &S.spare;
                *d2 = (uint16_t)pw;
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
        wv::sync();
        // (2) the list, one token per lane and trip.  The key of a run of <= 12 bytes is its bytes of the code ring (bf_flat_key.h): the key IS the
        // word.  Two 16-byte gathers per lane in flight; the ids go to their entries as whole rows.
        uint32_t *eout = ent + k;
#ifdef BF_EXPERIMENTS
        if (!(p.dbg & 3))
#endif
        for (int t0 = 0; t0 < ntok; t0 += 64) {
            const bool have = t0 + lane < ntok;
            const uint32_t ps = have ? (uint32_t)S.tq_pos[t0 + lane] : 0u;
            const uint32_t b6 = ps >> 10;
            const bool solo = b6 == WF_TQ_SOLO;
            const int blen = solo ? 1 : (int)b6;                                 // (0: a run of more than WF_RUN_MAX bytes; its document is handed on)
            const int s0 = c - 64 + (int)(ps & 0x3FFu);
            // the 12 codes behind the token's first byte, those behind the word cleared (the masks of a length: behind the table of the ASCII bytes).
            // A word with a code 0 inside (a class without a code, or no character: a continuation byte) matches no entry: an entry's bytes are codes up
            // to its length, which is part of the match.
            uint32_t w[3];
            __builtin_memcpy(w, S.cring + ((uint32_t)s0 & RMASK), 12);                 // (the duplicate bytes: no wrap)
            const int kn = (have && !solo && blen <= WF_KEY_CHARS) ? blen : 0;
            const uint32_t *km = lut + 128 + 4 * kn;
            uint64_t k0 = ((uint64_t)(w[0] & km[0])) | ((uint64_t)(w[1] & km[1]) << 32);
            uint32_t k1 = w[2] & km[2];
            if (kn == 0) { k0 = solo ? (WF_KEY_SOLO | ((uint64_t)(S.ring[(uint32_t)s0 & RMASK] & LX_T_CLS_MASK) << WF_KEY_SOLO_SHIFT)) : WF_KEY_NONE; k1 = 0u; }       // a one-element token: by class
            const uint32_t x = wf_mix(k0, k1, p.m0);
            WfRow A, B;
            BF_WF_LOAD_ROWS(A, B, (const WfRow *)p.W + wf_h(x, p.m1, p.wbits), (const WfRow *)p.W + wf_h(x, p.m2, p.wbits));     // both rows whole and in flight together: one trip to the table
            const uint32_t klen = (uint32_t)kn << WF_ROW_LEN_SHIFT;
            const bool hita = ((A.k0lo ^ (uint32_t)k0) | (A.k0hi ^ (uint32_t)(k0 >> 32)) | (A.k1 ^ k1) | ((A.id ^ klen) & ~WF_ROW_ID_MASK)) == 0u;
            const bool hitb = ((B.k0lo ^ (uint32_t)k0) | (B.k0hi ^ (uint32_t)(k0 >> 32)) | (B.k1 ^ k1) | ((B.id ^ klen) & ~WF_ROW_ID_MASK)) == 0u;
            const uint32_t ai = A.id & WF_ROW_ID_MASK, bi = B.id & WF_ROW_ID_MASK;
            const bool hit = have && (hita || hitb);
            if (hit) eout[t0 + lane] = hita ? ai : bi;
            if (p.espan) {
                // offsets API: where the token is in its document (the first byte of the last document that begins at or before it; a one-element
                // token is one character: its bytes from its first one)
                int ds = doc0;
                for (int d = dfirst; d < dnext; ++d) { const int o = off_rel(d); ds = s0 >= o ? o : ds; }
                int bl = blen > 0 ? blen : 1;
                if (!ascii_chunk && have && solo) { const uint32_t b = txt[s0]; bl = b < 0x80u ? 1 : b < 0xE0u ? 2 : b < 0xF0u ? 3 : 4; }      // (a one-element token lies in this chunk)
                if (have) esp[k + t0 + lane] = (uint32_t)(s0 - ds) | ((uint32_t)(bl - 1) << WF_SPAN_LEN_SHIFT);
            }
            if (STATS) st_hit += (unsigned long long)__builtin_popcountll(wv::ballot(hit));
            const bool rest = have && !hit;
            const unsigned long long RB = wv::ballot(rest), TL = wv::ballot(rest && blen == 0);
            if (RB) {
                // a run of more than WF_RUN_MAX bytes: its document is handed on
                for (unsigned long long tl = TL; tl;) { const int l = __builtin_ctzll(tl); tl &= tl - 1ull; mark(c - 64 + (int)(wv::bcast(ps, l) & 0x3FFu), WF_D_HARD); }
                const bool word = rest && blen != 0;
                const unsigned long long WB = wv::ballot(word);
                const int nw = __builtin_popcountll(WB);
                if (STATS) st_notes += (unsigned long long)nw;
                if (nrec + nw > WF_REC) flush_records();
                // a unit can read the word from the text when no lane it touches holds a byte >= 0x80 (lanes counted from the chunk before; a run is <= 48 bytes)
                bool plain = true;
                int rl = blen;                                                       // (a one-element token is one character: its bytes)
                if (na | na_prev) {
                    if (word && solo) { const uint32_t b = txt[s0]; rl = b < 0x80u ? 1 : b < 0xE0u ? 2 : b < 0xF0u ? 3 : 4; }
                    const int ls = (s0 - (c - WF_CHUNK)) >> 3, le = (s0 + rl - 1 - (c - WF_CHUNK)) >> 3;
                    const unsigned long long wlo_ = ls < 64 ? ((na_prev >> (ls & 63)) | ((ls & 63) ? na << (64 - (ls & 63)) : 0ull)) : (na >> ((ls - 64) & 63));
                    plain = ls >= 0 && (wlo_ & ((2ull << ((le - ls) & 63)) - 1ull)) == 0ull;
                }
                // the word's document: the last one that begins at or before it (the one open where the chunk begins: the one before the first that begins here)
                int dd = dfirst - 1;
                for (int d = dfirst; d < dnext; ++d) dd += s0 >= off_rel(d) ? 1 : 0;
                if (word) {
                    uint32_t *r = S.rec + 4 * (nrec + (int)wv::mbcnt(WB));          // (entry, first byte and document count from the range's first)
                    r[0] = (uint32_t)(k + t0 + lane); r[1] = (uint32_t)s0; r[2] = (uint32_t)rl | (plain ? WF_REC_PLAIN : 0u); r[3] = (uint32_t)dd;
                }
                nrec += nw;
            }
        }
        k += ntok; open_start = new_open; na_prev = na;
        if (p.espan && dnext > dfirst) doc0 = off_rel(dnext - 1);
    }

    BF_WVD void range(int64_t r)
    {
        dlo = p.range_doc[r];
        const int64_t dhi = p.range_doc[r + 1];
        if (dlo >= dhi) return;
        dn = (int)(dhi - dlo);
        b0 = p.doc_off[dlo]; len = (int)(p.doc_off[dhi] - b0);
        txt = p.text + b0; ent = p.ent + b0; home = p.home + b0; esp = p.espan ? p.espan + b0 : nullptr;
        k = kdoc = 0; dnext = 0; open_start = -1; doc0 = 0; cov_carry = loop_carry = 0; na_prev = 0; bad_lo = bad_hi = hard_lo = hard_hi = 0; wf_n = ws_n = 0;
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
        flush_records();
        if (lane == 0) { p.wrec_cnt[2 * r] = wf_n; p.wrec_cnt[2 * r + 1] = ws_n; }
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
// k_wp_units: the words of the list, NU per lane and 64 * NU per wave at a time.  The frame of ONE call of the vocabulary function on a word
// (FALexTools_t.h:229-393 at depth 1; the word is shorter than max-token-length, so every walk's limit is the word's end): first the walk from
// the state behind the left anchor at character 0, if the function has one; a walk that ends with a match is a piece and the next walk starts
// behind it (:390-393); the anchored walk without a match is followed by the plain walk at 0 (:293); any other walk without a match leaves a
// gap: the pieces cannot tile the word, its id is UnkId (tokdll:1252-1301).  Restated from bf_wave_body.h Unit / unit_step / unit_event.
// Positions are BYTES of the word.  A plain-ASCII word of <= 16 bytes is held in two registers (its characters' classes through the table of
// the ASCII bytes, the class of the next character fetched while the transition on this one is in flight); any other word is read from the text
// character by character (UTF-8 as the flat program has validated it, fused code-point map).  What a wave waits for is the longest chain of
// dependent gathers among its words: nothing else runs in this kernel, and all resident waves wait alike.  FAST: the list of the words in
// registers (no other code in the loop); else the list of the others.
// ----------------------------------------------------------------------------------------------------------------------
// MODE 0: the list of the plain-ASCII words of <= 16 bytes: the word in two registers, classes through the table of the ASCII bytes.
// MODE 1: the other list, its words of <= 16 bytes: the word in two registers, its characters' classes made up front (`cbuf`: 16 classes per
//         lane in LDS; a character outside ASCII: fused code-point map) -- positions are then CHARACTERS.
// MODE 2: the other list, what is left (longer words, words at the very end of the text): read from the text character by character.
template <int NU, bool STATS, int MODE, bool OFFS>
BF_WVD void wf_units(const WfUnitParams &p, const uint32_t *lut, uint16_t *cbuf, const uint32_t *wrec, int64_t b0, int64_t dlo, unsigned long long first, unsigned long long total, unsigned long long *rounds)
{
    static_assert(MODE == 0 || NU == 1, "one word per lane but on the first list");
    const int lane = wv::lane();
    const bool anchored0 = p.ini_l != LX_NO_STATE && p.max_token_length > 1;
    const uint64_t *T = p.T;
    uint32_t state[NU], ftag[NU], c_cur[NU], w3[NU]; int L[NU], j[NU], fp[NU], cnt[NU], clen[NU]; int32_t id0[NU];
    // offsets API: where the walk under way started (a piece is [from, fp)); the span of the first piece; the word's first byte in its document;
    // MODE 1: the bytes at which the word's characters start, its bytes
    int from[NU]; uint32_t sp0[NU]; uint32_t startm_keep = 0; int Lb = 0;
    constexpr bool offs = OFFS;
    int64_t ea[NU], pa[NU]; int xtra[NU];
    uint64_t t_lo[NU], t_hi[NU];
    bool anch[NU], act[NU], missed[NU], stale[NU];
    // MODE 2: the character at byte j of word u, read from the text: its class and its length
    auto fetch_mem = [&](int u) {
        const uint8_t *q = p.text + pa[u] + j[u];
        const int left = L[u] - j[u];
        const uint32_t b0 = q[0];
        int cl = b0 < 0x80u ? 1 : (b0 & 0xE0u) == 0xC0u ? 2 : (b0 & 0xF0u) == 0xE0u ? 3 : (b0 & 0xF8u) == 0xF0u ? 4 : 1;
        if (cl > left) cl = left;                                  // (cannot be in a document that has ids)
        uint32_t cls;
        if (b0 < 0x80u) cls = wf_lut_class(lut[b0]);
        else {
            int cp = cl == 2 ? (int)(b0 & 0x1Fu) : cl == 3 ? (int)(b0 & 0x0Fu) : (int)(b0 & 0x07u);
            for (int t = 1; t < cl; ++t) cp = (cp << 6) | (int)(q[t] & 0x3Fu);
            if (cp > 0x10FFFF) cp = 0;
            cls = wv_cpmap_get(p.cpmap, cp) & LX_T_CLS_MASK;
        }
        c_cur[u] = cls; clen[u] = cl;
    };
    auto byte_at = [&](int u, int i) -> uint32_t { return (uint32_t)((i < 8 ? t_lo[u] : t_hi[u]) >> (8 * (i & 7))) & 0xFFu; };
    // class of character i (MODE 0: byte i through the table; MODE 1: from the classes made up front)
    auto cls_at = [&](int u, int i) -> uint32_t {
        if (MODE == 0) return wf_lut_class(lut[byte_at(u, i) & 0x7Fu]);
        return (uint32_t)cbuf[lane * 16 + (i & 15)];
    };
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const unsigned long long ri = first + (unsigned long long)(64 * u + lane);
        bool have = ri < total;
        const uint32_t *r = wrec + 4 * (have ? ri : first);
        const uint32_t r0 = r[0], r1 = r[1], r2 = r[2]; w3[u] = r1 - r0;               // (home - entry: a word has no more pieces than bytes)
        ea[u] = b0 + (int64_t)r0; pa[u] = b0 + (int64_t)r1;
        L[u] = (int)(r2 & 0xFFu);
        const bool b16 = L[u] <= 16 && pa[u] + 16 <= p.total_bytes;
        if (MODE == 1) have = have && b16;
        if (MODE == 2) have = have && !b16;
        t_lo[u] = 0; t_hi[u] = 0;
        if (MODE != 2 && have) { __builtin_memcpy(&t_lo[u], p.text + pa[u], 8); __builtin_memcpy(&t_hi[u], p.text + pa[u] + 8, 8); }
        state[u] = anchored0 ? p.ini_l : p.ini; j[u] = 0; fp[u] = -1; cnt[u] = 0; ftag[u] = 0; id0[u] = 0; clen[u] = 1;
        anch[u] = anchored0; act[u] = have; missed[u] = false; stale[u] = MODE == 2 && have;
        c_cur[u] = 0;
        from[u] = 0; sp0[u] = 0; xtra[u] = 0;
    }
    { bool a = false;
#pragma unroll
      for (int u = 0; u < NU; ++u) a = a || act[u];
      if (!wv::any(a)) return; }
    if (MODE == 1) {
        // the classes of the word's characters, up front: the ASCII ones through the table, the others one per lane and trip through the code-point map
        uint32_t startm = 0, leadm = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const uint32_t b = byte_at(0, i);
            const bool in = act[0] && i < L[0];
            if (in && (b & 0xC0u) != 0x80u) startm |= 1u << i;
            if (in && b >= 0xC0u) leadm |= 1u << i;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool on = ((startm & ~leadm) >> i) & 1u;
            uint16_t *d = on ? &cbuf[lane * 16 + __builtin_popcount(startm & ((1u << i) - 1u))] : &cbuf[64 * 16];
            *d = (uint16_t)wf_lut_class(lut[byte_at(0, i) & 0x7Fu]);
        }
        for (uint32_t lm = leadm; wv::any(lm != 0);) {
            if (lm) {
                const int i = __builtin_ctz(lm); lm &= lm - 1u;
                uint32_t w = i < 8 ? (uint32_t)(t_lo[0] >> (8 * i)) : (uint32_t)(t_hi[0] >> (8 * (i - 8)));
                if (i > 4 && i < 8) w |= (uint32_t)(t_hi[0] << (64 - 8 * i));
                const uint32_t b0 = w & 0xFFu;
                const int cl = (b0 & 0xE0u) == 0xC0u ? 2 : (b0 & 0xF0u) == 0xE0u ? 3 : 4;
                int cp = cl == 2 ? (int)(b0 & 0x1Fu) : cl == 3 ? (int)(b0 & 0x0Fu) : (int)(b0 & 0x07u);
                for (int t = 1; t < cl; ++t) cp = (cp << 6) | (int)((w >> (8 * t)) & 0x3Fu);
                if (cp > 0x10FFFF) cp = 0;
                cbuf[lane * 16 + __builtin_popcount(startm & ((1u << i) - 1u))] = (uint16_t)(wv_cpmap_get(p.cpmap, cp) & LX_T_CLS_MASK);
            }
        }
        startm_keep = startm; Lb = L[0];
        L[0] = __builtin_popcount(startm);                                     // characters from here on
        wv::sync();
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) if (MODE != 2) c_cur[u] = cls_at(u, 0);
    bool any_act = true;
    while (any_act) {
        if (STATS) ++*rounds;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            uint64_t e64[NU]; uint32_t c_nxt[NU]; bool walking[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) { if (MODE == 2 && wv::any(stale[u])) { if (stale[u]) fetch_mem(u); stale[u] = false; } }
#pragma unroll
            for (int u = 0; u < NU; ++u) { walking[u] = act[u] && !missed[u] && j[u] < L[u]; e64[u] = 0; if (walking[u]) e64[u] = T[state[u] + c_cur[u]]; }      // (only the lanes that walk: a gather costs by the lane)
#pragma unroll
            for (int u = 0; u < NU; ++u) c_nxt[u] = MODE != 2 ? cls_at(u, j[u] + 1) : 0u;       // (the class of the next character while the transition on this one is in flight)
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const uint32_t e = (uint32_t)e64[u];
                const bool hit = walking[u] && (e & LX_T_CLS_MASK) == c_cur[u];
                const bool fin = hit && (int32_t)e < 0;
                const int jn = j[u] + clen[u];
                fp[u] = fin ? jn : fp[u]; ftag[u] = fin ? (uint32_t)(e64[u] >> 32) : ftag[u];
                state[u] = hit ? ((e >> LX_T_NEXT_SHIFT) & LX_T_NEXT_MASK) : state[u];
                j[u] = hit ? jn : j[u];
                c_cur[u] = (hit && MODE != 2) ? c_nxt[u] : c_cur[u];
                stale[u] = MODE == 2 && hit && jn < L[u];
                missed[u] = missed[u] || (walking[u] && !hit);
            }
        }
        any_act = false;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const bool ev = act[u] && (missed[u] || j[u] >= L[u]);
            if (wv::any(ev)) {
                if (ev) {
                    int32_t *hm = p.home + ea[u] + (int64_t)w3[u];
                    if (fp[u] >= 0) {                                        // fp: the position behind the last character of the match
                        const int32_t id = (int32_t)(ftag[u] & 0x7FFFFFFFu);
                        if (cnt[u] == 0) id0[u] = id;
                        else if (!offs) { if (cnt[u] == 1) hm[0] = id0[u]; hm[cnt[u]] = id; }
                        if (offs) {
                            // the piece's bytes [fb, tb) of the word (MODE 1 counts characters: the byte a character starts at is a bit of startm_keep)
                            int fb = from[u], tb = fp[u];
                            if (MODE == 1) {
                                uint32_t mf = startm_keep, mt = startm_keep;
                                for (int i = 0; i < fb; ++i) mf &= mf - 1u;
                                for (int i = 0; i < tb; ++i) mt &= mt - 1u;
                                fb = mf ? __builtin_ctz(mf) : Lb; tb = mt ? __builtin_ctz(mt) : Lb;
                            }
                            const uint32_t sp = ((p.espan[ea[u]] & WF_SPAN_POS_MASK) + (uint32_t)fb) | ((uint32_t)(tb - fb - 1) << WF_SPAN_LEN_SHIFT);
                            // (id, span) of a piece side by side: one scattered store of eight bytes, and one gather in the merge
                            uint32_t *hs = p.hspan + 2 * (ea[u] + (int64_t)w3[u]);
                            if (cnt[u] == 0) sp0[u] = sp;
                            else { if (cnt[u] == 1) { hs[0] = (uint32_t)id0[u]; hs[1] = sp0[u]; } hs[2 * cnt[u]] = (uint32_t)id; hs[2 * cnt[u] + 1] = sp; }
                        }
                        ++cnt[u];
                        const int nf = fp[u];
                        if (nf >= L[u]) {
                            p.ent[ea[u]] = cnt[u] == 1 ? (uint32_t)id0[u] : (WF_ENT_FLAG | ((uint32_t)cnt[u] << WF_ENT_CNT_SHIFT) | w3[u]);
                            xtra[u] = cnt[u] - 1;                              // (its document has that many ids more than entries)
                            act[u] = false;
                        } else { state[u] = p.ini; j[u] = nf; from[u] = nf; fp[u] = -1; anch[u] = false; missed[u] = false; }
                    } else if (anch[u]) { state[u] = p.ini; j[u] = 0; fp[u] = -1; anch[u] = false; missed[u] = false; }
                    else { p.ent[ea[u]] = WF_ENT_FLAG; act[u] = false; }        // a gap: UnkId
                    if (act[u]) { if (MODE != 2) c_cur[u] = cls_at(u, j[u]); else stale[u] = true; }
                }
            }
            any_act = any_act || wv::any(act[u]);
        }
    }
    // What the documents of these words have in ids beyond their entries: the records of a list are in the order of their documents, so the lanes
    // of one document are neighbours -- their sum by a segmented scan, one atomic per document (a third of one per word).
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        if (!wv::any(xtra[u] != 0)) continue;
        const unsigned long long ri = first + (unsigned long long)(64 * u + lane);
        const int doc = ri < total ? (int)wrec[4 * ri + 3] : -1 - lane;
        int x = xtra[u];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = wv::shfl_up(x, o), dp = wv::shfl_up(doc, o);
            if (lane >= o && dp == doc) x += t;
        }
        const int dn = wv::shfl_down(doc, 1);
        if ((lane == 63 || dn != doc) && x != 0) wv::atomic_add_i32(&p.extra[dlo + (int64_t)doc], x);
    }
    if (MODE == 1) wv::sync();                                               // (the next batch writes the classes anew)
}

// ----------------------------------------------------------------------------------------------------------------------
// k_wp_count / k_wp_merge: a wave takes 64 consecutive documents, one per lane for what is read once (first entry, number of entries, flags,
// place in the output), then works through them together.  An entry is an id, or -- bit 31 -- a word of n ids at its home (n = 0: UnkId).
// ----------------------------------------------------------------------------------------------------------------------
BF_WVD int wf_entry_ids(uint32_t e) { const int n = (int)((e & ~WF_ENT_FLAG) >> WF_ENT_CNT_SHIFT); return (e & WF_ENT_FLAG) ? (n ? n : 1) : 1; }

// counts[d] = ids of document d (tokdll:1308-1310: at most max_ids; 0 for invalid UTF-8: :1151-1153): its entries and what its words of several
// pieces have beyond one id each (k_wp_units has added that up in counts[d]); a document the wave program tokenised keeps the count that wrote
BF_WVD void wf_count_docs(const WfMergeParams &p, int64_t base)
{
    const int64_t d = base + wv::lane();
    if (d >= p.ndocs) return;
    const int st = *p.unsafe != 0 ? WF_D_HARD : p.dstat[d];
    const int old = p.counts[d];
    int cnt = (st & WF_D_HARD) ? old : (st & WF_D_BAD) ? 0 : p.ent_cnt[d] + old;
    if (!(st & WF_D_HARD) && cnt > p.max_ids) cnt = p.max_ids;
    p.counts[d] = cnt;
    if (p.counts_hard) p.counts_hard[d] = (st & WF_D_HARD) ? cnt : 0;
}

// a span (WF_SPAN_*) as the byte offsets of the id's first and last byte in its document (tokdll:1263-1297)
BF_WVD void wf_span_out(const WfMergeParams &p, int64_t at, uint32_t sp)
{
    const int32_t st = (int32_t)(sp & WF_SPAN_POS_MASK);
    p.starts_out[at] = st; p.ends_out[at] = st + (int32_t)(sp >> WF_SPAN_LEN_SHIFT);
}

// One entry vector of a document: ids to out[run ...); returns the ids it held.  OFFS: their spans to the same places of starts_out / ends_out.
template <bool OFFS>
BF_WVD int wf_merge_vec(const WfMergeParams &p, uint32_t e, bool have, int64_t eidx, int64_t dst, int run, int cap)
{
    const int x = have ? wf_entry_ids(e) : 0;
    const int inc = wv::incl_scan(x), pos = run + inc - x;
    const bool flag = have && (e & WF_ENT_FLAG) != 0u;
    const int nn = flag ? (int)((e & ~WF_ENT_FLAG) >> WF_ENT_CNT_SHIFT) : 0;
    const int64_t hi = eidx + (int64_t)(e & WF_ENT_DELTA_MASK);
    int32_t *out = p.ids_out + dst;
    if (have && nn == 0 && pos < cap) { out[pos] = flag ? p.unk : (int32_t)e; if (OFFS) wf_span_out(p, dst + pos, p.espan[eidx]); }
    for (int j = 0; j < nn && pos + j < cap; ++j) {
        if (OFFS) { out[pos + j] = (int32_t)p.hspan[2 * (hi + j)]; wf_span_out(p, dst + pos + j, p.hspan[2 * (hi + j) + 1]); }
        else out[pos + j] = p.home[hi + j];
    }
    return wv::bcast(inc, 63);
}

// ids a trip of the merge stages in LDS: 256 entries, their extra ids and the <= 3 ids carried from the trip before
template <bool OFFS> struct WfMergeLds { static constexpr int N = OFFS ? 512 : 1024; alignas(16) int32_t buf[N]; alignas(16) uint32_t sbuf[OFFS ? N : 4]; };
struct alignas(16) WfQuad { int32_t v[4]; };

template <bool OFFS>
BF_WVD void wf_merge_docs(const WfMergeParams &p, int64_t base, bool &over, WfMergeLds<OFFS> &M)
{
    const int lane = wv::lane();
    const int64_t d = base + lane;
    const bool unsafe = *p.unsafe != 0;
    const int nd = p.ndocs - base < 64 ? (int)(p.ndocs - base) : 64;
    int ec = 0, st = 0, cnt = 0; int64_t eo = 0, o = 0;
    bool capped = false;
    if (d < p.ndocs) {
        st = unsafe ? WF_D_HARD : p.dstat[d]; cnt = p.counts[d]; o = p.id_off[d];
        if (st & WF_D_HARD) { eo = wv_ids_slot(p.doc_off[d], d); ec = cnt; } else { eo = p.ent_off[d]; ec = p.ent_cnt[d]; }
        capped = cnt >= p.max_ids || o + cnt > p.ids_cap;
        if (o + cnt > p.ids_cap) { over = true; cnt = o < p.ids_cap ? (int)(p.ids_cap - o) : 0; }
    }
    // The usual block: no document flagged or cut, the entries of all 64 one contiguous run -- then so are their ids (id_off is the running sum
    // of the counts).  The run is streamed 256 entries per trip without a look at the documents: four consecutive entries per lane and load (the
    // next trip's on its way); the trip's ids are put together in LDS -- a plain id at its place, the pieces of a word of several from its home
    // by the lane that holds its entry -- and leave as whole aligned 16-byte rows (what bounds a streaming kernel here is the number of
    // vector-memory instructions a CU can issue: a store per word of several pieces was most of them); the <= 3 ids behind the last whole row
    // wait for the next trip.  OFFS: the spans travel the same way (a second buffer) and leave as two rows, first and last byte.
    const int64_t eo_n = wv::shfl_down(eo, 1);
    if (!wv::any(lane < nd && (st != 0 || capped))) {
      unsigned long long brk = wv::ballot(lane < nd && (lane + 1 == nd || eo_n != eo + ec));      // the lanes that end a contiguous piece (a range of the flat program ends there)
      for (int first = 0; brk;) {
        const int lastl = __builtin_ctzll(brk); brk &= brk - 1ull;
        const int64_t E0 = wv::bcast(eo, first), E1 = wv::bcast(eo + (int64_t)ec, lastl);
        const int64_t O0 = wv::bcast(o, first);
        first = lastl + 1;
        const int head = (int)(((uintptr_t)(p.ids_out + O0) >> 2) & 3u);
        int64_t oq = O0 - head;                         // M.buf[i] is ids_out[oq + i]; that address is 16-byte aligned
        const bool rows_ok = !OFFS || ((((uintptr_t)(p.starts_out + oq)) | ((uintptr_t)(p.ends_out + oq))) & 15u) == 0u;     // (the three outputs are aligned alike but for a caller's odd pointers)
        int lo = head, carry = head;                    // the first slot that is this piece's; the slots filled so far
        auto load4 = [&](const uint32_t *src, int64_t t, uint32_t (&e)[4]) {
            const int64_t a = t + 4 * lane;
            e[0] = e[1] = e[2] = e[3] = 0u;
            if (a + 4 <= E1) __builtin_memcpy(e, src + a, 16);
            else for (int u = 0; u < 4; ++u) if (a + u < E1) e[u] = src[a + u];
        };
        uint32_t en[4], sn[4] = {0u, 0u, 0u, 0u};
        load4(p.ent, E0, en);
        if (OFFS) load4(p.espan, E0, sn);
        for (int64_t t = E0; t < E1; t += 256) {
            const int64_t a = t + 4 * lane;
            uint32_t e[4] = {en[0], en[1], en[2], en[3]}, sp[4] = {sn[0], sn[1], sn[2], sn[3]};
            const int nin = a + 4 <= E1 ? 4 : (a < E1 ? (int)(E1 - a) : 0);
            if (t + 256 < E1) { load4(p.ent, t + 256, en); if (OFFS) load4(p.espan, t + 256, sn); }
            const bool flagged = ((e[0] | e[1] | e[2] | e[3]) & WF_ENT_FLAG) != 0u;
            const bool any_f = wv::any(flagged);
            int n[4], xs = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) { n[u] = u < nin ? wf_entry_ids(e[u]) : 0; xs += n[u]; }
            int pos0 = carry + 4 * lane, sum = E1 - t < 256 ? (int)(E1 - t) : 256;
            if (any_f) { const int inc = wv::incl_scan(xs); pos0 = carry + inc - xs; sum = wv::bcast(inc, 63); }
            if (carry + sum > WfMergeLds<OFFS>::N) {
                // more ids than the buffer holds (entries of four ids and more on average): what waits leaves, the trip's ids go out one by one
                if (lane >= lo && lane < carry) { p.ids_out[oq + lane] = M.buf[lane]; if (OFFS) wf_span_out(p, oq + lane, M.sbuf[lane]); }
                int pos = pos0;
                for (int u = 0; u < 4; ++u) {
                    if (u >= nin) break;
                    const uint32_t eu = e[u];
                    const int nn = (eu & WF_ENT_FLAG) ? (int)((eu & ~WF_ENT_FLAG) >> WF_ENT_CNT_SHIFT) : -1;       // -1: a plain id, 0: UnkId
                    if (nn <= 0) { p.ids_out[oq + pos] = nn < 0 ? (int32_t)eu : p.unk; if (OFFS) wf_span_out(p, oq + pos, sp[u]); }
                    else {
                        const int64_t hi = (a + u) + (int64_t)(eu & WF_ENT_DELTA_MASK);
                        for (int j = 0; j < nn; ++j) {
                            if (OFFS) { p.ids_out[oq + pos + j] = (int32_t)p.hspan[2 * (hi + j)]; wf_span_out(p, oq + pos + j, p.hspan[2 * (hi + j) + 1]); }
                            else p.ids_out[oq + pos + j] = p.home[hi + j];
                        }
                    }
                    pos += n[u];
                }
                const int64_t np = oq + carry + sum;
                const int h2 = (int)(((uintptr_t)(p.ids_out + np) >> 2) & 3u);
                oq = np - h2; lo = carry = h2;
                wv::sync();
                continue;
            }
            // ---- the trip's ids to their slots
            if (!flagged) {
#pragma unroll
                for (int u = 0; u < 4; ++u) if (u < nin) { M.buf[pos0 + u] = (int32_t)e[u]; if (OFFS) M.sbuf[pos0 + u] = sp[u]; }
            } else {
                int pos = pos0;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t eu = e[u];
                    const int nn = (eu & WF_ENT_FLAG) ? (int)((eu & ~WF_ENT_FLAG) >> WF_ENT_CNT_SHIFT) : -1;
                    if (u < nin && nn <= 0) { M.buf[pos] = nn < 0 ? (int32_t)eu : p.unk; if (OFFS) M.sbuf[pos] = sp[u]; }
                    pos += n[u];
                }
            }
            if (any_f) {
                // the words of several pieces, one per lane and turn: four pieces in one load (the homes have 64 entries of slack behind them)
                uint32_t fm = 0;
#pragma unroll
                for (int u = 0; u < 4; ++u) if (u < nin && (e[u] & WF_ENT_FLAG) && (e[u] & ~WF_ENT_FLAG) >> WF_ENT_CNT_SHIFT) fm |= 1u << u;
                const int p1 = pos0 + n[0], p2 = p1 + n[1], p3 = p2 + n[2];
                while (wv::any(fm != 0u)) {
                    if (fm) {
                        const int u = __builtin_ctz(fm); fm &= fm - 1u;
                        const uint32_t eu = u == 0 ? e[0] : u == 1 ? e[1] : u == 2 ? e[2] : e[3];
                        const int pu = u == 0 ? pos0 : u == 1 ? p1 : u == 2 ? p2 : p3;
                        const int nn = (int)((eu & ~WF_ENT_FLAG) >> WF_ENT_CNT_SHIFT);
                        const int64_t hi = (a + u) + (int64_t)(eu & WF_ENT_DELTA_MASK);
                        if (OFFS) {
                            const uint32_t *hp = p.hspan + 2 * hi;            // (id, span) pairs
                            uint32_t v[8];
                            __builtin_memcpy(v, hp, 32);
                            M.buf[pu] = (int32_t)v[0]; M.sbuf[pu] = v[1];
                            if (nn > 1) { M.buf[pu + 1] = (int32_t)v[2]; M.sbuf[pu + 1] = v[3]; }
                            if (nn > 2) { M.buf[pu + 2] = (int32_t)v[4]; M.sbuf[pu + 2] = v[5]; }
                            if (nn > 3) { M.buf[pu + 3] = (int32_t)v[6]; M.sbuf[pu + 3] = v[7]; }
                            for (int j = 4; j < nn; ++j) { M.buf[pu + j] = (int32_t)hp[2 * j]; M.sbuf[pu + j] = hp[2 * j + 1]; }
                        } else {
                            const int32_t *hm = p.home + hi;
                            int32_t v[4];
                            __builtin_memcpy(v, hm, 16);
                            M.buf[pu] = v[0];
                            if (nn > 1) M.buf[pu + 1] = v[1];
                            if (nn > 2) M.buf[pu + 2] = v[2];
                            if (nn > 3) M.buf[pu + 3] = v[3];
                            for (int j = 4; j < nn; ++j) M.buf[pu + j] = hm[j];
                        }
                    }
                }
            }
            wv::sync();
            // ---- whole rows out
            const int total = carry + sum;
            const bool last = t + 256 >= E1;
            const int nfull = last ? total : (total & ~3);
            for (int q = 4 * lane; q < nfull; q += 256) {
                const WfQuad v = *(const WfQuad *)(M.buf + q);
                const bool whole = q >= lo && q + 4 <= nfull;
                if (whole) *(WfQuad *)(p.ids_out + oq + q) = v;
                else for (int j = 0; j < 4; ++j) if (q + j >= lo && q + j < nfull) p.ids_out[oq + q + j] = v.v[j];
                if (OFFS) {
                    const WfQuad w = *(const WfQuad *)(M.sbuf + q);
                    WfQuad sa, sb;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { sa.v[j] = (int32_t)((uint32_t)w.v[j] & WF_SPAN_POS_MASK); sb.v[j] = sa.v[j] + (int32_t)((uint32_t)w.v[j] >> WF_SPAN_LEN_SHIFT); }
                    if (whole && rows_ok) { *(WfQuad *)(p.starts_out + oq + q) = sa; *(WfQuad *)(p.ends_out + oq + q) = sb; }
                    else for (int j = 0; j < 4; ++j) if (q + j >= lo && q + j < nfull) { p.starts_out[oq + q + j] = sa.v[j]; p.ends_out[oq + q + j] = sb.v[j]; }
                }
            }
            const int rest = total - nfull;
            const int32_t keep = lane < rest ? M.buf[nfull + lane] : 0;
            const uint32_t skeep = (OFFS && lane < rest) ? M.sbuf[nfull + lane] : 0u;
            wv::sync();
            if (lane < rest) { M.buf[lane] = keep; if (OFFS) M.sbuf[lane] = skeep; }
            oq += nfull; lo = 0; carry = rest;
        }
      }
      return;
    }
    if (cnt == 0) ec = 0;
    for (int i = 0; i < nd; ++i) {
        const int n = wv::bcast(ec, i);
        if (n == 0) continue;
        const int64_t src = wv::bcast(eo, i), dst = wv::bcast(o, i);
        const int cap = wv::bcast(cnt, i);
        if (wv::bcast(st, i) & WF_D_HARD) {                                   // tokenised by the wave program: its ids are in place
            if (!OFFS) for (int t = lane; t < cap; t += 64) p.ids_out[dst + t] = p.ids_tmp[src + t];      // (OFFS: k_compact_text copies them, and makes their byte offsets)
            continue;
        }
        int run = 0;
        for (int t = 0; t < n && run < cap; t += 64) run += wf_merge_vec<OFFS>(p, t + lane < n ? p.ent[src + t + lane] : 0u, t + lane < n, src + t + lane, dst, run, cap);
    }
}

} // namespace bfa
