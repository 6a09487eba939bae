// bf_bpe_wave_body.h -- BPE (bpe-opt flavour) as a wave program: the structure of bf_wave_body.h with another unit.
//
// Reproduces FATokenSegmentationTools_1best_bpe_t<int>::Process (cl/inc/FATokenSegmentationTools_1best_bpe_t.h:126-316) and the id
// loop of TextToIdsWithOffsets_sp (tokdll:1509-1532) on the class stream the prologue kernel produced (k_prep_sp), for models whose
// load-time analysis (bf_model.cpp, Model::bpe_wave_ok) proves (gpt2.bin, bpe_example*.bin, and -- the flavour with merge ranks, whose arcs are
// ordered by the entry's place in "rank descending, id ascending": prio / place_id, 20 bits of place in the keys here, a larger place hands the
// document back -- roberta.bin):
//   (a) m_fFastBpe (bpe-opt): the collection loop takes a whole word as ONE arc and jumps behind it (:189-206,228-230);
//   (b) no vocabulary entry has U+2581 behind its first position: an arc never leaves the stretch from one U+2581 to the next
//       ("word"), so every word is a segment of the merge procedure of its own (:234-313: an arc is applied depending on the
//       interior marks of the positions it covers only);
// What can still couple two words is the unknown arc of a start position without any arc, which merges with an unknown arc right before
// it (:212-225): a unit that meets such a start -- or a symbol outside the alphabet -- hands its document back.  So does a word of more
// than BW_WORD_MAX positions or with more arcs than the lane's window holds (flags[d] = 1, no ids): k_bpe_seg (bf_bpe_seg_body.h, one wave per
// document, keys with 22 bits of priority; launch_bpe_seg_flags) redoes those documents from scratch.  Measured with the oracle on the config-3 corpus (gpt2.bin): 80 % of the words are one entry,
// 5.7 % of the documents are handed back (a word with more than 32 arcs of more than one element).
//
//   fill    a chunk of 512 stream elements goes into the LDS ring, eight per lane; words = [a U+2581 (or the document's first
//           element) .. the element before the next U+2581]: starts and ends from one 8-bit mask per lane, two shuffles, one ballot,
//           one prefix sum -- the chunk-wide pass of bf_wave_body.h with one element kind;
//   table   (round 6) the words just queued, one per lane and 64 per trip: a word of U+2581 + at most 12 symbols (classes below 256) is looked up
//           in the word table by its symbols (lookup_words; bf_model.cpp build_bpe_word_table: every word the collection loop takes WHOLE,
//           :189-206 -- a function of the word alone): two 16-byte gathers per lane, a hit IS the word's id.  80 % of the words of running text
//           never reach a unit;
//   units   one word per lane.  First the walk of the whole word from its first element, output weights summed: if it ends on the
//           word's last element in a final state and an earlier prefix was an entry too (:189 count_at_start < narcs), the word is
//           that entry.  Otherwise the lane collects the arcs of every start position of the word into its LDS window
//           ([id:20 | start:6 | end:6], the key order of :238-255; one-element arcs are not stored: they never mark an interior
//           and any applied arc of the same start overrides them, so they are looked up again only for a position that ends up a
//           token of its own), sorts and applies them against a 64-bit interior mask (:274-296) and emits the pieces in position
//           order (:299-313);
//   retire  as in bf_wave_body.h: ids move from LDS / their provisional homes down to their place in the document.
// HOME form (round 6, the shipped one): nothing is retired in order.  Every word's ids stay at its HOME -- the cells of the document's staging slot
// under the word's own elements (a word has no more ids than elements; the fill writes BW_HOME_NONE into every cell first) -- the document's count
// is added up as its words report (table hits per trip, units by an LDS atomic) and a streaming kernel behind (k_bpe_home_gather) squeezes the cells
// that hold an id.  A word the table answers never enters the
// queue: the queue holds only the words a unit must walk, sixteen documents' worth instead of three, so the lanes of a round are busy -- with
// the in-order queue the table's hits sat in it waiting for their turn and the kernel's time did not move (18.8 vs 19.1 ms per 1 M documents).
// Include AFTER a definition of namespace wv (bf_kernels.hip on the device, tests/hosttest/wave_emu.h in the test simulator).
#pragma once
#include "bf_wave_body.h"
#include "bf_bpe_wave.h"

namespace bfa {

struct alignas(16) BwRow { uint32_t k0lo, k0hi, k1, id; };      // a row of the word table (bf_flat_key.h: the flat program's layout)
#if defined(__HIPCC__)
// both candidate rows whole and in flight together (left alone the compiler fetches the second only after the first has come back: bf_flat_body.h)
typedef uint32_t bw_u32x4 __attribute__((ext_vector_type(4)));
#define BF_BW_LOAD_ROWS(A, B, PA, PB) { bw_u32x4 ra_, rb_; const BwRow *pa_ = (PA), *pb_ = (PB); \
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %3, off\n\ts_waitcnt vmcnt(0)" : "=&v"(ra_), "=&v"(rb_) : "v"(pa_), "v"(pb_) : "memory"); \
    A.k0lo = ra_.x; A.k0hi = ra_.y; A.k1 = ra_.z; A.id = ra_.w; B.k0lo = rb_.x; B.k0hi = rb_.y; B.k1 = rb_.z; B.id = rb_.w; }
#else
#define BF_BW_LOAD_ROWS(A, B, PA, PB) { A = *(PA); B = *(PB); }
#endif

constexpr int BW_WORD_MAX = 62;          // positions of a word (the interior mask has 64 bits: position + 1 must fit)
constexpr int BW_PRIV = 16;              // arcs of more than one element in a lane's own window: 85 % of the words of the config-3 corpus that are not one entry
constexpr int BW_POOL_N = 5, BW_POOL_ARCS = 48;   // the wave's overflow windows: 5 of 48 more arcs each (12 % of those words have 17-24 arcs, 2.7 % 25-32,
constexpr int BW_WIN = BW_PRIV + BW_POOL_ARCS;    // 0.4 % 33-48, one in a million more than 64): BW_WIN = 64 = one arc per lane in solve_big(); a word
                                                  // with more is solved by its lane alone, its arcs in global memory (unit_huge)
constexpr uint32_t BW_TK_TS = 1u << 29;  // token flag: the word starts with U+2581 (token_start of :176)
constexpr uint32_t BW_DT_FALLBACK = 4;

template <int RING_, int QCAP_, int DTN_>
struct BwLds {
    static constexpr int RING = RING_, QCAP = QCAP_, DTN = DTN_;
    alignas(16) uint16_t ring[RING];
    int64_t dt_slot[DTN], dt_doc[DTN];
    alignas(8) WvTok q[QCAP];
    int32_t dt_cap[DTN], dt_cnt[DTN]; uint32_t dt_flags[DTN], dt_rbase[DTN];
    uint32_t win[BW_PRIV * 64];          // arc windows, structure-of-arrays: arc k (< BW_PRIV) of lane l at win[k * 64 + l]
    uint32_t pool[BW_POOL_ARCS * BW_POOL_N];   // overflow windows: arc BW_PRIV + k of the lane that holds window w at pool[k * BW_POOL_N + w]
    uint32_t pool_free;                  // bit w: overflow window w is free
    uint16_t qc[QCAP];
    int64_t dslot[WV_GRAB_MAX]; int32_t dlen[WV_GRAB_MAX];     // stream slot and length of the documents taken from the work counter
    uint32_t spare32; uint16_t spare;
};

constexpr int32_t BW_HOME_NONE = (int32_t)0x80000000;      // HOME form: a cell of the staging slot that holds no id (an id is a dictionary id + IdOffset: never this)

template <class LDS, int STEPS = 3, int UMIN = 4, bool HOME = false>
struct BpeWave {
    static constexpr int RING = LDS::RING, QCAP = LDS::QCAP, DTN = LDS::DTN;
    static constexpr uint32_t RMASK = RING - 1, QMASK = QCAP - 1, DMASK = DTN - 1;
    static_assert((RING & (RING - 1)) == 0 && RING >= 1024 && (QCAP & (QCAP - 1)) == 0 && QCAP >= 128 && (DTN & (DTN - 1)) == 0, "sizes");

    const BpeWaveParams &p; LDS &S;
    int lane;
    // ---- wave-uniform state (names as in bf_wave_body.h)
    uint32_t u_need, rhi, rlo, q_tail, q_issue, q_retire, dt_head, dt_tail;
    int64_t dbase; int di, dn; int st_round, st_wave, st_waves;
    bool have_doc, exiting;
    int64_t slot; int n; uint32_t rbase; int dec, done, open_start; uint32_t curk; bool fall;   // current document

    BF_WVD BpeWave(const BpeWaveParams &p_, LDS &S_) : p(p_), S(S_)
    {
        lane = wv::lane(); rhi = rlo = 0; u_need = 0xFFFFFFFFu; q_tail = q_issue = q_retire = 0; dt_head = dt_tail = 0;
        dbase = 0; di = dn = 0; st_round = st_wave = 0; st_waves = 1; have_doc = exiting = false;
        slot = 0; n = 0; rbase = 0; dec = done = 0; open_start = -1; curk = 0; fall = false;
    }
    BF_WVD int ring_free() const { return RING - (int)(rhi - rlo); }
    // what orders the arcs (:238-255; with merges ..._with_merges_t.h:242-262): the id, or the entry's place in the order by rank
    BF_WVD int32_t order_of(uint32_t mph) const { return p.prio ? (int32_t)(p.prio[mph] >> 1) : p.info[mph].id; }
    BF_WVD int id_of_order(int o) const { return p.place_id ? p.place_id[o] : o; }

    // ------------------------------------------------------------------------------------------------------------------
    // fill
    // ------------------------------------------------------------------------------------------------------------------
    BF_WVD void put_token(uint32_t t, int pos, int len, bool ts)
    {
        WvTok e; e.pos = rbase + (uint32_t)pos; e.w = (uint32_t)len | ((curk & 0xFFu) << 16) | (ts ? BW_TK_TS : 0u);
        S.q[t & QMASK] = e;
    }
    // 512 stream elements (or what is left of the document) -> ring, eight per lane
    BF_WVD void load_chunk()
    {
        const int pos = dec;
        const int total = n - pos < WV_CHUNK ? n - pos : WV_CHUNK;
        int nb = total - lane * 8; nb = nb < 0 ? 0 : (nb > 8 ? 8 : nb);
        const uint16_t *src = p.stream + slot + pos + lane * 8;
        const uint32_t r = rbase + (uint32_t)pos + (uint32_t)(lane * 8);
        if (nb == 8 && (r & 7u) == 0) {                                 // a whole row: 16 bytes in (the slot is 2-byte aligned only), one 16-byte row out
            uint32_t d[4];
            __builtin_memcpy(d, src, 16);
            uint32_t *dst = (uint32_t *)(S.ring + (r & RMASK));
            dst[0] = d[0]; dst[1] = d[1]; dst[2] = d[2]; dst[3] = d[3];
        } else for (int k = 0; k < nb; ++k) S.ring[(r + (uint32_t)k) & RMASK] = src[k];
        if (HOME) { int32_t *cell = p.ids_tmp + slot + pos + lane * 8; for (int k = 0; k < nb; ++k) cell[k] = BW_HOME_NONE; }      // no id here (yet)
        dec += total; rhi = rbase + (uint32_t)dec;
        wv::sync();
    }
    // the words that END in the next (up to) 512 elements are queued; a word that is still open at the end of the decoded part is
    // carried in open_start.  Returns false (nothing committed) when the queue lacks room.
    // `lim`: look at no more than that many elements (a chunk of one-letter words holds more tokens than the queue has room for)
    BF_WVD bool resolve_chunk(int lim)
    {
        const int cb = done;
        const int total = dec - cb < lim ? dec - cb : lim;
        const bool at_end = cb + total == n;                            // the document ends with these elements
        int nb = total - lane * 8; nb = nb < 0 ? 0 : (nb > 8 ? 8 : nb);
        const int lane0 = cb + lane * 8;
        const uint32_t r = rbase + (uint32_t)lane0;
        uint32_t dm = 0, bad = 0;
        for (int k = 0; k < 8; ++k) {
            const uint32_t e = k < nb ? (uint32_t)S.ring[(r + (uint32_t)k) & RMASK] : 0u;
            if (k < nb && e == p.cls_delim) dm |= 1u << k;
            if (k < nb && e >= SG_CLS_DELIM_ABSENT) bad |= 1u << k;    // a symbol outside the alphabet: an unknown arc (:212-225)
        }
        if (wv::any(bad != 0)) { fall = true; if (p.stats && lane == 0) wv::atomic_add(&p.stats[2], 1ull); }
        const uint32_t vm = nb >= 8 ? 0xFFu : ((1u << nb) - 1u);
        uint32_t h = dm & vm;                                           // word starts
        if (lane0 == 0 && nb > 0) h |= 1u;                              // the document's first element starts a word whatever it is
        uint32_t next_first = wv::shfl_down(h & 1u, 1);
        if (lane == 63) next_first = 0u;
        uint32_t en = vm & ((h >> 1) | (next_first << 7));              // the element before a start ends a word
        const int last_lane = (total - 1) >> 3; const uint32_t last_bit = 1u << ((total - 1) & 7);
        if (lane == last_lane && total > 0) { if (at_end) en |= last_bit; else en &= ~last_bit; }
        const bool stays_open = !at_end && total > 0;                   // the last element's word goes on (or ends exactly here: not known yet)
        const bool cont = open_start >= 0;
        const int hl = h ? lane0 + (31 - __builtin_clz(h)) : -1;        // this lane's last word start
        const unsigned long long HB = wv::ballot(h != 0);
        const unsigned long long hb_lt = HB & ((1ull << lane) - 1ull);
        const int hsrc = hb_lt ? 63 - __builtin_clzll(hb_lt) : 0;
        int hprev = wv::shfl(hl, hsrc);
        if (!hb_lt) hprev = open_start;
        const int new_open = stays_open ? (HB ? wv::bcast(hl, 63 - __builtin_clzll(HB)) : open_start) : -1;
        const bool carry_end = cont && wv::any(lane == 0 && (h & 1u) != 0);            // the open word ended just before these elements
        uint32_t tk = en;
        const int c = __builtin_popcount(tk);
        const int inc = wv::incl_scan(c);
        const uint32_t ntok = (uint32_t)wv::bcast(inc, 63) + (carry_end ? 1u : 0u);
        if ((q_tail - q_retire) + ntok > (uint32_t)QCAP) return false;
        uint32_t t = q_tail + (carry_end ? 1u : 0u) + (uint32_t)(inc - c);
        bool toolong = false;
        if (carry_end) {
            const int len = cb - open_start;
            if (len > BW_WORD_MAX) toolong = true;
            if (lane == 0) put_token(q_tail, open_start, len > BW_WORD_MAX ? BW_WORD_MAX : len, S.ring[(rbase + (uint32_t)open_start) & RMASK] == p.cls_delim);
        }
        while (wv::any(tk != 0)) {
            if (tk) {
                const int bit = __builtin_ctz(tk); tk &= tk - 1u;
                const int bpos = lane0 + bit;
                const uint32_t hm = h & ((2u << bit) - 1u);
                const int start = hm ? lane0 + (31 - __builtin_clz(hm)) : hprev;
                int len = bpos - start + 1;
                if (len > BW_WORD_MAX) { toolong = true; len = BW_WORD_MAX; }
                put_token(t, start, len, S.ring[(rbase + (uint32_t)start) & RMASK] == p.cls_delim);
                ++t;
            }
        }
        if (stays_open && cb + total - new_open > BW_WORD_MAX) toolong = true;       // (its head may leave the ring before its end is seen)
        if (wv::any(toolong)) { fall = true; if (p.stats && lane == 0) wv::atomic_add(&p.stats[3], 1ull); }
        wv::sync();
        q_tail += lookup_words(q_tail, ntok); done = cb + total;
        open_start = stays_open ? new_open : -1;
        if (fall) { n = dec; done = dec; open_start = -1; }             // the rest of a document that is handed back is not looked at
        return true;
    }

    // The words [first, first + n) of the queue, just put there by resolve_chunk: every one gets its count slot (0: a unit will take it), and a
    // word the table holds its id at once (count slot 2 = one id, in the token's `pos` -- what unit_event writes for a word taken whole).
    // The key of a word is the classes of the (at most 12) symbols behind its U+2581, one byte each; a class of 256 or more has no byte.
    // HOME form: the id of a word the table holds goes to the word's home at once and the word leaves the queue -- the words that stay move up;
    // returns how many stay.
    BF_WVD uint32_t lookup_words(uint32_t first, uint32_t n)
    {
        uint32_t kept = 0;
        for (uint32_t t0 = 0; t0 < n; t0 += 64u) {
            const bool have = t0 + (uint32_t)lane < n;
            const uint32_t sl = (first + t0 + (uint32_t)lane) & QMASK;
            WvTok e; e.pos = 0; e.w = 0;
            if (have) e = S.q[sl];
            const int len = (int)(e.w & WV_TK_LEN_MASK);
            const int kn = (have && p.W && (e.w & BW_TK_TS) && len >= 2 && len <= WF_KEY_CHARS + 1) ? len - 1 : 0;
            const uint32_t r0 = (e.pos + 1u) & RMASK;
            uint32_t c[6];
            if (r0 + 12u <= (uint32_t)RING) __builtin_memcpy(c, S.ring + r0, 24);
            else for (int k = 0; k < 6; ++k) c[k] = (uint32_t)S.ring[(r0 + 2u * (uint32_t)k) & RMASK] | ((uint32_t)S.ring[(r0 + 2u * (uint32_t)k + 1u) & RMASK] << 16);
            const uint32_t km0 = kn >= 4 ? 0xFFFFFFFFu : ((1u << (8 * kn)) - 1u);
            const uint32_t km1 = kn >= 8 ? 0xFFFFFFFFu : kn > 4 ? ((1u << (8 * (kn - 4))) - 1u) : 0u;
            const uint32_t km2 = kn >= 12 ? 0xFFFFFFFFu : kn > 8 ? ((1u << (8 * (kn - 8))) - 1u) : 0u;
            const uint32_t w0 = wv::perm(c[1], c[0], 0x06040200u) & km0, w1 = wv::perm(c[3], c[2], 0x06040200u) & km1, w2 = wv::perm(c[5], c[4], 0x06040200u) & km2;
            const uint32_t hb = (wv::perm(c[1], c[0], 0x07050301u) & km0) | (wv::perm(c[3], c[2], 0x07050301u) & km1) | (wv::perm(c[5], c[4], 0x07050301u) & km2);
            const uint64_t k0 = (uint64_t)w0 | ((uint64_t)w1 << 32);
            const uint32_t x = wf_mix(k0, w2, p.m0);
            BwRow A, B; A.k0lo = A.k0hi = A.k1 = A.id = 0; B = A;
            if (wv::any(kn != 0)) BF_BW_LOAD_ROWS(A, B, (const BwRow *)p.W + (kn ? wf_h(x, p.m1, p.wbits) : 0u), (const BwRow *)p.W + (kn ? wf_h(x, p.m2, p.wbits) : 0u));
            const uint32_t klen = (uint32_t)kn << WF_ROW_LEN_SHIFT;
            const bool hita = ((A.k0lo ^ w0) | (A.k0hi ^ w1) | (A.k1 ^ w2) | ((A.id ^ klen) & ~WF_ROW_ID_MASK)) == 0u;
            const bool hitb = ((B.k0lo ^ w0) | (B.k0hi ^ w1) | (B.k1 ^ w2) | ((B.id ^ klen) & ~WF_ROW_ID_MASK)) == 0u;
            const bool hit = kn != 0 && hb == 0u && (hita || hitb);
            if (p.stats && hit) wv::atomic_add(&p.stats[12], 1ull);
            if (HOME) {
                { const int nh = __builtin_popcountll(wv::ballot(hit)); if (lane == 0 && nh) S.dt_cnt[curk & DMASK] += nh; }      // the document's ids so far (settle() writes its count)
                if (hit) p.ids_tmp[slot + (int64_t)(e.pos - rbase)] = (int32_t)((hita ? A.id : B.id) & WF_ROW_ID_MASK);      // (the words of a chunk are the current document's)
                const unsigned long long keep = wv::ballot(have && !hit);
                wv::sync();                                              // every lane has read its token: the ones that stay move up
                if (have && !hit) { const uint32_t dl = (first + kept + wv::mbcnt(keep)) & QMASK; S.q[dl] = e; S.qc[dl] = 0; }
                kept += (uint32_t)__builtin_popcountll(keep);
            } else {
                if (hit) S.q[sl].pos = (hita ? A.id : B.id) & WF_ROW_ID_MASK;
                if (have) S.qc[sl] = hit ? (uint16_t)2 : (uint16_t)0;
                kept = n;
            }
        }
        wv::sync();
        return kept;
    }

    BF_WVD bool open_document(int64_t d, int64_t sl, int len)
    {
        if (len <= 0) { if (lane == 0) { p.counts[d] = 0; p.flags[d] = 0; } return false; }         // the prologue decided: TextToIds returns 0
        n = len; slot = sl;
        int cap = p.max_ids; if (cap > n) cap = n; if (cap < 0) cap = 0;
        curk = dt_tail++;
        const uint32_t ke = curk & DMASK;
        rhi = (rhi + 7u) & ~7u; rbase = rhi;
        if (lane == 0) { S.dt_slot[ke] = sl; S.dt_doc[ke] = d; S.dt_cap[ke] = cap; S.dt_cnt[ke] = 0; S.dt_flags[ke] = 0; S.dt_rbase[ke] = rbase; }
        dec = done = 0; open_start = -1; fall = false;
        wv::sync();
        return true;
    }
    BF_WVD void close_document()
    {
        if (lane == 0) S.dt_flags[curk & DMASK] |= WV_DT_CLOSED | (fall ? BW_DT_FALLBACK : 0u);
        wv::sync();
    }
    BF_WVD bool fill_step(int grab)
    {
        if (!have_doc) {
            if (exiting || dt_tail - dt_head >= (uint32_t)DTN) return false;
            if (di >= dn) {
                wv::sync();                                             // every lane has read the range before (dslot / dlen)
                unsigned long long base = 0;
                if (p.next_doc) {
                    if (lane == 0) base = wv::atomic_add(p.next_doc, (unsigned long long)grab);
                    base = wv::bcast(base, 0);
                } else { base = ((unsigned long long)st_wave + (unsigned long long)st_round * (unsigned long long)st_waves) * (unsigned long long)grab; ++st_round; }
                if ((int64_t)base >= p.ndocs) { exiting = true; return false; }
                dbase = (int64_t)base; di = 0; dn = dbase + grab < p.ndocs ? grab : (int)(p.ndocs - dbase);
                if (lane < dn) { const int64_t d = dbase + lane; S.dslot[lane] = (int64_t)p.slot_mul * (p.doc_off[d] + d); S.dlen[lane] = p.lens[d]; }
                wv::sync();
            }
            have_doc = open_document(dbase + di, S.dslot[di], S.dlen[di]);
            ++di;
            return true;
        }
        if (done < dec) {
            if (resolve_chunk(WV_CHUNK)) return true;
            if (q_tail != q_retire) return false;                       // the queue drains first
            return resolve_chunk(64);                                   // <= 65 tokens: fits an empty queue
        }
        if (dec < n) {
            if (ring_free() < WV_CHUNK || (q_tail - q_retire) + (uint32_t)(QCAP / 2) > (uint32_t)QCAP) return false;
            load_chunk(); return true;
        }
        close_document(); have_doc = false; return true;
    }

    // ------------------------------------------------------------------------------------------------------------------
    // units.  mode: 0 idle, 1 the walk of the whole word, 2 collecting the arcs of start position s0, 3 collected: waits for the
    // solve pass (sort / apply / emit).  As in bf_wave_body.h a round is: events (the lanes whose walk is over, by vote), new words
    // for idle lanes, STEPS transitions for every lane in straight-line code; the solve pass runs when enough lanes wait for it.
    // (Measured on MI355X with the event and solve code inside every transition: 58.9 ms per 1 M documents of config 3.)
    // ------------------------------------------------------------------------------------------------------------------
    struct Unit {
        int tok; uint32_t rs; int L; uint32_t ke;
        int64_t g;                       // HOME form: the word's first element in the class stream (the units read their symbols there, not in the ring)
        int mode, s0, j; uint32_t state; int sum; uint32_t seen, last_final, ovf; int narc, narc0, pw;      // seen / last_final / ovf: 0 or 1 (bf_wave.h wv_b)   // narc0: arcs before the walk of s0; pw: overflow window (-1: none)
        unsigned long long single;       // bit s: the element at position s is an entry by itself (its one-element arc is not stored)
    };
    // element `pos` of the unit's word
    BF_WVD uint32_t sym_at(const Unit &u, int pos) const { return HOME ? (uint32_t)p.stream[u.g + pos] : (uint32_t)S.ring[(u.rs + (uint32_t)pos) & RMASK]; }
    BF_WVD uint32_t *arc_at(const Unit &u, int a) { return a < BW_PRIV ? &S.win[a * 64 + lane] : &S.pool[(a - BW_PRIV) * BW_POOL_N + u.pw]; }
    BF_WVD void unit_finish(Unit &u, int cnt) {
        if (u.pw >= 0) { wv::lds_or(&S.pool_free, 1u << u.pw); u.pw = -1; }          // (lanes that finish in the same instruction return different windows)
        if (HOME && cnt > 0) wv::lds_add((uint32_t *)&S.dt_cnt[u.ke], (uint32_t)cnt);      // (several lanes may finish words of one document in the same instruction)
        S.qc[(uint32_t)u.tok & QMASK] = (uint16_t)(cnt + 1); u.tok = -1; u.mode = 0; u.j = u.L; }
    BF_WVD void unit_fallback(Unit &u, int why)
    {
        if (p.stats) wv::atomic_add(&p.stats[why], 1ull);
        S.dt_flags[u.ke] |= BW_DT_FALLBACK;                             // (several lanes may do this for one document: they write the same bit)
        unit_finish(u, 0);
    }
    BF_WVD void unit_begin(Unit &u, uint32_t t)
    {
        if (S.qc[t & QMASK] != 0) return;                                // the word table had it (lookup_words): the lane stays idle and asks again
        u.tok = (int)t;
        const WvTok e = S.q[t & QMASK];
        u.rs = e.pos; u.L = (int)(e.w & WV_TK_LEN_MASK); u.ke = (e.w >> 16) & DMASK;
        if (HOME) u.g = S.dt_slot[u.ke] + (int64_t)(u.rs - S.dt_rbase[u.ke]);
        u.state = p.initial; u.sum = 0; u.j = 0; u.s0 = 0; u.seen = 0; u.last_final = 0; u.ovf = 0; u.narc = 0; u.narc0 = 0; u.pw = -1; u.single = 0;
        u.mode = (e.w & BW_TK_TS) ? 1 : 2;                              // only a word that starts with U+2581 can be taken whole (:176,189)
        if (p.stats) wv::atomic_add(&p.stats[0], 1ull);
    }
    // One transition for every lane at once, straight-line: a lane that is not walking (idle, waiting for the solve pass, walk over)
    // gathers entry 0 and keeps what it has.  j >= L afterwards: the walk is over (a miss sets j = L) and waits for unit_event().
    // An arc of collection mode is stored as [MPH index : 20 | start : 6 | end : 6]; the solve pass turns the index into the id.
    BF_WVD void unit_step(Unit &u, uint32_t c_home = 0)
    {
        // (truth values as integers, combined in the vector unit: bf_wave.h wv_b)
        const uint32_t act = wv_b(u.tok >= 0) & wv_b(u.mode != 3) & wv_b(u.j < u.L);
        const uint32_t c = HOME ? c_home : (uint32_t)S.ring[(u.rs + (uint32_t)u.j) & RMASK];
        const uint32_t valid = act & wv_b(c < SG_CLS_DELIM_ABSENT);
        const uint64_t e = p.T[valid ? u.state + c : 0u];
        const uint32_t hit = valid & wv_b((uint32_t)(e & SG_CLS_MASK) == c);
        const uint32_t fin = hit & (uint32_t)((e >> 20) & 1ull);               // SG_FINAL
        u.state = hit ? (uint32_t)((e >> SG_NEXT_SHIFT) & SG_NEXT_MASK) : u.state;
        u.sum += (int)((uint32_t)(e >> SG_OW_SHIFT) & (0u - hit));
        const uint32_t m1 = wv_b(u.mode == 1), m2 = wv_b(u.mode == 2);
        u.last_final = (m1 & act) ? (fin & wv_b(u.j == u.L - 1) & u.seen) : u.last_final;   // :189: final on the word's last element, after an earlier arc
        const uint32_t arc = fin & m2 & wv_b(u.j > u.s0);
        const uint32_t room = (wv_b(u.narc < BW_PRIV) | (wv_b(u.pw >= 0) & wv_b(u.narc < BW_WIN))) & wv_b((uint32_t)u.sum < (1u << 20));
        const uint32_t put = arc & room, stop = arc & (room ^ 1u);
        uint32_t *dst = put ? arc_at(u, u.narc) : &S.spare32;
        *dst = ((uint32_t)u.sum << 12) | ((uint32_t)u.s0 << 6) | (uint32_t)u.j;
        u.narc += (int)put;
        u.ovf |= stop;
        u.single |= (unsigned long long)(fin & m2 & wv_b(u.j == u.s0)) << u.s0;
        u.seen |= fin;
        u.j = act ? ((hit & (stop ^ 1u)) ? u.j + 1 : u.L) : u.j;     // an arc that found no room ends the walk: unit_event gets a window or gives up
    }
    // A word with more than BW_WIN arcs (a run of one letter whose run lengths are all entries: one word in a million of the config-3
    // corpus): its lane does the whole word alone, the plain sequential program of :151-313 restricted to the word, with the arcs in global
    // memory -- 6 keys per position of the word in the batch's arc workspace, the per-position result behind them.  Slow (every access is
    // a round trip to memory) and rare; the alternative was to hand the document to a lane kernel that needs 13 us per byte.
    BF_WVD void unit_huge(Unit &u)
    {
        if (p.stats) wv::atomic_add(&p.stats[7], 1ull);
        const uint32_t ke = u.ke;
        const int64_t f = (int64_t)(u.rs - S.dt_rbase[ke]);
        uint32_t *buf = p.scratch + 6 * (S.dt_slot[ke] + f);
        const int cap = 6 * u.L - u.L;                                   // keys; the last L words hold the per-position result
        int n = 0; unsigned long long single = 0; bool give_up = false;
        for (int s0 = 0; s0 < u.L && !give_up; ++s0) {                   // :151-232 on the word (no unknown arc: a start without an arc gives up)
            uint32_t state = p.initial; int sum = 0; bool seen = false;
            for (int j = s0; j < u.L; ++j) {
                const uint32_t c = sym_at(u, j);
                if (c >= SG_CLS_DELIM_ABSENT) break;
                const uint64_t e = p.T[state + c];
                if ((e & SG_CLS_MASK) != c) break;
                state = (uint32_t)((e >> SG_NEXT_SHIFT) & SG_NEXT_MASK); sum += (int)(e >> SG_OW_SHIFT);
                if (e & SG_FINAL) {
                    seen = true;
                    if (j == s0) single |= 1ull << s0;
                    else {
                        const int32_t id = order_of((uint32_t)sum);
                        if (n >= cap || (uint32_t)id >= (1u << BPE_LOCAL_ID_BITS_W)) { give_up = true; break; }
                        buf[n++] = ((uint32_t)id << 12) | ((uint32_t)s0 << 6) | (uint32_t)j;
                    }
                }
            }
            if (!seen) give_up = true;
        }
        if (give_up) { unit_fallback(u, 4); return; }
        for (int a = 1; a < n; ++a) {                                    // :234-256 (insertion sort: the keys are distinct, any correct sort gives the same order)
            const uint32_t key = buf[a];
            int b = a - 1;
            while (b >= 0 && buf[b] > key) { buf[b + 1] = buf[b]; --b; }
            buf[b + 1] = key;
        }
        uint32_t *res = buf + n;                                         // per position: id << 6 | end of the last applied arc that starts there
        for (int q = 0; q < u.L; ++q) res[q] = 0xFFFFFFFFu;
        unsigned long long inter = 0;
        for (int a = 0; a < n; ++a) {                                    // :274-296
            const uint32_t key = buf[a];
            const int s = (int)((key >> 6) & 63u), e = (int)(key & 63u);
            if (!((inter >> s) & 1ull) && !((inter >> (e + 1)) & 1ull)) {
                inter |= ((1ull << (e + 1)) - 1ull) & ~((1ull << (s + 1)) - 1ull);
                res[s] = ((key >> 12) << 6) | (uint32_t)e;
            }
        }
        int32_t *home = p.ids_tmp + S.dt_slot[ke] + f;
        int cnt = 0; bool bad = false; int32_t first = 0;
        for (int pos = 0; pos < u.L;) {                                  // :299-313
            const uint32_t v = res[pos];
            int id, end = pos;
            if (v != 0xFFFFFFFFu) { id = id_of_order((int)(v >> 6)); end = (int)(v & 63u); }
            else {
                if (!((single >> pos) & 1ull)) { bad = true; break; }
                const uint32_t c = sym_at(u, pos);
                id = p.info[(int)(p.T[p.initial + c] >> SG_OW_SHIFT)].id;
            }
            if (cnt == 0) first = id + p.id_offset; else { if (cnt == 1) home[0] = first; home[cnt] = id + p.id_offset; }
            ++cnt;
            pos = end + 1;
        }
        if (bad || cnt == 0) { unit_fallback(u, 6); return; }
        if (cnt == 1) { if (HOME) home[0] = first; else S.q[(uint32_t)u.tok & QMASK].pos = (uint32_t)first; }
        unit_finish(u, cnt);
    }
    // the walk of the unit is over (ev): the whole word matched / go on collecting / all arcs collected
    BF_WVD void unit_event(Unit &u, bool ev)
    {
        const bool whole = ev && u.mode == 1 && u.last_final;
        if (wv::any(whole)) {
            if (whole) {                                                // the word is one entry
                const SegInfo r = p.info[u.sum];
                if (HOME) p.ids_tmp[S.dt_slot[u.ke] + (int64_t)(u.rs - S.dt_rbase[u.ke])] = r.id + p.id_offset;
                else S.q[(uint32_t)u.tok & QMASK].pos = (uint32_t)(r.id + p.id_offset);
                if (p.stats) wv::atomic_add(&p.stats[1], 1ull);
                unit_finish(u, 1);
            }
        }
        // a walk that ran out of window: with an overflow window of the wave it is repeated from its start position; a lane that holds one
        // already (BW_WIN arcs) gives up, a lane that gets none waits (mode 2 with j >= L: it asks again in the next round)
        const bool want = ev && !whole && u.mode == 2 && u.ovf && u.pw < 0 && (uint32_t)u.sum < (1u << 20);
        bool retry = false, wait = false;
        if (wv::any(want)) {
            const uint32_t freem = wv::uni(S.pool_free);
            const unsigned long long wm = wv::ballot(want);
            const int r = (int)wv::mbcnt(wm);
            uint32_t f = freem; int w = -1;
            for (int k = 0; k <= r && f; ++k) { w = __builtin_ctz(f); f &= f - 1u; if (k < r) w = -1; }
            if (want && w >= 0) { u.pw = w; retry = true; } else if (want) wait = true;
            const unsigned long long got = wv::ballot(want && w >= 0);
            uint32_t taken = 0, ff = freem;
            for (int k = __builtin_popcountll(got); k > 0 && ff; --k) { taken |= ff & (0u - ff); ff &= ff - 1u; }
            wv::sync();
            if (lane == 0) S.pool_free = freem & ~taken;
            wv::sync();
        }
        if (retry) { u.narc = u.narc0; u.ovf = 0; u.j = u.s0; u.state = p.initial; u.sum = 0; u.seen = 0; u.single &= ~(1ull << u.s0); }
        const bool huge = ev && !whole && !retry && !wait && u.mode == 2 && u.ovf && u.pw >= 0 && (uint32_t)u.sum < (1u << 20);   // more than BW_WIN arcs
        if (wv::any(huge)) { if (huge) unit_huge(u); }
        const bool bad = ev && !whole && !retry && !wait && !huge && u.mode == 2 && (!u.seen || u.ovf);      // a start without an arc (an unknown arc, :212-225)
        if (wv::any(bad)) { if (bad) unit_fallback(u, u.ovf ? 4 : 5); }
        const bool go = ev && !whole && !bad && !retry && !wait && !huge;
        const bool first = go && u.mode == 1;                           // not one entry: collect, from the word's first position
        const int ns0 = first ? 0 : u.s0 + 1;
        const bool more = go && ns0 < u.L;
        u.single = first ? 0ull : u.single; u.narc = first ? 0 : u.narc; u.ovf = first ? 0u : u.ovf;
        u.narc0 = go ? u.narc : u.narc0;
        u.s0 = go ? ns0 : u.s0; u.j = more ? ns0 : u.j; u.state = more ? p.initial : u.state; u.sum = more ? 0 : u.sum; u.seen = more ? 0u : u.seen;
        u.mode = go ? (more ? 2 : 3) : u.mode;
    }
    // the solve pass of the lanes in mode 3: ids of the collected arcs, sort by key (:238-255), apply against the interior mask
    // (:274-296), emit the pieces in position order (:299-313)
    // Words with at most BW_PRIV arcs, every waiting lane its own: ids of the arcs, Batcher's odd-even merge sort over the BW_PRIV slots
    // of the lane's window (the slots behind the last arc hold the largest key), apply, emit.  Fixed trip counts: a pass costs the same
    // whatever the lanes hold (by insertion sort it cost what its largest word cost -- measured: the pass was 3/4 of the kernel).
    BF_WVD void unit_solve_small(Unit &u)
    {
        const int na = u.narc;
        bool bigid = false;
#pragma unroll
        for (int a0 = 0; a0 < BW_PRIV; a0 += 8) {                        // eight I2Info gathers in flight, then their ids into the keys
            uint32_t key[8]; int32_t id[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) key[k] = a0 + k < na ? S.win[(a0 + k) * 64 + lane] : 0u;
#pragma unroll
            for (int k = 0; k < 8; ++k) id[k] = order_of(a0 + k < na ? (key[k] >> 12) : 0u);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (a0 + k < na && (uint32_t)id[k] >= (1u << BPE_LOCAL_ID_BITS_W)) bigid = true;
                S.win[(a0 + k) * 64 + lane] = a0 + k < na ? (((uint32_t)id[k] << 12) | (key[k] & 0xFFFu)) : 0xFFFFFFFFu;
            }
        }
        if (bigid) { unit_fallback(u, 4); return; }
        if (p.stats) wv::atomic_add(&p.stats[8], 1ull);
#pragma unroll
        for (int pp = 1; pp < BW_PRIV; pp <<= 1)
#pragma unroll
            for (int kq = pp; kq >= 1; kq >>= 1)
#pragma unroll
                for (int jj = kq % pp; jj <= BW_PRIV - 1 - kq; jj += 2 * kq)
#pragma unroll
                    for (int i2 = 0; i2 <= (kq - 1 < BW_PRIV - jj - kq - 1 ? kq - 1 : BW_PRIV - jj - kq - 1); ++i2)
                        if ((i2 + jj) / (2 * pp) == (i2 + jj + kq) / (2 * pp)) {
                            const uint32_t x = S.win[(i2 + jj) * 64 + lane], y = S.win[(i2 + jj + kq) * 64 + lane];
                            S.win[(i2 + jj) * 64 + lane] = x < y ? x : y; S.win[(i2 + jj + kq) * 64 + lane] = x < y ? y : x;
                        }
        unsigned long long inter = 0; uint32_t applied = 0;
#pragma unroll
        for (int a = 0; a < BW_PRIV; ++a) {                              // :274-296
            const uint32_t key = S.win[a * 64 + lane];
            const int s = (int)((key >> 6) & 63u), e = (int)(key & 63u);
            if (a < na && !((inter >> s) & 1ull) && !((inter >> (e + 1)) & 1ull)) {
                if (e > s) inter |= ((1ull << (e + 1)) - 1ull) & ~((1ull << (s + 1)) - 1ull);
                applied |= 1u << a;
            }
        }
        const uint32_t ke = u.ke;
        int32_t *home = p.ids_tmp + S.dt_slot[ke] + (int64_t)(u.rs - S.dt_rbase[ke]);
        int cnt = 0; bool bad = false; int32_t first = 0;
        for (int pos = 0; pos < u.L;) {                                  // :299-313: token by token (a token's positions behind its first are interior)
            int id = -1, end = pos;
#pragma unroll
            for (int a = 0; a < BW_PRIV; ++a) {                          // the LAST applied arc that starts here set tos / ids (:291-292)
                const uint32_t key = S.win[a * 64 + lane];
                if (((applied >> a) & 1u) && (int)((key >> 6) & 63u) == pos) { id = (int)(key >> 12); end = (int)(key & 63u); }
            }
            if (id < 0) {                                                // no applied arc of more than one element starts here: the one-element arc
                if (!((u.single >> pos) & 1ull)) { bad = true; break; }  // none: pTos[start] == 0 < start, the reference does not come back from here
                const uint32_t c = sym_at(u, pos);
                const uint64_t e1 = p.T[p.initial + c];
                id = p.info[(int)(e1 >> SG_OW_SHIFT)].id;
            } else id = id_of_order(id);
            if (cnt == 0) first = id + p.id_offset; else { if (cnt == 1) home[0] = first; home[cnt] = id + p.id_offset; }
            ++cnt;
            pos = end + 1;
        }
        if (bad || cnt == 0) { unit_fallback(u, 6); return; }
        if (cnt == 1) { if (HOME) home[0] = first; else S.q[(uint32_t)u.tok & QMASK].pos = (uint32_t)first; }
        unit_finish(u, cnt);
    }
    // Words with more arcs, one after the other by the whole wave: lane j takes arc j (BW_WIN <= 64), its rank among the keys from an
    // all-pairs comparison through broadcasts, the sorted keys go back into the word's windows, the apply runs on wave-uniform values,
    // lane q stands for position q of the word when the pieces are emitted (BW_WORD_MAX < 64).
    BF_WVD void solve_big(Unit &u, unsigned long long bigm)
    {
        while (bigm) {
            const int o = __builtin_ctzll(bigm); bigm &= bigm - 1ull;
            const int na = wv::bcast(u.narc, o), pw = wv::bcast(u.pw, o), L = wv::bcast(u.L, o), tok = wv::bcast(u.tok, o);
            const uint32_t rs = wv::bcast(u.rs, o), ke = wv::bcast(u.ke, o);
            const int64_t gg = HOME ? wv::bcast(u.g, o) : 0;
            const unsigned long long single = wv::bcast(u.single, o);
            uint32_t *slot = lane < BW_PRIV ? &S.win[lane * 64 + o] : &S.pool[(lane - BW_PRIV) * BW_POOL_N + pw];
            uint32_t key = 0xFFFFFFFFu;
            if (lane < na) key = *slot;
            int32_t id = 0;
            if (lane < na) id = order_of(key >> 12);
            const bool bigid = lane < na && (uint32_t)id >= (1u << BPE_LOCAL_ID_BITS_W);
            if (wv::any(bigid)) { if (lane == o) unit_fallback(u, 4); wv::sync(); continue; }
            if (lane < na) key = ((uint32_t)id << 12) | (key & 0xFFFu);
            if (p.stats && lane == o) wv::atomic_add(&p.stats[8 + (na <= 24 ? 1 : na <= 32 ? 2 : 3)], 1ull);
            int rank = 0;
            for (int t = 0; t < na; ++t) { const uint32_t kt = wv::bcast(key, t); rank += kt < key ? 1 : 0; }       // the keys of a word are distinct
            wv::sync();                                                  // every lane has read its arc: the windows take the sorted order
            if (lane < na) { uint32_t *dst = rank < BW_PRIV ? &S.win[rank * 64 + o] : &S.pool[(rank - BW_PRIV) * BW_POOL_N + pw]; *dst = key; }
            wv::sync();
            unsigned long long inter = 0; int my_id = -1;
            for (int r = 0; r < na; ++r) {                               // :274-296, in sorted order, on wave-uniform values
                const uint32_t kr = wv::uni(r < BW_PRIV ? S.win[r * 64 + o] : S.pool[(r - BW_PRIV) * BW_POOL_N + pw]);
                const int s = (int)((kr >> 6) & 63u), e = (int)(kr & 63u);
                if (!((inter >> s) & 1ull) && !((inter >> (e + 1)) & 1ull)) {
                    if (e > s) inter |= ((1ull << (e + 1)) - 1ull) & ~((1ull << (s + 1)) - 1ull);
                    if (lane == s) my_id = id_of_order((int)(kr >> 12));   // the last applied arc of a start stays (:291-292)
                }
            }
            const bool is_tok = lane < L && !((inter >> lane) & 1ull);   // :299-313: the non-interior positions, lane = position
            bool bad = false; int idv = my_id;
            if (is_tok && my_id < 0) {                                   // no applied arc of more than one element starts here: the one-element arc
                if (!((single >> lane) & 1ull)) bad = true;
                else { const uint32_t c = HOME ? (uint32_t)p.stream[gg + lane] : (uint32_t)S.ring[(rs + (uint32_t)lane) & RMASK]; const uint64_t e1 = p.T[p.initial + c]; idv = p.info[(int)(e1 >> SG_OW_SHIFT)].id; }
            }
            const unsigned long long mt = wv::ballot(is_tok);
            const int cnt = __builtin_popcountll(mt), k = __builtin_popcountll(mt & ((1ull << lane) - 1ull));
            if (wv::any(bad) || cnt == 0) { if (lane == o) unit_fallback(u, 6); wv::sync(); continue; }
            int32_t *home = p.ids_tmp + S.dt_slot[ke] + (int64_t)(rs - S.dt_rbase[ke]);
            if (is_tok) { if (cnt == 1 && !HOME) S.q[(uint32_t)tok & QMASK].pos = (uint32_t)(idv + p.id_offset); else home[k] = idv + p.id_offset; }
            wv::sync();
            if (lane == o) unit_finish(u, cnt);
            wv::sync();
        }
    }
    static constexpr int BPE_LOCAL_ID_BITS_W = 20;
    static constexpr int SOLVE_MIN = 16;                                 // lanes that wait for the solve pass before it runs (or nothing else is left to do)
    BF_WVD bool units_phase(Unit &u, bool drain)
    {
        bool ran = false;
        const uint32_t tail = wv::uni(q_tail);
        uint32_t issue = wv::uni(q_issue);
        for (;;) {
            const bool ev = u.tok >= 0 && u.mode != 3 && u.j >= u.L;
            if (wv::any(ev)) unit_event(u, ev);
            // words for idle lanes -- several times over: four words in five have their id from the table and leave their lane idle
            unsigned long long idle = wv::ballot(u.tok < 0);
            for (int rep = 0; rep < 8; ++rep) {
                const uint32_t avail = tail - issue;
                if (avail == 0 || idle == 0) break;
                const uint32_t r = wv::mbcnt(idle);
                const bool take = u.tok < 0 && r < avail;
                if (take) unit_begin(u, issue + r);
                const uint32_t k = (uint32_t)__builtin_popcountll(idle);
                issue += k < avail ? k : avail;
                idle = wv::ballot(u.tok < 0);
                if (__builtin_popcountll(idle) < 16) break;
            }
            const int nb = 64 - __builtin_popcountll(idle);
            const unsigned long long ready = wv::ballot(u.tok >= 0 && u.mode == 3);
            const int nready = __builtin_popcountll(ready);
            const bool walking = wv::any(u.tok >= 0 && u.mode != 3 && u.j < u.L);
            if (nready != 0 && (nready >= SOLVE_MIN || !walking)) {      // enough of them, or nobody walks (the others wait for an overflow window)
                const bool rdy = u.tok >= 0 && u.mode == 3;
                const unsigned long long bigm = wv::ballot(rdy && u.narc > BW_PRIV);
                if (rdy && u.narc <= BW_PRIV) unit_solve_small(u);
                wv::sync();
                if (bigm) solve_big(u, bigm);
                ran = true;
                continue;
            }
            if (nb == 0) break;
            if (!drain && issue == tail && nb < UMIN) break;
            if (HOME) {
                // the symbols of this round's transitions: one load (a lane that walks at all walks consecutive elements within a round)
                static_assert(!HOME || STEPS <= 4, "four symbols per load");
                uint64_t s4 = 0;
                if (u.tok >= 0 && u.mode != 3 && u.j < u.L) __builtin_memcpy(&s4, p.stream + u.g + u.j, 8);      // (the stream buffer is padded)
                for (int st = 0; st < STEPS; ++st) unit_step(u, (uint32_t)(s4 >> (16 * st)) & 0xFFFFu);
            } else for (int st = 0; st < STEPS; ++st) unit_step(u);
            ran = true;
        }
        q_issue = issue;
        u_need = 0xFFFFFFFFu;
        {
            uint32_t need = 0xFFFFFFFFu;
            if (!HOME && u.tok >= 0) need = u.rs - rlo;
            if (wv::any(need != 0xFFFFFFFFu)) u_need = wv::min_all(need);
        }
        wv::sync();
        return ran;
    }

    // ------------------------------------------------------------------------------------------------------------------
    // retire / settle: bf_wave_body.h
    // ------------------------------------------------------------------------------------------------------------------
    BF_WVD int retire(bool all)
    {
        if (HOME) {                                                     // nothing moves: the finished words at the head of the queue give their places back
            const uint32_t nav = q_issue - q_retire < 64u ? q_issue - q_retire : 64u;
            if (nav == 0) return 0;
            const unsigned long long fin = wv::ballot((uint32_t)lane < nav && S.qc[(q_retire + (uint32_t)lane) & QMASK] != 0);
            const int nret = fin == ~0ull ? 64 : __builtin_ctzll(~fin);
            q_retire += (uint32_t)nret;
            return nret;
        }
        const uint32_t navail = q_issue - q_retire < 64u ? q_issue - q_retire : 64u;
        if (navail == 0 || (!all && navail < 64u)) return 0;
        const uint32_t t = q_retire + (uint32_t)lane, sl = t & QMASK;
        const int cnt0 = (uint32_t)lane < navail ? (int)S.qc[sl] - 1 : -1;
        const unsigned long long fin = wv::ballot(cnt0 >= 0);
        const int nret = fin == ~0ull ? 64 : __builtin_ctzll(~fin);
        if (nret == 0 || (!all && nret < 64)) return 0;
        const bool act = lane < nret;
        const int cnt = act ? cnt0 : 0;
        const WvTok te = S.q[sl];
        const int k = act ? (int)((te.w >> 16) & 0xFFu) : -1;
        const uint32_t ke = (uint32_t)k & DMASK;
        int64_t sl64 = 0; int cap = 0, dcnt = 0; uint32_t f = 0; uint32_t w0 = 0;
        if (act) { sl64 = S.dt_slot[ke]; cap = S.dt_cap[ke]; dcnt = S.dt_cnt[ke]; w0 = te.pos; f = w0 - S.dt_rbase[ke]; }
        const int32_t *src = p.ids_tmp + sl64 + (int64_t)f;
        int32_t v0 = (int32_t)w0, v1 = 0, v2 = 0, v3 = 0;
        if (cnt > 1) { v0 = src[0]; v1 = src[1]; }
        if (cnt > 2) v2 = src[2];
        if (cnt > 3) v3 = src[3];
        const int inc = wv::incl_scan(cnt), exc = inc - cnt;
        const int kp = wv::shfl_up(k, 1), kn = wv::shfl_down(k, 1);
        const unsigned long long hm = wv::ballot(lane == 0 || k != kp);
        const int head = 63 - __builtin_clzll(hm & ((2ull << lane) - 1ull));
        const int segbase = wv::shfl(exc, head);
        const int pos = dcnt + (exc - segbase);
        wv::arrived(v0, v1, v2, v3);                                  // one wait for the loads above: the four stores below leave back to back (see k_wp_wave's retire pass)
        wv::sync();
        if (act && (lane == 63 || k != kn)) S.dt_cnt[ke] = pos + cnt;
        const int room = cap - pos;
        unsigned long long big = wv::ballot(cnt > 4);
        while (big) {
            const int l = __builtin_ctzll(big); big &= big - 1ull;
            const int bc = wv::bcast(cnt, l), br = wv::bcast(room, l);
            const int64_t bs = wv::bcast(sl64 + (int64_t)f, l), bd = wv::bcast(sl64 + (int64_t)pos, l);
            for (int o = 4; o < bc; o += 64) {
                const int i = o + lane;
                int32_t v = 0;
                if (i < bc) v = p.ids_tmp[bs + i];
                wv::sync();
                if (i < bc && i < br) p.ids_tmp[bd + i] = v;
                wv::sync();
            }
        }
        int32_t *dst = p.ids_tmp + sl64 + pos;
        if (cnt > 0 && room > 0) dst[0] = v0;
        if (cnt > 1 && room > 1) dst[1] = v1;
        if (cnt > 2 && room > 2) dst[2] = v2;
        if (cnt > 3 && room > 3) dst[3] = v3;
        q_retire += (uint32_t)nret;
        wv::sync();
        return nret;
    }
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
                const bool fb = (f & BW_DT_FALLBACK) != 0;
                p.counts[S.dt_doc[e]] = fb ? 0 : (c < cap ? c : cap);                  // (HOME form: dt_cnt = the table's hits + what the units reported)
                p.flags[S.dt_doc[e]] = fb ? 1 : 0;
            }
            dt_head = limit; moved = true;
        }
        const uint32_t old_lo = rlo;
        // (HOME form: a word is looked up when it is queued and a unit reads the class stream: the ring holds only what the fill has not resolved yet)
        uint32_t keep = (!HOME && q_issue != q_tail) ? S.q[q_issue & QMASK].pos - old_lo : (have_doc ? rbase + (uint32_t)(open_start >= 0 ? open_start : done) : rhi) - old_lo;
        if (u_need < keep) keep = u_need;
        rlo = old_lo + keep; u_need -= u_need == 0xFFFFFFFFu ? 0u : keep;
        return moved || keep != 0;
    }

    BF_WVD void run(int grab, int wave_id, int n_waves)
    {
        grab = grab < 1 ? 1 : (grab > WV_GRAB_MAX ? WV_GRAB_MAX : grab);
        st_wave = wave_id; st_waves = n_waves; st_round = 0;
        if (lane == 0) S.pool_free = (1u << BW_POOL_N) - 1u;
        wv::sync();
        Unit u; u.narc0 = 0; u.pw = -1; u.tok = -1; u.mode = 0; u.rs = 0; u.L = 0; u.ke = 0; u.s0 = u.j = 0; u.state = 0; u.sum = 0; u.seen = u.last_final = u.ovf = 0; u.narc = 0; u.single = 0;
        for (;;) {
            bool moved = settle();
            bool filled = false;
            while (fill_step(grab)) filled = true;
            const bool drain = !filled;
            if (units_phase(u, drain)) moved = true;
            for (;;) { const int r = retire(drain); if (r == 0) break; moved = true; if (r < 64) break; }
            if (settle()) moved = true;
            if (exiting && !have_doc && q_tail == q_retire && dt_head == dt_tail) break;
            if (!filled && !moved) { if (lane == 0) wv::atomic_or(p.status, BF_STATUS_INTERNAL); break; }
        }
    }
};

} // namespace bfa
