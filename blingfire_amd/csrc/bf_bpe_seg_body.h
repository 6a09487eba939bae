// bf_bpe_seg_body.h -- BPE (all three flavours) for ONE document by ONE wave, without a bound on the number of arcs.
//
// Reproduces FATokenSegmentationTools_1best_bpe_t<int>::Process (cl/inc/FATokenSegmentationTools_1best_bpe_t.h:126-316), its twin with
// merge ranks (cl/inc/FATokenSegmentationTools_1best_bpe_with_merges_t.h:129-323) and the id loop of TextToIdsWithOffsets_sp
// (tokdll:1509-1532) on the class stream of the prologue kernel -- for the documents the lane-per-document kernels cannot finish
// (bf_seg.h: they reserve 6 * L + 32 arcs per document; a run of one character whose run lengths are all vocabulary entries needs up to
// trie-depth arcs per position).  The reference collects into an unbounded std::vector (..._bpe_t.h:143-144,197); here the arcs live in
// a block claimed from a pool and every step is wave-cooperative, so that 10^4 .. 10^6 identical characters are a matter of
// milliseconds to seconds instead of a refusal:
//
//   collect  (:151-232) 64 start positions per window, one per lane.  What a start contributes is a function of its own walk -- its
//            finals, the whole-token replacement and the fast-forward of bpe-opt (:189-206,228-230: only a start on U+2581 can skip,
//            and it skips to the element before the next U+2581) -- so the walks of a window run side by side; what couples starts
//            (which ones the fast-forward skips, unknown starts merging into the arc pushed last, :212-225) is resolved per window
//            from ballots, with a wave-uniform carry between windows.  One-element arcs are not stored: they mark no interior and
//            any applied longer arc of the same start overrides them (a bit per position remembers that the reference has one).
//            Arcs are 64-bit keys [priority | start | flags | length - 1]; priority = 2 * id + 1 (with merges: 2 * the entry's place
//            in the order "rank descending, id ascending" + 1, a table made at load), the unknown arc's priority is computed by the
//            host for the call's UnkId so that it sorts where the reference's comparator puts it.  The walks run twice: once to
//            count (the block is then claimed at its exact size), once to write.
//   sort     (:234-256) stable LSD radix sort over the used start and priority bits, 8 bits per pass, ranks inside a chunk of 64 keys
//            from eight ballots (the order is total: two arcs never share priority and start);
//   apply    (:274-296) in sorted order, 64 arcs at a time: the interior marks only ever grow, so an arc that fails against the marks
//            at the beginning of its chunk fails for good; the survivors are taken in order (the lane-uniform loop only compares
//            registers: a survivor kills the later ones it covers), then all of them mark their interiors and their start's result at once;
//   emit     (:299-313) the non-interior positions in order: the id of the longest applied arc of the position (the last applied one:
//            a later arc of the same start can only be longer), else the one-element arc, else -- only at position 0 -- UnkId.
//            A position > 0 with nothing applied makes the reference walk backwards for ever: the document gets no ids and the
//            batch a status bit (BF_STATUS_DOC_FAILED).
// A document whose block does not fit the pool gets count 0 and BF_STATUS_POOL, and the claims are added up for the host, which grows
// the pool and runs the batch again (bf_capi.cpp): a failure is per document and never silent.
// Include AFTER a definition of namespace wv (bf_kernels.hip on the device, tests/hosttest/wave_emu.h in the test simulator).
#pragma once
#include "bf_wave.h"
#include "bf_seg.h"

namespace bfa {

constexpr uint64_t BS_LEN_MASK = 0xFFull, BS_F_UNK = 1ull << 8, BS_F_EXT = 1ull << 9;
constexpr int BS_START_SHIFT = 10, BS_PRIO_SHIFT = 40;
constexpr uint64_t BS_START_MASK = (1ull << 30) - 1ull;
constexpr int BS_PRIO_BITS_MAX = 22;
// res word of a position = the longest applied arc that starts there: [ext : 1 | length - 1 : 8 | unknown : 1 | priority : 22]; an extended
// arc (BS_F_EXT) is its start's longest by construction, so the largest word is the arc with the largest end; 0 = none
constexpr uint32_t BS_R_EXT = 1u << 31, BS_R_UNK = 1u << 22, BS_R_PRIO_MASK = (1u << 22) - 1u;
constexpr int BS_R_LEN_SHIFT = 23;
constexpr int BS_MAX_DEPTH = 256;          // arcs carry length - 1 in 8 bits (checked at load: Model::bpe_seg_ok)

struct BpeSegParams {
    const uint64_t *T; const SegInfo *info; uint32_t initial, cls_delim; int id_offset, kind;     // bf_seg.h SegTables
    const uint32_t *prio;            // with merges: priority of the entry with MPH index k (2 * place + 1); nullptr: 2 * id + 1
    const int32_t *place_id;         // with merges: id of the entry at a place of the order; nullptr: the place is the id
    uint32_t unk_prio; int prio_bits; // priority of the unknown arc for this call's UnkId; bits the priorities use
    const uint16_t *stream; const int32_t *lens; const int64_t *doc_off; int slot_mul;
    const int32_t *list; const unsigned int *list_n; const int32_t *narcs; int narcs_want;         // documents: list[i], i < *list_n, of which those with narcs[d] == narcs_want (narcs: optional)
    int64_t ndocs;                   // list == nullptr: every document 0 .. ndocs - 1
    int32_t *ids_tmp; int32_t *span_tmp; int32_t *counts; int max_ids, unk;
    unsigned long long *next_doc; int *status;
    uint8_t *pool; unsigned long long pool_bytes; unsigned long long *pool_used, *pool_need;      // *pool_need: bytes of the claims that did not fit
    unsigned long long *stats;       // optional: [0] documents, [1] arcs, [2] sort passes, [3] apply chunks, [4] survivors, [5] chunks without a survivor
};

struct BsLds { uint32_t hist[256]; };

template <class LDS>
struct BpeSeg {
    const BpeSegParams &p; LDS &S; int lane;
    // the document
    const uint16_t *src; int L; bool fast;
    uint32_t *res, *ext, *bmi, *bms; uint64_t *keys, *keys2; int64_t cap;

    BF_WVD BpeSeg(const BpeSegParams &p_, LDS &S_) : p(p_), S(S_)
    {
        lane = wv::lane(); src = nullptr; L = 0; fast = false; res = ext = bmi = bms = nullptr; keys = keys2 = nullptr; cap = 0;
    }

    BF_WVD uint32_t prio_of(int sum) const { return p.prio ? p.prio[sum] : 2u * (uint32_t)p.info[sum].id + 1u; }
    BF_WVD static uint64_t make_key(uint32_t prio, int s, int len1, uint64_t flags)
    {
        return ((uint64_t)prio << BS_PRIO_SHIFT) | ((uint64_t)(uint32_t)s << BS_START_SHIFT) | flags | (uint64_t)(uint32_t)len1;
    }
    // the marks are set with atomics (they execute in the L2): they are read past the CU's vector cache
    BF_WVD static bool bit(const uint32_t *bm, int q) { return (wv::load_l2(&bm[q >> 5]) >> (q & 31)) & 1u; }

    // ------------------------------------------------------------------------------------------------------------------
    // collect
    // ------------------------------------------------------------------------------------------------------------------
    struct Walk { int n, m, ff, last_sum, last_i; bool unknown, single; };
    // The walk of start s (:163-208), all lanes side by side.  n: arcs the start has pushed (the whole-token arc replaces what the start
    // pushed before it), m: those of more than one element, `single`: the one-element arc is among them.  With `out`: the arcs of more
    // than one element go to out[0 .. m) in the order the reference's vector holds them after the walk.
    BF_WVD Walk walk(bool valid, int s, uint64_t *out, int64_t out_cap)
    {
        Walk w; w.n = 0; w.m = 0; w.ff = s; w.last_sum = 0; w.last_i = -1; w.unknown = true; w.single = false;
        uint32_t state = p.initial; int sum = 0; int i = s; bool act = valid;
        const bool ts = valid && (uint32_t)src[valid ? s : 0] == p.cls_delim;
        while (wv::any(act)) {
            const uint32_t c = act ? (uint32_t)src[i] : 0u;
            const bool ok = act && c < SG_CLS_DELIM_ABSENT;
            const uint64_t e = p.T[ok ? state + c : 0u];
            const bool hit = ok && (e & SG_CLS_MASK) == c;
            const bool fin = hit && (e & SG_FINAL) != 0;
            if (hit) { state = (uint32_t)((e >> SG_NEXT_SHIFT) & SG_NEXT_MASK); sum += (int)(e >> SG_OW_SHIFT); }
            if (fin) {
                const bool boundary = i == L - 1 || (uint32_t)src[i + 1 < L ? i + 1 : i] == p.cls_delim;
                const bool apply = fast && ts && boundary && w.n > 0;                     // :189-191
                if (apply) { w.n = 1; w.m = 0; w.ff = i; w.single = false; }             // :203-206: what the start pushed so far is gone
                else ++w.n;
                if (i > s) {
                    if (out && w.m < out_cap) out[w.m] = make_key(prio_of(sum), s, i - s, 0);
                    ++w.m;
                } else w.single = true;
                w.last_sum = sum; w.last_i = i; w.unknown = false;
            }
            act = hit && i + 1 < L;
            i = act ? i + 1 : i;
        }
        return w;
    }

    // write == false: count only.  Returns the number of stored arcs.
    BF_WVD int64_t collect(bool write)
    {
        int64_t narcs = 0;
        int skip = -1;                               // starts <= skip are skipped by a fast-forward (:228-230)
        // the arc pushed last (:216): is there one, is its id UnkId (an unknown start then moves its end instead of pushing an arc of
        // its own), where is it among the stored arcs, which start owns it
        bool c_has = false, c_lastunk = false; int64_t c_idx = -1; int c_start = -1;
        for (int w0 = 0; w0 < L; w0 += 64) {
            const int s = w0 + lane; const bool valid = s < L;
            Walk w = walk(valid, s, nullptr, 0);
            // ---- which starts the loop of :151 visits: a visited start on U+2581 whose whole-token arc was found skips to its end
            {
                unsigned long long covered = 0; int sk = skip;
                unsigned long long cand = wv::ballot(valid && s > sk && w.ff > s);
                while (cand) {
                    const int j = __builtin_ctzll(cand);
                    const int fj = wv::bcast(w.ff, j);
                    const int hi = j + (fj - (w0 + j)) > 63 ? 63 : j + (fj - (w0 + j));       // lanes j + 1 .. hi are skipped
                    if (hi > j) covered |= (hi >= 63 ? ~0ull : ((2ull << hi) - 1ull)) & ~((2ull << j) - 1ull);
                    sk = fj > sk ? fj : sk;
                    cand &= ~covered; cand &= ~(1ull << j);
                }
                const bool vis0 = valid && s > skip && !((covered >> lane) & 1ull);
                skip = sk;
                w.unknown = w.unknown && vis0;
                if (!vis0) { w.n = 0; w.m = 0; w.single = false; w.last_i = -1; }
            }
            const bool vis = valid && (w.n > 0 || w.unknown);
            const unsigned long long V = wv::ballot(vis), U = wv::ballot(vis && w.unknown);
            // ---- the id of the arc a start pushes last: when it is UnkId the next start may move its end, so it is stored even as a one-element arc
            int lastid = 0;
            if (vis && !w.unknown) lastid = p.info[w.last_sum].id;
            const bool lastunk = vis && (w.unknown || lastid == p.unk);
            const bool keep_single = vis && !w.unknown && lastunk && w.last_i == s;
            // ---- unknown starts (:212-225).  After an unknown start the arc pushed last has id UnkId whatever it did, so only the first
            //      start of a run of unknown starts decides: a new arc, or the end of the arc pushed last moves; the rest of the run moves
            //      the same end.  newunk: this lane pushes the new arc
            const unsigned long long below = V & ((1ull << lane) - 1ull);
            const int pl = below ? 63 - __builtin_clzll(below) : 0;
            const bool p_lastunk = wv::shfl((int)lastunk, pl) != 0;
            const bool prev_has = below ? true : c_has, prev_lastunk = below ? p_lastunk : c_lastunk;
            const bool newunk = vis && w.unknown && !(prev_has && prev_lastunk);
            int M = 0;                                                    // arcs this lane stores
            if (vis && !w.unknown) M = w.m + (keep_single ? 1 : 0);
            if (newunk) M = 1;
            const int inc = wv::incl_scan(M);
            const int64_t base = narcs + (int64_t)(inc - M);
            const int total = wv::bcast(inc, 63);
            // the run of unknown starts a lane belongs to: `anchor` = the visited start before the run (-1: the run reaches the first visited
            // start of the window, the arc pushed last is the carry's), `head` = the run's first start; the arc the run extends is owned by
            // head (new arc), anchor or the carry; the run's last start of this window writes the end
            const unsigned long long NU = V & ~U;
            const unsigned long long nu_below = NU & ((1ull << lane) - 1ull);
            const int anchor = nu_below ? 63 - __builtin_clzll(nu_below) : -1;
            const unsigned long long u_after = anchor >= 0 ? (U & ~((2ull << anchor) - 1ull)) : U;
            const int head = u_after ? __builtin_ctzll(u_after) : 0;
            const bool head_new = wv::shfl((int)newunk, head) != 0;
            const unsigned long long above = V & ~((2ull << lane) - 1ull);
            const int nl = lane < 63 && above ? __builtin_ctzll(above) : 64;
            const bool run_end = vis && w.unknown && (nl == 64 || !((U >> nl) & 1ull));
            if (write && run_end) {
                if (head_new) ext[w0 + head] = (uint32_t)s;
                else if (anchor >= 0) ext[w0 + anchor] = (uint32_t)s;
                else if (c_start >= 0) ext[c_start] = (uint32_t)s;
            }
            // a start whose last arc is extended: the next visited start is unknown and pushes no arc of its own
            const bool nxt_new = wv::shfl((int)newunk, nl & 63) != 0;
            const bool extended = vis && !w.unknown && nl < 64 && ((U >> nl) & 1ull) && !nxt_new;
            // the carry's arc is extended when the window's first visited start is unknown and pushes no arc of its own
            const int fl = V ? __builtin_ctzll(V) : 0;
            const bool carry_ext = V != 0 && ((U >> fl) & 1ull) && wv::bcast((int)newunk, fl) == 0;
            if (write && carry_ext && c_idx >= 0 && c_idx < cap && lane == 0) keys[c_idx] |= BS_F_EXT;
            // ---- the arcs: the walk again, into place
            {
                const bool wr = write && vis && !w.unknown && M > 0;
                uint64_t *out = (wr && base < cap) ? keys + base : nullptr;
                const int64_t oc = wr ? (cap - base < (int64_t)w.m ? cap - base : (int64_t)w.m) : 0;
                if (write) (void)walk(wr && w.m > 0, s, out, oc);
                if (wr) {
                    const int64_t li = base + M - 1;                  // the arc the start pushed last
                    if (keep_single) { if (li < cap) keys[li] = make_key(prio_of(w.last_sum), s, 0, extended ? BS_F_EXT : 0); }
                    else if (extended && li < cap) keys[li] |= BS_F_EXT;
                    if (w.single && !keep_single) wv::atomic_or_u32(&bms[s >> 5], 1u << (s & 31));
                }
                if (write && vis && !w.unknown && M == 0 && w.single) wv::atomic_or_u32(&bms[s >> 5], 1u << (s & 31));
                if (write && newunk && base < cap) keys[base] = make_key(p.unk_prio, s, 0, BS_F_UNK | BS_F_EXT);
            }
            // ---- carry: the arc pushed last
            if (V) {
                const int hl = 63 - __builtin_clzll(V);
                const bool h_unknown = wv::bcast((int)w.unknown, hl) != 0, h_lastunk = wv::bcast((int)lastunk, hl) != 0;
                const int64_t h_base = wv::bcast(base, hl); const int h_M = wv::bcast(M, hl);
                const int h_anchor = wv::bcast(anchor, hl), h_head = wv::bcast(head, hl); const bool h_new = wv::bcast((int)head_new, hl) != 0;
                const int64_t hd_base = wv::bcast(base, h_head);
                const int64_t an_base = wv::bcast(base, h_anchor >= 0 ? h_anchor : 0); const int an_M = wv::bcast(M, h_anchor >= 0 ? h_anchor : 0);
                if (!h_unknown) { c_idx = h_M > 0 ? h_base + h_M - 1 : -1; c_start = w0 + hl; }
                else if (h_new) { c_idx = hd_base; c_start = w0 + h_head; }       // the window ends inside a run of unknown starts: the arc it extends stays the one pushed last
                else if (h_anchor >= 0) { c_idx = an_base + an_M - 1; c_start = w0 + h_anchor; }
                c_has = true; c_lastunk = h_lastunk;
            }
            narcs += total;
            wv::sync();
        }
        return narcs;
    }

    // ------------------------------------------------------------------------------------------------------------------
    // sort: keys[0 .. A) by (priority, start), ascending.  Returns the buffer that holds the result.
    // ------------------------------------------------------------------------------------------------------------------
    BF_WVD uint64_t *sort(int64_t A)
    {
        uint64_t *a = keys, *b = keys2;
        int sb = 1; while (sb < 30 && ((int64_t)1 << sb) < (int64_t)L) ++sb;
        int shifts[8]; int np = 0;
        for (int o = 0; o < sb; o += 8) shifts[np++] = BS_START_SHIFT + o;
        for (int o = 0; o < p.prio_bits; o += 8) shifts[np++] = BS_PRIO_SHIFT + o;
        for (int ps = 0; ps < np; ++ps) {
            const int sh = shifts[ps];
            if (p.stats && lane == 0) wv::atomic_add(&p.stats[2], 1ull);
            for (int k = lane; k < 256; k += 64) S.hist[k] = 0;
            wv::sync();
            for (int64_t i = lane; i < A; i += 64) wv::lds_add(&S.hist[(uint32_t)(a[i] >> sh) & 255u], 1u);
            wv::sync();
            {   // exclusive scan of the 256 counters, four per lane
                const uint32_t h0 = S.hist[lane * 4], h1 = S.hist[lane * 4 + 1], h2 = S.hist[lane * 4 + 2], h3 = S.hist[lane * 4 + 3];
                const int t = (int)(h0 + h1 + h2 + h3);
                const int inc = wv::incl_scan(t);
                const uint32_t e0 = (uint32_t)(inc - t);
                wv::sync();
                S.hist[lane * 4] = e0; S.hist[lane * 4 + 1] = e0 + h0; S.hist[lane * 4 + 2] = e0 + h0 + h1; S.hist[lane * 4 + 3] = e0 + h0 + h1 + h2;
            }
            wv::sync();
            for (int64_t c0 = 0; c0 < A; c0 += 64) {
                const int64_t i = c0 + lane; const bool v = i < A;
                const uint64_t key = v ? a[i] : 0ull;
                const uint32_t d = (uint32_t)(key >> sh) & 255u;
                unsigned long long same = wv::ballot(v);
#pragma unroll
                for (int bq = 0; bq < 8; ++bq) { const unsigned long long B = wv::ballot(v && ((d >> bq) & 1u)); same &= ((d >> bq) & 1u) ? B : ~B; }
                const uint32_t rank = (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull));
                const uint32_t at = v ? S.hist[d] : 0u;
                wv::sync();
                if (v) b[at + rank] = key;
                if (v && (lane == 63 || !(same >> (lane + 1)))) S.hist[d] = at + rank + 1u;          // the digit's highest lane moves its cursor
                wv::sync();
            }
            uint64_t *t = a; a = b; b = t;
        }
        return a;
    }

    // ------------------------------------------------------------------------------------------------------------------
    // apply (:274-296)
    // ------------------------------------------------------------------------------------------------------------------
    BF_WVD void apply(const uint64_t *a, int64_t A)
    {
        for (int64_t c0 = 0; c0 < A; c0 += 64) {
            const int64_t i = c0 + lane; const bool v = i < A;
            const uint64_t key = v ? a[i] : 0ull;
            const int s = (int)((key >> BS_START_SHIFT) & BS_START_MASK);
            int e = s + (int)(key & BS_LEN_MASK);
            if (v && (key & BS_F_EXT)) e = (int)ext[s];
            bool alive = v && !bit(bmi, s) && (e + 1 >= L || !bit(bmi, e + 1));
            const unsigned long long am = wv::ballot(alive);
            if (p.stats && lane == 0) { wv::atomic_add(&p.stats[3], 1ull); wv::atomic_add(&p.stats[4], (unsigned long long)__builtin_popcountll(am)); if (!am) wv::atomic_add(&p.stats[5], 1ull); }
            if (!am) continue;
            unsigned long long todo = am;
            while (todo) {
                const int j = __builtin_ctzll(todo); todo &= todo - 1ull;
                const int sj = wv::bcast(s, j), ej = wv::bcast(e, j);
                if (ej == sj) continue;                                   // marks no interior
                const bool kill = alive && lane > j && ((sj < s && s <= ej) || (sj < e + 1 && e + 1 <= ej));
                const unsigned long long km = wv::ballot(kill);
                alive = alive && !kill; todo &= ~km;
            }
            if (alive) {
                for (int w = (s + 1) >> 5; w <= (e >> 5) && s < e; ++w) {   // interior marks s + 1 .. e
                    const int lo = w == ((s + 1) >> 5) ? ((s + 1) & 31) : 0, hi = w == (e >> 5) ? (e & 31) : 31;
                    wv::atomic_or_u32(&bmi[w], (hi == 31 ? ~0u : ((1u << (hi + 1)) - 1u)) & ~((1u << lo) - 1u));
                }
                uint32_t r = ((uint32_t)(key & BS_LEN_MASK) << BS_R_LEN_SHIFT) | ((uint32_t)(key >> BS_PRIO_SHIFT) & BS_R_PRIO_MASK);
                if (key & BS_F_EXT) r |= BS_R_EXT;
                if (key & BS_F_UNK) r |= BS_R_UNK;
                wv::atomic_max_u32(&res[s], r);                           // pTos / pIds of :291-292: the last applied arc of a start is its longest
            }
            wv::sync();
        }
    }

    // ------------------------------------------------------------------------------------------------------------------
    // emit (:299-313 + tokdll:1512-1529).  Returns the id count, -1: a position without an arc (the reference does not terminate)
    // ------------------------------------------------------------------------------------------------------------------
    BF_WVD int emit(int32_t *ids, int32_t *spans)
    {
        int cnt = 0; bool err = false;
        for (int w0 = 0; w0 < L; w0 += 64) {
            const int q = w0 + lane; const bool valid = q < L;
            const bool tok = valid && !bit(bmi, q);
            const unsigned long long TM = wv::ballot(tok);
            const int k = cnt + __builtin_popcountll(TM & ((1ull << lane) - 1ull));
            if (tok) {
                const uint32_t r = wv::load_l2(&res[q]);
                int id;
                if (r != 0) {
                    if (r & BS_R_UNK) id = p.unk;
                    else { const uint32_t pr = r & BS_R_PRIO_MASK; id = p.place_id ? p.place_id[pr >> 1] : (int)(pr >> 1); }
                } else if (bit(bms, q)) {
                    const uint32_t c = (uint32_t)src[q];
                    id = p.info[(int)(p.T[p.initial + c] >> SG_OW_SHIFT)].id;
                } else { id = p.unk; if (q > 0) err = true; }           // pTos[0] == 0, pIds[0] == UnkId; elsewhere the reference walks backwards
                if (k < p.max_ids) {
                    ids[k] = id + p.id_offset;
                    if (spans) { spans[2 * k] = q; if (k > 0) spans[2 * (k - 1) + 1] = q - 1; }     // a token ends before the next one starts
                } else if (spans && k == p.max_ids) spans[2 * (k - 1) + 1] = q - 1;
            }
            cnt += __builtin_popcountll(TM);
        }
        if (spans && lane == 0 && cnt > 0 && cnt <= p.max_ids) spans[2 * (cnt - 1) + 1] = L - 1;
        if (wv::any(err)) return -1;
        return cnt;
    }

    // ------------------------------------------------------------------------------------------------------------------
    BF_WVD uint8_t *claim(unsigned long long bytes)
    {
        unsigned long long at = 0;
        if (lane == 0) at = wv::atomic_add(p.pool_used, bytes);
        at = wv::bcast(at, 0);
        if (at + bytes > p.pool_bytes) {                                // the claim is taken back: smaller documents behind this one still fit
            if (lane == 0) { wv::atomic_add(p.pool_used, 0ull - bytes); wv::atomic_add(p.pool_need, bytes); }
            return nullptr;
        }
        return p.pool + at;
    }
    BF_WVD void fail(int64_t d, int bits) { if (lane == 0) { p.counts[d] = 0; wv::atomic_or(p.status, bits); } }

    BF_WVD void run_doc(int64_t d)
    {
        L = p.lens[d];
        if (L <= 0) { if (lane == 0) p.counts[d] = 0; return; }
        const int64_t slot = (int64_t)p.slot_mul * (p.doc_off[d] + d);
        src = p.stream + slot;
        fast = p.kind == SG_KIND_BPE_OPT || p.kind == SG_KIND_BPE_MERGES;     // m_fFastBpe (..._bpe_t.h:110, ..._with_merges_t.h:113)
        if (p.stats && lane == 0) wv::atomic_add(&p.stats[0], 1ull);
        // ---- the arc count, then the document's block: res, ext (one word per position), the two bitmaps (L + 1 bits), two key arrays
        cap = 0; keys = keys2 = nullptr;
        const int64_t A = collect(false);
        const unsigned long long nw = (unsigned long long)((L + 32) >> 5);
        const unsigned long long words = 2ull * (unsigned long long)L + 2ull * nw;
        const unsigned long long bytes = ((words * 4ull + 15ull) & ~15ull) + 2ull * (unsigned long long)A * 8ull;
        uint8_t *blk = claim(bytes);
        if (!blk) { fail(d, BF_STATUS_POOL); return; }
        res = (uint32_t *)blk; ext = res + L; bmi = ext + L; bms = bmi + nw;
        keys = (uint64_t *)(blk + ((words * 4ull + 15ull) & ~15ull)); keys2 = keys + A; cap = A;
        for (unsigned long long q = (unsigned long long)lane; q < words; q += 64) res[q] = 0;
        wv::sync();
        const int64_t A2 = collect(true);
        if (A2 != A) { fail(d, BF_STATUS_INTERNAL); return; }
        if (p.stats && lane == 0) wv::atomic_add(&p.stats[1], (unsigned long long)A);
        wv::sync();
        const uint64_t *sorted = A > 1 ? sort(A) : keys;
        apply(sorted, A);
        const int cnt = emit(p.ids_tmp + slot, p.span_tmp ? p.span_tmp + 2 * slot : nullptr);
        if (cnt < 0) { fail(d, BF_STATUS_DOC_FAILED); return; }
        if (lane == 0) p.counts[d] = cnt < p.max_ids ? cnt : p.max_ids;
    }

    BF_WVD void run()
    {
        const int64_t n = p.list ? (int64_t)*p.list_n : p.ndocs;
        for (;;) {
            unsigned long long i = 0;
            if (lane == 0) i = wv::atomic_add(p.next_doc, 1ull);
            i = wv::bcast(i, 0);
            if ((int64_t)i >= n) break;
            const int64_t d = p.list ? (int64_t)p.list[i] : (int64_t)i;
            if (p.narcs && p.narcs[d] != p.narcs_want) continue;
            run_doc(d);
            wv::sync();
        }
    }
};

} // namespace bfa
