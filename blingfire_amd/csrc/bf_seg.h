// bf_seg.h -- per-document segmenter programs for the SentencePiece-style branch (one document per lane).
//
// Reproduce, on the re-laid-out dictionary tables of bf_model.h (64-bit displacement-packed Mealy entries
// + I2Info rows), the reference
//   FATokenSegmentationTools_1best_t<int>::Process               (Unigram-LM Viterbi 1-best)
//       blingfireclient.library/inc/FATokenSegmentationTools_1best_t.h:175-279 (AddArc 118-142, AddUnknownArc 145-171)
//   FATokenSegmentationTools_1best_bpe_t<int>::Process            (BPE, rank == id, optional bpe-opt shortcut)
//       blingfireclient.library/inc/FATokenSegmentationTools_1best_bpe_t.h:126-316
//   FATokenSegmentationTools_1best_bpe_with_merges_t<int>::Process (BPE with a separate float merge rank)
//       blingfireclient.library/inc/FATokenSegmentationTools_1best_bpe_with_merges_t.h:129-323
// and the id loop of TextToIdsWithOffsets_sp (blingfiretools/blingfiretokdll/blingfiretokdll.cpp:1509-1532).
// Input = the class stream the prep stage produced (dummy prefix, charmap, whitespace collapse already done,
// tokdll:1367-1496).
//
// Floating point: scores are float32 bit patterns widened to double and accumulated exactly as the reference
// does (same operations, same order, strict '<'); compiled with -ffp-contract=off.
//
// BF_HD code: the HIP kernels run it per lane; tests/hosttest compiles it for the host (test-only).
#pragma once
#include "bf_lex.h"

namespace bfa {

constexpr uint64_t SG_CLS_MASK = 0xFFFFFull, SG_FINAL = 1ull << 20, SG_NEXT_MASK = 0x1FFFFFull;
constexpr int SG_NEXT_SHIFT = 21, SG_OW_SHIFT = 42;
constexpr uint64_t SG_MISS = ~0ull;
constexpr uint32_t SG_CLS_NONE = 0xFFFFu;          // class stream: symbol not in the dictionary alphabet
constexpr uint32_t SG_CLS_DELIM_ABSENT = 0xFFFEu;  // class stream: U+2581 when the alphabet does not contain it
constexpr int SG_KIND_UNIGRAM = 1, SG_KIND_BPE = 2, SG_KIND_BPE_OPT = 3, SG_KIND_BPE_MERGES = 4;

struct SegInfo { int32_t id; uint32_t score_bits; };   // one I2Info row (FAMultiMap_pack_fixed.cpp:140-160): [id, float score]

struct SegTables {
    const uint64_t *T;          // displacement-packed Mealy transitions (bf_model.h T64 entry)
    const SegInfo *info;        // I2Info rows, key = MPH index (sum of output weights along the path)
    uint32_t initial;
    uint32_t cls_delim;         // class-stream value of U+2581
    int kind, id_offset;
    const uint32_t *score;      // Unigram lane program: the score bits of the I2Info rows alone (same key); the forward pass reads nothing else of a row
    uint32_t leaf_lo = 0, leaf_n = 0;   // states [leaf_lo, leaf_lo + leaf_n) have no transitions (bf_model.h PackedDfa): a walk that reaches one is over
};

struct SegArc { int32_t start, end, id; uint32_t rank_bits; };   // BPE arc (…_bpe_t.h:66-88, …_with_merges_t.h)

// FAMealyDfa_pack_triv::GetDestOw (cl/src/FAMealyDfa_pack_triv.cpp:69-244) on the packed table
BF_HD uint64_t sg_lookup(const SegTables &S, uint32_t state, uint32_t cls)
{
    if (cls >= SG_CLS_DELIM_ABSENT) return SG_MISS;
    const uint64_t e = S.T[state + cls];
    return (e & SG_CLS_MASK) == cls ? e : SG_MISS;
}
BF_HD float sg_bits_to_float(uint32_t b) { union { uint32_t u; float f; } x; x.u = b; return x.f; }

// ---------------------------------------------------------------------------------------------------
// Unigram-LM.  sc[] / bi[] are the End2BestArc array (…_1best_t.h:61-77,193), one entry per position.
// ---------------------------------------------------------------------------------------------------
struct SegBest { double score; int32_t begin, id; };   // one End2BestArc entry (…_1best_t.h:61-77), 16 bytes

template <class ClsAt, class IdOut>
BF_HD int seg_unigram_doc(const SegTables &S, ClsAt &cls_at, int L, SegBest *best, IdOut &out, int max_ids, int unk)
{
    if (L <= 0) return 0;                                              // …_1best_t.h:186-188
    const double neg_flt_max = -3.40282346638528859811704183484516925e+38;   // (double)-FLT_MAX
    { SegBest z; z.score = neg_flt_max; z.begin = -1; z.id = -1; for (int i = 0; i < L; ++i) best[i] = z; }
    for (int start = 0; start < L; ++start) {
        uint32_t state = S.initial; int sum = 0; bool unknown = true;
        SegBest pb; pb.score = 0; pb.begin = -1; pb.id = 0;
        if (0 < start) pb = best[start - 1];                           // final by now: every arc ending there started earlier
        const double prev = 0 < start ? pb.score : 0;
        for (int i = start; i < L; ++i) {
            const uint64_t e = sg_lookup(S, state, cls_at(i));
            if (e == SG_MISS) break;
            state = (uint32_t)((e >> SG_NEXT_SHIFT) & SG_NEXT_MASK);
            sum += (int)(e >> SG_OW_SHIFT);
            if (e & SG_FINAL) {                                        // AddArc (…_1best_t.h:118-142)
                const SegInfo r = S.info[sum];
                const float score = sg_bits_to_float(r.score_bits);
                const double cand = score + prev;
                SegBest b = best[i];
                if (b.score < cand) { b.begin = start; b.id = r.id; b.score = cand; best[i] = b; }
                unknown = false;
            }
        }
        if (unknown) {                                                 // AddUnknownArc (…_1best_t.h:145-171)
            const float unk_score = -100000.0f;
            const double cand = unk_score + prev;
            SegBest b = best[start];
            if (b.score < cand) {
                b.begin = start;
                if (0 < start && -1 == pb.id) b.begin = pb.begin;
                b.id = -1; b.score = cand; best[start] = b;
            }
        }
    }
    // follow the best path backwards (…_1best_t.h:237-265); the reference reverses the triple array afterwards,
    // here the token count is taken first so that each id goes straight to its forward position
    int cnt = 0;
    for (int end = L - 1; 0 <= end; end = best[end].begin - 1) ++cnt;
    int k = cnt - 1;
    for (int end = L - 1; 0 <= end; --k) {
        const SegBest b = best[end];
        const int id = b.id != -1 ? b.id : unk;
        if (k < max_ids) { out.put(k, id + S.id_offset); out.span(k, b.begin, end); }   // tokdll:1512-1529
        end = b.begin - 1;
    }
    const int n = cnt < max_ids ? cnt : max_ids;
    out.finish(n);
    return n;
}

// ---------------------------------------------------------------------------------------------------
// Unigram-LM as a resumable lane program (the default GPU form; bf_kernels.hip k_seg_unigram_lane drives it, tests/hosttest
// runs the same code on the host).  Same per-document order of operations as seg_unigram_doc above, cut into
//   wstep()  one trie transition of the forward pass (AddArc / AddUnknownArc / next start absorbed), and
//   bstep()  one hop of the backward pass,
// with changes of *mechanism* that keep every value identical:
//  * only `depth` End2BestArc entries are live at a time (an arc from `start` ends before start + depth, depth = longest
//    dictionary entry): score AND {begin, id} sit in a ring (LDS on the device).  An entry is final when `start` moves past its
//    position -- it is then written to memory ONCE, packed into 32 bits, QN positions per aligned group (16 bytes, or a whole 64-byte sector in the split form) (measured on MI355X:
//    8-byte records stored on every improvement cost 12.5 GB of write traffic per 0.3 GB of text and evicted the tables from L2);
//  * the relaxation of a final transition is DEFERRED by one step: the I2Info row is requested when the transition is taken
//    and consumed at the beginning of the next step, behind the issue of that step's trie gather -- the two dependent
//    gathers of a final transition overlap instead of adding up.  Relaxations still happen in arc order, and the pending one
//    is flushed before anything reads the score it may change (the end of the walk from `start` reads position `start`).
// Packed record of a position: [len - 1 : 12 | id + 1 : 20], len = position - begin + 1; id + 1 == 0 is the unknown arc (id -1);
// a length field of 4095 means "4096 or more" -- only possible for a merged run of unknown positions, every position of which
// carries the run's begin, so the backward pass hops 4095 positions back and adds up (bstep_len); 0xFFFFFFFF = no incoming arc
// (the reference's {-1, -1} sentinel).  The "id" of a record is the entry's MPH index (the key of its I2Info row): the forward pass then
// needs only the row's score -- a 4-byte array half the size of the rows, so that transitions + scores of xlm_roberta_base.bin fit the
// 4 MB of L2 of an XCD together -- and the id is looked up once per TOKEN in the backward pass instead of once per arc.  Needs fewer
// than 2^20 - 2 rows (checked at load; other models use the sequential form).
// ---------------------------------------------------------------------------------------------------
constexpr uint32_t UNI_REC_NONE = 0xFFFFFFFFu, UNI_LEN_MAX = 4095u;
constexpr int UNI_MAX_ID = (1 << 20) - 3;
BF_HD uint32_t uni_rec(int id, int len) { const uint32_t l = (uint32_t)(len - 1); return ((uint32_t)(id + 1) & 0xFFFFFu) | ((l < UNI_LEN_MAX ? l : UNI_LEN_MAX) << 20); }

template <class ClsAt, class Ring, int QN = 4>
struct UniLane {
    const SegTables &S; ClsAt &cls_at; Ring &ring; uint32_t *recs;
    int L, depth, start, i, sum; uint32_t state; bool unknown, pend; double prev; uint32_t pend_score; int pend_key; int pend_i;
    int unk_run;                                       // length of the unknown run that ends at start - 1 (0: that position is not unknown)
    uint32_t q[QN]; int qn;                            // final records of the last positions, not yet stored (qn of them, q[QN - 1] newest; static indices only: registers)
    int64_t abs0;                                      // absolute element index of position 0 (16-byte store groups are aligned on it)
    int end, cnt;                                      // backward pass

    BF_HD UniLane(const SegTables &S_, ClsAt &c, Ring &r) : S(S_), cls_at(c), ring(r), recs(nullptr) {}

    static BF_HD double neg_flt_max() { return -3.40282346638528859811704183484516925e+38; }   // (double)-FLT_MAX

    // Start a document of L >= 1 stream elements; recs_ has room for L records and is element abs0_ of a 16-byte aligned array.
    BF_HD void init(int L_, int depth_, uint32_t *recs_, int64_t abs0_)
    {
        L = L_; depth = depth_; recs = recs_; abs0 = abs0_;
        ring.fill(neg_flt_max());
        start = 0; i = 0; state = S.initial; sum = 0; unknown = true; prev = 0; pend = false; pend_i = 0; pend_score = 0; pend_key = 0;
        unk_run = 0; for (int k = 0; k < QN; ++k) q[k] = 0; qn = 0;
        end = 0; cnt = 0;
        cls_at.seek(0);
    }

    BF_HD void relax()                                 // AddArc (..._1best_t.h:118-142) of the pending final transition
    {
        const double cand = sg_bits_to_float(pend_score) + prev;
        if (ring.score(pend_i) < cand) ring.set(pend_i, cand, uni_rec(pend_key, pend_i - start + 1));
        pend = false;
    }

    // the record of position p (= start) is final: queue it, store whole aligned groups of four
    BF_HD void finalize(int p, uint32_t r)
    {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
        for (int k = 0; k + 1 < QN; ++k) q[k] = q[k + 1];
        q[QN - 1] = r; ++qn;
        const int64_t a = abs0 + p;
        if ((a & (QN - 1)) == QN - 1 || p == L - 1) {
            if (qn == QN && (a & (QN - 1)) == QN - 1) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                for (int k = 0; k < QN; k += 4) *(uint4 *)(recs + (p - (QN - 1) + k)) = make_uint4(q[k], q[k + 1], q[k + 2], q[k + 3]);
#else
                for (int k = 0; k < QN; ++k) recs[p - (QN - 1) + k] = q[k];
#endif
            } else {                                   // a group that the document only partly owns (its first / last positions)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
                for (int j = 0; j < QN; ++j) if (qn > j) recs[p - j] = q[QN - 1 - j];
            }
            qn = 0;
        }
    }

    // One trie transition.  Returns false once the forward pass is complete (follow with begin_back / bstep).
    BF_HD bool wstep()
    {
        const uint32_t c = cls_at(i);
        const bool valid = c < SG_CLS_DELIM_ABSENT;                     // sg_lookup: symbols outside the alphabet never match
        const uint64_t e = S.T[state + (valid ? c : 0u)];               // the gather is issued ...
        if (pend) relax();                                              // ... and the previous arc is relaxed while it travels
        const bool hit = valid && (e & SG_CLS_MASK) == c;
        bool ends = !hit;
        if (hit) {
            state = (uint32_t)((e >> SG_NEXT_SHIFT) & SG_NEXT_MASK);
            sum += (int)(e >> SG_OW_SHIFT);
            if (e & SG_FINAL) { pend_score = S.score[sum]; pend_key = sum; pend_i = i; pend = true; unknown = false; }   // requested now, used next step
            ++i;
            ends = i >= L;
        }
        if (ends) {
            if (pend) relax();                                          // only when the arc that ends the document is the last of its walk
            double fin = ring.score(start);
            uint32_t r = ring.rec(start);
            int run = 0;
            if (unknown) {                                              // AddUnknownArc (..._1best_t.h:145-171)
                const float unk_score = -100000.0f;
                const double cand = unk_score + prev;
                if (fin < cand) {                                       // begin = start, or the begin of the unknown arc that ends just before
                    run = unk_run + 1;
                    r = uni_rec(-1, run); fin = cand;
                }
            }
            if (!(neg_flt_max() < fin)) r = UNI_REC_NONE;               // no incoming arc at all (..._1best_t.h:61-77)
            unk_run = run;
            finalize(start, r);
            ++start;
            if (!(start < L)) return false;
            prev = fin;                                                 // End2BestArc[start - 1] is final by now
            ring.set(start + depth - 1, neg_flt_max(), UNI_REC_NONE);   // the position that enters the reach of this start
            i = start; state = S.initial; sum = 0; unknown = true;
            cls_at.seek(start);
        }
        return true;
    }

    BF_HD void begin_back() { end = L - 1; cnt = 0; }
    // One hop of the backward pass (..._1best_t.h:237-265) given r = recs[end] (read by the caller, so that a GPU driver can
    // request it early): id k-from-the-end goes to put(k, ...).  Returns false after the last hop.
    template <class IdPut>
    BF_HD bool bstep(uint32_t r, IdPut &put, int unk)
    {
        int id = -1, begin = -1;
        if (r != UNI_REC_NONE) {
            const int key = (int)(r & 0xFFFFFu) - 1;                    // the entry's MPH index, -1: the unknown arc
            if (key != -1) id = S.info[key].id;
            int64_t len = (int64_t)(r >> 20) + 1;
            if ((r >> 20) == UNI_LEN_MAX) {                             // a long unknown run: add up 4095-position hops
                int e = end; uint32_t rr = r; len = 0;
                while ((rr >> 20) == UNI_LEN_MAX && e - (int)UNI_LEN_MAX >= 0) { len += UNI_LEN_MAX; e -= (int)UNI_LEN_MAX; rr = recs[e]; }
                len += (int64_t)(rr >> 20) + 1;
            }
            begin = (int)((int64_t)end - len + 1);
        }
        put(cnt, (id != -1 ? id : unk) + S.id_offset, begin, end);
        ++cnt;
        end = begin - 1;
        return end >= 0;
    }
};

// ---------------------------------------------------------------------------------------------------
// Unigram-LM, cut form (round 6; bf_kernels_sp.hip k_uni_cut drives it, tests/hosttest runs the same code on the host): the lane program
// above WITHOUT the lattice in memory.  The forward pass is UniLane's (same walks, same deferred relaxation, same doubles in the same
// order); what changes is what is kept of the End2BestArc records and where.  A position p is a CUT when
//   (a) its final record is a dictionary entry (not the unknown arc, not "no incoming arc"), and
//   (b) no dictionary arc found from any start <= p ends behind p   (reach <= p: every walk from a start <= p is complete when p is final).
// Then the backward pass of the reference (..._1best_t.h:237-265) is certain to land on p: a hop over p would need an arc that spans
// p | p + 1 -- a dictionary arc contradicts (b), a merged unknown run (begin <= p < end) makes every position of the run an unknown record and
// contradicts (a).  So the tokens of [previous cut + 1, p] are final as soon as p is: they are read off the record ring (LDS) by the
// reference's own hops and leave the lane in forward order; the last position of a document always ends a chunk.  Nothing is proved about
// the model -- the criterion is evaluated on the lattice of the document itself (for the shipped SentencePiece models every position in front
// of a U+2581 qualifies, because no entry holds U+2581 behind its first symbol; text without blanks is cut wherever its arcs allow).
// A hop that lands on a position without incoming arc makes the reference emit <UnkId, -1, end> and STOP (..._1best_t.h:250-262): everything
// in front of it is dropped -- here the output restarts at index 0 with that token.
// What bounds the lane program is the latency of its dependent gathers times the waves a CU holds, and the waves are bounded by the LDS of
// the rings (measured on MI355X, configs 4 / 5: T = 22 ms + 756 ms / waves per CU).  So a record is ONE BYTE -- the length of the arc,
// which is all the hops need:
//     0           no incoming arc                      1 .. 32      a dictionary entry of that many symbols
//     0x80 | f    the unknown arc, f = min(run - 1, 127), run = length of the merged unknown run that ends here (f == 127: "128 or more" --
//                 every position of a run carries its length so far, so the hop goes 127 back and adds up, like UniLane's 4095)
// and WHICH entry a token is -- 20 bits of MPH index per position in the lane program above -- is not carried along at all.  A token leaves as
// one word:   0x80000000               the unknown id
//             0x40000000 | key         the entry's MPH index, when the lane still knows it (below)
//             (len - 1) << 25 | begin  otherwise: k_uni_ids walks the token's symbols once more (independent walks at full occupancy;
//                                      uni_token_id below).  A document of 2^25 elements or more walks them here instead.
// Nine chunks in ten are one token, or two, and leave by the short way (quick(), once per trip of the driver): the lane remembers the arc
// that last extended `reach` -- its start and MPH index, two registers -- and at a cut p that is the longest entry from the leftmost start that
// ends at p: if the best arc into p begins where it does, it is that arc (the arc from a start to an end is unique) and its key goes out.
// The other chunks (more tokens, an unknown run in front of the cut) wait for the wave's emission phase and are read off the ring by the
// reference's hops.
// The ring holds the records of [ring_lo, i]: a record slot is written by the relaxation that sets its score (a position no arc reached is
// "no incoming arc" by its score alone), so nothing is reserved ahead of the walk.  A lane whose pending region fills the ring with no cut to
// emit (a word of more than ~24 symbols, a long unknown run) SPILLS its oldest final records to the document's record bytes in memory;
// the hops read them back from there.  Rare: 0.1 % of the documents of running text.
// ---------------------------------------------------------------------------------------------------
enum { UC_DONE = 0, UC_MORE = 1, UC_STALL = 2 };
constexpr int UC_SPILL = 8;                           // records per spill
constexpr uint32_t UC_NONE = 0u, UC_UNK = 0x80u, UC_RUN_MAX = 127u, UC_TOK_UNK = 0x80000000u, UC_TOK_KEY = 0x40000000u;
constexpr int UC_BEGIN_BITS = 25;
#ifndef BF_UC_STAT
#define BF_UC_STAT(k)
#endif
BF_HD uint32_t uc_unk_rec(int run) { const uint32_t f = (uint32_t)(run - 1); return UC_UNK | (f < UC_RUN_MAX ? f : UC_RUN_MAX); }

template <class ClsAt, class Ring>
struct UniCut {
    const SegTables &S; ClsAt &cls_at; Ring &ring; uint8_t *recs;
    int L, depth, W, start, i, sum; uint32_t state; bool unknown, pend, walking; double prev; uint32_t pend_score; int pend_i;
    int unk_run;
    int reach, rk, rs;                                 // last position a dictionary arc found so far ends at; MPH index and start of the arc that moved it there
    int ck, cs;                                        // the same two of the arc that moved `reach` to the latest cut (cs == -1: none)
    int cut0, lastcut;                                 // first position not emitted yet; latest cut (cut0 - 1: nothing to emit)
    int ring_lo;                                       // first position whose record is in the ring (cut0 <= older ones: in recs[])
    int nout;                                          // tokens emitted so far (not limited by max_ids)
    int tok_align;                                     // index of the document's token 0 in a 16-byte aligned array, mod 4 (emit() stores aligned groups of four)

    BF_HD UniCut(const SegTables &S_, ClsAt &c, Ring &r) : S(S_), cls_at(c), ring(r), recs(nullptr) {}
    static BF_HD double neg_flt_max() { return -3.40282346638528859811704183484516925e+38; }

    // Start a document of L >= 1 stream elements; recs_ has room for L record bytes (touched by spills only)
    BF_HD void init(int L_, int depth_, int W_, uint8_t *recs_)
    {
        L = L_; depth = depth_; W = W_; recs = recs_;
        ring.fill(neg_flt_max());
        start = 0; prev = 0; pend = false; pend_i = 0; pend_score = 0;
        unk_run = 0; reach = -1; rk = 0; rs = 0; ck = 0; cs = -1; cut0 = 0; lastcut = -1; ring_lo = 0; nout = 0; tok_align = 0;
        i = 0; state = S.initial; sum = 0; unknown = true; walking = true;
        cls_at.seek(0);
    }
    BF_HD void relax()
    {
        const double cand = sg_bits_to_float(pend_score) + prev;
        if (ring.score(pend_i) < cand) ring.set(pend_i, cand, (uint32_t)(pend_i - start + 1));
        pend = false;
    }
    BF_HD bool pending() const { return lastcut >= cut0; }
    BF_HD bool room(int k) const { return i + k - 1 - ring_lo < W; }     // room in the ring for k more steps (a step may write the record of position i)
    BF_HD bool stalled() const { return !room(1); }
    BF_HD uint32_t rec_at(int e) const { return e >= ring_lo ? ring.rec(e) : (uint32_t)recs[e]; }
    // a lane without room for k more steps and with nothing to emit: its oldest final records go to memory (the walk is at most depth - 1
    // positions ahead of `start`, so W >= depth + UC_SPILL + k leaves at least UC_SPILL final ones in the ring)
    BF_HD void spill(int k = 1)
    {
        if (room(k)) return;                                            // (the short way out may have made room since the walk stalled)
        for (int q = 0; q < UC_SPILL; ++q) recs[ring_lo + q] = (uint8_t)ring.rec(ring_lo + q);
        ring_lo += UC_SPILL;
    }

    // the word of a token that is a dictionary entry of `len` symbols from `begin`, key unknown
    BF_HD uint32_t walk_word(int begin, int len)
    {
        if (begin < (1 << UC_BEGIN_BITS)) return ((uint32_t)(len - 1) << UC_BEGIN_BITS) | (uint32_t)begin;
        uint32_t st = S.initial; int sm = 0;                            // (a document this long: the walk k_uni_ids would do)
        for (int j = 0; j < len; ++j) { const uint64_t e = S.T[st + cls_at(begin + j)]; st = (uint32_t)((e >> SG_NEXT_SHIFT) & SG_NEXT_MASK); sm += (int)(e >> SG_OW_SHIFT); }
        return UC_TOK_KEY | (uint32_t)sm;
    }

    // One trie transition of the walk from `start` (the walk must be on: `walking`): the gather for the symbol at position i, the
    // relaxation of the arc the step before found while it travels (AddArc, ..._1best_t.h:118-142), the transition.  The walk is over
    // (`walking` false) behind a symbol without transition, at the end of the document, and in a state without transitions (the next
    // GetDestOw would fail, ..._1best_t.h:209-212); finish_start() then ends the start position.
    // (Measured and removed again, round 6: the first transition of the NEXT walk gathered one walk ahead -- it depends on nothing but the symbol
    //  there -- takes a third of the steps out and made the kernel 4 ms slower: its time is instruction issue and the gathers' rate, not the
    //  length of a lane's chain.)
    BF_HD void step()
    {
        const uint32_t c = cls_at(i);
        const bool valid = c < SG_CLS_DELIM_ABSENT;
        const uint64_t e = S.T[state + (valid ? c : 0u)];               // the gather is issued ...
        if (pend) relax();                                              // ... and the previous arc is relaxed while it travels
        const bool hit = valid && (uint32_t)(e & SG_CLS_MASK) == c;
        if (hit) {
            state = (uint32_t)((e >> SG_NEXT_SHIFT) & SG_NEXT_MASK);
            sum += (int)(e >> SG_OW_SHIFT);
            if (e & SG_FINAL) {
                pend_score = S.score[sum]; pend_i = i; pend = true; unknown = false;       // requested now, relaxed at the next step
                const bool ext = i > reach;
                reach = ext ? i : reach; rk = ext ? sum : rk; rs = ext ? start : rs;
            }
            ++i;
        }
        walking = hit && i < L && !(state - S.leaf_lo < S.leaf_n);
    }

    // The walk from `start` is over: the position is final (AddUnknownArc ..._1best_t.h:145-171 if no arc began here), a cut is noted, the
    // next walk is set up.  UC_DONE behind the document's last position.
    BF_HD int finish_start()
    {
        if (pend) relax();
        double fin = ring.score(start);
        uint32_t r = ring.rec(start);
        int run = 0;
        if (unknown) {
            const float unk_score = -100000.0f;
            const double cand = unk_score + prev;
            if (fin < cand) { run = unk_run + 1; r = uc_unk_rec(run); fin = cand; }
        }
        const bool none = !(neg_flt_max() < fin);                       // no incoming arc at all (..._1best_t.h:61-77); its record slot was never written
        if (none) r = UC_NONE;
        unk_run = run;
        if (run != 0 || none) ring.setrec(start, r);
        const bool piece = !none && run == 0;
        if ((piece && reach <= start) || start == L - 1) {              // a cut: remember what is known about the arc that ends here (quick())
            lastcut = start; ck = rk; cs = (piece && reach == start) ? rs : -1;
        }
        ++start;
        if (!(start < L)) return UC_DONE;
        prev = fin;
        ring.setscore(start + depth - 1, neg_flt_max());                // the position that enters the reach of this start
        i = start; state = S.initial; sum = 0; unknown = true; walking = true;
        cls_at.advance(start);
        return UC_MORE;
    }

    // step() and finish_start() as one call (the host drivers; the device runs a few steps, then the finish, once per trip).
    // UC_STALL: the ring is full -- emit() or spill() first (nothing was done).
    BF_HD int wstep()
    {
        if (stalled()) return UC_STALL;
        step();
        return walking ? (int)UC_MORE : finish_start();
    }

    // The short way out, once per trip of the driver: the chunk [cut0, lastcut] when it is ONE token, or two, both dictionary entries -- nine
    // chunks in ten of running text.  The last token's key is at hand when the arc that moved `reach` to the cut begins where the token does (it
    // is the longest entry from the leftmost start that ends there; the arc from a start to an end is unique): its word carries the key, every
    // other word names its symbols.  put(index in the document's token sequence, word).  Anything else stays for emit().
    template <class Put>
    BF_HD void quick(Put &put)
    {
        if (!pending() || lastcut < ring_lo || L > (1 << UC_BEGIN_BITS)) return;
        const uint32_t r = ring.rec(lastcut);
        if (r == UC_NONE || (r & UC_UNK)) return;
        const int b1 = lastcut - (int)r + 1;                            // the last token begins here
        const uint32_t w1 = cs == b1 ? (UC_TOK_KEY | (uint32_t)ck) : (((r - 1u) << UC_BEGIN_BITS) | (uint32_t)b1);
        int n = 0; uint32_t w0 = w1;
        if (b1 == cut0) n = 1;
        else if (b1 > cut0 && b1 - 1 >= ring_lo) {
            const uint32_t r0 = ring.rec(b1 - 1);
            if (r0 != UC_NONE && !(r0 & UC_UNK) && b1 - (int)r0 == cut0) { n = 2; w0 = ((r0 - 1u) << UC_BEGIN_BITS) | (uint32_t)cut0; }
        }
        if (n == 0) return;
        put(nout, w0);
        if (n == 2) put(nout + 1, w1);
        BF_UC_STAT(cs == b1 ? 0 : 1); if (n == 2) BF_UC_STAT(1);
        nout += n; cut0 = lastcut + 1; ring_lo = ring_lo < cut0 ? cut0 : ring_lo;
    }

    // length of the token whose last position is e, r = its record
    BF_HD int tok_len(int e, uint32_t r) const
    {
        if (!(r & UC_UNK)) return (int)r;
        int len = 0;
        while ((r & 0x7Fu) == UC_RUN_MAX && e - (int)UC_RUN_MAX >= 0 && (r & UC_UNK)) { len += (int)UC_RUN_MAX; e -= (int)UC_RUN_MAX; r = rec_at(e); }
        return len + (int)(r & 0x7Fu) + 1;
    }

    // The tokens of [cut0, lastcut], first to last: put(index in the document's token sequence, word).  Two passes over the records: the
    // hops of the backward pass count the tokens, the same hops place them.
    template <class Put>
    BF_HD void emit(Put &put)
    {
        if (!pending()) return;
        int n = 0; bool restart = false;
        for (int e = lastcut; e >= cut0;) {
            const uint32_t r = rec_at(e);
            ++n;
            if (r == UC_NONE) { restart = true; break; }               // <UnkId, -1, end>, and the reference stops
            e -= tok_len(e, r);
        }
        const int base = restart ? 0 : nout;
        // the words leave last to first, four at a time when they make an aligned group of the document's token array (put.quad: one 16-byte store;
        // a 4-byte store per token made the kernel write 22 GB for 4.6 GB of tokens -- every store its own sector, round 6 counters)
        uint32_t g0 = 0, g1 = 0, g2 = 0, g3 = 0; int gn = 0;            // g0: the word at the lowest index so far
        int k = n - 1;
        for (int e = lastcut; k >= 0; --k) {
            const uint32_t r = rec_at(e);
            uint32_t w = UC_TOK_UNK; int len = 0;
            if (r != UC_NONE) { len = tok_len(e, r); if (!(r & UC_UNK)) w = walk_word(e - len + 1, len); BF_UC_STAT(2); }
            g3 = g2; g2 = g1; g1 = g0; g0 = w; ++gn;
            const int idx = base + k;
            if (((idx + tok_align) & 3) == 0 || k == 0 || r == UC_NONE) {
                if (gn == 4 && ((idx + tok_align) & 3) == 0) put.quad(idx, g0, g1, g2, g3);
                else { put(idx, g0); if (gn > 1) put(idx + 1, g1); if (gn > 2) put(idx + 2, g2); if (gn > 3) put(idx + 3, g3); }
                gn = 0;
            }
            if (r == UC_NONE) break;
            e -= len;
        }
        nout = base + n;
        cut0 = lastcut + 1;
        if (ring_lo < cut0) ring_lo = cut0;
    }
};

// The id of a token word (what k_uni_ids does per lane): the key, or the walk of the token's symbols, gives the entry's MPH index, the id
// column of I2Info the id (..._1best_t.h:118-142 stored it with the arc; -1 and the unknown arc: UnkId, tokdll:1512-1516 adds IdOffset to
// either).  A walked token is an arc the forward pass found, so every transition exists; ids[] = Model::i2info_id.
template <class ClsAt>
BF_HD int uni_token_id(const SegTables &S, const int32_t *ids, ClsAt &cls_at, uint32_t word, int unk)
{
    int id = -1;
    if (word & UC_TOK_KEY) id = ids[word & 0xFFFFFu];
    else if (!(word & UC_TOK_UNK)) {
        const int begin = (int)(word & ((1u << UC_BEGIN_BITS) - 1u)), len = (int)(word >> UC_BEGIN_BITS) + 1;
        uint32_t state = S.initial; int sum = 0;
        for (int j = 0; j < len; ++j) {
            const uint64_t e = S.T[state + cls_at(begin + j)];
            state = (uint32_t)((e >> SG_NEXT_SHIFT) & SG_NEXT_MASK);
            sum += (int)(e >> SG_OW_SHIFT);
        }
        id = ids[sum];
    }
    return (id != -1 ? id : unk) + S.id_offset;
}

// ---------------------------------------------------------------------------------------------------
// BPE (both flavours).  arcs[] has room for arc_cap entries; tos/idsv/inter are the three work arrays of
// …_bpe_t.h:258-296.  Returns -1 if arc_cap is exceeded (the host turns that into a loud error).
// ---------------------------------------------------------------------------------------------------
BF_HD bool sg_arc_less(const SegArc &a, const SegArc &b, bool merges)
{
    if (merges) {                                                      // …_with_merges_t.h:242-262: bigger ranks first
        const float ra = sg_bits_to_float(a.rank_bits), rb = sg_bits_to_float(b.rank_bits);
        if (ra > rb) return true;
        if (!(ra == rb)) return false;
    }
    if (a.id < b.id) return true;                                      // …_bpe_t.h:238-255: smaller ids first
    if (a.id != b.id) return false;
    return a.start < b.start;                                          // then left-most first
}

// in-place heap sort (the order is total on the arcs that can occur, so any correct sort equals std::qsort)
BF_HD void sg_sort_arcs(SegArc *a, int n, bool merges)
{
    for (int root0 = n / 2 - 1; root0 >= 0; --root0) {
        int root = root0; const SegArc v = a[root];
        for (;;) {
            int child = 2 * root + 1;
            if (child >= n) break;
            if (child + 1 < n && sg_arc_less(a[child], a[child + 1], merges)) ++child;
            if (!sg_arc_less(v, a[child], merges)) break;
            a[root] = a[child]; root = child;
        }
        a[root] = v;
    }
    for (int end = n - 1; end > 0; --end) {
        const SegArc v = a[end]; a[end] = a[0];
        int root = 0;
        for (;;) {
            int child = 2 * root + 1;
            if (child >= end) break;
            if (child + 1 < end && sg_arc_less(a[child], a[child + 1], merges)) ++child;
            if (!sg_arc_less(v, a[child], merges)) break;
            a[root] = a[child]; root = child;
        }
        a[root] = v;
    }
}

// Phase A: collect the arcs of one document (…_bpe_t.h:151-232).  Returns the arc count, or -1 if arc_cap is exceeded.
template <class ClsAt>
BF_HD int seg_bpe_collect(const SegTables &S, ClsAt &cls_at, int L, SegArc *arcs, int arc_cap, int unk)
{
    const bool merges = S.kind == SG_KIND_BPE_MERGES;
    const bool fast = S.kind == SG_KIND_BPE_OPT || merges;             // m_fFastBpe (…_bpe_t.h:110, …_with_merges_t.h:113)
    int narcs = 0;
    for (int start = 0; start < L; ++start) {
        uint32_t state = S.initial; int sum = 0; bool unknown = true;
        const bool token_start = cls_at(start) == S.cls_delim;
        const int count_at_start = narcs;
        int ff = start;
        for (int i = start; i < L; ++i) {
            const uint64_t e = sg_lookup(S, state, cls_at(i));
            if (e == SG_MISS) break;
            state = (uint32_t)((e >> SG_NEXT_SHIFT) & SG_NEXT_MASK);
            sum += (int)(e >> SG_OW_SHIFT);
            if (e & SG_FINAL) {
                const SegInfo r = S.info[sum];
                const bool apply = fast && token_start && ((i < L - 1) ? cls_at(i + 1) == S.cls_delim : true) && count_at_start < narcs;
                SegArc a; a.start = start; a.end = i; a.id = r.id; a.rank_bits = r.score_bits;
                if (!apply) {
                    if (narcs >= arc_cap) return -1;
                    arcs[narcs++] = a;
                } else {                                               // whole-token arc replaces the pieces (…_bpe_t.h:189-206)
                    arcs[count_at_start] = a; narcs = count_at_start + 1; ff = i;
                }
                unknown = false;
            }
        }
        if (unknown) {                                                 // …_bpe_t.h:212-225
            if (0 < narcs && unk == arcs[narcs - 1].id) arcs[narcs - 1].end = start;
            else {
                if (narcs >= arc_cap) return -1;
                SegArc a; a.start = start; a.end = start; a.id = unk; a.rank_bits = 0;   // rank 0.0f
                arcs[narcs++] = a;
            }
        }
        if (fast) start = ff;                                          // …_bpe_t.h:228-230
    }
    return narcs;
}

// Phase B, sequential form: sort, apply the merges in order, emit (…_bpe_t.h:234-313 + tokdll:1512-1516).
// Returns the id count, or -2 when a token start has no applied arc (the reference would loop forever there).
template <class IdOut>
BF_HD int seg_bpe_finish(const SegTables &S, int L, SegArc *arcs, int narcs, int32_t *tos, int32_t *idsv, uint8_t *inter,
                         IdOut &out, int max_ids, int unk, bool presorted = false)
{
    const bool merges = S.kind == SG_KIND_BPE_MERGES;
    if (!presorted) sg_sort_arcs(arcs, narcs, merges);
    for (int i = 0; i < L; ++i) { tos[i] = 0; idsv[i] = unk; inter[i] = 0; }
    for (int k = 0; k < narcs; ++k) {                                  // …_bpe_t.h:274-296
        const int s = arcs[k].start, e = arcs[k].end;
        if (0 == inter[s] && (e + 1 == L || 0 == inter[e + 1])) {
            tos[s] = e; idsv[s] = arcs[k].id;
            for (int q = s + 1; q <= e; ++q) inter[q] = 1;
        }
    }
    int cnt = 0;
    for (int start = 0; start < L; ++start) {                          // …_bpe_t.h:299-313
        const int e = tos[start];
        if (e < start) return -2;
        if (cnt < max_ids) { out.put(cnt, idsv[start] + S.id_offset); out.span(cnt, start, e); }
        ++cnt;
        start = e;
    }
    const int n = cnt < max_ids ? cnt : max_ids;
    out.finish(n);
    return n;
}

template <class ClsAt, class IdOut>
BF_HD int seg_bpe_doc(const SegTables &S, ClsAt &cls_at, int L, SegArc *arcs, int arc_cap, int32_t *tos, int32_t *idsv,
                      uint8_t *inter, IdOut &out, int max_ids, int unk)
{
    if (L <= 0) return 0;
    const int narcs = seg_bpe_collect(S, cls_at, L, arcs, arc_cap, unk);
    if (narcs < 0) return -1;
    return seg_bpe_finish(S, L, arcs, narcs, tos, idsv, inter, out, max_ids, unk);
}

// A document whose arcs do not fit the 6 * L + 32 the batch workspace reserves per document (a long run of one character whose
// run-length tokens are all in the vocabulary: '-' * 15 with gpt2.bin).  The reference collects into a std::vector
// (..._bpe_t.h:144,197); here the document gets its arcs and the three work arrays of ..._bpe_t.h:258-296 from a pool through
// `claim(bytes)` (nullptr = the pool is exhausted -> -1, a loud error) and runs the plain sequential program.  The claim is sized by
// an upper bound of the arc count: every final state of every start, plus one unknown arc per start without any (the whole-token
// replacement and the fast-forward of ..._bpe_t.h:189-206,228-230 only ever remove arcs).
constexpr long long SEG_BIG_MAX_ARCS = 1ll << 18;        // ~10 k characters of one run; one lane sorts them in a few seconds
template <class ClsAt, class IdOut, class Claim>
BF_HD int seg_bpe_doc_big(const SegTables &S, ClsAt &cls_at, int L, Claim &claim, IdOut &out, int max_ids, int unk)
{
    if (L <= 0) return 0;
    long long bound = 0;
    for (int start = 0; start < L; ++start) {
        uint32_t state = S.initial; bool any = false;
        for (int i = start; i < L; ++i) {
            const uint64_t e = sg_lookup(S, state, cls_at(i));
            if (e == SG_MISS) break;
            state = (uint32_t)((e >> SG_NEXT_SHIFT) & SG_NEXT_MASK);
            if (e & SG_FINAL) { ++bound; any = true; }
        }
        if (!any) ++bound;
    }
    if (bound > SEG_BIG_MAX_ARCS) return -1;                         // sequential work: a loud error beyond that rather than a launch that runs for a minute
    const size_t arcs_bytes = ((size_t)bound * sizeof(SegArc) + 15) & ~(size_t)15, ints_bytes = ((size_t)L * 4 + 15) & ~(size_t)15,
                 flag_bytes = ((size_t)L + 15) & ~(size_t)15;
    uint8_t *mem = claim(arcs_bytes + 2 * ints_bytes + flag_bytes);
    if (!mem) return -1;
    SegArc *arcs = (SegArc *)mem;
    int32_t *tos = (int32_t *)(mem + arcs_bytes), *idsv = (int32_t *)(mem + arcs_bytes + ints_bytes);
    uint8_t *inter = mem + arcs_bytes + 2 * ints_bytes;
    return seg_bpe_doc(S, cls_at, L, arcs, (int)bound, tos, idsv, inter, out, max_ids, unk);
}

// Sort key of an arc as unsigned integers, ascending == the reference's comparator order:
//   hi = merge rank, descending (0 for plain BPE);  lo = (id ascending) << 32 | start ascending
BF_HD uint32_t sg_key_hi(const SegArc &a, bool merges)
{
    if (!merges) return 0;
    uint32_t b = a.rank_bits;
    if ((b << 1) == 0) b = 0;                                          // -0.0f == 0.0f
    const uint32_t asc = (b & 0x80000000u) ? ~b : (b | 0x80000000u);   // monotone in the float value
    return ~asc;                                                       // bigger ranks first
}
BF_HD uint64_t sg_key_lo(const SegArc &a) { return ((uint64_t)((uint32_t)a.id ^ 0x80000000u) << 32) | (uint32_t)a.start; }

// ---------------------------------------------------------------------------------------------------
// Dictionary key -> info id (reference FADictInterpreter_t<int>::GetInfoId, cl/inc/FADictInterpreter_t.h:334-366, configured
// without a transformation like blingfiretokdll would: SetConf(conf, NULL)) on the same packed Mealy table:
//   :347-349  empty key / longer than FALimits::MaxWordSize (300) -> -1
//   :203-205  an l2r dictionary without ignore-case is looked up AS IS (m_NoNorm): the [pos-dict] charmap is NOT applied
//   :211-279  otherwise (r2l): FANormalizeWord with the charmap (a result longer than 600 elements counts as empty,
//             FAUtils_cl.h:441-487), then the symbols are fed last to first
//   :283-301  FAMphInterpretTools_t::GetId (cl/inc/FAMphInterpretTools_t.h:97-122): every symbol must have a transition, the
//             output weights add up to the MPH index K, the last state must be final; info id = K2I[K]
// One key per lane on the GPU (bf_kernels.hip k_dict_ids); tests/hosttest runs the same code on the host.
// ---------------------------------------------------------------------------------------------------
struct DictTables {
    const uint64_t *T; uint32_t initial; int initial_final;   // Mealy table; whether the initial state is final (empty normalised key)
    const uint16_t *cls_l1; const uint32_t *cls_pages;         // code point -> class of the dictionary alphabet (bf_model.h dict_clsmap), 0xFFFFF = none
    const uint16_t *nrm_l1; const uint32_t *nrm_pages; const int32_t *nrm_pool;   // [pos-dict] charmap (bf_model.h dict_charmap), nullptr = none
    const int32_t *k2i; int k2i_n;
    int r2l;                                                    // PARAM_DIRECTION != l2r
    int ignore_case;                                            // the dictionary folds its keys (then the charmap is applied in place: a buffer of 300)
};
constexpr int DICT_MAX_WORD = 300, DICT_NORM_BUF = 600;
constexpr uint32_t DICT_CLS_NONE = 0xFFFFFu, DICT_NORM_NONE = 0xFFFFFFFFu;

BF_HD uint32_t dict_map_get(const uint16_t *l1, const uint32_t *pages, int cp, uint32_t def)
{
    if ((unsigned)cp > 0x10FFFFu) return def;
    return pages[(uint32_t)l1[cp >> 8] * 256u + (uint32_t)(cp & 255)];
}

struct DictWalk {
    uint32_t state; int sum; bool ok, fin;
    BF_HD void start(const DictTables &D) { state = D.initial; sum = 0; ok = true; fin = D.initial_final != 0; }
    BF_HD void feed(const DictTables &D, int sym)
    {
        if (!ok) return;
        const uint32_t c = dict_map_get(D.cls_l1, D.cls_pages, sym, DICT_CLS_NONE);
        if (c == DICT_CLS_NONE) { ok = false; return; }
        const uint64_t e = D.T[state + c];
        if ((e & SG_CLS_MASK) != c) { ok = false; return; }
        state = (uint32_t)((e >> SG_NEXT_SHIFT) & SG_NEXT_MASK);
        sum += (int)(e >> SG_OW_SHIFT);
        fin = (e & SG_FINAL) != 0;
    }
    BF_HD int id(const DictTables &D) const { return (ok && fin && sum >= 0 && sum < D.k2i_n) ? D.k2i[sum] : -1; }
};

BF_HD int dict_info_id(const DictTables &D, const int32_t *key, int n)
{
    if (n <= 0 || n > DICT_MAX_WORD) return -1;
    DictWalk w; w.start(D);
    if (!D.r2l && !D.nrm_l1) { for (int i = 0; i < n; ++i) w.feed(D, key[i]); return w.id(D); }      // m_NoNorm (FADictInterpreter_t.h:203-205): the key as it is
    // normalised keys (r2l dictionaries, ignore-case ones in either direction; the map holds fold + charmap): length of the normalised word
    // first (FANormalizeWord gives up beyond the buffer), then its symbols -- first to last, or last to first for r2l
    int len = n;
    if (D.nrm_l1) {
        len = 0;
        for (int i = 0; i < n; ++i) {
            const uint32_t v = dict_map_get(D.nrm_l1, D.nrm_pages, key[i], DICT_NORM_NONE);
            const int c = v == DICT_NORM_NONE ? 1 : (int)(v >> 24) == 11 ? 1 : (int)(v >> 24);
            len += c;
        }
        // (an ignore-case dictionary normalises IN PLACE, through a buffer of MaxWordLen = 300: FADictInterpreter_t.h:231-247, FANormalizeWord with
        // pIn == pOut; the r2l-only way has the 600 of its own buffer)
        if (len > (D.ignore_case ? DICT_MAX_WORD : DICT_NORM_BUF)) return w.id(D);              // counts as the empty word
    }
    for (int t = 0; t < n; ++t) {
        const int i = D.r2l ? n - 1 - t : t;
        const uint32_t v = D.nrm_l1 ? dict_map_get(D.nrm_l1, D.nrm_pages, key[i], DICT_NORM_NONE) : DICT_NORM_NONE;
        if (v == DICT_NORM_NONE) { w.feed(D, key[i]); continue; }
        const int c = (int)(v >> 24); const uint32_t pay = v & 0xFFFFFFu;
        if (c == 1) w.feed(D, (int)pay);
        else if (c == 11) w.feed(D, D.nrm_pool[pay]);
        else for (int q = 0; q < c; ++q) w.feed(D, D.nrm_pool[pay + (uint32_t)(D.r2l ? c - 1 - q : q)]);
    }
    return w.id(D);
}

} // namespace bfa
