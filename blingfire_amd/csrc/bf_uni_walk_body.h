// bf_uni_walk_body.h -- Unigram-LM, the part of FATokenSegmentationTools_1best_t<int>::Process that does not depend on the scores:
// which dictionary entries start at which position (cl/inc/FATokenSegmentationTools_1best_t.h:195-235, the walk of :203-224 for every
// start).  One wave takes a document and walks 64 START POSITIONS AT A TIME, one per lane: 64 independent table gathers per
// instruction, no per-lane state beyond the walk itself, so the occupancy that hides the gather latency is not bought with LDS (the
// lane-per-document program keeps a window of scores per lane in LDS: 13 waves per CU, waves 73 % parked on memory, profiles/r02_*).
// What the walks find leaves as ARC RECORDS in the order the reference adds its arcs -- by start, then by end -- two words each:
// [id + 1 | length - 1 | last-of-its-start] and the score bits (bf_seg.h UniArcLane, which runs the relaxations, reads them: one
// document per lane, no table gather left).  A start without any entry leaves the "unknown" record (AddUnknownArc, :145-171).
//
// Records of a round (64 starts) are contiguous; rounds are found through a table (a wave takes its space from the pool in large
// pieces with one atomic each).  A document whose records do not fit the pool is flagged: the lane-per-document program redoes it.
// Include AFTER a definition of namespace wv (bf_kernels.hip on the device, tests/hosttest/wave_emu.h in the test simulator).
#pragma once
#include "bf_wave.h"
#include "bf_seg.h"

namespace bfa {

struct UniWalkParams {
    const uint64_t *T; const SegInfo *info; uint32_t initial;
    const uint16_t *stream; const int32_t *lens; const int64_t *doc_off; int slot_mul; int64_t ndocs;
    const int32_t *perm;             // optional: the order in which the documents are taken (longest first)
    uint64_t *pool; unsigned long long pool_recs; unsigned long long *pool_cursor;    // arc records
    uint64_t *rounds;                // rounds[uw_round_base(slot, d) + r] = index of the first record of the document's round r
    int32_t *flags;                  // [ndocs] 1: the document's records did not fit the pool
    unsigned long long *next_doc;
    unsigned long long *stats;       // optional: [0] rounds, [1] walk steps (wave instructions), [2] records, [3] transitions made, [4] pieces of pool taken
};

BF_WV int64_t uw_round_base(int64_t slot, int64_t d) { return (slot >> 6) + d; }
constexpr unsigned long long UW_PIECE = 8192;          // records a wave takes from the pool at a time (64 KB)

template <int ROWS> struct UwLds { uint32_t stage[ROWS * 64]; };      // what a lane's walk found: [MPH index : 22 | length - 1 : 5], row k = its k-th entry

template <class LDS, int ROWS>
struct UniWalk {
    const UniWalkParams &p; LDS &S; int lane;
    unsigned long long piece, piece_left;               // the wave's piece of the pool: next free record, records left

    BF_WVD UniWalk(const UniWalkParams &p_, LDS &S_) : p(p_), S(S_) { lane = wv::lane(); piece = 0; piece_left = 0; }

    // false: the pool is exhausted
    BF_WVD bool take(unsigned long long n)
    {
        if (n <= piece_left) return true;
        const unsigned long long want = n > UW_PIECE ? n : UW_PIECE;
        unsigned long long at = 0;
        if (lane == 0) at = wv::atomic_add(p.pool_cursor, want);
        at = wv::bcast(at, 0);
        if (at + want > p.pool_recs) return false;
        piece = at; piece_left = want;
        if (p.stats && lane == 0) wv::atomic_add(&p.stats[4], 1ull);
        return true;
    }

    BF_WVD void run_doc(int64_t d)
    {
        const int L = p.lens[d];
        if (lane == 0) p.flags[d] = 0;
        if (L <= 0) return;
        const int64_t slot = (int64_t)p.slot_mul * (p.doc_off[d] + d);
        const uint16_t *src = p.stream + slot;
        uint64_t *rt = p.rounds + uw_round_base(slot, d);
        for (int w0 = 0, r = 0; w0 < L; w0 += 64, ++r) {
            const int s = w0 + lane; const bool valid = s < L;
            // ---- the walks (:203-224): every lane from its own start, side by side
            uint32_t state = p.initial; int sum = 0, n = 0, j = 0; bool act = valid;
            unsigned long long steps = 0, trans = 0;
            while (wv::any(act)) {
                const uint32_t c = act ? (uint32_t)src[s + j] : 0u;
                const bool ok = act && c < SG_CLS_DELIM_ABSENT;                 // symbols outside the alphabet never match
                const uint64_t e = p.T[ok ? state + c : 0u];
                const bool hit = ok && (e & SG_CLS_MASK) == c;
                if (hit) { state = (uint32_t)((e >> SG_NEXT_SHIFT) & SG_NEXT_MASK); sum += (int)(e >> SG_OW_SHIFT); }
                const bool fin = hit && (e & SG_FINAL) != 0 && n < ROWS;
                if (fin) { S.stage[n * 64 + lane] = (uint32_t)sum | ((uint32_t)j << 22); ++n; }
                if (p.stats) { ++steps; trans += (unsigned long long)__builtin_popcountll(wv::ballot(hit)); }
                act = hit && s + j + 1 < L;
                j = act ? j + 1 : j;
            }
            // ---- records: a start's entries in order of their end, the starts in order; a start without an entry leaves the unknown record
            const int cnt = valid ? (n > 0 ? n : 1) : 0;
            const int inc = wv::incl_scan(cnt);
            const int total = wv::bcast(inc, 63);
            if (!take((unsigned long long)total)) { if (lane == 0) p.flags[d] = 1; return; }
            if (lane == 0) rt[r] = piece;
            uint64_t *out = p.pool + piece + (unsigned long long)(inc - cnt);
            for (int k = 0; wv::any(k < n); ++k) {
                if (k < n) {
                    const uint32_t v = S.stage[k * 64 + lane];
                    const SegInfo q = p.info[v & 0x3FFFFFu];
                    const uint32_t w0r = ((uint32_t)(q.id + 1) & 0xFFFFFu) | ((v >> 22) << 20) | (k == n - 1 ? UA_LAST : 0u);
                    out[k] = (uint64_t)w0r | ((uint64_t)q.score_bits << 32);
                }
            }
            if (valid && n == 0) out[0] = (uint64_t)(UA_UNK | UA_LAST);
            piece += (unsigned long long)total; piece_left -= (unsigned long long)total;
            if (p.stats && lane == 0) { wv::atomic_add(&p.stats[0], 1ull); wv::atomic_add(&p.stats[1], steps); wv::atomic_add(&p.stats[2], (unsigned long long)total); wv::atomic_add(&p.stats[3], trans); }
            wv::sync();                                                  // the stage is reused by the next round
        }
    }

    BF_WVD void run()
    {
        for (;;) {
            unsigned long long i = 0;
            if (lane == 0) i = wv::atomic_add(p.next_doc, 1ull);
            i = wv::bcast(i, 0);
            if ((int64_t)i >= p.ndocs) break;
            run_doc(p.perm ? (int64_t)p.perm[i] : (int64_t)i);
        }
    }
};

} // namespace bfa
