// bf_uni_walk_body.h -- Unigram-LM, the part of FATokenSegmentationTools_1best_t<int>::Process that does not depend on the scores:
// which dictionary entries start at which position (cl/inc/FATokenSegmentationTools_1best_t.h:195-235, the walk of :203-224 for every
// start).  One wave takes a document and walks 64 * NS START POSITIONS AT A TIME, NS per lane: NS independent table gathers in flight per
// lane and step, no per-lane state beyond the walks themselves, so the latency of a gather is covered inside the wave instead of being
// bought with resident waves (the lane-per-document program keeps a window of scores per lane in LDS: 13 waves per CU, waves 73 %
// parked on memory, profiles/r02_*).  The elements a walk reads come from a window of the class stream in LDS (one coalesced load per
// block of starts): a step is one LDS read and one gather.
// What the walks find leaves as ARC RECORDS in the order the reference adds its arcs -- by start, then by end -- two words each:
// [id + 1 | length - 1 | last-of-its-start] and the score bits (bf_seg.h UniArcLane, which runs the relaxations, reads them: one
// document per lane, no table gather left).  A start without any entry leaves the "unknown" record (AddUnknownArc, :145-171).
//
// Records of a round (64 consecutive starts) are contiguous; rounds are found through a table (a wave takes its space from the pool in
// large pieces with one atomic each).  A document whose records do not fit the pool -- or with a start that has more entries than the
// stage holds (ROWS) -- is flagged: the lane-per-document program redoes it.
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
    int32_t *flags;                  // [ndocs] 1: the document's records did not fit the pool / the stage
    unsigned long long *next_doc;
    unsigned long long *stats;       // optional: [0] rounds, [1] walk steps (wave instructions), [2] records, [3] transitions made, [4] pieces of pool taken,
                                     // [5] documents flagged for the stage, [8 + min(n, 7)] starts with n entries
};

BF_WV int64_t uw_round_base(int64_t slot, int64_t d) { return (slot >> 6) + d; }
constexpr unsigned long long UW_PIECE = 8192;          // records a wave takes from the pool at a time (64 KB)
constexpr int UW_AHEAD = 32;                           // elements behind the last start of a block a walk may read (>= the longest entry, bf_seg.h UA_MAX_DEPTH)

// what a lane's walks found: [MPH index : 22 | length - 1 : 5], row k of start-slot t = the k-th entry of that start; the block's window of the class stream
template <int ROWS, int NS> struct UwLds { uint32_t stage[NS * ROWS * 64]; uint16_t cls[64 * NS + UW_AHEAD]; };

template <class LDS, int ROWS, int NS>
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
        for (int w0 = 0; w0 < L; w0 += 64 * NS) {
            // ---- the block's elements: [w0, w0 + 64 * NS + UW_AHEAD), behind the document's end a symbol no entry has
            for (int q = lane; q < 64 * NS + UW_AHEAD; q += 64) S.cls[q] = w0 + q < L ? src[w0 + q] : (uint16_t)SG_CLS_NONE;
            wv::sync();
            // ---- the walks (:203-224): NS starts per lane, start-slot t = start w0 + 64 t + lane
            uint32_t state[NS]; int sum[NS], n[NS]; bool act[NS]; bool over = false;
#pragma unroll
            for (int t = 0; t < NS; ++t) { state[t] = p.initial; sum[t] = 0; n[t] = 0; act[t] = w0 + 64 * t + lane < L; }
            unsigned long long steps = 0, trans = 0;
            for (int j = 0;; ++j) {
                bool any_act = false;
#pragma unroll
                for (int t = 0; t < NS; ++t) any_act = any_act || act[t];
                if (!wv::any(any_act)) break;
                uint32_t c[NS]; uint64_t e[NS]; bool ok[NS];
#pragma unroll
                for (int t = 0; t < NS; ++t) {
                    c[t] = act[t] ? (uint32_t)S.cls[64 * t + lane + j] : (uint32_t)SG_CLS_NONE;
                    ok[t] = c[t] < SG_CLS_DELIM_ABSENT;                           // symbols outside the alphabet never match
                }
#pragma unroll
                for (int t = 0; t < NS; ++t) e[t] = p.T[ok[t] ? state[t] + c[t] : 0u];     // NS gathers in flight
#pragma unroll
                for (int t = 0; t < NS; ++t) {
                    const bool hit = ok[t] && (e[t] & SG_CLS_MASK) == c[t];
                    if (hit) { state[t] = (uint32_t)((e[t] >> SG_NEXT_SHIFT) & SG_NEXT_MASK); sum[t] += (int)(e[t] >> SG_OW_SHIFT); }
                    const bool fin = hit && (e[t] & SG_FINAL) != 0;
                    if (fin && n[t] < ROWS) { S.stage[(t * ROWS + n[t]) * 64 + lane] = (uint32_t)sum[t] | ((uint32_t)j << 22); ++n[t]; }
                    else if (fin) over = true;
                    if (p.stats) trans += (unsigned long long)__builtin_popcountll(wv::ballot(hit));
                    act[t] = hit && j + 1 < UW_AHEAD;
                }
                if (p.stats) ++steps;
            }
            if (wv::any(over)) {                                                // a start with more entries than the stage holds
                if (lane == 0) { p.flags[d] = 1; if (p.stats) wv::atomic_add(&p.stats[5], 1ull); }
                wv::sync();
                return;
            }
            // ---- records: a start's entries in order of their end, the starts in order; a start without an entry leaves the unknown record
            int cnt[NS], inc[NS]; int total = 0;
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                const bool valid = w0 + 64 * t + lane < L;
                cnt[t] = valid ? (n[t] > 0 ? n[t] : 1) : 0;
                inc[t] = wv::incl_scan(cnt[t]) + total;
                total = wv::bcast(inc[t], 63);
                if (p.stats && valid) wv::atomic_add(&p.stats[8 + (n[t] < 7 ? n[t] : 7)], 1ull);
            }
            if (!take((unsigned long long)total)) { if (lane == 0) p.flags[d] = 1; wv::sync(); return; }
#pragma unroll
            for (int t = 0; t < NS; ++t) {
                if (w0 + 64 * t < L) {                                          // (wave-uniform) round r = w0 / 64 + t exists
                    const int first = t == 0 ? 0 : wv::bcast(inc[t - 1], 63);
                    if (lane == 0) rt[w0 / 64 + t] = piece + (unsigned long long)first;
                    uint64_t *out = p.pool + piece + (unsigned long long)(inc[t] - cnt[t]);
                    // two entries per trip: both I2Info rows are requested before either record is stored
                    for (int k = 0; wv::any(k < n[t]); k += 2) {
                        const bool a = k < n[t], b = k + 1 < n[t];
                        const uint32_t va = a ? S.stage[(t * ROWS + k) * 64 + lane] : 0u, vb = b ? S.stage[(t * ROWS + k + 1) * 64 + lane] : 0u;
                        const SegInfo qa = p.info[va & 0x3FFFFFu], qb = p.info[vb & 0x3FFFFFu];
                        if (a) out[k] = (uint64_t)(((uint32_t)(qa.id + 1) & 0xFFFFFu) | ((va >> 22) << 20) | (k == n[t] - 1 ? UA_LAST : 0u)) | ((uint64_t)qa.score_bits << 32);
                        if (b) out[k + 1] = (uint64_t)(((uint32_t)(qb.id + 1) & 0xFFFFFu) | ((vb >> 22) << 20) | (k + 1 == n[t] - 1 ? UA_LAST : 0u)) | ((uint64_t)qb.score_bits << 32);
                    }
                    if (w0 + 64 * t + lane < L && n[t] == 0) out[0] = (uint64_t)(UA_UNK | UA_LAST);
                }
            }
            piece += (unsigned long long)total; piece_left -= (unsigned long long)total;
            if (p.stats && lane == 0) { wv::atomic_add(&p.stats[0], (unsigned long long)((L - w0 + 63) / 64 < NS ? (L - w0 + 63) / 64 : NS)); wv::atomic_add(&p.stats[1], steps); wv::atomic_add(&p.stats[2], (unsigned long long)total); wv::atomic_add(&p.stats[3], trans); }
            wv::sync();                                                  // the stage and the window are reused by the next block
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
