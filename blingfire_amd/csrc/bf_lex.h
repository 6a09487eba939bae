// bf_lex.h -- per-document WordPiece lexer program (one document per lane).
//
// Reproduces, on the re-laid-out tables of bf_model.h, the reference
//   FALexTools_t<int>::Process_int          (blingfireclient.library/inc/FALexTools_t.h:205-400)
//   + the TextToIdsWithOffsets_wp post-pass (blingfiretools/blingfiretokdll/blingfiretokdll.cpp:1207-1313)
// fused: <tag,from,to> triples are consumed by the post-pass as they are produced,
// so no triple buffer exists.  Recursion (_call functions) is an explicit frame stack.
//
// The code is written once as BF_HD functions: the HIP kernel (bf_kernels.hip) runs it per
// lane; tests/hosttest compiles the same header for the host to fuzz it against the oracle
// without a GPU (test-only: the product library never executes it on the CPU).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BF_HD __host__ __device__ __forceinline__
#else
#define BF_HD inline
#endif

namespace bfa {

constexpr int LEX_MAX_DEPTH = 4;          // frames; LoadModel refuses lexers with a deeper max-depth
constexpr uint32_t LX_CLS_NONE = 0xFFFFu; // class-stream value: code point not in the alphabet
constexpr uint32_t LX_T_CLS_MASK = 0x1FFFu, LX_T_FINAL = 1u << 13;
constexpr int LX_T_NEXT_SHIFT = 14;
constexpr uint32_t LX_INFO_SIMPLE = 0x80000000u;
constexpr int WBD_WORD_TAG = 1, WBD_IGNORE_TAG = 4;   // reference tokdll:39-40

struct LexTables {
    const uint32_t *T;        // displacement-packed transitions (bf_model.h T32 entry)
    const uint32_t *info;     // action info of final states, indexed by state base
    const int32_t *acts;      // general action records [left,right,tag,nfn,(fn,ini)*]
    uint32_t initial;
    uint32_t cls_any, cls_l, cls_r;   // LX_CLS_NONE when the symbol is not in the alphabet
    int max_depth, max_token_length;
};

// one DFA transition: returns the table entry or ~0u on a miss
BF_HD uint32_t lx_lookup(const LexTables &L, uint32_t state, uint32_t cls)
{
    if (cls >= LX_T_CLS_MASK) return 0xFFFFFFFFu;
    const uint32_t e = L.T[state + cls];
    return (e & LX_T_CLS_MASK) == cls ? e : 0xFFFFFFFFu;
}
// GetDest(State, Iw) with the IW_ANY retry of FALexTools_t.h:265-270
BF_HD uint32_t lx_dest(const LexTables &L, uint32_t state, uint32_t cls)
{
    uint32_t e = lx_lookup(L, state, cls);
    if (e == 0xFFFFFFFFu && L.cls_any != LX_CLS_NONE) e = lx_lookup(L, state, L.cls_any);
    return e;
}

// streaming form of the _wp post-pass (tokdll:1210-1311)
struct WpSink {
    int32_t *ids; int max_ids; int unk;
    int out_count; int scanning; int tok_to, expected, nsub, word_out;
    BF_HD void init(int32_t *ids_, int max_ids_, int unk_)
    { ids = ids_; max_ids = max_ids_; unk = unk_; out_count = 0; scanning = 0; tok_to = expected = nsub = word_out = 0; }
    BF_HD void finalize_word()
    {
        if (nsub > 0 && expected - 1 == tok_to) {       // sub-tokens tile the word exactly
            const int c = word_out + nsub;
            out_count = c < max_ids ? c : max_ids;
        } else if (word_out < max_ids) {                 // otherwise one UNK (tokdll:1282-1301)
            ids[word_out] = unk; out_count = word_out + 1;
        }
        scanning = 0;
    }
    // returns false once the id array is full (tokdll:1308-1310): nothing can change afterwards
    BF_HD bool push(int tag, int from, int to)
    {
        if (scanning) {
            if (tag > WBD_IGNORE_TAG && expected == from) {   // tokdll:1239
                const int k = word_out + nsub;
                if (k < max_ids) ids[k] = tag;
                nsub++; expected = to + 1;
                return true;
            }
            finalize_word();
            if (out_count >= max_ids) return false;
        }
        if (tag == WBD_WORD_TAG) { scanning = 1; tok_to = to; expected = from; nsub = 0; word_out = out_count; }
        return out_count < max_ids || scanning;   // a pending word at out_count == max_ids cannot happen (checked above)
    }
    BF_HD int finish() { if (scanning) finalize_word(); return out_count; }
};

struct LexFrame {      // caller state saved across a _call (FALexTools_t.h:350-382)
    uint32_t ini; int off, n, from, once, a_idx, a_end, to2, fn_once, fp_r, fn_from, emit_mark;
};

// Runs the lexer + post-pass over one document's class stream cls_at(0..n-1).
// Returns the number of ids written (<= max_ids); ids beyond it are untouched.
template <class ClsAt>
BF_HD int lex_doc(const LexTables &L, ClsAt cls_at, int n, int32_t *ids, int max_ids, int unk)
{
    WpSink sink; sink.init(ids, max_ids, unk);
    if (n <= 0 || L.max_depth < 1) return 0;
    const int max_triples = 2 * n;            // WbdRes holds 6*BuffSize ints = 2*BuffSize triples (tokdll:1194)
    int emitted = 0, last_to = 0;
    LexFrame st[LEX_MAX_DEPTH - 1];
    int d = 0;                                // RecDepth - 1
    // current frame
    uint32_t ini = L.initial; int off = 0, fn_ = n, from = -1, once = 0;
    // continuation of the action being executed in the current frame
    int a_idx = 0, a_end = 0, to2 = 0, fn_once = 0, fp_r = 0, fn_from = 0;
    for (;;) {
        if (from >= fn_) {
            // ---- Process_int returns (FALexTools_t.h:399); resume the caller's function loop
            if (d == 0) break;
            --d;
            const LexFrame &f = st[d];
            ini = f.ini; off = f.off; fn_ = f.n; from = f.from; once = f.once;
            a_idx = f.a_idx + 2; a_end = f.a_end; to2 = f.to2; fn_once = f.fn_once; fp_r = f.fp_r; fn_from = f.fn_from;
            if (emitted > f.emit_mark) {                      // FnOutSize > 0 (FALexTools_t.h:372-381)
                fn_from = last_to + 1 - off;
                if (fn_from > to2) a_idx = a_end;
            }
        } else {
            // ---- one start position (FALexTools_t.h:229-290)
            uint32_t state = ini, fs = 0; int fp = -1;
            int j = from;
            int bound = from + L.max_token_length; if (fn_ < bound) bound = fn_;
            if (j == -1) {
                const uint32_t e = lx_dest(L, ini, L.cls_l);
                if (e == 0xFFFFFFFFu) { ++from; continue; }
                state = e >> LX_T_NEXT_SHIFT; j = 0;
            }
            for (; j < bound; ++j) {
                const uint32_t e = lx_dest(L, state, cls_at(off + j));
                if (e == 0xFFFFFFFFu) break;
                state = e >> LX_T_NEXT_SHIFT;
                if (e & LX_T_FINAL) { fs = state; fp = j; }
            }
            if (j == fn_) {
                const uint32_t e = lx_dest(L, state, L.cls_r);
                if (e != 0xFFFFFFFFu && (e & LX_T_FINAL)) { fs = e >> LX_T_NEXT_SHIFT; fp = j; }
            }
            if (fp == -1) { ++from; continue; }
            // ---- a match (FALexTools_t.h:293-342)
            const uint32_t inf = L.info[fs];
            int left = 0, right = 0, tag;
            if (inf & LX_INFO_SIMPLE) { tag = (int)(inf & 0x7FFFFFFFu); a_idx = a_end = 0; fn_once = 0; }
            else {
                const int32_t *a = L.acts + inf;
                left = a[0]; right = a[1]; tag = a[2];
                a_idx = (int)inf + 4; a_end = a_idx + 2 * a[3]; fn_once = a[3] > 1;
            }
            int from2 = from + left; if (from2 < 0) from2 = 0; else if (fn_ <= from2) from2 = fn_ - 1;
            to2 = fp - right; if (to2 < 0) to2 = 0; else if (fn_ <= to2) to2 = fn_ - 1;
            fp_r = fp - right;
            if (tag != 0) {
                if (emitted >= max_triples) break;            // output buffer full (FALexTools_t.h:337-340): nothing more can be added
                ++emitted; last_to = to2 + off;
                if (!sink.push(tag, from2 + off, to2 + off)) break;   // id array full (tokdll:1308-1310)
            }
            fn_from = from2;                                  // FALexTools_t.h:347
        }
        // ---- (rest of) the action's function list (FALexTools_t.h:350-382)
        if (a_idx < a_end) {
            if (L.max_depth < d + 2) a_idx = a_end;           // callee returns 0 at once (FALexTools_t.h:222-224)
            else {
                LexFrame &f = st[d];
                f.ini = ini; f.off = off; f.n = fn_; f.from = from; f.once = once;
                f.a_idx = a_idx; f.a_end = a_end; f.to2 = to2; f.fn_once = fn_once; f.fp_r = fp_r; f.fn_from = fn_from; f.emit_mark = emitted;
                const int fn = L.acts[a_idx];
                ini = (uint32_t)L.acts[a_idx + 1];
                off = fn_from + off; fn_ = to2 - fn_from + 1; from = -1; once = (fn == 0) ? 0 : fn_once;
                ++d;
                continue;
            }
        }
        if (once) { from = fn_; continue; }                   // "called once": return (FALexTools_t.h:385-387)
        if (fp_r > from) from = fp_r;                         // FALexTools_t.h:390-393
        ++from;
    }
    return sink.finish();
}

} // namespace bfa
