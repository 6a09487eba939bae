// bf_lex.h -- per-document WordPiece lexer program (one document per lane).
//
// Reproduces, on the re-laid-out tables of bf_model.h, the reference
//   FALexTools_t<int>::Process_int          (blingfireclient.library/inc/FALexTools_t.h:205-400)
//   + the TextToIdsWithOffsets_wp post-pass (blingfiretools/blingfiretokdll/blingfiretokdll.cpp:1207-1313)
// fused: <tag,from,to> triples are consumed by the post-pass as they are produced, so no triple
// buffer exists.  Recursion (_call functions) is an explicit frame stack.
//
// The program is a resumable state machine (LexLane): prepare() runs everything between two DFA
// walks (calls/returns, start-position advance), step() is exactly one DFA transition, after_walk()
// consumes the result of a walk.  A sequential driver (lex_doc) and the divergence-aware GPU driver
// (bf_kernels.hip: lanes vote on when to run the "event" code) call the same methods, so the order of
// operations per document is always the reference's.
//
// Written once as BF_HD code: the HIP kernels run it per lane; tests/hosttest compiles the same
// header for the host to fuzz it against the oracle without a GPU (test-only: the product library
// never executes it on the CPU).
#pragma once
#include <stdint.h>
#include "bf_layout.h"

#if defined(__HIPCC__)
#define BF_HD __host__ __device__ __forceinline__
#else
#define BF_HD inline
#endif

namespace bfa {

constexpr int LEX_MAX_DEPTH = 4;          // frames; LoadModel refuses lexers with a deeper max-depth
constexpr uint32_t LX_INFO_SIMPLE = 0x80000000u;
constexpr uint32_t LX_NO_STATE = 0xFFFFFFFFu;
constexpr int LX_ACT_FN_STRIDE = 3;
constexpr int WBD_WORD_TAG = 1, WBD_IGNORE_TAG = 4;   // reference tokdll:39-40

struct LexTables {
    const uint64_t *T;        // displacement-packed transitions: low word = bf_layout.h entry, high word = action info of
                              // the destination state when it is final (so a match needs no second gather; a 4-byte
                              // table + an info gather per match measured 5 % slower on MI355X)
    const int32_t *acts;      // general action records [left,right,tag,nfn,(fn,ini,ini_l)*]
    uint32_t initial;
    uint32_t initial_l;       // state after feeding the left anchor to `initial` (LX_NO_STATE: no such transition); the same
                              // is precomputed per callable function (ini_l above), so the anchor costs no gather
    uint32_t cls_any, cls_l, cls_r;   // LX_CLS_NONE when the symbol is not in the alphabet
    int max_depth, max_token_length;
    int max_frames;           // saved frames the call graph can need (= call depth - 1, computed at load; <= LEX_MAX_DEPTH - 1)
    // The model's "loop state" (LX_NO_STATE: none): a state that goes to ITSELF on every class flagged LX_C_LOOP in the class
    // stream -- for the BERT lexers the state of the top-level rule `(AllLetters)+` after its first letter.  Walking through
    // it is the same as skipping the run of flagged elements (fp / finfo follow the last one when the state is final), so
    // step() does that from the class window alone, without one table gather per character.
    uint32_t loop_state, loop_info; int loop_final;
    // "Two-level" lexer (load-time fact, bf_model.cpp): every action with functions is [left 0, right 0, tag != 0, ONE function]
    // and no rule of a called function calls anything -- the shape of every WordPiece model (top-level rules find the word and
    // call FnTokWord, whose rules are the vocabulary).  The call stack then never holds more than the top-level frame, which is a
    // constant: prepare2() / after_walk2() run the same algorithm without the frame stack and the general action decoding.
    int two_level;
    // No state reachable from a called function's initial states has a transition on the right anchor (load-time fact; true for the
    // WordPiece vocabularies, whose rules are `^piece` / `piece`): feeding the anchor at the end of a function's input can only miss
    // (FALexTools_t.h:280-290 would look it up and find nothing), so walks inside a function stop at the last letter instead.
    int fn_no_ra;
};

// where table entries come from: a policy, so that the host build can count lookups per table index
// (tools/lookup_profile.py).  An LDS-resident table prefix behind this policy was measured on MI355X (89 % of the
// lookups served from LDS, one 1024-thread block per CU) and was slower: see DESIGN.md section 5.
struct TabDirect {
    const uint64_t *T;
    BF_HD uint64_t operator()(uint32_t idx) const
    {
#ifdef BF_LEX_PROFILE_HOOK
        BF_LEX_PROFILE_HOOK(idx);                  // host-only instrumentation (tools/lookup_profile.py)
#endif
        return T[idx];
    }
};

// LDS-resident table (models whose transitions fit: wbd.bin is 3.4 K entries = 27 KB): a gather costs ~7 CU-cycles per
// wave instead of 60-100 from L2 (tools/microbench/gather.hip).  Entries at or past `n` are the empty probe tail.
struct TabLds {
    const uint64_t *lds; uint32_t n;
    BF_HD uint64_t operator()(uint32_t idx) const { return idx < n ? lds[idx] : (uint64_t)LX_T_CLS_MASK; }
};

// one DFA transition: the table entry + whether it is a hit.  Branch-free and unclamped: the table is padded by
// 0x2000 entries past the largest base (bf_model.cpp) and cls <= LX_CLS_NONE.
template <class Tab>
BF_HD uint64_t lx_lookup(const Tab &tab, uint32_t state, uint32_t cls, bool &hit)
{
    const uint64_t e = tab(state + cls);
    hit = ((uint32_t)e & LX_T_CLS_MASK) == cls;
    return e;
}
// GetDest(State, Iw) with the IW_ANY retry of FALexTools_t.h:265-270.  HAS_ANY is a compile-time fact of the
// model (IW_ANY is in the alphabet or not) so that models without it pay nothing for the retry.
template <bool HAS_ANY, class Tab>
BF_HD uint64_t lx_dest(const Tab &tab, uint32_t cls_any, uint32_t state, uint32_t cls, bool &hit)
{
    uint64_t e = lx_lookup(tab, state, cls, hit);
    if (HAS_ANY) { if (!hit) e = lx_lookup(tab, state, cls_any, hit); }
    return e;
}

// direct id output (host emulation): ids[k] = v
struct IdOutDirect {
    int32_t *ids;
    int32_t *spans;    // optional [2*k], [2*k+1]: first / last stream position of id k (TextToIdsWithOffsets)
    BF_HD void put(int k, int32_t v) { ids[k] = v; }
    BF_HD void span(int k, int from, int to) { if (spans) { spans[2 * k] = from; spans[2 * k + 1] = to; } }
    BF_HD void finish(int) {}
};

struct LexFrame {      // caller state saved across a _call (FALexTools_t.h:350-382)
    uint32_t ini; int off, n, from, once, a_idx, a_end, to2, fn_once, fp_r, fn_from, emit_mark;
};
constexpr int LEX_FRAME_WORDS = 12;

struct FramesArray {   // host emulation only (a dynamically indexed private struct array mis-executed on the device)
    LexFrame st[LEX_MAX_DEPTH - 1];
    BF_HD void save(int d, const LexFrame &f) { st[d] = f; }
    BF_HD void load(int d, LexFrame &f) const { f = st[d]; }
};

// SINGLE: the lane runs ONE top-level start position `p0` (set after init(): from = p0 = the position) and everything that position
// causes (its walk, its action, the functions the action calls), then prepare() returns false with `from` = the next top-level
// start position (FALexTools_t.h:229-252, 385-393).  What a top-level start position does depends on nothing before it -- the
// loop of Process_int carries only FromPos -- which is what the long-document path of the words modes (lex_long_* below) builds on.
template <class ClsAt, class IdOut, class Frames, bool HAS_ANY, class Tab = TabDirect, bool SINGLE = false>
struct LexLane {
    const LexTables &L; ClsAt &cls_at; IdOut &ids; Frames &frames; Tab tab;
    int p0 = 0;                                                   // SINGLE: the start position this lane runs
    // ---- streaming _wp post-pass (tokdll:1210-1311)
    int max_ids, unk;
    int out_count, scanning, tok_from, tok_to, expected, nsub, word_out;
    // ---- lexer
    int max_triples, emitted, last_to, d, n_doc;
    uint32_t ini, ini_l; int off, fn_, from, once;                // current frame (ini_l is only needed by its first walk)
    int a_idx, a_end, to2, fn_once, fp_r, fn_from;                // action being executed in it
    uint32_t state, finfo; int j, lim, fp;                        // current walk (finfo: action info of the deepest final state;
                                                                  // lim: the walk goes on while the next position is < lim)
    bool stop;                                                    // nothing can change any more
    int words;                                                    // 0: ids (_wp post-pass); 1: TextToWords (raw non-IGNORE <tag,from,to> tokens); 2: TextToSentences (all of them)

    BF_HD LexLane(const LexTables &L_, ClsAt &c, IdOut &o, Frames &f) : L(L_), cls_at(c), ids(o), frames(f), tab{L_.T} {}
    BF_HD LexLane(const LexTables &L_, ClsAt &c, IdOut &o, Frames &f, const Tab &t) : L(L_), cls_at(c), ids(o), frames(f), tab(t) {}

    BF_HD void sink_finalize_word()
    {
        if (nsub > 0 && expected - 1 == tok_to) {       // sub-tokens tile the word exactly (tokdll:1252)
            const int c = word_out + nsub;
            out_count = c < max_ids ? c : max_ids;
        } else if (word_out < max_ids) {                 // otherwise one UNK (tokdll:1282-1301)
            ids.put(word_out, unk); ids.span(word_out, tok_from, tok_to);      // offsets of the whole word (tokdll:1289-1297)
            out_count = word_out + 1;
        }
        scanning = 0;
    }
    // returns false once the id array is full (tokdll:1308-1310): nothing can change afterwards
    BF_HD bool sink_push(int tag, int from_, int to_)
    {
        if (scanning) {
            if (tag > WBD_IGNORE_TAG && expected == from_) {   // tokdll:1239
                const int k = word_out + nsub;
                if (k < max_ids) { ids.put(k, tag); ids.span(k, from_, to_); }
                nsub++; expected = to_ + 1;
                return true;
            }
            sink_finalize_word();
            if (out_count >= max_ids) return false;
        }
        if (tag == WBD_WORD_TAG) { scanning = 1; tok_from = from_; tok_to = to_; expected = from_; nsub = 0; word_out = out_count; }
        return out_count < max_ids || scanning;
    }

    // Start a document of n normalised characters.  Follow with prepare().
    BF_HD void init(int n, int max_ids_, int unk_, int words_ = 0)
    {
        max_ids = max_ids_; unk = unk_; words = words_;
        out_count = 0; scanning = 0; tok_from = tok_to = expected = nsub = word_out = 0;
        // WbdRes holds 6*BuffSize ints = 2*BuffSize triples for TextToIds (tokdll:1194), 3*BuffSize ints for TextToWords (tokdll:494-499)
        max_triples = words_ ? n : 2 * n; n_doc = n;
        emitted = 0; last_to = 0; d = 0;
        ini = L.initial; ini_l = L.initial_l; off = 0; fn_ = n; from = -1; once = 0;
        a_idx = a_end = 0; to2 = 0; fn_once = 0; fp_r = 0; fn_from = 0;
        state = finfo = 0; j = 0; lim = 0; fp = -1;
        stop = (n <= 0 || L.max_depth < 1);
    }

    // Everything between two walks.  Returns true when a walk is set up (call step() until it returns
    // false, then after_walk()), false when the document is finished (call finish()).
    BF_HD bool prepare()
    {
        if (stop) return false;
        for (;;) {
            if (SINGLE) { if (d == 0 && from != p0) return false; }
            if (from >= fn_) {
                // ---- Process_int returns (FALexTools_t.h:399); resume the caller's function loop
                if (d == 0) return false;
                --d;
                LexFrame f; frames.load(d, f);
                ini = f.ini; off = f.off; fn_ = f.n; from = f.from; once = f.once;
                a_idx = f.a_idx + LX_ACT_FN_STRIDE; a_end = f.a_end; to2 = f.to2; fn_once = f.fn_once; fp_r = f.fp_r; fn_from = f.fn_from;
                if (emitted > f.emit_mark) {                      // FnOutSize > 0 (FALexTools_t.h:372-381)
                    fn_from = last_to + 1 - off;
                    if (fn_from > to2) a_idx = a_end;
                }
                after_action();
                continue;
            }
            // ---- set up one start position (FALexTools_t.h:229-252)
            state = ini; fp = -1; finfo = 0; j = from;
            set_lim(from);
            if (j < 0) {
                // the left anchor (from == -1: the first walk of a frame, FALexTools_t.h:244-252) is taken here, so that
                // step() only ever sees letters and the right anchor; its destination is a fact of the frame's initial
                // state, resolved at LoadModel (IW_ANY retry included); no finality check after it
                if (ini_l == LX_NO_STATE || !(0 < lim)) { ++from; continue; }         // the walk ended without a match
                state = ini_l; j = 0;
                cls_at.prefetch(off);                                 // start the class-window refill for the first letter now
                return true;
            }
            if (!(j < lim)) { ++from; continue; }                 // MaxTokenLength == 0: no letters, j != InSize
            cls_at.prefetch(off + j);
            return true;
        }
    }

    // The walk started at `f` reads letters while position < min(InSize, f + MaxTokenLength) and feeds the right
    // anchor only when the input was exhausted (position == InSize): one bound `lim` for "next position < lim".
    BF_HD void set_lim(int f)
    {
        const int b = f + L.max_token_length;
        const int ra = (d > 0 && L.fn_no_ra) ? 0 : 1;          // inside a function whose rules never use the right anchor: no anchor step
        lim = b < fn_ ? b : fn_ + ra;
    }

    // Exactly one DFA transition on a letter or on the right anchor.  Returns true while the walk continues.
    // Written with selects instead of branches: on the GPU this is the hot loop body (VALU-issue bound) and every
    // divergent branch costs scalar instructions for the whole wave.  Reads position off + InSize under the right
    // anchor (value unused): class streams carry at least one element of padding.
    // Optional accelerator, valid at any point of a walk: while the walk sits in the model's loop state, every element flagged
    // LX_C_LOOP keeps it there (k transitions of FALexTools_t.h:255-277 at once; a final loop state is the deepest final state
    // seen after each of them), so the run is skipped from the class window alone.  Returns false if the walk ended that way.
    // step() starts with it.
    BF_HD bool ff()
    {
        const int hi = lim < fn_ ? lim : fn_;          // letters are read at positions < hi
        if (state == L.loop_state && j < hi) {
            int k = cls_at.run(off + j);
            k = k < hi - j ? k : hi - j;
            if (k > 0) {
                j += k;
                if (L.loop_final) { fp = j - 1; finfo = L.loop_info; }
                if (!(j < lim)) return false;
            }
        }
        return true;
    }

    BF_HD bool step()
    {
        // measured on MI355X (1.25 M documents, bert_base_tok.bin): fast-forward inside every step 6.5 ms, none 6.74, once per vote 7.01
        if (!ff()) return false;
        const bool ra = j >= fn_;                      // feeding the right anchor (FALexTools_t.h:280-290)
        uint32_t c = cls_at(off + j) & LX_T_CLS_MASK;  // a letter (FALexTools_t.h:255-277)
        c = ra ? L.cls_r : c;
        uint64_t e64 = tab(state + c);                 // the gather is issued ...
        cls_at.prefetch(off + j + 1);                  // ... and the refill of the class window for the next letter (when it
                                                       // crosses a block) travels with it instead of in front of the next gather
        bool hit = ((uint32_t)e64 & LX_T_CLS_MASK) == c;
        if (HAS_ANY) { if (!hit) e64 = lx_lookup(tab, state, L.cls_any, hit); }       // IW_ANY retry (FALexTools_t.h:265-270)
        const uint32_t e = (uint32_t)e64;
        const bool fin = hit && (int32_t)e < 0;
        fp = fin ? j : fp;
        finfo = fin ? (uint32_t)(e64 >> 32) : finfo;
        const bool adv = hit && !ra;
        state = adv ? ((e >> LX_T_NEXT_SHIFT) & LX_T_NEXT_MASK) : state;
        const int jn = j + 1;
        j = adv ? jn : j;
        return adv && jn < lim;
    }

    // step() that also absorbs the cheapest event: a walk that ended WITHOUT a match simply restarts at the next
    // start position (what after_walk() + prepare() would do: FALexTools_t.h:229, 293), as long as the frame has
    // input left.  Lanes then stay in the walk loop instead of waiting for the next event vote.
    BF_HD bool step_r()
    {
        bool cont = step();
        const int nf = from + 1;
        if (!cont && fp == -1 && nf < fn_ && L.max_token_length > 0 && !(SINGLE && d == 0)) {
            from = nf; state = ini; finfo = 0; j = nf;
            set_lim(nf);
            cont = true;
        }
        return cont;
    }

    // Consume the result (fp, finfo) of the walk that just ended.  Follow with prepare().
    BF_HD void after_walk()
    {
        if (fp == -1) { ++from; return; }
        // ---- a match (FALexTools_t.h:293-342)
        const uint32_t inf = finfo;
        int left = 0, right = 0, tag;
        if (inf & LX_INFO_SIMPLE) { tag = (int)(inf & 0x7FFFFFFFu); a_idx = a_end = 0; fn_once = 0; }
        else {
            const int32_t *a = L.acts + inf;
            left = a[0]; right = a[1]; tag = a[2];
            a_idx = (int)inf + 4; a_end = a_idx + LX_ACT_FN_STRIDE * a[3]; fn_once = a[3] > 1;
        }
        int from2 = from + left; if (from2 < 0) from2 = 0; else if (fn_ <= from2) from2 = fn_ - 1;
        to2 = fp - right; if (to2 < 0) to2 = 0; else if (fn_ <= to2) to2 = fn_ - 1;
        fp_r = fp - right;
        if (tag != 0) {
            if (emitted >= max_triples) { stop = true; return; }      // output buffer full (FALexTools_t.h:337-340)
            ++emitted; last_to = to2 + off;
            if (words) {                                              // every non-IGNORE token is a word (tokdll:511-517)
                if ((words == 2 || tag != WBD_IGNORE_TAG) && out_count < max_ids) { ids.put(out_count, tag); ids.span(out_count, from2 + off, to2 + off); ++out_count; }
            } else if (!sink_push(tag, from2 + off, to2 + off)) { stop = true; return; }   // id array full (tokdll:1308-1310)
        }
        fn_from = from2;                                              // FALexTools_t.h:347
        after_action();
    }

    // (Rest of) the action's function list, then the start-position update (FALexTools_t.h:350-393).
    BF_HD void after_action()
    {
        if (a_idx < a_end) {
            if (L.max_depth < d + 2 || d + 1 > L.max_frames) a_idx = a_end;   // callee returns 0 at once (FALexTools_t.h:222-224)
            else {
                LexFrame f;
                f.ini = ini; f.off = off; f.n = fn_; f.from = from; f.once = once;
                f.a_idx = a_idx; f.a_end = a_end; f.to2 = to2; f.fn_once = fn_once; f.fp_r = fp_r; f.fn_from = fn_from; f.emit_mark = emitted;
                frames.save(d, f);
                const int fn = L.acts[a_idx];
                ini = (uint32_t)L.acts[a_idx + 1]; ini_l = (uint32_t)L.acts[a_idx + 2];
                off = fn_from + off; fn_ = to2 - fn_from + 1; from = -1; once = (fn == 0) ? 0 : fn_once;
                ++d;
                return;
            }
        }
        if (once) { from = fn_; return; }                             // "called once": return (FALexTools_t.h:385-387)
        if (fp_r > from) from = fp_r;                                 // FALexTools_t.h:390-393
        ++from;
    }

    // ---- two-level form (L.two_level): the same operations as prepare() / after_walk() / after_action() for the only two
    //      situations that can occur -- the top-level frame (d == 0) and one function frame on top of it (d == 1).
    //      `to2` doubles as the saved top-level resume position while the function runs.
    BF_HD bool prepare2()
    {
        if (stop) return false;
        for (;;) {
            if (from >= fn_) {
                if (d == 0) return false;                          // Process_int returns (FALexTools_t.h:399)
                // the function returns into the top-level frame; its action has no further function and left = right = 0:
                // after_action() there is "from = max(fp, from) + 1" (FALexTools_t.h:390-393), precomputed at the call
                d = 0; ini = L.initial; off = 0; fn_ = n_doc; from = to2;
                continue;
            }
            state = ini; fp = -1; finfo = 0; j = from;
            set_lim(from);
            if (j < 0) {
                if (ini_l == LX_NO_STATE || !(0 < lim)) { ++from; continue; }
                state = ini_l; j = 0;
                cls_at.prefetch(off);
                return true;
            }
            if (!(j < lim)) { ++from; continue; }
            cls_at.prefetch(off + j);
            return true;
        }
    }
    BF_HD void after_walk2()
    {
        if (fp == -1) { ++from; return; }
        const uint32_t inf = finfo;
        const bool call = !(inf & LX_INFO_SIMPLE);
        int tag; uint32_t f_ini = 0, f_ini_l = 0;
        if (!call) tag = (int)(inf & 0x7FFFFFFFu);
        else { const int32_t *a = L.acts + inf; tag = a[2]; f_ini = (uint32_t)a[5]; f_ini_l = (uint32_t)a[6]; }
        int from2 = from; if (from2 < 0) from2 = 0;                  // left = right = 0: the clamps of FALexTools_t.h:316-329 only see from == -1 ...
        const int t2 = fp < fn_ ? fp : fn_ - 1;                      // ... and a match on the right anchor (FinalPos == InSize, :280-290)
        // tag != 0 always (a SIMPLE action has tag > 0, a calling one tag != 0: checked at load)
        if (emitted >= max_triples) { stop = true; return; }         // output buffer full (FALexTools_t.h:337-340)
        ++emitted; last_to = t2 + off;
        // the streaming post-pass (sink_push / sink_finalize_word above) with its single possible id store hoisted out, so that the
        // store (an LDS chunk buffer with a flush path on the GPU) is instantiated once
        const int tf = from2 + off, tt = t2 + off;
        bool put_on = false, full = false; int put_k = 0, put_v = 0, put_f = 0, put_t = 0;
        if (words) {
            if ((words == 2 || tag != WBD_IGNORE_TAG) && out_count < max_ids) { put_on = true; put_k = out_count; put_v = tag; put_f = tf; put_t = tt; ++out_count; }
        } else {
            bool consumed = false;
            if (scanning) {
                if (tag > WBD_IGNORE_TAG && expected == tf) {            // a sub-token of the open word (tokdll:1239)
                    const int k = word_out + nsub;
                    if (k < max_ids) { put_on = true; put_k = k; put_v = tag; put_f = tf; put_t = tt; }
                    nsub++; expected = tt + 1; consumed = true;
                } else {                                                 // the open word is complete (tokdll:1252-1301)
                    if (nsub > 0 && expected - 1 == tok_to) { const int c = word_out + nsub; out_count = c < max_ids ? c : max_ids; }
                    else if (word_out < max_ids) { put_on = true; put_k = word_out; put_v = unk; put_f = tok_from; put_t = tok_to; out_count = word_out + 1; }
                    scanning = 0;
                    full = out_count >= max_ids;
                }
            }
            if (!consumed && !full) {
                if (tag == WBD_WORD_TAG) { scanning = 1; tok_from = tf; tok_to = tt; expected = tf; nsub = 0; word_out = out_count; }
                full = !(out_count < max_ids || scanning);
            }
        }
        if (put_on) { ids.put(put_k, put_v); ids.span(put_k, put_f, put_t); }
        if (full) { stop = true; return; }                           // id array full (tokdll:1308-1310)
        if (call && d == 0 && L.max_depth >= 2) {
            // enter the function on [from2, t2] (FALexTools_t.h:350-382); remember where the top level goes on
            to2 = fp + 1;                                            // where the top level goes on
            d = 1; ini = f_ini; ini_l = f_ini_l; off = from2 + off; fn_ = t2 - from2 + 1; from = -1;
            return;
        }
        from = fp + 1;                                               // FALexTools_t.h:390-393 with right = 0
    }

    BF_HD int finish()
    {
        if (scanning && !words) sink_finalize_word();
        ids.finish(out_count);
        return out_count;
    }
};

// Sequential driver (host emulation).  Returns the number of ids written (<= max_ids).
template <bool HAS_ANY, class ClsAt, class IdOut, class Frames>
BF_HD int lex_doc_t(const LexTables &L, ClsAt &cls_at, int n, IdOut &out, int max_ids, int unk, Frames &frames, int words = 0)
{
    LexLane<ClsAt, IdOut, Frames, HAS_ANY> lane(L, cls_at, out, frames);
    lane.init(n, max_ids, unk, words);
    if (L.two_level) {
        while (lane.prepare2()) {
            while (lane.step_r()) {}
            lane.after_walk2();
        }
        return lane.finish();
    }
    while (lane.prepare()) {
        while (lane.step_r()) {}
        lane.after_walk();
    }
    return lane.finish();
}

template <class ClsAt, class IdOut, class Frames>
BF_HD int lex_doc(const LexTables &L, ClsAt &cls_at, int n, IdOut &out, int max_ids, int unk, Frames &frames, int words = 0)
{
    return L.cls_any != LX_CLS_NONE ? lex_doc_t<true>(L, cls_at, n, out, max_ids, unk, frames, words)
                                    : lex_doc_t<false>(L, cls_at, n, out, max_ids, unk, frames, words);
}

// ------------------------------------------------------------------------------------------
// Long documents in the words modes (TextToWords / TextToSentences, tokdll:415-614): a document does not have to walk on one lane.
// The top-level loop of Process_int (FALexTools_t.h:229-393) carries nothing from one start position to the next but the
// position itself, and the words modes copy every triple out as it comes (no post-pass state), so
//   1. every position p of the document (-1 = the left anchor, 0 .. n-1) is run on its own as if the loop had arrived there
//      (lex_one_start with IdOutNull): where the loop goes next, how many triples it produces, how many of them are output;
//   2. the chain -1 -> next(-1) -> ... is followed once, adding the counts up: the positions the reference really visits and
//      the output index of each one's first token;
//   3. the visited positions are run again, writing their tokens at those indices.
// The triple buffer of the reference holds MaxTriples = n triples in these modes (tokdll:494-499, FALexTools_t.h:337-340): step 2
// finds the position at which it would fill, step 3 runs that position with the room that is left (LexLong::room) and nothing
// after it.  Step 1 reports a position that fills the buffer alone as n + 1 triples.
// ------------------------------------------------------------------------------------------
struct IdOutNull {
    BF_HD void put(int, int32_t) {}
    BF_HD void span(int, int, int) {}
    BF_HD void finish(int) {}
};

// keeps the first two tokens a start position writes: most positions write one (a few two), and the pass that writes the output then
// needs no second walk for them
struct IdOutFirst {
    int32_t tag0 = 0, tag1 = 0; int from0 = 0, to0 = 0, from1 = 0, to1 = 0;       // (scalars: a two-element array went to scratch on the device)
    BF_HD void put(int k, int32_t v) { tag0 = k == 0 ? v : tag0; tag1 = k == 1 ? v : tag1; }
    BF_HD void span(int k, int f, int t) { from0 = k == 0 ? f : from0; to0 = k == 0 ? t : to0; from1 = k == 1 ? f : from1; to1 = k == 1 ? t : to1; }
    BF_HD void finish(int) {}
    // a span relative to position p as one word (0x80000000 | first - p << 16 | last - p), 0 when it does not pack
    static BF_HD int pack(int from, int to, int p)
    {
        const int df = from - p, dt = to - p;
        return (df >= 0 && df < 0x8000 && dt >= 0 && dt < 0x10000) ? (int)(0x80000000u | ((unsigned)df << 16) | (unsigned)dt) : 0;
    }
    BF_HD int packed0(int p) const { return pack(from0, to0, p); }
    BF_HD int packed1(int p) const { return pack(from1, to1, p); }
};

struct LexStart { int next, n_out, n_emit; };       // of one start position: the next one, tokens output, triples produced

template <bool HAS_ANY, class ClsAt, class IdOut, class Frames, class Tab>
BF_HD LexStart lex_one_start(const LexTables &L, ClsAt &cls_at, int n, int p0, IdOut &out, Frames &frames, const Tab &tab, int words, int max_triples)
{
    LexLane<ClsAt, IdOut, Frames, HAS_ANY, Tab, true> lane(L, cls_at, out, frames, tab);
    lane.init(n, 0x7fffffff, 0, words);
    lane.from = p0; lane.p0 = p0; lane.max_triples = max_triples;
    while (lane.prepare()) {
        while (lane.step_r()) {}
        lane.after_walk();
    }
    LexStart r;
    r.next = lane.stop ? n : lane.from; r.n_out = lane.out_count; r.n_emit = lane.stop ? max_triples + 1 : lane.emitted;
    return r;
}

// One visit of step 2: position `pos` with the result `r` of step 1; `ob` / `eb` = tokens output / triples produced before it.
// Returns true while the chain goes on (pos = the next visited position).  room >= 0: the triple buffer fills at this
// position -- it is the last one visited and step 3 runs it with that much room.
BF_HD bool lex_chain_visit(int n, int max_triples, const LexStart &r, int &pos, int &ob, int &eb, int &room)
{
    if (eb + r.n_emit > max_triples) { room = max_triples - eb; return false; }
    room = -1; ob += r.n_out; eb += r.n_emit; pos = r.next;
    return pos < n;
}

} // namespace bfa
