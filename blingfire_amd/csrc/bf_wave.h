// bf_wave.h -- WordPiece TextToIds for "unit form" lexers (every BERT model), one WAVE per stream of documents.
//
// Reproduces the same reference path as bf_lex.h --
//   FAStrUtf8ToArray (cl/src/FAUtf8Utils.cpp:233-270) -> FANormalize (cl/inc/FAUtils_cl.h:311-369)
//   -> FALexTools_t<int>::Process_int (cl/inc/FALexTools_t.h:205-400) -> the _wp post-pass (tokdll:1207-1313)
// -- but organised around what the load-time analysis proves for these models (bf_model.h Model::wave_ok):
//
//   * the lexer is two-level: top-level rules find a token <tag, from, to> and call ONE function (the vocabulary) on it;
//   * a top-level token is a unit of its own for the post-pass: top-level tags are 1..4 (WORD .. IGNORE), vocabulary tags are
//     > 4, so a word's sub-tokens are exactly the tokens its own function call emits, and "the sub-tokens tile the word, else
//     UNK" (tokdll:1239-1301) is decided inside the unit;
//   * no left anchor at the top level, no right-anchor transition behind a letter, no IW_ANY, 1:1 charmap.
//
// So the work splits into three data-parallel phases per wave, none of which keeps a class stream in HBM:
//   decode   64 lanes x 8 bytes: UTF-8 -> code point -> fused charmap/class map -> a ring of 16-bit elements in LDS
//            (class | what a walk that starts on this class does: WK_LOOP / WK_NOMATCH / WK_SOLO / WK_GENERAL);
//   phase A  top level, one start position per lane: runs of WK_LOOP elements are words (one ballot, no table access), WK_SOLO /
//            WK_NOMATCH elements are decided by their class alone, WK_GENERAL elements walk the automaton; the chain of start
//            positions is closed from the masks (or serially in the rare window that holds a multi-character general token);
//            tokens are queued in LDS;
//   phase B  vocabulary, one WORD per lane slot (two slots per lane, so two independent table gathers are in flight per lane):
//            the function frame of Process_int on the word, leftmost-longest piece by piece, result = pieces or UNK;
//   phase C  in queue order: segmented prefix sum of the id counts per document -> ids to the document's staging slot.
// Text is read once, coalesced (8 consecutive bytes per lane); ids leave in document order.
//
// The same source runs on the GPU (bf_kernels.hip supplies namespace wv with the wave intrinsics) and, for tests only, on the host
// inside a 64-fibre wave simulator (tests/hosttest/wave_emu.h) where it is fuzzed against the oracle.  Rule of the house: every
// wv:: collective is called in wave-uniform control flow.
#pragma once
#include <stdint.h>
#include "bf_layout.h"
#include "bf_lex.h"
#include "bf_batch.h"

#if defined(__HIPCC__)
#define BF_WV __host__ __device__ __forceinline__
#define BF_WVD __device__ __forceinline__
#else
#define BF_WV inline
#define BF_WVD inline
#endif

namespace bfa {

// A per-lane truth value as an INTEGER (0 / 1) in a vector register.  The compiler keeps a per-lane `bool` as a 64-bit lane mask in scalar
// registers, and every `&&` / `||` / `!` on it is a scalar instruction: measured on MI355X (tools/microbench/valu_issue.hip), a CU issues
// ~1 scalar but ~1.8 vector wave-instructions per cycle, and the wave programs are bound by the scalar stream (k_bpe_wave: 9.2 k scalar
// against 7.8 k vector instructions per document).  Truth values that are combined a lot are therefore made integers once (wv_b) and
// combined with & | ^ in the vector unit; the empty asm keeps the compiler from turning the integer logic back into mask logic.
#if defined(__HIPCC__)
__host__ __device__ __forceinline__ uint32_t wv_b(bool b)
{
    uint32_t x = b ? 1u : 0u;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(x));
#endif
    return x;
}
#else
inline uint32_t wv_b(bool b) { return b ? 1u : 0u; }
#endif

// what a top-level walk that STARTS on an element of this class does (bf_model.cpp, "unit form")
constexpr uint32_t WK_GENERAL = 0;   // walk the automaton
constexpr uint32_t WK_LOOP = 1;      // enters the closed, final loop state: the token is the run of WK_LOOP elements (capped by max-length)
constexpr uint32_t WK_NOMATCH = 2;   // no transition from the initial state: no token, next start = next element
constexpr uint32_t WK_SOLO = 3;      // a final state without transitions: a one-element token
constexpr int WK_SHIFT = 14;

constexpr int WV_CHUNK = 512;        // bytes decoded per step (8 per lane)
constexpr int WV_ACTS_MAX = 64;      // ints of action records a unit-form lexer may have (bf_model.cpp)
constexpr uint32_t WV_DT_CLOSED = 1, WV_DT_BAD = 2;

// what the rare paths need (non-ASCII characters, status reports, experiments): the kernel keeps a copy in LDS instead of scalar registers
struct WpWaveCold {
    DevCpMap cpmap;                  // fused code point -> charmap -> class map
    const uint8_t *kind;             // [nclasses] WK_* per class
    int nclasses;
    int *status;                     // batch status word (Batch::status)
    unsigned long long *stats;       // optional (experiments): [0] unit rounds [1] wide passes [2] general windows [3] tokens [4] busy units summed over the rounds
                                     // [5] retire rounds [6] words of more than four pieces [7] trips without progress [8] decode steps
    int no_fast;                     // tests: every token carries its action explicitly (the path of lexers whose run / solo actions differ)
};

struct WpWaveParams {
    const uint64_t *T;               // lexer table, bf_layout.h entries (high word: action info of a final destination)
    uint32_t initial, loop_info, solo_info;
    int max_token_length;
    const uint8_t *text; const int64_t *doc_off; int64_t ndocs, total_bytes;       // the batch (bf_batch.h Batch)
    int32_t *ids_tmp;                // staging: document d writes ids_tmp[ids_slot(doc_off[d], d) ..)
    // offsets API (the OFFS instance of the program, else unused): [2k], [2k + 1] = first / last character of staged id k (indexed like ids_tmp);
    // k_compact_text turns them into byte offsets from the text itself (tokdll:1263-1273)
    int32_t *span_tmp;
    int32_t *counts;                 // [ndocs]
    int max_ids, unk;
    unsigned long long *next_doc;    // work counter
    const int32_t *doc_list = nullptr; const unsigned int *list_n = nullptr;   // the LIST instance (the documents the flat program hands back, bf_flat.h): their numbers; else unused
    const int32_t *acts; int acts_n; // action records [left, right, tag, nfn, (fn, ini, ini_l)*] (<= WV_ACTS_MAX ints: staged in LDS)
    WpWaveCold cold;
};

BF_WV int64_t wv_ids_slot(int64_t doc_off_d, int64_t d) { return ((doc_off_d + 7) & ~(int64_t)7) + 8 * d; }

BF_WV uint32_t wv_cpmap_get(const DevCpMap &m, int cp) { return m.pages[(uint32_t)m.l1[cp >> 8] * 256u + (uint32_t)(cp & 255)]; }

// ring element of a code point: class | kind << 14
BF_WV uint32_t wv_element(const WpWaveCold &p, int cp)
{
    const uint32_t c = wv_cpmap_get(p.cpmap, cp) & LX_T_CLS_MASK;
    const uint32_t k = c < (uint32_t)p.nclasses ? (uint32_t)p.kind[c] : WK_NOMATCH;      // LX_CLS_NONE: not in the alphabet
    return c | (k << WK_SHIFT);
}

// per-workgroup table of the 128 ASCII code points (LDS)
BF_WV void wv_init_ascii(const WpWaveCold &p, uint16_t *ascii, int tid, int nthreads)
{
    for (int i = tid; i < 128; i += nthreads) ascii[i] = (uint16_t)wv_element(p, i);
}

} // namespace bfa
