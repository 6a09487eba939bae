// bf_flat.h -- WordPiece TextToIds for "flat form" lexers (every BERT model): the text buffer of a batch as ONE byte stream.
//
// The same reference path as bf_wave.h --
//   FAStrUtf8ToArray (cl/src/FAUtf8Utils.cpp:233-270) -> FANormalize (cl/inc/FAUtils_cl.h:311-369)
//   -> FALexTools_t<int>::Process_int (cl/inc/FALexTools_t.h:205-400) -> the _wp post-pass (tokdll:1207-1313)
// -- organised around two more facts (bf_model.cpp "flat form"):
//
//   * 97 % of the words of running text are ONE vocabulary entry, and whether a word is one is a function of the word alone: the walk of
//     the vocabulary function from its first state consumes the whole word and ends on a final state.  All such words of <= 12 characters are
//     enumerated at load into a two-choice hash table keyed BY THE WORD (a byte per character, bf_flat_key.h): a word is resolved by two
//     independent 16-byte gathers instead of a chain of dependent transitions, and a hit needs no verification (the key is the word);
//   * nothing but the position of a word in its document's id list couples the words of a batch.  So a wave takes a contiguous RANGE of
//     documents as one stream of 512-byte chunks that ignore document boundaries (a boundary is a bit in a mask that cuts runs), gives every
//     token an ENTRY (range-dense: entry k of the range = its k-th token) and lets the merge kernel (k_wp_merge) put the entries of a
//     document in their final place; the words the table does not hold (3 %: long words, words of several pieces, words with
//     characters outside ASCII) become records of the range's own list and are walked by a kernel of their own (k_wp_units).
//
// What the program does not resolve it hands back PER DOCUMENT (a flag in dstat[]): a run of more than WF_RUN_MAX bytes, an element whose
// top-level token the automaton itself must decide (WK_GENERAL).  Those documents are tokenised by the wave program (bf_wave.h) afterwards.
// Invalid UTF-8 is detected here (same rules as bf_wave_body.h decode_chunk) and gives the document 0 ids.
//
// Runs on the GPU (bf_kernels.hip) and, for tests, in the wave simulator (tests/hosttest/wave_emu.h).
#pragma once
#include <stdint.h>
#include "bf_wave.h"
#include "bf_flat_key.h"

namespace bfa {

struct WfParams {
    const uint64_t *T;               // lexer table (bf_layout.h)
    const uint64_t *W; int wbits; uint32_t m0, m1, m2;     // word table (bf_model.h Model::flat_tab)
    uint32_t ini, ini_l; int max_token_length, unk;
    const uint8_t *text; const int64_t *doc_off; int64_t ndocs, total_bytes;
    const int64_t *range_doc; int nranges;                 // range r = documents [range_doc[r], range_doc[r + 1])  (k_wp_pre)
    unsigned long long *next_range;                        // work counter (nullptr: ranges dealt out round-robin)
    const int *unsafe;               // k_wp_pre: the batch is not fit for this program (every document goes to the wave program)
    uint32_t *ent;                   // [total_bytes + 64] entries: entry doc_off[first document of the range] + k = the range's k-th token
    int32_t *home;                   // [total_bytes + 64] ids of the words of two and more pieces: piece j of the word that starts at byte p -> home[p + j]
    int64_t *ent_off; int32_t *ent_cnt;                    // per document: its first entry, its number of entries
    int32_t *dstat;                  // per document, zero before the launch: WF_D_* bits
    uint32_t *espan;                 // offsets API (else nullptr), parallel to ent: where the token is in its document (WF_SPAN_*: first byte | bytes - 1 << 24)
    // the words the table did not answer (16-byte records, bf_flat_body.h).  A range owns the records [first byte / 4, last byte / 4): the words a unit
    // holds in registers from the front, the others from the back; wrec_cnt[2 r], [2 r + 1] = how many of each (written by the range's wave)
    uint32_t *wrec; int32_t *wrec_cnt;
    WpWaveCold cold;                 // code-point map, class kinds, status word, optional counters
    int dbg = 0;                     // measurements (BF_EXPERIMENTS builds; wrong results by design): 1 = no look-up pass, 2 = no token list either, 4 = look-up without the table gathers
};

// the words of the list, walked (k_wp_units)
struct WfUnitParams {
    const uint64_t *T; uint32_t ini, ini_l; int max_token_length;
    const uint8_t *text; int64_t total_bytes;
    const uint32_t *wrec; const int32_t *wrec_cnt; const int64_t *range_doc; const int64_t *doc_off; int nranges;      // the ranges' lists (WfParams)
    uint32_t *ent; int32_t *home;
    int32_t *extra;                  // per document, zero before the launch: the ids its words of several pieces have beyond one each (what k_wp_count adds to the entries)
    const uint32_t *espan; uint32_t *hspan;                 // offsets API (else nullptr): the words' spans (WfParams); INSTEAD of home: (id, span) of piece j of the word at byte p -> hspan[2 (p + j)], [2 (p + j) + 1]
    DevCpMap cpmap; const uint8_t *kind; int nclasses;      // fused code point -> charmap -> class map, kinds of the classes (the table of the ASCII bytes is made of them)
    unsigned long long *stats;       // optional: [8] rounds [9] batches
};

// the kernels behind the program (count -> scan -> merge)
struct WfMergeParams {
    const int64_t *doc_off; int64_t ndocs;
    const uint32_t *ent; const int32_t *home; const int64_t *ent_off; const int32_t *ent_cnt; const int32_t *dstat; const int *unsafe;
    const int32_t *ids_tmp;          // staging of the documents the wave program tokenised (bf_wave.h wv_ids_slot)
    int32_t *counts;                 // [ndocs] in: the wave program's counts of the documents it tokenised, of all others the ids beyond one per entry (k_wp_units); out (k_wp_count): every document's ids
    const int64_t *id_off; int32_t *ids_out; int64_t ids_cap; int *status;
    int max_ids, unk;
    // offsets API (else nullptr): the spans of the entries and of the pieces at the homes; the byte offsets of every id (tokdll:1263-1297).  The documents
    // the wave program tokenised are not copied here then: counts_hard[d] = their count (0 for all others) for the kernel that does (k_compact_text)
    const uint32_t *espan, *hspan; int32_t *starts_out, *ends_out; int32_t *counts_hard;
};

} // namespace bfa
