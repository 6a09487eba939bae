// bf_kernels.h -- device-side parameter blocks + launch wrappers (implemented in bf_kernels.hip).
#pragma once
#include <stdint.h>
#include <hip/hip_runtime_api.h>
#include "bf_lex.h"
#include "bf_seg.h"
#include "bf_batch.h"
#include "bf_wave.h"
#include "bf_bpe_wave.h"
#include "bf_flat.h"

namespace bfa {

struct WpPrepParams {
    Batch b;
    DevCpMap cpmap;             // fused charmap+class map (bf_model.h Model::wbd_cpmap)
    const uint16_t *multi_pool; // records for 0 / 2..10 output chars
    int has_multi;
    uint16_t *cls;              // [total_bytes] class stream: document d occupies cls[doc_off[d] ..)
    int32_t *src_off;           // optional (offsets API): byte offset in the document of the source character of every stream element
    int32_t *nchars;            // [ndocs] normalised length, 0 = "TextToIds returns 0"
    // optional (k_prep_wp, the one-pass form): documents of more than PREP_LONG_BYTES bytes are listed here and decoded by sixteen waves
    // each (k_prep_wp_long) instead of one; long_count is zeroed per launch
    int64_t *long_list; unsigned int *long_count; int64_t long_cap;
};

// Long documents of the words modes (bf_lex.h: lex_one_start / lex_chain_visit).  A document of more than `thresh` characters is
// listed (k_lex_long_list) and leaves the lane kernel alone; its n + 1 start positions are cells of a dense "chunk space" (64 cells
// per chunk, a document owns whole chunks: cell = 64 * chunk0 + position + 1) that three kernels go over:
//   k_lex_long<false>   every cell runs its start position without output (spec), then the wave resolves its own chunk: from every cell,
//                       the first cell beyond the chunk the chain reaches and the counts summed on the way there (jump);
//   k_lex_long_chain    one workgroup per document goes from chunk to chunk, one hop each: where the chain enters a chunk and the
//                       output index / triples so far at that point (entry);
//   k_lex_long<true>    every entered chunk follows the chain inside itself, the visited cells run again and write their tokens into
//                       the document's staging slot.
struct LexLongDoc { int64_t doc, chunk0; };       // doc < 0: a listed document that did not fit the workspace (the lane kernel keeps it)
struct LexLongParams {
    int thresh;                       // 0: off
    int cap_shift;                    // test knob (BfSetVariant 0x20000000): the triple buffer holds n >> cap_shift triples instead of n (lane kernel too)
    int64_t cap_docs, cap_chunks;     // capacity of `list` / of the cell arrays in chunks
    unsigned long long *hdr;          // documents listed << 32 | chunks handed out (zeroed per launch)
    LexLongDoc *list;
    int32_t *spec;                    // [4 * cell]: next position, tokens output, triples produced, tag of the first token
    int32_t *jump;                    // [4 * cell]: first cell of the document beyond this chunk on the chain from here (LEX_CHAIN_END: none), tokens / triples summed up to there,
                                      // the span of the position's first token when it writes one or two (0x80000000 | first - position << 16 | last - position; else 0):
                                      // those are written without a second walk
    int32_t *tok2;                    // [2 * cell]: tag and span of the second token of a position that writes two (the span of the first in `jump` is 0 unless both pack)
    // documents of more than `big_cells` cells (0: none) take the chain in two levels: jump2 = the same as jump for the document's
    // 64-chunk super-chunks (k_lex_long_jump2), entry2[first chunk of a super-chunk] = the CELL OF THE DOCUMENT the chain enters it at
    // (-1: it does not), tokens / triples before it (k_lex_long_chain2); k_lex_long_chain3 fills `entry` per super-chunk
    int big_cells;
    int32_t *jump2;                   // [4 * cell]
    int32_t *entry2;                  // [4 * chunk] (used at the first chunk of every super-chunk)
    int32_t *entry;                   // [4 * chunk]: cell of the chunk the chain enters at (-1: it does not), tokens output / triples produced before it, -
};
constexpr int LEX_CHAIN_END = 0x40000000, LEX_COUNT_SAT = 0x3fffffff;
constexpr unsigned long long LEX_LONG_CHUNK_MASK = (1ull << 32) - 1;

struct WpLexParams {
    LexTables L;
    Batch b;
    const uint16_t *cls;
    const int32_t *nchars;
    int32_t *ids_tmp;           // [total_bytes] staging: document d writes ids_tmp[doc_off[d] ..)
    int32_t *span_tmp;          // optional (offsets API): [2k],[2k+1] = first / last stream position of staged id k
    int32_t *counts;            // [ndocs]
    int max_ids, unk;
    unsigned long long *next_doc; // work counter (persistent variants)
    int *status;
    int ev_thresh, fetch_thresh;  // vote thresholds of the divergence-aware driver
    int acts_n;                   // ints in L.acts (staged in LDS when small)
    int words;                    // TextToWords mode (bf_lex.h LexLane::words)
    unsigned long long *stats;    // optional instrumentation counters (experiments), else nullptr
    int table_n;                  // entries of L.T that can hold a transition (the branch-free probe tail behind them is all empty):
                                  // when they fit, the kernel stages them in LDS (bf_kernels.hip TabLds)
    LexLongParams lg;             // words modes: the long-document path (thresh 0 = every document on a lane)
};

// _sp branch: element slot of document d (stream, DP arrays and the id staging slot share it):
//   first element = slot_mul * (doc_off[d] + d), capacity slot_mul * (nbytes + 1)
struct SpPrepParams {
    Batch b;
    DevCpMap cpmap;             // fused charmap + element-code map (bf_model.h Model::sp_cpmap)
    const uint16_t *multi_pool;
    int has_multi, use_bytes, has_charmap;
    uint16_t delim_code;
    int prefix_n; uint16_t prefix[10];
    int slot_mul;
    uint16_t *stream;           // element codes after prefix / charmap / whitespace collapse
    int32_t *src_off;           // optional (offsets API): byte offset of the source character of every kept element (-1: dummy prefix)
    int32_t *lens;              // [ndocs] stream length, 0 = "TextToIds returns 0"
    int old_form;               // experiments: the byte-per-lane kernel
    int waves;                  // experiments: waves per SIMD the eight-bytes-per-lane kernel is compiled for (0 = the compiler's choice)
};

struct SpSegParams {
    SegTables S;
    Batch b;
    const uint16_t *stream; const int32_t *lens;
    int slot_mul;
    int32_t *ids_tmp; int32_t *counts;
    int32_t *span_tmp;          // optional (offsets API)
    int max_ids, unk;
    // scratch (indexed by element slot): unigram sc/bi; bpe arcs (6 per element + 32 per document), tos/idsv/inter
    SegBest *best;
    SegArc *arcs; int32_t *tos; int32_t *idsv; uint8_t *inter;
    int *status;
    // documents are handed to lanes in order of stream length (counting sort), so the 64 documents of a wave
    // are of similar size: perm[i] = document processed by thread i; hist = 1024 bucket counters / cursors
    int32_t *perm; unsigned int *hist;
    int32_t *narcs;             // [ndocs] BPE arc count per document (-1 = capacity exceeded); Unigram: first id index within the slot
    int trie_depth;             // longest dictionary entry (bounds every arc length)
    int lane_ok;                // Unigram: the model fits the lane program (bf_seg.h UniLane: entries <= 32 symbols, ids < 2^20 - 2)
    int uni_cut = 0;            // Unigram: the cut form (bf_seg.h UniCut; ids only): tokens leave as words (bf_seg.h: key / symbols to walk / unknown), left-aligned in the slot -> k_uni_ids
    int64_t bm_words;           // BPE apply: words per bitmap (the two bitmaps live in the `tos` buffer)
    int32_t *fb_list; unsigned int *fb_count;   // BPE: documents k_bpe_fused hands to the full path (set by launch_seg_sp)
    uint8_t *big_pool; unsigned long long big_cap; unsigned long long *big_used, *big_need;   // BPE: pool of the documents whose arcs exceed the per-document reserve (k_bpe_seg); *big_need: bytes that did not fit
    const uint32_t *bpe_prio; const int32_t *bpe_place_id; uint32_t bpe_unk_prio; int bpe_prio_bits;   // bf_bpe_seg_body.h: the arc order as integers (bf_model.h)
    unsigned long long *seg_stats;   // optional (experiments): counters of k_bpe_seg
    int variant;
    int tune;                   // experiments: vote threshold of the lane-local BPE solve / Unigram transitions per trip (0 = default)
    int tune2;                  // experiments: resident waves per CU of the persistent segmenter kernels (0 = what fits)
    unsigned long long *next_doc;
    void *ev_dom0 = nullptr, *ev_dom1 = nullptr;   // optional (hipEvent_t): recorded around the dominant kernel of the step (the Unigram forward pass)
};

// key -> info lookup (reference FADictInterpreter_t<int>::GetInfo) for a batch of keys: key k = keys[key_off[k] .. key_off[k+1])
struct DictParams {
    DictTables D;
    const int32_t *rows; int stride, min_key, nrows;          // I2Info rows: [count or -1, values...]
    const int32_t *keys; const int64_t *key_off; int64_t nkeys;
    int32_t *info_ids;            // [nkeys] GetInfoId
    int32_t *ret;                 // [nkeys] GetInfo return value: value count, -1 = no such key / row
    int32_t *counts;              // [nkeys] values written for the key (max(ret, 0)) -> scanned into val_off
    const int64_t *val_off; int32_t *vals; int64_t vals_cap;
};
void launch_dict_ids(const DictParams &p, hipStream_t s);
void launch_dict_fill(const DictParams &p, hipStream_t s);

// the ids of the tokens of the Unigram cut form, written to their place in the caller's array (bf_kernels_sp.hip k_uni_ids)
struct UniIdsParams {
    Batch b; const uint64_t *T; uint32_t initial; const int32_t *ids;     // dictionary transitions; the id column of I2Info
    const uint16_t *stream; const int32_t *lens; int slot_mul;
    const int32_t *toks; const int32_t *counts; const int64_t *id_off;
    int32_t *ids_out; int64_t ids_cap; int unk, id_offset; int *status;
};
void launch_uni_ids(const UniIdsParams &p, hipStream_t s);

// the HOME form of the BPE wave program (bf_bpe_wave_body.h): ids at their words' homes -> counts -> (scan) -> the caller's array (bf_kernels_sp.hip)
struct BpeHomeParams {
    Batch b; const int32_t *ids_tmp; const int32_t *lens; const int32_t *flags; int slot_mul;
    int32_t *counts; const int64_t *id_off; int32_t *ids_out; int64_t ids_cap; int max_ids; int *status;
};
bool bpe_wave_home(int tune);
void launch_bpe_home_gather(const BpeHomeParams &p, hipStream_t s);

struct ScanParams { const int32_t *counts; int64_t ndocs; int64_t *id_off; int64_t *block_sums; int nblocks; };

struct CompactParams {
    Batch b; const int32_t *ids_tmp; const int32_t *counts; const int64_t *id_off;
    int32_t *ids_out; int64_t ids_cap; int *status;
    int slot_mul;               // 0: _wp slots (align8(doc_off)+8d); >0: _sp slots (slot_mul*(doc_off+d))
    const int32_t *first;       // optional: index of the first id inside the slot (Unigram flat kernel writes ids right-aligned)
    // offsets API (all optional): stream positions of the staged ids -> byte offsets in the original text (tokdll:1263-1273,1519-1529)
    const int32_t *span_tmp; const int32_t *src_off; int32_t *starts_out; int32_t *ends_out;
    const unsigned int *only_if = nullptr;      // optional: nothing to do when *only_if == 0 (the documents the flat program hands back: usually none)
};

// flags: one bit per 16-byte chunk of the text (u64 per KiB, + 1), or nullptr for the one-pass wave-per-document form
void launch_prep_wp(const WpPrepParams &p, int64_t total_bytes, unsigned long long *flags, hipStream_t s);
void launch_lex_wp(const WpLexParams &p, int variant, hipStream_t s);
void launch_lex_long_list(const WpLexParams &p, hipStream_t s);      // before launch_lex_wp: lists the long documents (counts[doc] = -1)
void launch_lex_long(const WpLexParams &p, hipStream_t s);           // after it: the three kernels of the long-document path
void launch_wp_wave(const WpWaveParams &p, int variant, hipStream_t s);     // unit-form lexers (bf_wave.h): replaces prep + lexer
// the flat program (bf_flat.h): ranges + fitness of the batch, the program, the documents it hands back, counts, merge
int wp_flat_ranges(int64_t ndocs, int64_t total_bytes);
void launch_wp_pre(const int64_t *doc_off, int64_t ndocs, int64_t total_bytes, int nranges, int64_t *range_doc, int *unsafe, hipStream_t s);
void launch_wp_flat(const WfParams &p, int variant, hipStream_t s);
void launch_wp_units(const WfUnitParams &p, int variant, hipStream_t s);
void launch_wp_hardlist(const int32_t *dstat, const int *unsafe, int64_t ndocs, int32_t *list, unsigned int *list_n, hipStream_t s);
void launch_wp_count(const WfMergeParams &p, hipStream_t s);
void launch_wp_merge(const WfMergeParams &p, hipStream_t s);
void launch_bpe_wave(const BpeWaveParams &p, int tune, hipStream_t s);                  // bpe-opt models (bf_bpe_wave_body.h)
void launch_bpe_seg_flags(const SpSegParams &p, const int32_t *flags, int32_t *list, unsigned int *count, hipStream_t s);   // the documents it hands back: bf_bpe_seg_body.h
void launch_prep_sp(const SpPrepParams &p, hipStream_t s);
void launch_seg_sp(const SpSegParams &p, hipStream_t s);
void launch_scan(const ScanParams &p, hipStream_t s);

// IdsToText (reference tokdll:1689-1745) as a batch: ids of sequence d are ids[id_off[d] .. id_off[d+1])
struct I2tParams {
    const uint32_t *tok_off; const uint8_t *tok_data; int ntok;   // [i2w] string array
    int min_id, max_id, skip_special;
    const int32_t *ids; const int64_t *id_off; int64_t nseq;
    int32_t *lens;               // [nseq] bytes of text per sequence (no terminator); 0 for an empty / failed sequence
    const int64_t *text_off;     // [nseq + 1] exclusive scan of lens (copy pass)
    uint8_t *text; int64_t text_cap;
    int *status;                 // bit 2: a sequence contains an id the model does not know (the reference returns 0 for it)
};
void launch_i2t_len(const I2tParams &p, hipStream_t s);

// TextToWords output assembly as a batch (reference tokdll:502-565): words of document d = spans word_off[d] .. word_off[d+1]
// (byte offsets relative to the document), joined by ' ', a ' ' or NUL inside a word written as '_'
struct W2tParams {
    const uint8_t *text; const int64_t *doc_off; int64_t ndocs;
    const int64_t *word_off; const int32_t *starts, *ends;
    int32_t *lens; const int64_t *text_off; uint8_t *out; int64_t out_cap;
    const int32_t *nvalid;       // sentences only: characters decoded per document (<= 0: the document is rejected)
    // optional (the copy kernels): documents of more than W2T_LONG_TOKENS tokens are listed here and assembled by sixteen waves each
    // (k_w2t_copy_long); long_count is zeroed by the caller
    int64_t *long_list; unsigned int *long_count; int64_t long_cap;
};
void launch_w2t_len(const W2tParams &p, hipStream_t s);
// TextToSentences output assembly (reference tokdll:257-339) on the same parameters: ends[] = last byte of every token of the
// sentence breaker (starts[] unused); a sentence runs from the byte after the previous token to the token's last byte, the rest
// of the document is the last sentence; leading white space dropped, '\n' / NUL -> ' ', sentences joined by '\n'
void launch_s2t_len(const W2tParams &p, hipStream_t s);
void launch_s2t_copy(const W2tParams &p, hipStream_t s);
void launch_w2t_copy(const W2tParams &p, hipStream_t s);

// NormalizeSpaces (reference tokdll:629-679) and TextToHashes (tokdll:683-815) as batches; both are model-free
struct NormSpParams {
    const uint8_t *text; const int64_t *doc_off; int64_t ndocs;
    int u_space, usp_len; uint32_t usp_bytes;   // uSpace, its UTF-8 length (0 if it cannot be encoded) and bytes (little end first)
    int32_t *lens;               // [ndocs] output bytes (0 on error)
    int32_t *aux;                // [ndocs] bit 0: error (empty / invalid UTF-8); bits 1..: normalised spaces left after the trim (saturating)
    const int64_t *out_off; uint8_t *out; int64_t out_cap;      // write pass
};
void launch_normsp(const NormSpParams &p, bool write, hipStream_t s);
struct HashParams {
    const uint8_t *text; const int64_t *doc_off; int64_t ndocs;
    int ngrams, bucket;
    int32_t *lens;               // [ndocs] (spaces + 1) * ngrams
    const int64_t *out_off; int32_t *out; int64_t out_cap;
};
void launch_hash_count(const HashParams &p, hipStream_t s);
void launch_hash_fill(const HashParams &p, hipStream_t s);
void launch_i2t_copy(const I2tParams &p, hipStream_t s);
void launch_compact(const CompactParams &p, hipStream_t s);
int scan_nblocks(int64_t ndocs);

} // namespace bfa
