// bf_bpe_wave.h -- launch parameters of the BPE wave program (bf_bpe_wave_body.h; see there)
#pragma once
#include <stdint.h>
#include "bf_seg.h"
#include "bf_flat_key.h"

namespace bfa {

struct BpeWaveParams {
    const uint64_t *T; const SegInfo *info; uint32_t initial, cls_delim; int id_offset;
    const uint32_t *prio; const int32_t *place_id;   // with merge ranks (bf_model.h bpe_prio / bpe_place_id): the arcs are ordered by the entry's place in "rank descending, id ascending"; nullptr: by id
    const uint16_t *stream; const int32_t *lens; const int64_t *doc_off; int slot_mul; int64_t ndocs;
    int32_t *ids_tmp; int32_t *counts; int32_t *flags; int max_ids; unsigned long long *next_doc; int *status;
    uint32_t *scratch;               // 6 words per stream cell (the batch's arc workspace): the arcs of a word with more than 64 of them (unit_huge)
    // the word table (bf_model.cpp build_bpe_word_table; nullptr: none): words the collection takes whole, keyed by their symbols behind the U+2581
    const uint64_t *W = nullptr; int wbits = 0; uint32_t m0 = 0, m1 = 0, m2 = 0;
    unsigned long long *stats;       // optional (tests, experiments): [0] words, [1] taken whole, [2..7] documents handed back because of: a symbol outside the
                                     // alphabet, a word too long, a window overflow, a start without an arc, a position without an applied arc; [7] words solved by unit_huge; [12] words the table answered
};

} // namespace bfa
