// bf_internal.h -- exported, but NOT part of the drop-in surface: experiment / measurement knobs used by bench.py and tools/.
#pragma once
#include "../../include/blingfiretokdll_amd.h"

extern "C" {
/* instrumentation counters of the lexer kernel when BF_LEX_STATS=1 is set in the environment (accumulated since LoadModel):
 * [0] walk trips  [1] lane-steps (table probes / fast-forward runs)  [2] event rounds  [3] event lanes  [4] fetch rounds
 * [5] idle lane-steps  [6..8] cycles in walk / event / fetch */
BF_API int BfLexStats(void *ModelPtr, unsigned long long *out, int n);
/* selects a kernel variant (3 = default): lexer 1 = sequential driver, 4 = no loop-state fast-forward, 5 = no LDS-resident table;
 * Unigram 1 = sequential, 2 = flat, 6 = round-1 ring kernel; bits 8.. carry tuning values.  Returns the previous value. */
BF_API int BfSetVariant(void *ModelPtr, int variant);
/* largest chunk of the pipelined host-buffer path of TextToIdsBatch (default 128 MiB; batches of at least that size take it, cut into
 * chunks of half to all of it; 0 = never).
 * Tests use small values to put chunk boundaries everywhere.  Returns the previous value. */
BF_API int64_t BfSetHostChunkBytes(void *ModelPtr, int64_t bytes);
}
