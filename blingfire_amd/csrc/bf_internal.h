// bf_internal.h -- exported, but NOT part of the drop-in surface: experiment / measurement knobs used by bench.py and tools/.
#pragma once
#include "../../include/blingfiretokdll_amd.h"

extern "C" {
/* instrumentation counters of the tokenising kernel of a WordPiece model (BF_LEX_STATS=1 in the environment at LoadModel, or
 * BfSetLexStats), accumulated since then.  Wave program (bf_wave.h WpWaveCold::stats): [0] unit rounds [1] wide passes [2] general windows
 * [3] tokens [4] busy units summed over the rounds [5] retire rounds [6] words of more than four pieces [7] trips without progress
 * [8] decode steps [9] table gathers issued (64 per transition step and unit) [10] transitions made (walking lanes).
 * Lane-per-document kernels: [0] walk trips [1] lane-steps [2] event rounds [3] event lanes [4] fetch rounds [5] idle lane-steps
 * [6..8] cycles in walk / event / fetch */
BF_API int BfLexStats(void *ModelPtr, unsigned long long *out, int n);
/* experiments / A-B runs.  Low byte: 3 = default (flat-form WordPiece models: the flat program, bf_flat.h, for batches of >= 1,024 documents and
 * >= 1 MiB, else the wave program, bf_wave.h); 4 = the flat program for every batch; 5 = never; 2 = the lane-per-document WordPiece kernels
 * (bf_lex.h) also for unit-form lexers; bit 0x40 = BPE without its wave program.  The higher bits carry tuning values (bf_kernels.hip
 * launch_wp_wave / launch_wp_flat / launch_lex_wp, bf_kernels_sp.hip launch_seg_sp); the ones that select a measurement instance need a build
 * with BF_EXPERIMENTS, else BF_E_UNSUPPORTED (-5) comes back and the setting stays.  The words modes (TextToWords / TextToSentences): bit
 * 0x40000000 = no long-document path (every document walks on one lane), bits 12..15 = k > 0: documents of more than 8 << k characters take it
 * (default: total_bytes / 24,000, at least 16), bit 0x20000000 = a test knob: the triple
 * buffer holds n / 8 triples instead of n (FALexTools_t.h:337-340), so that tests reach the position at which it fills, bit 0x10000000 = another: room for 40 chunks of long documents only (the ones
 * that do not fit stay on lanes), bit 0x08000000 = a third: the two-level chain of very long documents from 256 characters on instead of 512 K.
 * Returns the previous value. */
BF_API int BfSetVariant(void *ModelPtr, int variant);
/* switches the instrumented kernel instances on / off for this handle and clears the counters; returns the previous setting */
BF_API int BfSetLexStats(void *ModelPtr, int on);
/* name of the dominant kernel of a plain TextToIds batch of this model (as the last batch ran it); BfLastKernelMs value [5] is its time */
BF_API const char *BfTokeniseKernel(void *ModelPtr);
/* every kernel of a step, by the segment of BfLastKernelMs it is timed in: "prep: ... | tokenise: ... | scan: ... | compact: ..." */
BF_API const char *BfStepKernels(void *ModelPtr);
/* largest chunk of the pipelined host-buffer path of TextToIdsBatch (default 128 MiB; batches of at least that size take it, cut into
 * chunks of half to all of it; 0 = never).
 * Tests use small values to put chunk boundaries everywhere.  Returns the previous value. */
BF_API int64_t BfSetHostChunkBytes(void *ModelPtr, int64_t bytes);
}
