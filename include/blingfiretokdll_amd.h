/*
 * blingfiretokdll_amd.h -- C-ABI of the MI355X-native drop-in for BlingFire's
 * libblingfiretokdll (TextToIds hot path).
 *
 * The library is built as blingfire_amd/libblingfiretokdll.so and exports the SAME
 * unmangled extern "C" symbols the reference's wrappers bind by name
 * (reference: blingfiretools/blingfiretokdll/blingfiretokdll.h:23-104,
 *  blingfiretools/blingfiretokdll/blingfiretokdll.def:3-26,
 *  dist-pypi/blingfire/__init__.py:16-22,243-253, nuget/lib/BlingFireUtils.cs:23-35,195-215).
 *
 * All compute happens on the GPU (HIP, gfx950).  There is no CPU fallback: without a HIP
 * device LoadModel/SetModel print a diagnostic to stderr and return NULL.
 *
 * Plain pointers and sizes only; no C++/torch types.  Batch entry points (additive, not in
 * the reference) take either host buffers or device buffers + a hipStream_t passed as void*.
 */
#ifndef BLINGFIRETOKDLL_AMD_H
#define BLINGFIRETOKDLL_AMD_H

#include <stdint.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

/* The library is built with -fvisibility=hidden: only the entry points declared here (and the experiment knobs of
 * blingfire_amd/csrc/bf_internal.h) are exported. */
#if defined(__GNUC__)
#define BF_API __attribute__((visibility("default")))
#else
#define BF_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- reference entry points (same names, argument meaning and error behaviour) ---- */

/* reference tokdll:107-111: returns 18000 for v0.1.8-compatible behaviour */
BF_API int GetBlingFireTokVersion(void);

/* reference tokdll:1077-1094.  Returns an opaque handle or NULL.  (The reference throws a C++
 * exception through the C boundary for a nonexistent file; this returns NULL instead.) */
BF_API void *LoadModel(const char *pszLdbFileName);

/* reference tokdll:1056-1070.  The image is copied (the reference borrows it). */
BF_API void *SetModel(const unsigned char *pImgBytes, int ModelByteCount);

/* reference tokdll:1650-1662.  Returns 1, or 0 for a NULL handle. */
BF_API int FreeModel(void *ModelPtr);

/* reference tokdll:1619-1646.  Writes at most MaxIdsArrLength ids, leaves the rest of pIdsArr
 * untouched, returns the number written; 0 on any error (NULL model/text, n <= 0, n > 1e9,
 * invalid UTF-8 for non-byte models). */
BF_API int TextToIds(void *ModelPtr, const char *pInUtf8Str, int InUtf8StrByteCount,
              int32_t *pIdsArr, const int MaxIdsArrLength, const int UnkId);

/* reference tokdll:1320-1331 and 1541-1552: the two algorithm-specific spellings */
BF_API int TextToIds_wp(void *ModelPtr, const char *pInUtf8Str, int InUtf8StrByteCount,
                 int32_t *pIdsArr, const int MaxIdsArrLength, const int UnkId);
BF_API int TextToIds_sp(void *ModelPtr, const char *pInUtf8Str, int InUtf8StrByteCount,
                 int32_t *pIdsArr, const int MaxIdsArrLength, const int UnkId);

/* reference tokdll:1562-1609 (dispatch), 1108-1314 (_wp), 1349-1535 (_sp): ids plus, for every id, the byte offset of its first
 * character and the INCLUSIVE byte offset of the last byte of its last character in the caller's string (a BOM counts; the
 * dummy prefix has offset -1).  NULL pStartOffsets / pEndOffsets = ids only, exactly like the reference.  One deviation: for a
 * token made of the dummy prefix alone the reference adds the UTF-8 size of the byte BEFORE the caller's buffer (undefined);
 * this library reports end = -1. */
BF_API int TextToIdsWithOffsets(void *ModelPtr, const char *pInUtf8Str, int InUtf8StrByteCount, int32_t *pIdsArr,
                         int *pStartOffsets, int *pEndOffsets, const int MaxIdsArrLength, const int UnkId);
BF_API int TextToIdsWithOffsets_wp(void *ModelPtr, const char *pInUtf8Str, int InUtf8StrByteCount, int32_t *pIdsArr,
                            int *pStartOffsets, int *pEndOffsets, const int MaxIdsArrLength, const int UnkId);
BF_API int TextToIdsWithOffsets_sp(void *ModelPtr, const char *pInUtf8Str, int InUtf8StrByteCount, int32_t *pIdsArr,
                            int *pStartOffsets, int *pEndOffsets, const int MaxIdsArrLength, const int UnkId);

/* reference tokdll:610-614, 585-591, 569-575, 415-566: splits text into words with a lexer model (hModel == NULL: the built-in
 * wbd.bin, embedded like the reference embeds it).  Output = words joined by ' ' (inner spaces -> '_') + terminating 0; returns
 * the byte count needed (terminator included; copied only if it fits), 0 for empty input, -1 on error (invalid UTF-8, ...).
 * pStartOffsets / pEndOffsets (MaxOutUtf8StrByteCount entries each, may be NULL) receive the byte span of every word.
 * The tokenisation runs on the GPU; only the output string is assembled on the host. */
BF_API int TextToWords(const char *pInUtf8Str, int InUtf8StrByteCount, char *pOutUtf8Str, const int MaxOutUtf8StrByteCount);
BF_API int TextToWordsWithModel(const char *pInUtf8Str, int InUtf8StrByteCount, char *pOutUtf8Str, const int MaxOutUtf8StrByteCount, void *hModel);
BF_API int TextToWordsWithOffsets(const char *pInUtf8Str, int InUtf8StrByteCount, char *pOutUtf8Str, int *pStartOffsets, int *pEndOffsets,
                           const int MaxOutUtf8StrByteCount);
BF_API int TextToWordsWithOffsetsWithModel(const char *pInUtf8Str, int InUtf8StrByteCount, char *pOutUtf8Str, int *pStartOffsets,
                                    int *pEndOffsets, const int MaxOutUtf8StrByteCount, void *hModel);

/* additive: TextToWords for many documents at once.  The output string of document d -- exactly
 * what TextToWordsWithModel writes for it, without the terminating 0; nothing for a document it would reject -- is
 * text_out[text_offsets_out[d] .. text_offsets_out[d+1]).  ModelPtr NULL = the built-in wbd.bin.  Returns the total byte count
 * or BF_E_* (BF_E_CAPACITY: the offsets are valid and tell the size).  The Device form takes device pointers and a hipStream_t
 * and never writes past text_cap; with d_text_out == NULL it only computes the offsets.  Tokenisation AND string assembly
 * (a variable-length byte gather) run on the GPU. */
BF_API int64_t TextToWordsBatch(void *ModelPtr, const char *text, const int64_t *doc_offsets, int64_t ndocs, char *text_out, int64_t text_cap,
                         int64_t *text_offsets_out);
BF_API int TextToWordsBatchDevice(void *ModelPtr, const char *d_text, const int64_t *d_doc_offsets, int64_t ndocs, int64_t total_bytes,
                           char *d_text_out, int64_t text_cap, int64_t *d_text_offsets_out, void *stream);
/* the same for TextToSentences (ModelPtr NULL = the built-in sbd.bin): per document the string TextToSentencesWithModel writes */
BF_API int64_t TextToSentencesBatch(void *ModelPtr, const char *text, const int64_t *doc_offsets, int64_t ndocs, char *text_out, int64_t text_cap,
                             int64_t *text_offsets_out);
BF_API int TextToSentencesBatchDevice(void *ModelPtr, const char *d_text, const int64_t *d_doc_offsets, int64_t ndocs, int64_t total_bytes,
                               char *d_text_out, int64_t text_cap, int64_t *d_text_offsets_out, void *stream);

/* reference tokdll:163-402 (blingfiretokdll.def: TextToSentences, TextToSentencesWithModel, TextToSentencesWithOffsets,
 * TextToSentencesWithOffsetsWithModel): sentence breaking with the model behind hModel (a LoadModel handle of a [wbd]-type
 * model such as sbd.bin; NULL = the built-in sbd.bin, embedded like the reference embeds it).  Output = sentences joined by
 * '\n' (a '\n' inside a sentence -> ' ', leading white space dropped) + terminating 0; same return convention and offset
 * arrays as TextToWords.  The sentence boundaries come from the GPU lexer; only the output string is assembled on the host. */
BF_API int TextToSentences(const char *pInUtf8Str, int InUtf8StrByteCount, char *pOutUtf8Str, const int MaxOutUtf8StrByteCount);
BF_API int TextToSentencesWithModel(const char *pInUtf8Str, int InUtf8StrByteCount, char *pOutUtf8Str, const int MaxOutUtf8StrByteCount, void *hModel);
BF_API int TextToSentencesWithOffsets(const char *pInUtf8Str, int InUtf8StrByteCount, char *pOutUtf8Str, int *pStartOffsets, int *pEndOffsets,
                               const int MaxOutUtf8StrByteCount);
BF_API int TextToSentencesWithOffsetsWithModel(const char *pInUtf8Str, int InUtf8StrByteCount, char *pOutUtf8Str, int *pStartOffsets,
                                        int *pEndOffsets, const int MaxOutUtf8StrByteCount, void *hModel);

/* reference tokdll:629-679 (blingfiretokdll.def: NormalizeSpaces; model-free): every run of white space -> one uSpace (default
 * U+2581 in the reference's header), leading white space dropped, one trailing uSpace trimmed; returns the output byte count
 * (0-terminated when there is room), -1 for empty / invalid UTF-8 input or when the output does not fit.  Runs as a batch of
 * one on the GPU; NormalizeSpacesBatch (additive) does many documents, output layout like TextToWordsBatch (a document the
 * single call rejects yields nothing). */
BF_API int NormalizeSpaces(const char *pInUtf8Str, int InUtf8StrByteCount, char *pOutUtf8Str, const int MaxOutUtf8StrByteCount, const int uSpace);
BF_API int64_t NormalizeSpacesBatch(const char *text, const int64_t *doc_offsets, int64_t ndocs, char *text_out, int64_t text_cap,
                             int64_t *text_offsets_out, int uSpace);

/* reference tokdll:773-806 (blingfiretokdll.def: TextToHashes; model-free): fasttext-style hashes of the space-separated tokens
 * of an already tokenised string, followed by the word n-gram hashes modulo bucketSize; returns the number of hashes,
 * InUtf8StrByteCount * wordNgrams ("requested size") when MaxHashArrLength is too small, -1 on error.  Deviation: wordNgrams <= 0
 * is refused (-1); the reference would write past the array.  TextToHashesBatch (additive): document d's hashes =
 * hashes_out[hash_offsets_out[d] .. hash_offsets_out[d+1]), (spaces + 1) * wordNgrams of them. */
BF_API int TextToHashes(const char *pInUtf8Str, int InUtf8StrByteCount, int32_t *pHashArr, const int MaxHashArrLength, int wordNgrams, int bucketSize);
BF_API int64_t TextToHashesBatch(const char *text, const int64_t *doc_offsets, int64_t ndocs, int32_t *hashes_out, int64_t hashes_cap,
                          int64_t *hash_offsets_out, int wordNgrams, int bucketSize);

/* reference tokdll:1689-1745 (blingfiretokdll.def: IdsToText): text of an id sequence.  ModelPtr = a LoadModel handle of a
 * model with an [i2w] section (a *.i2w file, or a .bin that carries one).  Ids outside the model's regular range are left
 * out when SkipSpecialTokens is set; a leading space is not written; returns the byte count needed including the
 * terminating 0 (written when it fits), 0 on error (no [i2w], unknown id, ...).  Runs as a batch of one on the GPU. */
BF_API int IdsToText(void *ModelPtr, const int32_t *pIdsArr, const int IdsCount, char *pOutUtf8Str, const int MaxOutUtf8StrByteCount,
              bool SkipSpecialTokens);

/* additive: many sequences at once.  ids of sequence d = ids[id_offsets[d] .. id_offsets[d+1]); its text (no terminator) =
 * text_out[text_offsets_out[d] .. text_offsets_out[d+1]), exactly the bytes IdsToText produces for it; a sequence with an
 * unknown id yields no text.  Returns the total byte count or BF_E_* (BF_E_CAPACITY if text_cap is too small: the offsets
 * are valid then and tell the size).  The Device form takes device pointers and a hipStream_t; with d_text_out == NULL it
 * only computes the offsets (size query). */
BF_API int64_t IdsToTextBatch(void *ModelPtr, const int32_t *ids, const int64_t *id_offsets, int64_t nseq, char *text_out, int64_t text_cap,
                       int64_t *text_offsets_out, int skip_special);
BF_API int IdsToTextBatchDevice(void *ModelPtr, const int32_t *d_ids, const int64_t *d_id_offsets, int64_t nseq, char *d_text_out,
                         int64_t text_cap, int64_t *d_text_offsets_out, int skip_special, void *stream);

/* reference tokdll:1669-1679 */
BF_API int SetNoDummyPrefix(void *ModelPtr, bool fNoDummyPrefix);

/* reference tokdll:818-915 (blingfiretokdll.def: WordHyphenationWithModel).  The hyphenation engine is NOT on the TextToIds path
 * and is not part of this library: the symbol exists so that consumers that bind every export by name
 * resolve, and fails loudly -- 0 for an empty input like the reference, -1 (the reference's error value) otherwise, with one
 * diagnostic on stderr. */
BF_API int WordHyphenationWithModel(const char *pInUtf8Str, int InUtf8StrByteCount, char *pOutUtf8Str, const int MaxOutUtf8StrByteCount,
                             void *hModel, const int uHy);

/* ---- additive batch entry points (a GPU wants batches; semantics = "for every document,
 *      exactly what TextToIds(h, doc, len, buf, max_ids_per_doc, unk) would have written",
 *      concatenated in document order) ---- */

/* Host buffers.  text = concatenated documents; doc_offsets[ndocs+1] = byte offsets.
 * ids_out[ids_cap] receives the concatenated ids, id_offsets_out[ndocs+1] their boundaries.
 * Returns the total number of ids, or a negative error code (BF_E_*); with BF_E_CAPACITY id_offsets_out is still complete
 * (id_offsets_out[ndocs] = the ids_cap that would have sufficed).  Large batches (>= 128 MiB of text) are pipelined in chunks of
 * whole documents through page-locked staging; the call returns when everything is in the caller's arrays.
 * BPE models (gpt2.bin, roberta.bin, ...): every document has an answer, as with the reference, whose arc list is an unbounded
 * std::vector (FATokenSegmentationTools_1best_bpe_t.h:143-144,197).  A document that needs more candidate arcs than the batch
 * workspace reserves per document (e.g. nothing but one long run of '-') is tokenised by one wave out of a pool of the handle
 * (64 MiB at first, BfSetBpePoolBytes); when a batch needs more, this call grows the pool and runs again.  The ...BatchDevice form
 * cannot allocate: there such a document gets 0 ids and BfLastStatus reports bit 64 (size the pool with BfSetBpePoolBytes first). */
BF_API int64_t TextToIdsBatch(void *ModelPtr, const char *text, const int64_t *doc_offsets, int64_t ndocs,
                       int32_t *ids_out, int64_t ids_cap, int64_t *id_offsets_out,
                       int max_ids_per_doc, int unk);

/* Device buffers (all five pointers are device memory of the model's GPU).  Work is enqueued on
 * `stream` (a hipStream_t; NULL = the default stream) and the call returns without synchronising:
 * 0 = enqueued, negative = error.  The total id count is d_id_offsets_out[ndocs].  ids_cap must be
 * >= min(2 * (total_bytes + ndocs), ndocs * max_ids_per_doc) to be safe for any input (WordPiece models never
 * need more than total_bytes); ids beyond ids_cap are dropped
 * and reported by BfLastStatus.  d_text needs no padding.
 * PRECONDITION of every ...BatchDevice call: d_doc_offsets[0] == 0 and d_doc_offsets[ndocs] == total_bytes (offsets are relative to
 * d_text; pass d_text + first_byte and rebased offsets for a sub-range).  The kernels index their workspaces by these offsets; a
 * document whose range falls outside [0, total_bytes] is treated as empty and reported through BfLastStatus (bit 3).
 * A handle owns ONE set of device workspaces: the ...Device calls on one handle must be ordered on the device (the same
 * stream, or streams the caller synchronises); for concurrent batches use one handle per stream (LoadModel is cheap:
 * a few MB of tables).  The host-buffer calls serialise on the handle's mutex and synchronise before they return. */
BF_API int TextToIdsBatchDevice(void *ModelPtr, const char *d_text, const int64_t *d_doc_offsets, int64_t ndocs,
                         int64_t total_bytes, int32_t *d_ids_out, int64_t ids_cap, int64_t *d_id_offsets_out,
                         int max_ids_per_doc, int unk, void *stream);

/* Batch forms of TextToIdsWithOffsets: starts_out / ends_out are parallel to ids_out (same offsets array). */
BF_API int64_t TextToIdsWithOffsetsBatch(void *ModelPtr, const char *text, const int64_t *doc_offsets, int64_t ndocs,
                                  int32_t *ids_out, int32_t *starts_out, int32_t *ends_out, int64_t cap,
                                  int64_t *id_offsets_out, int max_ids_per_doc, int unk);
BF_API int TextToIdsWithOffsetsBatchDevice(void *ModelPtr, const char *d_text, const int64_t *d_doc_offsets, int64_t ndocs,
                                    int64_t total_bytes, int32_t *d_ids_out, int32_t *d_starts_out, int32_t *d_ends_out,
                                    int64_t cap, int64_t *d_id_offsets_out, int max_ids_per_doc, int unk, void *stream);

/* additive: the reference's dictionary interpreter, FADictInterpreter_t<int>::GetInfo
 * (blingfireclient.library/inc/FADictInterpreter_t.h:31-66,334-390; FAMphInterpretTools_t.h:97-122), for many keys at once over the
 * [pos-dict] of a tokenizer model (gpt2.bin, xlm_roberta_base.bin, ...) -- the same Mealy automaton / K2I / I2Info the segmenters use.
 * Key k = keys[key_offsets[k] .. key_offsets[k+1]) as int code points (the reference's Ty = int; byte-encoded models such as gpt2.bin
 * store bytes and U+2581 as their symbols), configured like blingfiretokdll would configure it (SetConf without a transformation):
 * an l2r dictionary is looked up as is (the [pos-dict] charmap is not applied, FADictInterpreter_t.h:203-205), an r2l one after
 * charmap normalisation and reversal, an ignore-case one (either direction) after folding every symbol (FAUtf32ToLower, :231-238) and
 * then the charmap.  ret_out[k] = GetInfo's return value (value count, -1 = no such key: not in the dictionary,
 * empty or longer than 300 symbols); info_ids_out[k] = GetInfoId (-1 = none); the values of key k (for the tokenizer dictionaries:
 * [token id, float score bits]) = values_out[value_offsets_out[k] .. value_offsets_out[k+1]).  ret_out / info_ids_out may be NULL.
 * Returns the total value count or BF_E_* (BF_E_CAPACITY: the offsets are valid and tell the size).  The Device form takes device
 * pointers + a hipStream_t, never writes past values_cap and returns 0 / BF_E_*; with d_values_out == NULL it only fills the
 * per-key results and the offsets. */
BF_API int64_t DictGetInfoBatch(void *ModelPtr, const int32_t *keys, const int64_t *key_offsets, int64_t nkeys, int32_t *ret_out, int32_t *info_ids_out,
                                int32_t *values_out, int64_t values_cap, int64_t *value_offsets_out);
BF_API int DictGetInfoBatchDevice(void *ModelPtr, const int32_t *d_keys, const int64_t *d_key_offsets, int64_t nkeys, int32_t *d_ret_out,
                                  int32_t *d_info_ids_out, int32_t *d_values_out, int64_t values_cap, int64_t *d_value_offsets_out, void *stream);

/* Per-kernel GPU time of the last batch call on this handle, measured with HIP events recorded on the
 * call's own stream.  Synchronises with those events.  Fills up to n floats (milliseconds):
 * [0] prep (decode+normalise+classify)  [1] tokenise (lexer / segmenter)  [2] scan  [3] compact  [4] total  [5] the dominant kernel of the
 * tokenise segment alone.
 * Returns the number of values written, or a negative error. */
BF_API int BfLastKernelMs(void *ModelPtr, float *ms, int n);

/* Status word of the last batch call (synchronises): 0 = ok; bit 0 (1) = ids_cap overflow; bit 1 (2) and bit 4 (16) = an internal
 * limit the load-time checks exclude was met (the results of the batch are not to be trusted; the host-buffer calls return
 * BF_E_INTERNAL); bit 3 (8) = a document's byte range was outside [0, total_bytes] (Device calls: see the precondition above) and was
 * treated as empty.  Per-document, the rest of the batch is valid: bit 5 (32) = a BPE document on which the reference itself does not
 * terminate (a skipped start position left without an arc, FATokenSegmentationTools_1best_bpe_t.h:299-313) got 0 ids; bit 6 (64) = a
 * BPE document's arcs did not fit the pool and it got 0 ids (only the ...BatchDevice forms: the host-buffer calls grow the pool and
 * run again; BfSetBpePoolBytes). */
BF_API int BfLastStatus(void *ModelPtr);

/* Last load error message of the calling thread ("" if none). */
BF_API const char *BfLastError(void);

/* Model facts: 0 = WordPiece lexer, 1 = Unigram-LM, 2 = BPE, 3 = BPE-opt, 4 = BPE with merge ranks */
BF_API long long BfBpeFallbackDocs(void *ModelPtr);   /* diagnostics: documents of the last BPE batch that took the full (sort + apply) path */
BF_API int BfModelKind(void *ModelPtr);

/* Optional: size every workspace of the handle for batches of up to max_docs documents / max_bytes bytes of text now, so that
 * later ...BatchDevice calls of that size allocate nothing (workspaces only ever grow; growing means hipMalloc, which
 * synchronises the device and is not allowed inside a stream capture).  want_offsets != 0 also sizes the offsets API.
 * Returns 0 or BF_E_*. */
BF_API int BfReserve(void *ModelPtr, int64_t max_docs, int64_t max_bytes, int want_offsets);
/* BPE models: the size of the pool from which documents with very many candidate arcs claim their working memory (about 16 bytes per
 * arc: a run of 10^6 identical characters whose run lengths are vocabulary entries takes ~350 MB).  Takes effect with the next batch
 * (the pool only grows).  Returns the previous size or BF_E_*. */
BF_API int64_t BfSetBpePoolBytes(void *ModelPtr, int64_t bytes);

/* Multi-GPU (SURVEY.md section 8b "SetDevices", 8e): range-shards the HOST-buffer batch calls of this handle (TextToIdsBatch,
 * TextToIdsWithOffsetsBatch) over n devices of this node.  The tables are replicated on every listed device; a batch is split into
 * n contiguous, byte-balanced document ranges (BfShardRanges), each tokenised by a host thread of its own on its own device, stream set
 * and workspace; ids and offsets come back exactly as one device would have returned them (no collective: documents are independent).
 * A device may be listed more than once (logical shards on one device).  BfSetDevices(h, &own_device, 1) ends sharding.
 * The ...BatchDevice calls (their buffers live on ONE device) and the single-document calls stay on the handle's own device; a caller
 * that keeps its shards resident on the devices itself takes the per-range handles with BfShardHandle(h, g) (NULL past the last range;
 * owned by h, released by FreeModel(h) / the next BfSetDevices) and calls TextToIdsBatchDevice on each from a thread of its own.
 * Reference semantics preserved per document: blingfiretokdll.cpp:1619-1646.  Returns 0 or BF_E_*. */
BF_API int BfSetDevices(void *ModelPtr, const int *device_ids, int n);
BF_API void *BfShardHandle(void *ModelPtr, int g);
/* the ranges a batch is split into over G devices: bounds[0 .. G], range g = documents [bounds[g], bounds[g + 1]): bounds[g] is the
 * document boundary closest to g / G of the text (pure host arithmetic, no device needed) */
BF_API int BfShardRanges(const int64_t *doc_offsets, int64_t ndocs, int G, int64_t *bounds);

#define BF_E_ARG      (-1)   /* bad argument */
#define BF_E_DEVICE   (-2)   /* HIP error */
#define BF_E_CAPACITY (-3)   /* ids_cap too small */
#define BF_E_INTERNAL (-4)   /* internal capacity exceeded */
#define BF_E_UNSUPPORTED (-5)

#ifdef __cplusplus
}
#endif
#endif
