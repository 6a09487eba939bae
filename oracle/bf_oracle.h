/*
 * bf_oracle.h -- TEST INFRASTRUCTURE ONLY (not part of the product).
 *
 * Plain-C CPU restatement of the reference BlingFire TextToIds hot path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call this.  The product library (blingfire_amd/libblingfiretokdll.so)
 * never does: it computes on the GPU or fails loudly.
 *
 * Parity status: PINNED.  tests/test_oracle.py checks this restatement
 *  (a) against every known-answer vector the reference holds for the path
 *      (README.md:111-141, README.md:232-268, blingfiretokdll.cpp:1103-1106,
 *      blingfiretokdll.cpp:1341-1347, ldbsrc/gpt2/README.TXT:40-47), and
 *  (b) differentially against oracle/_ref/libblingfiretokdll_ref.so, the
 *      unmodified reference compiled from /root/reference by oracle/Makefile.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference; "tokdll" = blingfiretools/blingfiretokdll/blingfiretokdll.cpp,
 * "cl" = blingfireclient.library).
 */
#ifndef BF_ORACLE_H
#define BF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bfo_model bfo_model;

/* tokdll:1077-1094 LoadModel (returns NULL instead of throwing on a bad path) */
bfo_model *bfo_load_model(const char *path);
/* tokdll:1056-1070 SetModel: the image is copied (the reference borrows it) */
bfo_model *bfo_set_model(const unsigned char *img, size_t size);
/* tokdll:1650-1662 FreeModel */
int bfo_free_model(bfo_model *m);

/* tokdll:1619-1646 TextToIds (dispatches to _wp / _sp exactly like the reference) */
int bfo_text_to_ids(const bfo_model *m, const char *utf8, int n,
                    int32_t *ids, int max_ids, int unk);

/* tokdll:1562-1609 TextToIdsWithOffsets: byte offsets (inclusive ends) of every id in the original string */
int bfo_text_to_ids_with_offsets(const bfo_model *m, const char *utf8, int n,
                                 int32_t *ids, int *starts, int *ends, int max_ids, int unk);

/* tokdll:415-566 TextToWordsWithOffsetsWithModel (the model is explicit here; the reference's NULL = its built-in wbd.bin) */
int bfo_text_to_words_with_offsets(const bfo_model *m, const char *utf8, int n, char *out, int *starts, int *ends, int max_out);

/* tokdll:163-355 TextToSentencesWithOffsetsWithModel (the model is explicit here; the reference's NULL = its built-in sbd.bin) */
int bfo_text_to_sentences_with_offsets(const bfo_model *m, const char *utf8, int n, char *out, int *starts, int *ends, int max_out);

/* tokdll:1689-1745 IdsToText (model: a .bin with an [i2w] section or a separate .i2w file) */
int bfo_ids_to_text(const bfo_model *m, const int32_t *ids, int n, char *out, int max_out, int skip_special);
int bfo_has_i2w(const bfo_model *m);
int bfo_i2w_count(const bfo_model *m);

/* tokdll:629-679 NormalizeSpaces and tokdll:683-815 TextToHashes (both model-free) */
int bfo_normalize_spaces(const char *utf8, int n, char *out, int max_out, int u_space);
int bfo_text_to_hashes(const char *utf8, int n, int32_t *out, int max_out, int ngrams, int bucket);

/* tokdll:1669-1679 SetNoDummyPrefix */
int bfo_set_no_dummy_prefix(bfo_model *m, int flag);

/* ---- building blocks, exported so tests can prove GPU-table equivalence ---- */

/* which: 0 = [wbd] Moore/RS DFA, 1 = [pos-dict] RS/Mealy DFA */
int bfo_has_dfa(const bfo_model *m, int which);
/* cl/src/FARSDfa_pack_triv.cpp:92-95 */
int bfo_dfa_initial(const bfo_model *m, int which);
/* cl/src/FARSDfa_pack_triv.cpp:128-138 */
int bfo_dfa_is_final(const bfo_model *m, int which, int state);
/* cl/src/FARSDfa_pack_triv.cpp:141-399 */
int bfo_dfa_dest(const bfo_model *m, int which, int state, int iw);
/* cl/src/FAState2Ow_pack_triv.cpp:34-130 */
int bfo_state2ow(const bfo_model *m, int state);
/* cl/src/FAMealyDfa_pack_triv.cpp:69-244 */
int bfo_mealy_dest_ow(const bfo_model *m, int state, int iw, int *ow);
/* cl/inc/FAIwMap_pack.h:55-110 (class of a code point in the [wbd] DFA, -1 = none) */
int bfo_wbd_iw_class(const bfo_model *m, int iw);
/* cl/src/FAMultiMap_pack.cpp:106-126: lexer action vector; returns count or -1 */
int bfo_wbd_action(const bfo_model *m, int key, int *out, int max_out);
/* cl/src/FAMultiMap_pack_fixed.cpp:67-137 applied to the model's charmap
 * (which: 0 = [wbd] charmap, 1 = [pos-dict] charmap); returns count or -1 */
int bfo_charmap_get(const bfo_model *m, int which, int key, int *out, int max_out);
/* cl/src/FAMultiMap_pack_fixed.cpp:140-160 on I2Info: returns count, fills id/score bits */
int bfo_i2info_get(const bfo_model *m, int key, int *id, uint32_t *score_bits);

/* cl/inc/FADictInterpreter_t.h:334-366 / :369-390 over the model's [pos-dict] (word = int code points, like the reference's
 * Ty = int instantiation; no transformation, as tokdll would configure it): info id or -1; value count or -1 */
int bfo_dict_get_info_id(const bfo_model *m, const int *word, int n);
int bfo_dict_get_info(const bfo_model *m, const int *word, int n, int *out, int max_out);

/* model facts (for tests / reports) */
int bfo_model_kind(const bfo_model *m);      /* 0 = _wp lexer, 1 = unigram, 2 = bpe, 3 = bpe-opt, 4 = bpe-with-merges */
int bfo_model_uses_bytes(const bfo_model *m);
int bfo_model_id_offset(const bfo_model *m);

/* cl/src/FAUtf8Utils.cpp:233-270 FAStrUtf8ToArray: returns count or -1 */
int bfo_utf8_to_utf32(const char *s, int len, int *out, int max_out);

/* cl/inc/FALexTools_t.h:403-421 Process on an UTF-32 array (after normalisation);
 * writes <tag,from,to> triples, returns number of ints written or -1 */
int bfo_lex_process(const bfo_model *m, const int *in, int n, int *out, int max_out);
/* cl/src/FAUtf32Utils.cpp:45-81 FAUtf32ToLower (the fold of ignore-case lexers and dictionaries) */
int bfo_tolower_sym(int cp);

#ifdef __cplusplus
}
#endif
#endif
