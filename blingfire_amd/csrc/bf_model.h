// bf_model.h -- host-side model loader: BlingFire .bin image -> GPU table layout.
//
// The .bin format is consumed unchanged (format spec read from the reference:
// blingfirecompile.library/inc/FADfaPack_triv.h:27-96, blingfireclient.library/src/FALDB.cpp:24-64).
// Nothing here walks the packed bytes at tokenisation time: LoadModel decodes every
// automaton ONCE into an abstract (state, symbol) -> (dst, final, ow) relation and
// re-lays it out as a displacement-packed ("comb") transition table, so that one DFA
// step on the GPU is ONE gather:   e = T[state + class];  hit <=> e.cls == class.
// State ids are the displacement bases, which are unique per state.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace bfa {

// ---- enum values baked into .bin files (reference blingfireclient.library/inc/FAFsmConst.h:67-80,152-273,364-371,402-415)
enum : int {
    IW_ANY = 0, IW_L_ANCHOR = 1, IW_R_ANCHOR = 2, IW_EPSILON = 3,
    DFA_DEAD_STATE = -2,
    TRS_NONE = 0, TRS_RANGE = 1, TRS_IMPL = 2, TRS_PARA = 4, TRS_IWIA = 6,
    FUNC_POS_DICT = 12, FUNC_WBD = 19, FUNC_GLOBAL = 20,
    PARAM_DIRECTION = 11, PARAM_FSM = 2, PARAM_MAP_MODE = 16, PARAM_NO_TR = 18, PARAM_IGNORE_CASE = 22,
    PARAM_ARRAY = 24, PARAM_MULTI_MAP = 25, PARAM_FSM_TYPE = 26, PARAM_DEPTH = 38, PARAM_CHARMAP = 47,
    PARAM_MAX_LENGTH = 69, PARAM_VERIFY_LDB_BIN = 70, PARAM_TOKENIZATION_TYPE = 71, PARAM_ID_OFFSET = 72,
    PARAM_USE_BYTE_ENCODING = 73, PARAM_NO_DUMMY_PREFIX = 74,
    MODE_PACK_TRIV = 1, MODE_PACK_FIXED = 3, TYPE_MOORE_DFA = 3, TYPE_MOORE_MULTI_DFA = 4, TYPE_MEALY_DFA = 7,
    TOKENIZE_BPE = 3, TOKENIZE_BPE_OPT = 4, TOKENIZE_BPE_OPT_WITH_MERGES = 5,
};

// kinds of TextToIds algorithm a model selects (reference tokdll:959-974, 1619-1646)
enum ModelKind : int { KIND_WP = 0, KIND_UNIGRAM = 1, KIND_BPE = 2, KIND_BPE_OPT = 3, KIND_BPE_MERGES = 4, KIND_I2W = 5 /* [i2w] only: IdsToText */ };

// Abstract automaton decoded from a packed dump (test hooks compare it with the oracle's readers).
struct RawDfa {
    int initial = -1;                  // state index
    bool remap = false;                // symbols are equivalence classes of the dump's own Iw map
    std::vector<int> state_off;        // state index -> byte offset in the dump (the reference's state id)
    std::vector<uint8_t> is_final;
    std::vector<int> ow;               // Moore output weight (action id) or -1
    std::vector<uint32_t> tr_begin;    // CSR over transitions, size = nstates + 1
    std::vector<int> tr_sym;           // symbol (class if remap, raw Iw otherwise), ascending per state
    std::vector<int> tr_dst;           // destination state index, or DFA_DEAD_STATE
    std::vector<int> tr_ow;            // Mealy output weight per transition (empty for Moore)
    // Iw map of the dump (remap only): intervals [from,to] -> class+1 values
    std::vector<int> iw_from, iw_to; std::vector<std::vector<int>> iw_cls;   // class or -1 per code point
    int class_of(int iw) const;        // -1 = not in alphabet
    int find_state(int off) const;     // byte offset -> index or -1
    std::vector<std::pair<int,int>> off_index; // sorted (off, idx)
};

// Two-level code-point map: l1[cp >> 8] -> page; pages[page*256 + (cp & 255)] -> value.
// Page 0 is the all-default page.  Covers 0 .. 0x10FFFF (4352 l1 entries).
struct TwoLevelMap {
    std::vector<uint16_t> l1;
    std::vector<uint32_t> pages;
    uint32_t def = 0;
    void init(uint32_t default_value);
    void set(int cp, uint32_t v);
    uint32_t get(int cp) const { return (cp < 0 || cp > 0x10FFFF) ? def : pages[(size_t)l1[cp >> 8] * 256 + (cp & 255)]; }
};

constexpr uint32_t CLS_NONE = 0x1FFEu;        // "code point not in the alphabet" in lexer class streams (= bf_lex.h LX_CLS_NONE)

// Displacement-packed transition table.
//  T32 entry (lexer / Moore):  [12:0] class, [13] dst-is-final, [31:14] dst base
//  T64 entry (dictionary / Mealy): [19:0] class, [20] dst-is-final, [41:21] dst base, [63:42] output weight
struct PackedDfa {
    bool wide = false;                 // false: t32, true: t64
    std::vector<uint32_t> t32;
    std::vector<uint64_t> t64;
    std::vector<uint32_t> state_base;  // state index -> base (unique)
    uint32_t dead_base = 0;            // base of the synthetic dead state
    uint32_t leaf_lo = 0, leaf_n = 0;  // wide tables: the states without transitions have the bases [leaf_lo, leaf_lo + leaf_n)
    uint32_t initial_base = 0;
    int nclasses = 0;                  // classes are 0 .. nclasses-1
    std::vector<int> sym_of_class;     // non-remap: class -> raw symbol
    uint32_t table_len() const { return (uint32_t)(wide ? t64.size() : t32.size()); }
    // host-side lookup identical to the device step (for equivalence tests): returns dst base or -1 (miss)
    long step(uint32_t base, uint32_t cls, int *final_out, int *ow_out) const;
};

constexpr uint32_t T32_CLS_MASK = 0x1FFFu, T32_FINAL_BIT = 1u << 13;
constexpr int T32_NEXT_SHIFT = 14;
constexpr uint64_t T64_CLS_MASK = 0xFFFFFull, T64_FINAL_BIT = 1ull << 20;
constexpr int T64_NEXT_SHIFT = 21, T64_OW_SHIFT = 42;
constexpr uint64_t T64_NEXT_MASK = 0x1FFFFFull;

// lexer action record in acts_pool: [left, right, tag, nfn, (fn_id, fn_initial_base, base after the left anchor) * nfn]
constexpr uint32_t INFO_SIMPLE_BIT = 0x80000000u; // info[base] = SIMPLE | tag   (left=right=0, no functions, tag != 0)

struct Model {
    std::vector<uint8_t> image;        // private copy of the .bin
    std::vector<size_t> dump_off;
    int kind = KIND_WP;
    std::string error;                 // why load failed

    // ---- [i2w] id -> token text (reference tokdll:998-1045, FAStringArray_pack.cpp:23-50), for IdsToText
    bool has_i2w = false;
    std::vector<uint32_t> i2w_off;     // [count + 1] byte offsets into i2w_data
    std::vector<uint8_t> i2w_data;
    int min_token_id = 0, max_token_id = 1000000000;   // regular (non special) ids (FALimits::MaxArrSize default)

    // ---- [wbd] lexer (reference FAWbdConfKeeper.cpp:56-232, FALexTools_t.h:129-202)
    bool has_wbd = false;
    int max_depth = 2, max_token_length = 300; bool ignore_case = false;
    bool lexer_void = false;           // a moore-multi-dfa [wbd]: the reference's lexer answers -1 to every input (bf_model.cpp)
    int lex_frames = 0;                // saved frames the call graph can need: min(max_depth, call depth) - 1
    RawDfa wbd_raw; PackedDfa wbd;
    std::vector<uint32_t> wbd_info;    // indexed by base: action info for final states
    std::vector<uint64_t> wbd_t2;      // device form: low = bf_layout.h entry, high = wbd_info[destination] when the destination is final
    std::vector<int32_t> acts_pool;
    uint32_t cls_any = CLS_NONE, cls_l = CLS_NONE, cls_r = CLS_NONE;
    uint32_t initial_l = 0xFFFFFFFFu;  // state after the left anchor from the initial state (0xFFFFFFFF: none)
    // "loop state" of the lexer (bf_lex.h LexTables::loop_state): the state with the most self-loop transitions (>= 8), e.g.
    // the state of `(AllLetters)+` after its first letter in the BERT lexers (774 self-loops).  Classes that loop there carry
    // LX_C_LOOP in the fused code-point maps below (and so in the class stream); 0xFFFFFFFF = the lexer has none.
    uint32_t loop_base = 0xFFFFFFFFu, loop_info = 0; bool loop_final = false;
    std::vector<uint8_t> loop_cls;     // [nclasses] 1 = self-loop of the loop state
    bool fn_no_ra = false;             // bf_lex.h LexTables::fn_no_ra: no right-anchor transition below any function's initial states
    bool two_level = false;            // bf_lex.h LexTables::two_level: calls are one level deep, one function per action, no contexts
    // fused "code point -> charmap -> (cp<3 ? 3 : cp) -> class" map:
    //   value = CLS_NONE | class (count 1 implicit) or FUSED_MULTI | pool offset for 0 or 2..10 outputs
    TwoLevelMap wbd_cpmap;
    std::vector<uint16_t> wbd_multi_pool;   // [count, cls0, cls1 ...] records for 1:n / deleted chars
    bool wbd_has_charmap = false, wbd_charmap_multi = false;
    // "Unit form" (bf_wave.h): the lexer is two-level, every top-level token is a unit of its own for the WordPiece post-pass
    // (top-level tags in 1..4, vocabulary tags > 4, no left anchor at the top level, no right-anchor transition behind a letter,
    // no IW_ANY), so top-level tokens can be found for many start positions at once and every word can be looked up by a lane
    // of its own.  wave_kind[class]: what a walk that STARTS on this class does (bf_wave.h WK_*), proven from the automaton.
    bool wave_ok = false; std::string wave_why;        // wave_why: the first condition that failed (diagnostics, tests)
    bool bpe_wave_ok = false;          // bpe-opt or merge-rank model whose entries never contain U+2581 behind their first symbol: bf_bpe_wave_body.h applies
    std::vector<uint8_t> wave_kind;
    uint32_t wave_solo_info = 0;                       // action info of every WK_SOLO token
    // "Flat form" (bf_flat.h): a unit-form lexer whose run and solo tokens both call ONE vocabulary function.  The whole-word answers of that
    // function -- every word of <= WF_KEY_CHARS characters over the classes 0 .. 126 that the walk from the function's first state consumes
    // entirely and ends on a final state (= the word is ONE piece: FALexTools_t.h:255-277 keeps the LAST final position, tokdll:1239-1301
    // then has a single sub-token that tiles the word) -- are enumerated at load into a two-choice hash table keyed by the word itself
    // (one byte per character, bf_flat_key.h: the key IS the word, a hit needs no second check).  flat_tab: entry e = [2e] k0, [2e + 1] k1 | id << 32; k0 0 = empty.
    bool flat_ok = false; uint32_t flat_ini = 0, flat_ini_l = 0xFFFFFFFFu;
    std::vector<uint64_t> flat_tab; int flat_bits = 0; uint32_t flat_m0 = 0, flat_m1 = 0, flat_m2 = 0; int flat_words = 0;
    // the word table of the BPE wave program (bf_model.cpp build_bpe_word_table): words that the bpe-opt collection takes whole, keyed by their symbols
    std::vector<uint64_t> bpe_tab; int bpe_tab_bits = 0; uint32_t bpe_tab_m0 = 0, bpe_tab_m1 = 0, bpe_tab_m2 = 0; int bpe_tab_words = 0;
    // TextToWords view of the same lexer (tokdll:415-566): NO charmap, U+0000 is fed as U+0020 -> plain code point -> class map
    TwoLevelMap words_cpmap;

    // ---- [pos-dict] segmenters (reference FADictConfKeeper.cpp:57-228)
    bool has_seg = false;
    int tok_algo = 0, id_offset = 0; bool use_bytes = false, no_dummy_prefix = false;
    RawDfa dict_raw; PackedDfa dict;
    std::vector<int32_t> i2info_id; std::vector<uint32_t> i2info_score;  // I2Info rows (key = MPH index)
    std::vector<uint8_t> i2info_valid;
    int i2info_min_key = 0;
    // "code point -> charmap" map for the _sp prologue: value = NORM_NONE (copy), or count<<24 | (value | pool offset)
    TwoLevelMap dict_charmap; std::vector<int32_t> dict_norm_pool; bool dict_has_charmap = false;
    bool dict_ignore_case = false; TwoLevelMap dict_lookup_map;      // ignore-case [pos-dict]: key symbol -> charmap entry of its fold (entries as in dict_charmap), key lookup only
    TwoLevelMap dict_clsmap;           // code point (or byte) -> class of the dictionary alphabet, CLS_NONE_W if absent
    int trie_max_depth = 0;            // longest path from the initial state (bounds every arc length)
    int max_info_id = 0;               // largest token id in I2Info (the Unigram lane program packs id + 1 into 20 bits)
    // ---- key -> info lookup over the same [pos-dict] (reference FADictInterpreter_t.h:334-390; additive DictGetInfoBatch)
    int dict_direction = 0;            // PARAM_DIRECTION: 0 = l2r (FAFsmConst.h DIR_L2R); else keys are normalised and reversed
    std::vector<int32_t> k2i;          // K2I array, decoded (reference FAArray_pack.cpp:27-95): MPH index -> info id
    std::vector<int32_t> info_rows;    // I2Info rows, decoded: row r = [count or -1, values ...], info_stride ints each
    int info_stride = 0, info_min_key = 0;
    // device forms
    std::vector<uint64_t> seg_info;    // I2Info rows as (id | score_bits << 32), key = MPH index
    std::vector<uint32_t> seg_score;   // their score bits alone (Unigram lane program, bf_seg.h SegTables::score)
    // bf_bpe_seg_body.h: the arc order of the BPE flavours as one integer per entry.  Plain BPE / bpe-opt sort by id (..._bpe_t.h:238-255):
    // priority = 2 * id + 1, no table.  With merges (..._with_merges_t.h:242-262: rank descending, then id ascending): bpe_prio[MPH index]
    // = 2 * place + 1, place = the entry's position in that order; bpe_place_id[place] = its id.  Even priorities are for the unknown arc
    // of a call (bpe_unk_prio).  bpe_seg_ok: ids and depth fit the arc keys (ids in [0, 2^20), entries of <= 256 symbols).
    std::vector<uint32_t> bpe_prio; std::vector<int32_t> bpe_place_id; std::vector<uint32_t> bpe_place_rank;   // bpe_place_rank: rank bits of the entry at a place
    int bpe_prio_bits = 0; bool bpe_seg_ok = false;
    // fused "code point (or byte) -> charmap -> element code" map of the _sp prologue.  Element codes (u16):
    //   class of the dictionary alphabet | SP_NONE (not in alphabet) | SP_WS (whitespace, tokdll.h:17-21) |
    //   SP_DELIM_ABSENT (U+2581 when the alphabet lacks it).  Value = code, or FUSED_MULTI | pool offset
    //   ([count, codes...]) for 0 or 2..10 outputs.
    TwoLevelMap sp_cpmap; std::vector<uint16_t> sp_multi_pool; bool sp_has_multi = false;
    uint16_t sp_delim_code = 0xFFFE;   // element code of U+2581
    std::vector<uint16_t> sp_prefix;   // element codes of the normalised dummy prefix (tokdll:1372)
};

constexpr uint32_t FUSED_MULTI = 0x80000000u;
constexpr uint32_t NORM_NONE = 0xFFFFFFFFu;
constexpr uint32_t CLS_NONE_W = 0xFFFFFu;     // T64 class "none"
constexpr uint16_t SP_NONE = 0xFFFF, SP_DELIM_ABSENT = 0xFFFE, SP_WS = 0xFFFD;

// Parses and re-lays-out a model image.  Returns false and sets m.error on failure.
bool build_model(Model &m, const uint8_t *img, size_t size);
bool load_file(const char *path, std::vector<uint8_t> &out);
// priority of the unknown arc (id UnkId, rank 0.0f) among the entries' priorities: where the reference's comparator sorts it
uint32_t bpe_unk_prio(const Model &m, int unk);

} // namespace bfa
