/*
 * bf_oracle.c -- TEST INFRASTRUCTURE ONLY.  See bf_oracle.h.
 *
 * Plain-C restatement of the reference TextToIds path.  It walks the packed
 * .bin image byte by byte the way the reference readers do; it shares no code
 * with the GPU product (which re-lays the tables out), so agreement between
 * the two is evidence, not tautology.  Citations are into /root/reference.
 */
#include "bf_oracle.h"

#include <float.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- constants: cl/inc/FAFsmConst.h:67-80,152-273,364-371,402-415 ---- */
enum {
    IW_ANY = 0, IW_L_ANCHOR = 1, IW_R_ANCHOR = 2, IW_EPSILON = 3,
    DFA_DEAD_STATE = -2,
    TRS_NONE = 0, TRS_RANGE = 1, TRS_IMPL = 2, TRS_PARA = 4, TRS_IWIA = 6,
    FUNC_POS_DICT = 12, FUNC_WBD = 19, FUNC_GLOBAL = 20,
    PARAM_FSM = 2, PARAM_MAP_MODE = 16, PARAM_IGNORE_CASE = 22, PARAM_ARRAY = 24,
    PARAM_MULTI_MAP = 25, PARAM_FSM_TYPE = 26, PARAM_DEPTH = 38, PARAM_CHARMAP = 47,
    PARAM_MAX_LENGTH = 69, PARAM_VERIFY_LDB_BIN = 70, PARAM_TOKENIZATION_TYPE = 71,
    PARAM_ID_OFFSET = 72, PARAM_USE_BYTE_ENCODING = 73, PARAM_NO_DUMMY_PREFIX = 74,
    MODE_PACK_TRIV = 1, MODE_PACK_MPH = 2, MODE_PACK_FIXED = 3,
    TYPE_MOORE_DFA = 3, TYPE_MEALY_DFA = 7,
    TOKENIZE_BPE = 3, TOKENIZE_BPE_OPT = 4, TOKENIZE_BPE_OPT_WITH_MERGES = 5,
    MAX_ARR_SIZE = 1000000000,  /* cl/inc/FALimits.h:26 */
    MAX_WORD_LEN = 300,         /* cl/inc/FALimits.h:35 */
    WBD_WORD_TAG = 1, WBD_IGNORE_TAG = 4, /* tokdll:39-40 */
    SP_DELIM = 0x2581,          /* blingfiretokdll.h:12 */
    MIN_ACT_SIZE = 3            /* cl/inc/FALexTools_t.h:109 */
};

/* unaligned little-endian readers (the reference casts pointers; dumps may be
 * unaligned, cl/inc/FAEncodeUtils.h:211 warns) */
static int rd_i32(const uint8_t *p) { int32_t v; memcpy(&v, p, 4); return v; }
static unsigned rd_u32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static unsigned rd_u16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
static int rd_i16(const uint8_t *p) { int16_t v; memcpy(&v, p, 2); return v; }

/* cl/inc/FAEncodeUtils.h:292-310 FADecode_UC_US_UI (native little-endian) */
static unsigned dec_uc_us_ui(const uint8_t *p, int size)
{
    if (size == 1) return p[0];
    if (size == 2) return rd_u16(p);
    return rd_u32(p);
}

/* cl/inc/FAEncodeUtils.h:418-451 FADecode_1_2_3_4_idx (big-endian) */
static unsigned dec_1234_idx(const uint8_t *p, unsigned idx, int size)
{
    if (size == 1) return p[idx];
    if (size == 2) { p += 2u * idx; return ((unsigned)p[0] << 8) | p[1]; }
    if (size == 3) { p += 3u * idx; return ((unsigned)p[0] << 16) | ((unsigned)p[1] << 8) | p[2]; }
    p += 4u * idx;
    return ((unsigned)p[0] << 24) | ((unsigned)p[1] << 16) | ((unsigned)p[2] << 8) | p[3];
}

/* cl/inc/FAEncodeUtils.h:456-501 FADecodeDst_idx: all-ones = dead state */
static int dec_dst_idx(const uint8_t *p, int idx, int size)
{
    unsigned v = dec_1234_idx(p, (unsigned)idx, size);
    if (size == 3 && v == 0x00ffffffu) return DFA_DEAD_STATE;
    if (size == 4 && v == 0xffffffffu) return DFA_DEAD_STATE;
    if (size == 2 && v == 0x0000ffffu) return DFA_DEAD_STATE;
    if (size == 1 && v == 0x000000ffu) return DFA_DEAD_STATE;
    return (int)v;
}

/* cl/inc/FAUtils_cl.h:85-137 FAFind_log on a sorted-unique array of `size`-byte
 * unsigned elements: index of val or -1 (the index==value shortcut and the
 * binary/linear split are pure optimisations) */
static int find_exact(const uint8_t *arr, int count, int size, unsigned val)
{
    int lo = 0, hi = count - 1;
    while (lo <= hi) {
        int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
        unsigned cur = dec_uc_us_ui(arr + (size_t)mid * size, size);
        if (cur == val) return mid;
        if (val < cur) hi = mid - 1; else lo = mid + 1;
    }
    return -1;
}

/* cl/inc/FAUtils_cl.h:141-201 FAFindEqualOrLess_log: last index with arr[i] <= val, or -1 */
static int find_eq_or_less_u(const uint8_t *arr, int count, int size, unsigned val)
{
    int lo = 0, hi = count - 1, res = -1;
    while (lo <= hi) {
        int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
        unsigned cur = dec_uc_us_ui(arr + (size_t)mid * size, size);
        if (cur <= val) { res = mid; lo = mid + 1; } else hi = mid - 1;
    }
    return res;
}
static int find_eq_or_less_i32(const uint8_t *arr, int count, int val)
{
    int lo = 0, hi = count - 1, res = -1;
    while (lo <= hi) {
        int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
        int cur = rd_i32(arr + 4 * (size_t)mid);
        if (cur <= val) { res = mid; lo = mid + 1; } else hi = mid - 1;
    }
    return res;
}

/* ---------------- packed containers ---------------- */

/* cl/inc/FAChains_pack_triv.h:81-308, cl/src/FAChains_pack_triv.cpp:19-29 */
typedef struct { const uint8_t *img; int size_of_value; int max_count; } chains_t;
static void chains_set(chains_t *c, const uint8_t *img)
{
    c->img = img; c->size_of_value = rd_i32(img); c->max_count = rd_i32(img + 4);
}
/* UnPack(Offset, pValues, MaxCount) cl/inc/FAChains_pack_triv.h:87-141 */
static int chains_unpack(const chains_t *c, int off, int *out, int max_out)
{
    const uint8_t *p = c->img + off;
    int count, i;
    if (c->size_of_value == 1) {
        count = (int8_t)p[0];
        if (out && max_out >= count) for (i = 0; i < count; ++i) out[i] = (int8_t)p[1 + i];
    } else if (c->size_of_value == 2) {
        count = rd_i16(p);
        if (out && max_out >= count) for (i = 0; i < count; ++i) out[i] = rd_i16(p + 2 + 2 * i);
    } else {
        count = rd_i32(p);
        if (out && max_out >= count) for (i = 0; i < count; ++i) out[i] = rd_i32(p + 4 + 4 * i);
    }
    return count;
}
/* UnPack(Offset, Idx) cl/inc/FAChains_pack_triv.h:166-230: element or -1 */
static int chains_unpack_idx(const chains_t *c, int off, int idx)
{
    const uint8_t *p = c->img + off;
    if (c->size_of_value == 1) { int n = (int8_t)p[0]; return idx < n ? (int8_t)p[1 + idx] : -1; }
    if (c->size_of_value == 2) { int n = rd_i16(p); return idx < n ? rd_i16(p + 2 + 2 * idx) : -1; }
    { int n = rd_i32(p); return idx < n ? rd_i32(p + 4 + 4 * idx) : -1; }
}

/* cl/src/FAMultiMap_pack.cpp:22-126 */
typedef struct { const uint8_t *offsets; unsigned max_key; int size_of_offset; chains_t values; int set; } mmap_t;
static void mmap_set(mmap_t *m, const uint8_t *dump)
{
    unsigned off = 0;
    m->max_key = rd_u32(dump); off += 4;
    m->size_of_offset = (int)rd_u32(dump + off); off += 4;
    m->offsets = dump + off;
    off += (unsigned)m->size_of_offset * (1 + m->max_key);
    if (off % 4) off += 4 - off % 4;                 /* FAMultiMap_pack.cpp:44-49 */
    chains_set(&m->values, dump + off);
    m->set = 1;
}
static int mmap_get(const mmap_t *m, int key, int *out, int max_out)
{
    unsigned vo;
    if (!(0 <= key && m->max_key >= (unsigned)key)) return -1;
    vo = dec_1234_idx(m->offsets, (unsigned)key, m->size_of_offset);
    if (vo == 0) return -1;
    return chains_unpack(&m->values, (int)(vo - 1), out, max_out);
}

/* cl/src/FAMultiMap_pack_fixed.cpp:25-160 */
typedef struct { const uint8_t *data; int size_of_value, max_count, min_key, max_key, size_of_arr; int set; } mmapf_t;
static void mmapf_set(mmapf_t *m, const uint8_t *dump)
{
    m->size_of_value = (int)rd_u32(dump);
    m->max_count = rd_i32(dump + 4);
    m->size_of_arr = (m->max_count + 1) * m->size_of_value;
    m->min_key = rd_i32(dump + 8);
    m->max_key = rd_i32(dump + 12);
    m->data = dump + 16;
    m->set = 1;
}
/* Get(Key, pValues, MaxCount) FAMultiMap_pack_fixed.cpp:67-137 */
static int mmapf_get(const mmapf_t *m, int key, int *out, int max_out)
{
    const uint8_t *arr;
    int count, i;
    if (!(m->min_key <= key && key <= m->max_key)) return -1;
    arr = m->data + (unsigned)m->size_of_arr * (unsigned)(key - m->min_key); /* 32-bit unsigned :79 */
    if (m->size_of_value == 1) {
        count = (int8_t)arr[0];
        if (count > m->max_count) return -1;
        if (out && max_out >= count) for (i = 0; i < count; ++i) out[i] = (int8_t)arr[1 + i];
    } else if (m->size_of_value == 2) {
        count = rd_i16(arr);
        if (count > m->max_count) return -1;
        if (out && max_out >= count) for (i = 0; i < count; ++i) out[i] = rd_i16(arr + 2 + 2 * i);
    } else {
        count = rd_i32(arr);
        if (count > m->max_count) return -1;
        if (out && max_out >= count) for (i = 0; i < count; ++i) out[i] = rd_i32(arr + 4 + 4 * i);
    }
    return count;
}

/* cl/inc/FAIwMap_pack.h:55-110, cl/src/FAIwMap_pack.cpp:35-88 (the 65,535-entry
 * cache is memoisation of the same function and is not restated) */
typedef struct { int size_of_new_iw, interval_count; const uint8_t *from, *to_off, *new_iws; } iwmap_t;
static void iwmap_set(iwmap_t *m, const uint8_t *img)
{
    unsigned off = 0;
    m->size_of_new_iw = rd_i32(img); off += 4;
    m->interval_count = rd_i32(img + off); off += 4;
    m->from = img + off; off += 4u * (unsigned)m->interval_count;
    m->to_off = img + off; off += 8u * (unsigned)m->interval_count;
    m->new_iws = img + off;
}
static int iwmap_get(const iwmap_t *m, int old_iw)
{
    int idx = find_eq_or_less_i32(m->from, m->interval_count, old_iw);
    int from_iw, end_iw, ioff;
    unsigned v;
    if (idx == -1) return -1;
    from_iw = rd_i32(m->from + 4 * (size_t)idx);
    end_iw = rd_i32(m->to_off + 8 * (size_t)idx);
    ioff = rd_i32(m->to_off + 8 * (size_t)idx + 4);
    if (old_iw > end_iw) return -1;
    v = dec_1234_idx(m->new_iws + ioff, (unsigned)(old_iw - from_iw), m->size_of_new_iw);
    return v ? (int)v - 1 : -1;
}

/* cl/src/FARSDfa_pack_triv.cpp:27-76 (+ Mealy view cl/src/FAMealyDfa_pack_triv.cpp:24-66) */
typedef struct {
    const uint8_t *img; int dst_size; int ows_offset; int remap; iwmap_t iwmap; int initial;
    chains_t ows; int set;
} dfa_t;
static void dfa_set(dfa_t *d, const uint8_t *img)
{
    unsigned off = 0, iwc;
    d->img = img;
    d->dst_size = rd_i32(img); off += 4;
    if (d->dst_size < 1 || d->dst_size > 4) d->dst_size = 3;
    d->ows_offset = rd_i32(img + off); off += 4;
    iwc = rd_u32(img + off); off += 4;
    d->remap = (iwc & 0x80000000u) != 0;
    off += 4u * (iwc & 0x7fffffffu);
    if (d->remap) {
        int sz = rd_i32(img + off); off += 4;
        iwmap_set(&d->iwmap, img + off);
        off += (unsigned)sz;
    }
    d->initial = (int)off;
    if (d->ows_offset) chains_set(&d->ows, img + d->ows_offset);
    d->set = 1;
}

/* cl/src/FARSDfa_pack_triv.cpp:128-138 */
static int dfa_is_final(const dfa_t *d, int state)
{
    if (state < 0) return 0;
    return (d->img[state] & 0x80) != 0;
}

/* cl/src/FARSDfa_pack_triv.cpp:141-399 */
static int dfa_dest(const dfa_t *d, int state, int iw)
{
    const uint8_t *p;
    unsigned info;
    int iw_size, tr, new_iw, idx;
    if (state < 0) return -1;
    if (d->remap) { new_iw = iwmap_get(&d->iwmap, iw); if (new_iw == -1) return -1; }
    else new_iw = iw;
    p = d->img + state;
    info = *p++;
    iw_size = (int)((info & 0x18) >> 3) + 1;
    tr = (int)(info & 7);
    switch (tr) {
    case TRS_PARA: {
        unsigned cnt;
        if (iw_size == 1 && (0xFFFFFF00u & (unsigned)new_iw)) return -1;
        if (iw_size == 2 && (0xFFFF0000u & (unsigned)new_iw)) return -1;
        cnt = 1 + dec_uc_us_ui(p, iw_size); p += iw_size;
        idx = find_exact(p, (int)cnt, iw_size, (unsigned)new_iw);
        p += (size_t)cnt * iw_size;
        if (idx == -1) return -1;
        return dec_dst_idx(p, idx, d->dst_size);
    }
    case TRS_IWIA: {
        unsigned base = dec_uc_us_ui(p, iw_size), mx; int dst;
        p += iw_size; mx = dec_uc_us_ui(p, iw_size); p += iw_size;
        if (new_iw < (int)base || new_iw > (int)mx) return -1;
        dst = dec_dst_idx(p, new_iw - (int)base, d->dst_size);
        return dst == 0 ? -1 : dst;
    }
    case TRS_RANGE: {
        unsigned cnt;
        if (iw_size == 1 && (0xFFFFFF00u & (unsigned)new_iw)) return -1;
        if (iw_size == 2 && (0xFFFF0000u & (unsigned)new_iw)) return -1;
        cnt = 1 + dec_uc_us_ui(p, iw_size); p += iw_size;
        idx = find_eq_or_less_u(p, (int)cnt, iw_size, (unsigned)new_iw);
        if (idx == -1) return -1;
        p += (size_t)cnt * iw_size;
        if (dec_uc_us_ui(p + (size_t)idx * iw_size, iw_size) < (unsigned)new_iw) return -1;
        p += (size_t)cnt * iw_size;
        return dec_dst_idx(p, idx, d->dst_size);
    }
    case TRS_IMPL: {
        int ow_code = (int)((info & 0x60) >> 5);
        int ow_size = ow_code == 3 ? 4 : ow_code;
        if (iw_size == 1) { if (new_iw == (int)p[0]) return state + 1 + 1 + ow_size; }
        else if (iw_size == 2) { if (new_iw == (int)rd_u16(p)) return state + 1 + 2 + ow_size; }
        else { if ((unsigned)new_iw == rd_u32(p)) return state + 1 + 4 + ow_size; }
        return -1;
    }
    default: return -1;
    }
}

/* cl/src/FAState2Ow_pack_triv.cpp:34-130 */
static int dfa_state2ow(const dfa_t *d, int state)
{
    const uint8_t *p = d->img + state;
    unsigned info = *p;
    int ow_code = (int)((info & 0x60) >> 5), iw_size, tr;
    if (ow_code == 0) return -1;
    p++;
    iw_size = (int)((info & 0x18) >> 3) + 1;
    tr = (int)(info & 7);
    switch (tr) {
    case TRS_PARA: { unsigned c = dec_uc_us_ui(p, iw_size); p += iw_size; p += (size_t)(c + 1) * (d->dst_size + iw_size); break; }
    case TRS_IWIA: { unsigned b = dec_uc_us_ui(p, iw_size), m; p += iw_size; m = dec_uc_us_ui(p, iw_size); p += iw_size;
                     p += (size_t)d->dst_size * (m - b + 1); break; }
    case TRS_RANGE: { unsigned c = dec_uc_us_ui(p, iw_size); p += iw_size; p += (size_t)(c + 1) * (d->dst_size + 2 * iw_size); break; }
    case TRS_IMPL: p += iw_size; break;
    default: break;
    }
    if (ow_code == 1) return (int8_t)p[0];
    if (ow_code == 2) return rd_i16(p);
    return rd_i32(p);
}

/* cl/src/FAMealyDfa_pack_triv.cpp:69-244 (no Iw remap, :56-57) */
static int mealy_dest_ow(const dfa_t *d, int state, int iw, int *pow)
{
    const uint8_t *p, *pows = NULL;
    unsigned info;
    int iw_size, ow_code, tr, idx, dest = -1;
    if (state < 0) return -1;
    p = d->img + state;
    info = *p++;
    iw_size = (int)((info & 0x18) >> 3) + 1;
    ow_code = (int)((info & 0x60) >> 5);
    tr = (int)(info & 7);
    switch (tr) {
    case TRS_PARA: {
        unsigned cnt;
        if (iw_size == 1 && (0xFFFFFF00u & (unsigned)iw)) return -1;
        if (iw_size == 2 && (0xFFFF0000u & (unsigned)iw)) return -1;
        cnt = 1 + dec_uc_us_ui(p, iw_size); p += iw_size;
        idx = find_exact(p, (int)cnt, iw_size, (unsigned)iw);
        p += (size_t)cnt * iw_size;
        if (idx == -1) return -1;
        if (ow_code != 0) pows = p + (size_t)d->dst_size * cnt;
        dest = dec_dst_idx(p, idx, d->dst_size);
        break;
    }
    case TRS_IMPL: {
        int ow_size = ow_code == 3 ? 4 : ow_code;
        idx = 0;
        if (iw_size == 1) { if (iw == (int)p[0]) { pows = p + 1; dest = state + 1 + 1 + ow_size; } else return -1; }
        else if (iw_size == 2) { if (iw == (int)rd_u16(p)) { pows = p + 2; dest = state + 1 + 2 + ow_size; } else return -1; }
        else { if ((unsigned)iw == rd_u32(p)) { pows = p + 4; dest = state + 1 + 4 + ow_size; } else return -1; }
        break;
    }
    default: return -1;    /* IWIA / RANGE unsupported for Mealy (:204-210), NONE has no arcs */
    }
    if (ow_code > 0 && pows) {
        int ows_off = ow_code == 1 ? (int8_t)pows[0] : ow_code == 2 ? rd_i16(pows) : rd_i32(pows);
        *pow = chains_unpack_idx(&d->ows, ows_off, idx);
    } else *pow = -1;
    return dest;
}

/* ---------------- model (tokdll:47-96 FAModelData, tokdll:918-1048 SetModelData) ---------------- */

struct bfo_model {
    uint8_t *img; size_t size;
    int dump_count; const uint8_t **dumps;
    mmap_t conf;                 /* cl/src/FALDB.cpp:24-64, dump 0 */
    /* [wbd] cl/src/FAWbdConfKeeper.cpp:56-232 */
    int has_wbd; dfa_t wbd_dfa; mmap_t acts; mmapf_t wbd_charmap;
    int max_depth, max_token_length, ignore_case /* [wbd] */, dict_ignore_case /* [pos-dict] */, lexer_void /* moore-multi-dfa [wbd] */;
    int *fn2ini; int fn2ini_size;
    /* [pos-dict] cl/src/FADictConfKeeper.cpp:57-228 */
    int has_seg; dfa_t dict_dfa; mmapf_t i2info; mmapf_t dict_charmap;
    int tok_algo, id_offset, use_bytes, no_dummy_prefix, fsm_type, k2i_count;
    const uint8_t *k2i; int direction;   /* K2I array image (cl/src/FAArray_pack.cpp:27-65), PARAM_DIRECTION (0 = l2r, FAFsmConst.h DIR_L2R) */
    /* [i2w] tokdll:998-1045 */
    int has_i2w, i2w_count, min_token_id, max_token_id; const uint8_t *i2w_offs, *i2w_data;
};

/* cl/src/FAUtils_cl.cpp:101-159 FAGetCrc32 -- standard reflected CRC-32 table, seedable */
static unsigned crc32_update(const uint8_t *p, size_t n, unsigned crc)
{
    static unsigned table[256]; static int init = 0;
    size_t i;
    if (!init) {
        unsigned c, k, j;
        for (k = 0; k < 256; ++k) { c = k; for (j = 0; j < 8; ++j) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[k] = c; }
        init = 1;
    }
    crc = ~crc;
    for (i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xff] ^ (crc >> 8);
    return ~crc;
}

/* cl/src/FALDB.cpp:119-191: boolean params carry no value slot */
static int is_boolean_param(int p)
{
    /* PARAM_REVERSE 10, NO_TR 18, IGNORE_CASE 22, DICT_MODE 31, NORMALIZE 35, DO_W2B 37,
     * LOG_SCALE 40, USE_NFST 46, VERIFY_LDB_BIN 70 (cl/inc/FAFsmConst.h:194-265) */
    return p == 10 || p == 18 || p == PARAM_IGNORE_CASE || p == 31 || p == 35 || p == 37 ||
           p == 40 || p == 46 || p == PARAM_VERIFY_LDB_BIN;
}

/* cl/src/FAWbdConfKeeper.cpp:246-314 CalcFnIniStates */
static void calc_fn_ini(bfo_model *m)
{
    int initial = m->wbd_dfa.initial;
    int state_r = dfa_dest(&m->wbd_dfa, initial, IW_R_ANCHOR);
    int act[64], n, id = 0, max_fn = -1, i, f;
    if (state_r == -1) return;
    while ((n = mmap_get(&m->acts, id++, act, 64)) != -1) {
        if (n > 64) continue;
        for (i = 2; i < n; ++i) if (act[i] == 0 && i + 1 < n) { i++; break; }
        for (; i < n; ++i) if (act[i] > max_fn) max_fn = act[i];
    }
    if (max_fn == -1) return;
    m->fn2ini_size = max_fn + 1;
    m->fn2ini = (int *)malloc(sizeof(int) * (size_t)m->fn2ini_size);
    m->fn2ini[0] = initial;
    for (f = 1; f <= max_fn; ++f) m->fn2ini[f] = dfa_dest(&m->wbd_dfa, state_r, f);
}

static int set_model_data(bfo_model *m)
{
    const uint8_t *img = m->img;
    int count, i, n, vals[256];
    if (m->size < 8) return 0;
    count = rd_i32(img);
    if (count <= 0 || count > 3 * 64) return 0;        /* FALimits::MaxLdbDumpCount */
    if (m->size < 4 + 4 * (size_t)count) return 0;
    m->dump_count = count;
    m->dumps = (const uint8_t **)malloc(sizeof(void *) * (size_t)count);
    for (i = 0; i < count; ++i) {
        int off = rd_i32(img + 4 + 4 * i);
        if (off < 0 || (size_t)off >= m->size) return 0;
        m->dumps[i] = img + off;
    }
    mmap_set(&m->conf, m->dumps[0]);

    /* cl/src/FALDB.cpp:67-116 IsValidBinary */
    n = mmap_get(&m->conf, FUNC_GLOBAL, vals, 256);
    for (i = 0; i < n; ++i) {
        if (vals[i] == PARAM_VERIFY_LDB_BIN) {
            const uint8_t *v = m->dumps[count - 1];
            if (count < 2) return 0;
            if (rd_u32(v) == 0) {
                unsigned size = 0, crc = 0; int k;
                for (k = 0; k < count - 1; ++k) {
                    int sz = (int)(m->dumps[k + 1] - m->dumps[k]);
                    if (sz < 0) return 0;
                    size += (unsigned)sz; crc = crc32_update(m->dumps[k], (size_t)sz, crc);
                }
                if (size != rd_u32(v + 4) || crc != rd_u32(v + 8)) return 0;
            }
        } else if (!is_boolean_param(vals[i])) i++;
    }

    /* [wbd] */
    m->max_depth = 2; m->max_token_length = MAX_WORD_LEN;  /* FAWbdConfKeeper defaults */
    n = mmap_get(&m->conf, FUNC_WBD, vals, 256);
    if (n != -1) {
        int have_fsm = 0;
        m->has_wbd = 1;
        for (i = 0; i < n; ++i) {
            switch (vals[i]) {
            case PARAM_MAP_MODE: ++i; break;
            case PARAM_DEPTH: m->max_depth = vals[++i]; break;
            case PARAM_MAX_LENGTH: m->max_token_length = vals[++i]; break;
            case PARAM_IGNORE_CASE: m->ignore_case = 1; break;
            case PARAM_FSM_TYPE:
                /* FAWbdConfKeeper.cpp:212-227: moore-multi-dfa gets a State2Ows map and no State2Ow one -> FALexTools_t::Process
                 * returns -1 for every input (FALexTools_t.h:134, 412-414); any other type is a LogAssert */
                ++i;
                if (vals[i] == 4 /* TYPE_MOORE_MULTI_DFA */) m->lexer_void = 1;
                else if (vals[i] != TYPE_MOORE_DFA) return 0;
                break;
            case PARAM_FSM: dfa_set(&m->wbd_dfa, m->dumps[vals[++i]]); have_fsm = 1; break;
            case PARAM_MULTI_MAP: mmap_set(&m->acts, m->dumps[vals[++i]]); break;
            case PARAM_CHARMAP: mmapf_set(&m->wbd_charmap, m->dumps[vals[++i]]); break;
            default: ++i; break;  /* PARAM_WORD/XWORD/SEG/IGNORE/PUNKT/EOS/EOP/MAX_TAG/ACT_DATA: tag ids, one value */
            }
        }
        if (have_fsm && m->acts.set) calc_fn_ini(m);
    }

    /* [pos-dict] */
    n = mmap_get(&m->conf, FUNC_POS_DICT, vals, 256);
    if (n != -1) {
        int mode = MODE_PACK_TRIV;
        m->has_seg = 1; m->fsm_type = TYPE_MOORE_DFA;
        for (i = 0; i < n; ++i) {
            switch (vals[i]) {
            case PARAM_IGNORE_CASE: m->dict_ignore_case = 1; break;
            case 18 /* PARAM_NO_TR */: break;
            case 11 /* PARAM_DIRECTION */: m->direction = vals[++i]; break;
            case PARAM_USE_BYTE_ENCODING: m->use_bytes = 1; break;
            case PARAM_NO_DUMMY_PREFIX: m->no_dummy_prefix = 1; break;
            case PARAM_TOKENIZATION_TYPE: m->tok_algo = vals[++i]; break;
            case PARAM_ID_OFFSET: m->id_offset = vals[++i]; break;
            case PARAM_FSM_TYPE: m->fsm_type = vals[++i]; break;
            case PARAM_MAP_MODE: mode = vals[++i]; break;
            case PARAM_FSM: dfa_set(&m->dict_dfa, m->dumps[vals[++i]]); break;
            case PARAM_ARRAY: { const uint8_t *d = m->dumps[vals[++i]]; m->k2i = d; m->k2i_count = rd_i32(d + 12); break; } /* cl/src/FAArray_pack.cpp:27-65 */
            case PARAM_CHARMAP: mmapf_set(&m->dict_charmap, m->dumps[vals[++i]]); break;
            case PARAM_MULTI_MAP:
                if (mode != MODE_PACK_FIXED) return 0;      /* all tokenizer models use fixed-dump */
                mmapf_set(&m->i2info, m->dumps[vals[++i]]); break;
            default: return 0;
            }
        }
        if (m->fsm_type != TYPE_MEALY_DFA || !m->dict_dfa.set || !m->i2info.set) return 0;
    }

    /* [i2w] (tokdll:998-1045): FAStringArray_pack image = [count][offsets: count + 1][data] (cl/src/FAStringArray_pack.cpp:23-50) */
    m->min_token_id = 0; m->max_token_id = MAX_ARR_SIZE;
    n = mmap_get(&m->conf, 35 /* FUNC_I2W */, vals, 256);
    if (n != -1) {
        for (i = 0; i < n; ++i) {
            if (vals[i] == 75 /* PARAM_STRING_ARRAY */ && i + 1 < n) {
                const uint8_t *d;
                int dn = vals[++i];
                if (dn < 0 || dn >= count) return 0;
                d = m->dumps[dn];
                m->i2w_count = rd_i32(d); m->i2w_offs = d + 4; m->i2w_data = d + 4 + 4 * ((size_t)m->i2w_count + 1); m->has_i2w = 1;
                if (m->i2w_count < 0) return 0;
            } else if (vals[i] == 76 /* PARAM_TOKENID_MIN */ && i + 1 < n) m->min_token_id = vals[++i];
            else if (vals[i] == 77 /* PARAM_TOKENID_MAX */ && i + 1 < n) m->max_token_id = vals[++i];
        }
    }
    return 1;
}

bfo_model *bfo_set_model(const unsigned char *img, size_t size)
{
    bfo_model *m;
    if (!img || !size) return NULL;
    m = (bfo_model *)calloc(1, sizeof(*m));
    m->img = (uint8_t *)malloc(size + 16);
    memcpy(m->img, img, size); memset(m->img + size, 0, 16);
    m->size = size;
    if (!set_model_data(m)) { bfo_free_model(m); return NULL; }
    return m;
}

bfo_model *bfo_load_model(const char *path)
{
    FILE *f = path ? fopen(path, "rb") : NULL;
    long sz; uint8_t *buf; bfo_model *m;
    if (!f) return NULL;
    fseek(f, 0, SEEK_END); sz = ftell(f); fseek(f, 0, SEEK_SET);
    if (sz <= 0) { fclose(f); return NULL; }
    buf = (uint8_t *)malloc((size_t)sz);
    if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(buf); return NULL; }
    fclose(f);
    m = bfo_set_model(buf, (size_t)sz);
    free(buf);
    return m;
}

int bfo_free_model(bfo_model *m)
{
    if (!m) return 0;
    free(m->img); free(m->dumps); free(m->fn2ini); free(m);
    return 1;
}

int bfo_set_no_dummy_prefix(bfo_model *m, int flag) { if (!m) return 0; m->no_dummy_prefix = flag != 0; return 1; }

/* ---------------- UTF-8 (cl/src/FAUtf8Utils.cpp) ---------------- */

/* FAUtf8Size(int) :44-58 */
static int utf8_size_of(int sym)
{
    unsigned s = (unsigned)sym;
    if (s <= 0x7f) return 1;
    if (s <= 0x7ff) return 2;
    if (s <= 0xffff) return 3;
    if (s <= 0x10ffff) return 4;
    return 0;
}

/* FAUtf8ToInt(pBegin,pEnd,pResult) :121-196: returns bytes consumed or 0 on error */
static int utf8_to_int(const uint8_t *p, const uint8_t *end, int *res)
{
    int ch, n, i, ret;
    if (end <= p) return 0;
    ch = *p;
    if ((ch & 0x80) == 0) { *res = ch; return 1; }
    if ((ch & 0xE0) == 0xC0) { n = 2; ch &= ~0xE0; }
    else if ((ch & 0xF0) == 0xE0) { n = 3; ch &= ~0xF0; }
    else if ((ch & 0xF8) == 0xF0) { n = 4; ch &= ~0xF8; }
    else return 0;
    if ((size_t)(end - p) < (size_t)n) return 0;
    ret = ch;
    for (i = 1; i < n; ++i) {
        ch = p[i];
        if ((ch & 0xC0) != 0x80) return 0;
        ret = (ret << 6) | (ch & 0x3f);
    }
    if (n != utf8_size_of(ret)) return 0;              /* overlong / > 10FFFF */
    if (0xD800 == (0xFFFFF800 & (unsigned)ret)) return 0; /* surrogate */
    *res = ret;
    return n;
}

/* FAStrUtf8ToArray(pStr,Len,pArray,MaxSize) :233-270 */
int bfo_utf8_to_utf32(const char *s, int len, int *out, int max_out)
{
    const uint8_t *p = (const uint8_t *)s, *end = p + len;
    int i = 0;
    if (len >= 3 && p[0] == 0xEF && p[1] == 0xBB && p[2] == 0xBF) p += 3;
    while (p < end && i < max_out) {
        int n = utf8_to_int(p, end, out + i);
        if (!n) return -1;
        p += n; i++;
    }
    return i;
}

/* FAStrUtf8ToArray with offsets :273-313 (offsets are relative to the ORIGINAL start: a BOM counts) */
static int utf8_to_utf32_off(const char *s, int len, int *out, int *offs, int max_out)
{
    const uint8_t *begin = (const uint8_t *)s, *p = begin, *end = p + len;
    int i = 0;
    if (len >= 3 && p[0] == 0xEF && p[1] == 0xBB && p[2] == 0xBF) p += 3;
    while (p < end && i < max_out) {
        int n = utf8_to_int(p, end, out + i);
        if (!n) return -1;
        offs[i] = (int)(p - begin);
        p += n; i++;
    }
    return i;
}

/* FAStrUtf8AsBytesToArray :316-345 (+ offsets variant :348-381) */
static int utf8_as_bytes(const char *s, int len, int *out, int *offs, int max_out)
{
    const uint8_t *begin = (const uint8_t *)s, *p = begin, *end = p + len;
    int i = 0;
    if (len >= 3 && p[0] == 0xEF && p[1] == 0xBB && p[2] == 0xBF) p += 3;
    while (p < end && i < max_out) { if (offs) offs[i] = (int)(p - begin); out[i++] = *p++; }
    return i;
}

/* FAUtf8Size(const char*) cl/src/FAUtf8Utils.cpp:23-42 */
static int utf8_size_at(const char *p)
{
    int ch = *(const unsigned char *)p;
    if ((ch & 0x80) == 0x00) return 1;
    if ((ch & 0xE0) == 0xC0) return 2;
    if ((ch & 0xF0) == 0xE0) return 3;
    if ((ch & 0xF8) == 0xF0) return 4;
    return 0;
}

/* cl/inc/FAUtils_cl.h:311-369 FANormalize (offs != NULL: the offsets form :374-436) */
static int normalize(const int *in, int n, int *out, int *offs, int max_out, const mmapf_t *map)
{
    int norm[10], o = 0, i, j;
    for (i = 0; i < n; ++i) {
        int c = mmapf_get(map, in[i], norm, 10);
        if (c == -1) { if (o < max_out) { out[o] = in[i]; if (offs) offs[o] = i; } o++; }
        else if (c == 1) { if (o < max_out) { out[o] = norm[0]; if (offs) offs[o] = i; } o++; }
        else if (c > 1 && c <= 10) {
            int copy = max_out - o;
            if (c < copy) copy = c;
            for (j = 0; j < copy; ++j) { out[o + j] = norm[j]; if (offs) offs[o + j] = i; }
            o += c;
        }
    }
    return o;
}

/* ---------------- lexer: cl/inc/FALexTools_t.h:205-421 ---------------- */

/* cl/src/FAUtf32Utils.cpp:45-81 FAUtf32ToLower, reached with ignore-case models only: its three-level table is DATA, taken here as
 * runs observed from the compiled reference (tools/make_tolower.py -> bf_tolower_tab.h; tests/test_ignore_case.py compares
 * bfo_tolower with the reference's function on every code point). */
#include "bf_tolower_tab.h"
int bfo_tolower_sym(int cp) { return bfo_tolower(cp); }

static int lex_process_int(const bfo_model *m, int initial, int offset, const int *in, int n,
                           int *out, int max_out, int depth, int once)
{
    const dfa_t *d = &m->wbd_dfa;
    int out_size = 0, from;
    if (m->max_depth < depth) return 0;
    for (from = -1; from < n; ++from) {
        int state = initial, final_state = -1, final_pos = -1, j = from, dst;
        int bound = from + m->max_token_length;
        if (n < bound) bound = n;
        if (j == -1) {
            state = dfa_dest(d, initial, IW_L_ANCHOR);
            if (state == -1) { state = dfa_dest(d, initial, IW_ANY); if (state == -1) continue; }
            j++;
        }
        for (; j < bound; ++j) {
            int iw = in[j];
            if (iw < IW_EPSILON) iw = IW_EPSILON;
            if (m->ignore_case) iw = bfo_tolower(iw);                 /* FALexTools_t.h:262-264 */
            dst = dfa_dest(d, state, iw);
            if (dst == -1) { dst = dfa_dest(d, state, IW_ANY); if (dst == -1) break; }
            if (dfa_is_final(d, dst)) { final_state = dst; final_pos = j; }
            state = dst;
        }
        if (j == n) {
            dst = dfa_dest(d, state, IW_R_ANCHOR);
            if (dst == -1) dst = dfa_dest(d, state, IW_ANY);
            if (dst != -1 && dfa_is_final(d, dst)) { final_state = dst; final_pos = j; }
        }
        if (final_pos != -1) {
            int act[64], act_size, ow, left, right, tag, from2, to2, fn_idx, fn_once, fn_from;
            ow = dfa_state2ow(d, final_state);
            act_size = mmap_get(&m->acts, ow, act, 64);
            if (act_size < MIN_ACT_SIZE || act_size > 64) return out_size; /* validated at load in the reference */
            left = act[0]; right = act[1]; tag = act[2];
            from2 = from + left; if (from2 < 0) from2 = 0; else if (n <= from2) from2 = n - 1;
            to2 = final_pos - right; if (to2 < 0) to2 = 0; else if (n <= to2) to2 = n - 1;
            fn_idx = MIN_ACT_SIZE;
            if (tag != 0) {
                if (out_size + 3 <= max_out) { out[out_size++] = tag; out[out_size++] = from2 + offset; out[out_size++] = to2 + offset; }
                else return out_size;
                fn_idx = MIN_ACT_SIZE + 1;
            }
            fn_once = 1 < (act_size - fn_idx);
            fn_from = from2;
            for (; fn_idx < act_size; ++fn_idx) {
                int fn = act[fn_idx], r;
                if (fn < 0 || fn >= m->fn2ini_size) break;
                r = lex_process_int(m, m->fn2ini[fn], fn_from + offset, in + fn_from, to2 - fn_from + 1,
                                    out + out_size, max_out - out_size, depth + 1, fn == 0 ? 0 : fn_once);
                if (r > 0) {
                    out_size += r;
                    fn_from = out[out_size - 1] + 1 - offset;
                    if (fn_from > to2) break;
                }
            }
            if (once) return out_size;
            if (final_pos - right > from) from = final_pos - right;
        }
    }
    return out_size;
}

int bfo_lex_process(const bfo_model *m, const int *in, int n, int *out, int max_out)
{
    if (!m || !m->has_wbd || !m->wbd_dfa.set || !m->acts.set || m->lexer_void) return -1;      /* FALexTools_t.h:412-414 (no State2Ow: lexer_void) */
    return lex_process_int(m, m->wbd_dfa.initial, 0, in, n, out, max_out, 1, 0);
}

/* tokdll:1108-1314 TextToIdsWithOffsets_wp (starts/ends may be NULL: ids only) */
static int text_to_ids_wp(const bfo_model *m, const char *s, int n, int32_t *ids, int *starts, int *ends, int max_ids, int unk)
{
    int *buf, *norm = NULL, *res, len, res_max, res_size, out = 0, i;
    int *offs = NULL, *noffs = NULL;
    const int need_off = starts && ends;
    const int *in;
    if (n <= 0 || n > MAX_ARR_SIZE || !s || !m) return 0;
    buf = (int *)malloc(sizeof(int) * (size_t)n);
    if (need_off) { offs = (int *)malloc(sizeof(int) * (size_t)n); len = utf8_to_utf32_off(s, n, buf, offs, n); }
    else len = bfo_utf8_to_utf32(s, n, buf, n);
    if (len <= 0 || len > n) { free(buf); free(offs); return 0; }
    in = buf;
    if (m->wbd_charmap.set) {
        norm = (int *)malloc(sizeof(int) * (size_t)n);
        if (need_off) noffs = (int *)malloc(sizeof(int) * (size_t)n);
        len = normalize(buf, len, norm, noffs, n, &m->wbd_charmap);
        if (len <= 0 || len > n) { free(buf); free(norm); free(offs); free(noffs); return 0; }
        in = norm;
    }
#define BFO_OFF(k) (offs[m->wbd_charmap.set ? noffs[(k)] : (k)])
    res_max = len * 6;
    res = (int *)calloc((size_t)res_max + 8, sizeof(int)); /* std::vector<int>(n) zero-fills, tokdll:1195 */
    res_size = bfo_lex_process(m, in, len, res, res_max);
    if (res_size > res_max || res_size % 3 != 0 || res_size < 0) { free(buf); free(norm); free(res); free(offs); free(noffs); return 0; }
    for (i = 0; i < res_size; i += 3) {
        int tag = res[i];
        if (tag == WBD_IGNORE_TAG) continue;
        if (tag == WBD_WORD_TAG) {
            int tok_from = res[i + 1], tok_to = res[i + 2], j = i + 3, nsub = 0, covered = 0;
            if (j < res_size) {
                int expected = tok_from, st = res[j], sf = res[j + 1], sto = res[j + 2], k;
                /* restated literally, including the '<=' (tokdll:1239) and the stale
                 * SubToken* values once j reaches res_size (tokdll:1244-1248) */
                while (j <= res_size && st > WBD_IGNORE_TAG && expected == sf) {
                    expected = sto + 1; nsub++; j += 3;
                    if (j < res_size) { st = res[j]; sf = res[j + 1]; sto = res[j + 2]; }
                }
                if (nsub > 0 && expected - 1 == tok_to) {
                    for (k = 0; k < nsub && out < max_ids; ++k) {
                        int ti = (k + 1) * 3 + i;
                        ids[out] = res[ti];
                        if (need_off) {                                        /* tokdll:1263-1273 */
                            int to_off = BFO_OFF(res[ti + 2]), sz;
                            starts[out] = BFO_OFF(res[ti + 1]);
                            sz = utf8_size_at(s + to_off);
                            ends[out] = to_off + (0 < sz ? sz - 1 : 0);
                        }
                        out++;
                    }
                    covered = 1;
                }
            }
            if (!covered && out < max_ids) {
                ids[out] = unk;
                if (need_off) {                                                /* tokdll:1289-1297 */
                    int to_off = BFO_OFF(tok_to), sz;
                    starts[out] = BFO_OFF(tok_from);
                    sz = utf8_size_at(s + to_off);
                    ends[out] = to_off + (0 < sz ? sz - 1 : 0);
                }
                out++;
            }
            i = j - 3;
        }
        if (out >= max_ids) break;
    }
#undef BFO_OFF
    free(buf); free(norm); free(res); free(offs); free(noffs);
    return out;
}

/* ---------------- segmenters ---------------- */

/* blingfiretokdll.h:17-21 __FAIsWhiteSpace__ */
static int is_ws(int c)
{
    return c <= 0x20 || c == 0xa0 || (c >= 0x2000 && c <= 0x200f) || c == 0x202f || c == 0x205f ||
           c == 0x2060 || c == 0x2420 || c == 0x2424 || c == 0x3000 || c == 0xfeff;
}

/* cl/src/FAMultiMap_pack_fixed.cpp:140-160 pointer form (requires 4-byte values) */
static int i2info_get(const mmapf_t *m, int key, int *id, uint32_t *bits)
{
    const uint8_t *arr; int count;
    if (!(m->min_key <= key && key <= m->max_key && m->size_of_value == 4)) return -1;
    arr = m->data + (unsigned)m->size_of_arr * (unsigned)(key - m->min_key);
    count = rd_i32(arr);
    if (count > m->max_count) return -1;
    *id = rd_i32(arr + 4);
    *bits = count >= 2 ? rd_u32(arr + 8) : 0;
    return count;
}

typedef struct { int begin, id; double score; } uarc_t;

/* cl/inc/FATokenSegmentationTools_1best_t.h:175-279 (AddArc 118-142, AddUnknownArc 145-171) */
static int seg_unigram(const bfo_model *m, const int *in, int n, int *out, int max_out, int unk)
{
    const dfa_t *d = &m->dict_dfa;
    const float unk_score = -100000.0f;
    uarc_t *arcs; int start, i, actual = 0, end;
    if (n <= 0) return 0;
    arcs = (uarc_t *)malloc(sizeof(uarc_t) * (size_t)n);
    for (i = 0; i < n; ++i) { arcs[i].begin = -1; arcs[i].id = -1; arcs[i].score = -FLT_MAX; }
    for (start = 0; start < n; ++start) {
        int state = d->initial, sum = 0, ow = 0, unknown = 1;
        for (i = start; i < n; ++i) {
            state = mealy_dest_ow(d, state, in[i], &ow);
            if (state == -1) break;
            sum += ow;
            if (dfa_is_final(d, state)) {
                int id; uint32_t bits; float score; double prev; uarc_t *a = arcs + i;
                if (i2info_get(&m->i2info, sum, &id, &bits) != 2) { free(arcs); return -1; }
                memcpy(&score, &bits, 4);
                prev = 0 < start ? arcs[start - 1].score : 0;
                if (a->score < score + prev) { a->begin = start; a->id = id; a->score = score + prev; }
                unknown = 0;
            }
        }
        if (unknown) {
            uarc_t *a = arcs + start, *pa = a - 1;
            double prev = 0 < start ? pa->score : 0;
            if (a->score < unk_score + prev) {
                a->begin = start; a->id = -1; a->score = unk_score + prev;
                if (0 < start && pa->id == -1) a->begin = pa->begin;
            }
        }
    }
    end = n - 1;
    while (0 <= end) {
        const uarc_t *a = arcs + end;
        if (actual + 3 <= max_out) { out[actual] = end; out[actual + 1] = a->begin; out[actual + 2] = a->id != -1 ? a->id : unk; }
        actual += 3;
        end = a->begin - 1;
    }
    if (max_out >= actual) for (i = 0; i < actual / 2; ++i) { int t = out[i]; out[i] = out[actual - i - 1]; out[actual - i - 1] = t; }
    free(arcs);
    return actual;
}

typedef struct { int start, end, id; float rank; } barc_t;

static int cmp_bpe(const void *a, const void *b)
{   /* cl/inc/FATokenSegmentationTools_1best_bpe_t.h:238-255 */
    const barc_t *x = (const barc_t *)a, *y = (const barc_t *)b;
    if (x->id < y->id) return -1;
    if (x->id == y->id) { if (x->start < y->start) return -1; if (x->start == y->start) return 0; return 1; }
    return 1;
}
static int cmp_bpe_merges(const void *a, const void *b)
{   /* cl/inc/FATokenSegmentationTools_1best_bpe_with_merges_t.h:242-262 */
    const barc_t *x = (const barc_t *)a, *y = (const barc_t *)b;
    if (x->rank > y->rank) return -1;
    if (x->rank == y->rank) {
        if (x->id < y->id) return -1;
        if (x->id == y->id) { if (x->start < y->start) return -1; if (x->start == y->start) return 0; }
    }
    return 1;
}

/* cl/inc/FATokenSegmentationTools_1best_bpe_t.h:126-316 and
 * cl/inc/FATokenSegmentationTools_1best_bpe_with_merges_t.h:129-323 (with_merges != 0) */
static int seg_bpe(const bfo_model *m, const int *in, int n, int *out, int max_out, int unk, int with_merges)
{
    const dfa_t *d = &m->dict_dfa;
    /* with_merges has no m_fFastBpe assignment other than the same test (…_with_merges_t.h SetConf) */
    int fast = m->tok_algo == TOKENIZE_BPE_OPT || (with_merges && m->tok_algo == TOKENIZE_BPE_OPT_WITH_MERGES);
    barc_t *arcs; size_t narcs = 0, cap = (size_t)n + 16, k;
    int start, i, actual = 0, *tos, *idsv; unsigned char *inter;
    if (n <= 0) return 0;
    arcs = (barc_t *)malloc(sizeof(barc_t) * cap);
    for (start = 0; start < n; ++start) {
        int state = d->initial, sum = 0, ow = 0, unknown = 1;
        int token_start = SP_DELIM == in[start];
        size_t count_at_start = narcs;
        int ff = start;
        for (i = start; i < n; ++i) {
            state = mealy_dest_ow(d, state, in[i], &ow);
            if (state == -1) break;
            sum += ow;
            if (dfa_is_final(d, state)) {
                int id; uint32_t bits; float rank = 0; int apply;
                if (i2info_get(&m->i2info, sum, &id, &bits) < 1) { free(arcs); return -1; }
                memcpy(&rank, &bits, 4);
                apply = fast && token_start && ((i < n - 1) ? SP_DELIM == in[i + 1] : 1) && count_at_start < narcs;
                if (!apply) {
                    if (narcs == cap) { cap *= 2; arcs = (barc_t *)realloc(arcs, sizeof(barc_t) * cap); }
                    arcs[narcs].start = start; arcs[narcs].end = i; arcs[narcs].id = id; arcs[narcs].rank = rank; narcs++;
                } else {
                    arcs[count_at_start].start = start; arcs[count_at_start].end = i; arcs[count_at_start].id = id; arcs[count_at_start].rank = rank;
                    narcs = count_at_start + 1;
                    ff = i;
                }
                unknown = 0;
            }
        }
        if (unknown) {
            if (0 < narcs && unk == arcs[narcs - 1].id) arcs[narcs - 1].end = start;
            else {
                if (narcs == cap) { cap *= 2; arcs = (barc_t *)realloc(arcs, sizeof(barc_t) * cap); }
                arcs[narcs].start = start; arcs[narcs].end = start; arcs[narcs].id = unk; arcs[narcs].rank = 0.0f; narcs++;
            }
        }
        if (fast) start = ff;
    }
    qsort(arcs, narcs, sizeof(barc_t), with_merges ? cmp_bpe_merges : cmp_bpe);
    tos = (int *)calloc((size_t)n * 3, sizeof(int));
    idsv = tos + n;
    for (i = 0; i < n; ++i) idsv[i] = unk;
    inter = (unsigned char *)(tos + 2 * (size_t)n);
    for (k = 0; k < narcs; ++k) {
        int s = arcs[k].start, e = arcs[k].end;
        if (0 == inter[s] && (e + 1 == n || 0 == inter[e + 1])) {
            tos[s] = e; idsv[s] = arcs[k].id;
            if (e - s > 0) memset(inter + s + 1, 1, (size_t)(e - s));
        }
    }
    for (start = 0; start < n; start++) {
        int e = tos[start], id = idsv[start];
        if (actual + 3 <= max_out) { out[actual] = id; out[actual + 1] = start; out[actual + 2] = e; }
        actual += 3;
        start = e;
    }
    free(arcs); free(tos);
    return actual;
}

/* tokdll:1349-1535 TextToIdsWithOffsets_sp (starts/ends may be NULL: ids only) */
static int text_to_ids_sp(const bfo_model *m, const char *s, int n, int32_t *ids, int *starts, int *ends, int max_ids, int unk)
{
    int *buf, *norm = NULL, *in, *res, len, off, i, j, res_max, res_size, out = 0;
    int *offs = NULL, *noffs = NULL, *adj = NULL;
    const int need_off = starts && ends;
    if (n <= 0 || n > MAX_ARR_SIZE || !s || !m) return 0;
    buf = (int *)malloc(sizeof(int) * ((size_t)n + 1));
    buf[0] = SP_DELIM;
    if (need_off) { offs = (int *)malloc(sizeof(int) * ((size_t)n + 1)); offs[0] = -1; }   /* tokdll:1387 */
    off = m->no_dummy_prefix ? 0 : 1;
    if (m->use_bytes) len = utf8_as_bytes(s, n, buf + off, need_off ? offs + off : NULL, n);
    else len = need_off ? utf8_to_utf32_off(s, n, buf + off, offs + off, n) : bfo_utf8_to_utf32(s, n, buf + off, n);
    if (len <= 0 || len > n) { free(buf); free(offs); return 0; }
    len += off;
    in = buf;
    if (m->dict_charmap.set) {
        int max_norm = (n + 1) * 2, actual;
        norm = (int *)malloc(sizeof(int) * (size_t)max_norm);
        if (need_off) noffs = (int *)malloc(sizeof(int) * (size_t)max_norm);
        actual = normalize(buf, len, norm, noffs, max_norm, &m->dict_charmap);
        if (actual <= 0 || actual > max_norm) { free(buf); free(norm); free(offs); free(noffs); return 0; }
        len = actual; in = norm;
    }
    adj = need_off ? (m->dict_charmap.set ? noffs : offs) : NULL;                          /* tokdll:1460 */
    for (i = 0, j = 0; i < len; ++i) {            /* tokdll:1462-1488 */
        int c = in[i];
        if (!is_ws(c)) { in[j] = c; if (adj) adj[j] = adj[i]; j++; }
        else if (0 == j || SP_DELIM != in[j - 1]) { in[j] = SP_DELIM; if (adj) adj[j] = adj[i]; j++; }
    }
    if (1 < j && in[j - 1] == SP_DELIM) j--;      /* tokdll:1491-1493 */
    len = j;
    res_max = len * 3;
    res = (int *)malloc(sizeof(int) * (size_t)(res_max > 0 ? res_max : 1));
    if (m->tok_algo == TOKENIZE_BPE || m->tok_algo == TOKENIZE_BPE_OPT) res_size = seg_bpe(m, in, len, res, res_max, unk, 0);
    else if (m->tok_algo == TOKENIZE_BPE_OPT_WITH_MERGES) res_size = seg_bpe(m, in, len, res, res_max, unk, 1);
    else res_size = seg_unigram(m, in, len, res, res_max, unk);
    if (res_size > res_max || res_size % 3 != 0 || res_size < 0) { free(buf); free(norm); free(res); free(offs); free(noffs); return 0; }
    for (i = 0; i < res_size && out < max_ids; i += 3) {
        ids[out] = res[i] + m->id_offset;
        if (need_off) {                                                        /* tokdll:1519-1529 */
            int from_off = offs[m->dict_charmap.set ? noffs[res[i + 1]] : res[i + 1]];
            int to_off = offs[m->dict_charmap.set ? noffs[res[i + 2]] : res[i + 2]];
            /* a token made of the dummy prefix alone has offset -1: the reference then reads the byte BEFORE the
             * caller's string (undefined); the restatement treats its size as 0 */
            int sz = to_off >= 0 ? utf8_size_at(s + to_off) : 0;
            starts[out] = from_off;
            ends[out] = to_off + (0 < sz ? sz - 1 : 0);
        }
        out++;
    }
    free(buf); free(norm); free(res); free(offs); free(noffs);
    return out;
}

int bfo_text_to_ids(const bfo_model *m, const char *utf8, int n, int32_t *ids, int max_ids, int unk)
{
    return bfo_text_to_ids_with_offsets(m, utf8, n, ids, NULL, NULL, max_ids, unk);
}

/* tokdll:1562-1609 TextToIdsWithOffsets */
int bfo_text_to_ids_with_offsets(const bfo_model *m, const char *utf8, int n, int32_t *ids, int *starts, int *ends, int max_ids, int unk)
{
    if (!m) return 0;
    if (!m->has_seg) return m->has_wbd ? text_to_ids_wp(m, utf8, n, ids, starts, ends, max_ids, unk) : 0;
    return text_to_ids_sp(m, utf8, n, ids, starts, ends, max_ids, unk);
}

/* tokdll:415-566 TextToWordsWithOffsetsWithModel: the lexer of the model (NO charmap, U+0000 -> U+0020), every
 * non-IGNORE token re-encoded as UTF-8 with inner ' ' -> '_', joined by ' ', terminated by 0.  Returns the byte
 * count needed (terminator included), -1 on error, 0 for empty input; copies only if it fits. */
int bfo_text_to_words_with_offsets(const bfo_model *m, const char *s, int n, char *out, int *starts, int *ends, int max_out)
{
    int *buf, *offs, *res, len, res_size, i, words = 0, added = 0, pos = 0;
    char *tmp;
    if (!m || !m->has_wbd) return -1;
    if (n == 0) return 0;
    if (n < 0 || n > MAX_ARR_SIZE || !s) return -1;
    buf = (int *)malloc(sizeof(int) * (size_t)n);
    offs = (int *)malloc(sizeof(int) * (size_t)n);
    if (starts) memset(starts, 0, sizeof(int) * (size_t)(max_out > 0 ? max_out : 0));   /* tokdll:469-474 */
    if (ends) memset(ends, 0, sizeof(int) * (size_t)(max_out > 0 ? max_out : 0));
    len = utf8_to_utf32_off(s, n, buf, offs, n);
    if (len <= 0 || len > n) { free(buf); free(offs); return -1; }
    for (i = 0; i < len; ++i) if (buf[i] == 0) buf[i] = 0x20;                           /* tokdll:482 */
    res = (int *)calloc((size_t)len * 3 + 8, sizeof(int));
    res_size = bfo_lex_process(m, buf, len, res, len * 3);
    if (res_size > len * 3 || res_size % 3 != 0 || res_size < 0) { free(buf); free(offs); free(res); return -1; }
    {   /* tokens may overlap (a token and the sub-tokens a _call produced from it): size the buffer from the triples */
        size_t need = 2;
        for (i = 0; i < res_size; i += 3) if (res[i] != WBD_IGNORE_TAG && res[i + 2] >= res[i + 1]) need += 4 * (size_t)(res[i + 2] - res[i + 1] + 1) + 1;
        tmp = (char *)malloc(need);
    }
    for (i = 0; i < res_size; i += 3) {
        int from, to, k;
        if (res[i] == WBD_IGNORE_TAG) continue;
        from = res[i + 1]; to = res[i + 2];
        if (starts && words < max_out) starts[words] = offs[from];
        if (ends && words < max_out) { int sz = utf8_size_at(s + offs[to]); ends[words] = offs[to] + (0 < sz ? sz - 1 : 0); }
        words++;
        if (added) tmp[pos++] = ' ';
        for (k = from; k <= to; ++k) {                                                  /* FAArrayToStrUtf8 cl/src/FAUtf8Utils.cpp:530-557 */
            unsigned c = (unsigned)buf[k];
            if (c < 0x80) tmp[pos++] = (char)(c == ' ' ? '_' : c);                      /* tokdll:543 */
            else if (c < 0x800) { tmp[pos++] = (char)(0xC0 | (c >> 6)); tmp[pos++] = (char)(0x80 | (c & 0x3F)); }
            else if (c < 0x10000) { tmp[pos++] = (char)(0xE0 | (c >> 12)); tmp[pos++] = (char)(0x80 | ((c >> 6) & 0x3F)); tmp[pos++] = (char)(0x80 | (c & 0x3F)); }
            else { tmp[pos++] = (char)(0xF0 | (c >> 18)); tmp[pos++] = (char)(0x80 | ((c >> 12) & 0x3F)); tmp[pos++] = (char)(0x80 | ((c >> 6) & 0x3F)); tmp[pos++] = (char)(0x80 | (c & 0x3F)); }
        }
        added = 1;
    }
    tmp[pos++] = 0;
    if (pos <= max_out && out) memcpy(out, tmp, (size_t)pos);
    free(buf); free(offs); free(res); free(tmp);
    return pos;
}

/* tokdll:163-355 TextToSentencesWithOffsetsWithModel: every triple's To ends a sentence (Tag and From are ignored,
 * tokdll:262-266), a sentence starts right after the previous one, leading white space is dropped (tokdll:138-150,270),
 * '\n' inside a sentence becomes ' ' (tokdll:296), the rest of the paragraph is the last sentence (tokdll:307-339) */
int bfo_text_to_sentences_with_offsets(const bfo_model *m, const char *s, int n, char *out, int *starts, int *ends, int max_out)
{
    int *buf, *offs, *res, len, res_size, i, sents = 0, added = 0, pos = 0, prev_end = -1, pass;
    char *tmp;
    if (!m || !m->has_wbd) return -1;
    if (n == 0) return 0;
    if (n < 0 || n > MAX_ARR_SIZE || !s) return -1;
    buf = (int *)malloc(sizeof(int) * (size_t)n);
    offs = (int *)malloc(sizeof(int) * (size_t)n);
    if (starts) memset(starts, 0, sizeof(int) * (size_t)(max_out > 0 ? max_out : 0));   /* tokdll:220-225 */
    if (ends) memset(ends, 0, sizeof(int) * (size_t)(max_out > 0 ? max_out : 0));
    len = utf8_to_utf32_off(s, n, buf, offs, n);
    if (len <= 0 || len > n) { free(buf); free(offs); return -1; }
    for (i = 0; i < len; ++i) if (buf[i] == 0) buf[i] = 0x20;                           /* tokdll:233 */
    res = (int *)calloc((size_t)len * 3 + 8, sizeof(int));
    res_size = bfo_lex_process(m, buf, len, res, len * 3);
    if (res_size > len * 3 || res_size % 3 != 0 || res_size < 0) { free(buf); free(offs); free(res); return -1; }
    tmp = (char *)malloc(4 * (size_t)len + (size_t)len + 8);
    for (pass = 0; pass <= res_size; pass += 3) {
        int from, to, sl, delta, k;
        if (pass < res_size) { from = prev_end + 1; to = res[pass + 2]; prev_end = to; }
        else { if (!(prev_end + 1 < len)) break; from = prev_end + 1; to = len - 1; }  /* tokdll:307-311 */
        sl = to - from + 1;
        for (delta = 0; delta < sl && is_ws(buf[from + delta]); ++delta) {}
        if (!(delta < sl)) continue;
        if (starts && sents < max_out) starts[sents] = offs[from + delta];
        if (ends && sents < max_out) { int sz = utf8_size_at(s + offs[to]); ends[sents] = offs[to] + (0 < sz ? sz - 1 : 0); }
        sents++;
        if (added) tmp[pos++] = '\n';
        for (k = from + delta; k <= to; ++k) {                                          /* FAArrayToStrUtf8 cl/src/FAUtf8Utils.cpp:530-557 */
            unsigned c = (unsigned)buf[k];
            if (c < 0x80) tmp[pos++] = (char)(c == '\n' ? ' ' : c);                     /* tokdll:296 */
            else if (c < 0x800) { tmp[pos++] = (char)(0xC0 | (c >> 6)); tmp[pos++] = (char)(0x80 | (c & 0x3F)); }
            else if (c < 0x10000) { tmp[pos++] = (char)(0xE0 | (c >> 12)); tmp[pos++] = (char)(0x80 | ((c >> 6) & 0x3F)); tmp[pos++] = (char)(0x80 | (c & 0x3F)); }
            else { tmp[pos++] = (char)(0xF0 | (c >> 18)); tmp[pos++] = (char)(0x80 | ((c >> 12) & 0x3F)); tmp[pos++] = (char)(0x80 | ((c >> 6) & 0x3F)); tmp[pos++] = (char)(0x80 | (c & 0x3F)); }
        }
        if (pass < res_size) added = 1;
    }
    tmp[pos++] = 0;
    if (pos <= max_out && out) memcpy(out, tmp, (size_t)pos);
    free(buf); free(offs); free(res); free(tmp);
    return pos;
}

/* tokdll:1689-1745 IdsToText */
int bfo_ids_to_text(const bfo_model *m, const int32_t *ids, int n, char *out, int max_out, int skip_special)
{
    int i, actual = 0;
    if (!m) return 0;
    if (n == 0 || !ids) return 0;
    if (!m->has_i2w) return 0;
    for (i = 0; i < n; ++i) {
        const int id = ids[i];
        unsigned b, e; const uint8_t *tok; int len;
        if (skip_special && (id < m->min_token_id || id > m->max_token_id)) continue;     /* tokdll:1712-1714 */
        if (id < 0 || id >= m->i2w_count) return 0;                                       /* unknown id (tokdll:1719-1721) */
        b = rd_u32(m->i2w_offs + 4 * (size_t)id); e = rd_u32(m->i2w_offs + 4 * ((size_t)id + 1));
        tok = m->i2w_data + b; len = (int)(e - b);
        if (actual == 0 && len > 0 && tok[0] == 0x20) { tok++; len--; }                   /* no space in the leading position */
        if (len > 0 && max_out - actual >= len) memcpy(out + actual, tok, (size_t)len);
        actual += len;
    }
    if (max_out > actual) out[actual] = 0;
    return actual + 1;
}
int bfo_has_i2w(const bfo_model *m) { return m && m->has_i2w; }
int bfo_i2w_count(const bfo_model *m) { return m ? m->i2w_count : 0; }

/* tokdll:629-679 NormalizeSpaces (model-free): strict UTF-8 decode, every run of white space becomes one uSpace unless
 * nothing was written yet or the previous written character already equals uSpace, one trailing uSpace is trimmed, re-encode;
 * -1 on empty / invalid input or when the output does not fit (FAArrayToStrUtf8 cl/src/FAUtf8Utils.cpp:530-557) */
static int int_to_utf8(int c, char *p, int room)
{
    unsigned u = (unsigned)c;
    if (u <= 0x7F && room > 0) { p[0] = (char)u; return 1; }
    if (u <= 0x7FF && room > 1) { p[0] = (char)(0xC0 | (u >> 6)); p[1] = (char)(0x80 | (u & 0x3F)); return 2; }
    if (u <= 0xFFFF && room > 2) {
        if ((u & 0xFFFFF800u) == 0xD800u) return -1;                                    /* surrogate (cl/src/FAUtf8Utils.cpp:498-501) */
        p[0] = (char)(0xE0 | (u >> 12)); p[1] = (char)(0x80 | ((u >> 6) & 0x3F)); p[2] = (char)(0x80 | (u & 0x3F)); return 3;
    }
    if (u <= 0x10FFFF && room > 3) { p[0] = (char)(0xF0 | (u >> 18)); p[1] = (char)(0x80 | ((u >> 12) & 0x3F)); p[2] = (char)(0x80 | ((u >> 6) & 0x3F)); p[3] = (char)(0x80 | (u & 0x3F)); return 4; }
    return -1;
}
int bfo_normalize_spaces(const char *s, int n, char *out, int max_out, int u_space)
{
    int *buf, len, i = 0, j = 0, pos = 0;
    if (n == 0) return -1;                                                              /* tokdll:634-636 */
    if (n < 0 || !s) return -1;
    buf = (int *)malloc(sizeof(int) * (size_t)n);
    len = bfo_utf8_to_utf32(s, n, buf, n);
    if (len <= 0 || len > n) { free(buf); return -1; }
    while (i < len) {                                                                   /* tokdll:651-664 */
        const int c = buf[i++];
        if (!is_ws(c)) buf[j++] = c;
        else if (0 < j && u_space != buf[j - 1]) buf[j++] = u_space;
    }
    if (1 < j && buf[j - 1] == u_space) j--;                                            /* tokdll:667-669 */
    for (i = 0; i < j; ++i) {
        const int k = int_to_utf8(buf[i], out + pos, max_out - pos);
        if (k < 0) { free(buf); return -1; }
        pos += k;
    }
    if (pos < max_out) out[pos] = 0;
    free(buf);
    return pos;
}

/* tokdll:683-815 TextToHashes (model-free; the text is already tokenised, tokens separated by single spaces): fasttext-style
 * FNV-1a hash of every token (bytes sign-extended, tokdll:684-692), then word n-grams h = h * 116049371 + next, stored modulo
 * the bucket count behind the unigram hashes (tokdll:699-714); the n-gram arithmetic runs on sign-extended int32 values */
static uint32_t ft_hash(const char *s, int n) { uint32_t h = 2166136261u; int i; for (i = 0; i < n; ++i) { h ^= (uint32_t)(int8_t)s[i]; h *= 16777619u; } return h; }
int bfo_text_to_hashes(const char *s, int n, int32_t *out, int max_out, int ngrams, int bucket)
{
    int tokens, i, count = 0, pos, wlen = 0; const char *w;
    const int32_t eos = (int32_t)ft_hash("</s>", 4);
    if (ngrams <= 0 && n < 0) return -1;                                                /* tokdll:786-789 */
    if (n == 0) tokens = 0; else { tokens = 1; for (i = 0; i < n; ++i) if (s[i] == ' ') tokens++; }   /* tokdll:718-737 */
    if (tokens * ngrams >= max_out) return n * ngrams;                                  /* tokdll:795-798 */
    w = s;
    for (pos = 0; pos <= n; ++pos) {                                                    /* tokdll:743-771 */
        if (pos == n || s[pos] == ' ') { out[count++] = (int32_t)ft_hash(w, wlen); w = s + pos + 1; wlen = 0; }
        else ++wlen;
    }
    {
        const int tc = count; int k, jn;
        for (k = 0; k < tc; ++k) {
            uint64_t h = (uint64_t)(int64_t)out[k];
            for (jn = k + 1; jn < k + ngrams; ++jn) {
                const uint64_t t = (jn < tc) ? (uint64_t)(int64_t)out[jn] : (uint64_t)(int64_t)eos;
                h = h * 116049371ull + t;
                out[(jn - k) * tc + k] = (int32_t)(h % (uint64_t)(int64_t)bucket);
            }
        }
        count += (ngrams - 1) * tc;
    }
    return count;
}

/* ---------------- exported building blocks ---------------- */

static const dfa_t *pick(const bfo_model *m, int which) { return which ? &m->dict_dfa : &m->wbd_dfa; }
int bfo_has_dfa(const bfo_model *m, int which) { return m && pick(m, which)->set; }
int bfo_dfa_initial(const bfo_model *m, int which) { return pick(m, which)->initial; }
int bfo_dfa_is_final(const bfo_model *m, int which, int state) { return dfa_is_final(pick(m, which), state); }
int bfo_dfa_dest(const bfo_model *m, int which, int state, int iw) { return dfa_dest(pick(m, which), state, iw); }
int bfo_state2ow(const bfo_model *m, int state) { return dfa_state2ow(&m->wbd_dfa, state); }
int bfo_mealy_dest_ow(const bfo_model *m, int state, int iw, int *ow) { return mealy_dest_ow(&m->dict_dfa, state, iw, ow); }
int bfo_wbd_iw_class(const bfo_model *m, int iw) { return m->wbd_dfa.remap ? iwmap_get(&m->wbd_dfa.iwmap, iw) : iw; }
int bfo_wbd_action(const bfo_model *m, int key, int *out, int max_out) { return mmap_get(&m->acts, key, out, max_out); }
int bfo_charmap_get(const bfo_model *m, int which, int key, int *out, int max_out)
{
    const mmapf_t *c = which ? &m->dict_charmap : &m->wbd_charmap;
    return c->set ? mmapf_get(c, key, out, max_out) : -1;
}
int bfo_i2info_get(const bfo_model *m, int key, int *id, uint32_t *bits) { return i2info_get(&m->i2info, key, id, bits); }
int bfo_model_kind(const bfo_model *m)
{
    if (!m->has_seg) return 0;
    if (m->tok_algo == TOKENIZE_BPE) return 2;
    if (m->tok_algo == TOKENIZE_BPE_OPT) return 3;
    if (m->tok_algo == TOKENIZE_BPE_OPT_WITH_MERGES) return 4;
    return 1;
}
int bfo_model_uses_bytes(const bfo_model *m) { return m->use_bytes; }
int bfo_model_id_offset(const bfo_model *m) { return m->id_offset; }

/* ---------------- dictionary key -> info lookup ---------------- */

/* cl/src/FAArray_pack.cpp:68-95 GetAt on the K2I image (header :27-65: M, SizeOfIndex, SizeOfValue, Count) */
static int k2i_get_at(const uint8_t *img, int idx)
{
    const int M = rd_i32(img), soi = rd_i32(img + 4), sov = rd_i32(img + 8), count = rd_i32(img + 12);
    const uint8_t *p = img + 16;
    if (M == 1) return (int)dec_1234_idx(p, (unsigned)idx, sov);
    {
        const uint8_t *data = p + (size_t)((count + M - 1) / M) * (size_t)soi;
        const int chain = (int)dec_1234_idx(p, (unsigned)(idx / M), soi);
        return (int)dec_1234_idx(data + (size_t)chain * (size_t)(M * sov), (unsigned)(idx % M), sov);
    }
}

/* cl/inc/FADictInterpreter_t.h:334-366 GetInfoId for a Mealy [pos-dict] configured like tokdll configures it (SetConf with no
 * transformation, tokdll:953-956 / FADictInterpreter_t.h:155-206): returns the info id or -1.
 *   :347-349  empty / longer than FALimits::MaxWordSize (300) -> -1
 *   :203-205  m_NoNorm = no transformation && !ignore-case && direction == l2r: then the word is looked up AS IS -- the
 *             [pos-dict] charmap is NOT applied (it only is inside Normalize(), i.e. for r2l / ignore-case dictionaries)
 *   :211-279  Normalize: lower-casing (ignore-case), FANormalizeWord with the charmap (FAUtils_cl.h:441-487: a result longer
 *             than the buffer -- 600 elements, 300 when the word was lower-cased first: then it is normalised in place -- counts as length 0), reversal for r2l
 *   :283-301  GetInfoId_mph: FAMphInterpretTools_t::GetId (FAMphInterpretTools_t.h:97-122: walk every symbol, add the output
 *             weights, the last state must be final) then K2I */
int bfo_dict_get_info_id(const bfo_model *m, const int *in, int n)
{
    int buf[600], tmp[600];
    const int *w = in;
    int len = n, i, state, ow, id = 0;
    if (!m || !m->has_seg || !m->k2i || n <= 0 || n > 300 || !in) return -1;
    if (m->dict_ignore_case || m->direction != 0) {           /* !m_NoNorm: Normalize (:211-279) */
        int low[300];
        if (m->dict_ignore_case) { for (i = 0; i < n; ++i) low[i] = bfo_tolower(in[i]); w = low; }      /* :231-238 */
        /* FAUtils_cl.h:441-487: in place (the ignore-case way: pSrc == pOut) the result goes through Tmp[MaxWordLen = 300], else into the 600 of pOut */
        if (m->dict_charmap.set) { const int lim = m->dict_ignore_case ? 300 : 600; len = normalize(w, n, tmp, NULL, 600, &m->dict_charmap); if (len < 0 || len > lim) len = 0; w = tmp; }
        if (m->direction != 0) { for (i = 0; i < len; ++i) buf[i] = w[len - 1 - i]; w = buf; }
    }
    state = m->dict_dfa.initial;
    for (i = 0; i < len; ++i) {
        state = mealy_dest_ow(&m->dict_dfa, state, w[i], &ow);
        if (state == -1) return -1;
        id += ow;
    }
    if (!dfa_is_final(&m->dict_dfa, state)) return -1;
    if (id < 0 || id >= m->k2i_count) return -1;   /* DebugLogAssert in the reference (:295) */
    return k2i_get_at(m->k2i, id);
}

/* cl/inc/FADictInterpreter_t.h:369-390 GetInfo: info id -> I2Info row (FAMultiMap_pack_fixed.cpp:67-137: the values are copied
 * only if max_out >= count; returns count, -1 = no such word / row) */
int bfo_dict_get_info(const bfo_model *m, const int *in, int n, int *out, int max_out)
{
    const int id = bfo_dict_get_info_id(m, in, n);
    if (id == -1) return -1;
    return mmapf_get(&m->i2info, id, out, max_out);
}
