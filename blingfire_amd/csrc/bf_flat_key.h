// bf_flat_key.h -- the word table of the flat program (bf_flat.h): constants and hash functions shared by the loader (bf_model.cpp, host
// only) and the kernels.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define BF_FK __host__ __device__ inline
#else
#define BF_FK inline
#endif

namespace bfa {

constexpr int WF_CHUNK = 512;            // bytes per step (8 per lane)
constexpr int WF_RING = 1024;            // byte positions whose class is kept (two chunks: a run may begin in the chunk before)
constexpr int WF_REC_SHIFT = 4;           // a range of b bytes owns b >> WF_REC_SHIFT records of the word list (round 6: one per 16 bytes instead of one per 4 -- 1.3 instead of 4 bytes of
                                         // workspace per byte of text; the metric's corpus needs one per 167 bytes, a range whose words do not fit hands their documents on)
constexpr int WF_REC = 64;               // words that wait for a unit, at most (one per lane)
constexpr int WF_RUN_MAX = 48;           // bytes of the longest run the program resolves itself
constexpr int WF_KEY_CHARS = 12;         // one byte per character (its code: class + 1, 1 .. 127)
constexpr uint32_t WF_CONT = 0xFFFFu;    // ring: a byte that starts no character
constexpr uint32_t WF_ENT_FLAG = 0x80000000u;   // entry with bit 31: [30:25] number of ids (0: one id, UnkId), [24:0] home index - entry index
constexpr int WF_ENT_CNT_SHIFT = 25;
constexpr uint32_t WF_ENT_DELTA_MASK = (1u << WF_ENT_CNT_SHIFT) - 1u;
constexpr int32_t WF_D_BAD = 1, WF_D_HARD = 2;  // dstat[] bits: invalid UTF-8 (0 ids); handed to the wave program
constexpr int WF_SPAN_LEN_SHIFT = 24;    // a span (offsets API): [21:0] first byte in the document, [29:24] bytes - 1
constexpr uint32_t WF_SPAN_POS_MASK = (1u << 22) - 1u;
constexpr int64_t WF_DOC_MAX = 1 << 22;  // a batch with a longer document (or with offsets out of order) is not taken (k_wp_pre sets *unsafe)
constexpr int64_t WF_RANGE_MAX = 1 << 22; // bytes a range aims at, at most (a range ends with a whole document: < WF_RANGE_MAX + WF_DOC_MAX bytes)
// A key is 12 bytes: the codes of the word's characters, first character in the lowest byte, 0 behind the word (k0: characters 0..7, k1: 8..11).
// A code is never 0 and never has bit 7: byte 0 == 0x80 marks the key of a one-element token (class in bits 8..), byte 0 == 0xFF is no key at all.
constexpr uint64_t WF_KEY_SOLO = 0x80ull, WF_KEY_NONE = 0xFFull;
constexpr int WF_KEY_SOLO_SHIFT = 8;

// ---- the word table.  Entry e = 16 bytes {k0 (8), k1 (4), id | characters << 24 (4)}; k0 == 0: empty; two candidate entries per key.
constexpr int WF_ROW_LEN_SHIFT = 24;
constexpr uint32_t WF_ROW_ID_MASK = (1u << WF_ROW_LEN_SHIFT) - 1u;
BF_FK uint32_t wf_mix(uint64_t k0, uint32_t k1, uint32_t m0) { return (uint32_t)k0 ^ ((uint32_t)(k0 >> 32) * m0) ^ (k1 * 0x85EBCA6Bu); }
BF_FK uint32_t wf_h(uint32_t x, uint32_t m, int bits) { return (x * m) >> (32 - bits); }
// code of a class inside a key (0: the class has none)
BF_FK uint32_t wf_code(uint32_t cls, uint32_t kind) { return (kind == 1u /* WK_LOOP */ && cls < 127u) ? cls + 1u : 0u; }

} // namespace bfa
